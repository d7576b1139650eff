"""MER2023 corpus loader — mirror of MERBench/toolkit/dataloader/mer2023.py:12-155: label parsing, 5-fold split
with python `random.shuffle` (bit-exact under a fixed random.seed), three unshuffled test loaders, metrics."""
import random

import numpy as np
from sklearn.metrics import accuracy_score, f1_score, mean_squared_error
from torch.utils.data import DataLoader
from torch.utils.data.sampler import SubsetRandomSampler

from ... import config
from ..data import get_datasets
from ..globals import emo2idx_mer


class MER2023:
    def __init__(self, args):
        self.args = args
        self.debug = args.debug
        self.num_folder = 5
        self.batch_size = args.batch_size
        self.num_workers = args.num_workers
        self.label_path = config.PATH_TO_LABEL[args.dataset]
        self.dataset = args.dataset
        assert self.dataset in ['MER2023']
        args.output_dim1 = 6
        args.output_dim2 = 1
        args.metric_name = 'emoval'

    def _loader(self, dataset, sampler=None):
        return DataLoader(dataset, batch_size=self.batch_size, sampler=sampler, num_workers=self.num_workers,
                          collate_fn=dataset.collater, shuffle=False if sampler is None else None, pin_memory=True)

    def get_loaders(self):
        names, labels = self.read_names_labels(self.label_path, 'train', debug=self.debug)
        print(f'train: sample number {len(names)}')
        train_dataset = get_datasets(self.args, names, labels)
        train_eval_idxs = self.random_split_indexes(len(names), self.num_folder)
        train_loaders, eval_loaders = [], []
        for train_idxs, eval_idxs in train_eval_idxs:
            train_loaders.append(self._loader(train_dataset, SubsetRandomSampler(train_idxs)))
            eval_loaders.append(self._loader(train_dataset, SubsetRandomSampler(eval_idxs)))
        test_loaders = []
        for data_type in ['test1', 'test2', 'test3']:
            names, labels = self.read_names_labels(self.label_path, data_type, debug=self.debug)
            print(f'{data_type}: sample number {len(names)}')
            test_loaders.append(self._loader(get_datasets(self.args, names, labels)))
        return train_loaders, eval_loaders, test_loaders

    def read_names_labels(self, label_path, data_type, debug=False):
        assert data_type in ['train', 'test1', 'test2', 'test3']
        corpus = np.load(label_path, allow_pickle=True)[f'{data_type}_corpus'].tolist()
        names, labels = [], []
        for name in corpus:
            names.append(name)
            labels.append(corpus[name])
        for ii, label in enumerate(labels):
            val = -10 if ('val' not in label or label['val'] == '') else label['val']
            labels[ii] = {'emo': emo2idx_mer[label['emo']], 'val': val}
        if debug:
            names, labels = names[:100], labels[:100]
        return names, labels

    def random_split_indexes(self, whole_num, num_folder):
        indices = np.arange(whole_num)
        random.shuffle(indices)
        each = int(whole_num / num_folder)
        whole_folder = [indices[each * ii: each * (ii + 1)] for ii in range(num_folder - 1)]
        whole_folder.append(indices[each * (num_folder - 1):])
        assert len(whole_folder) == num_folder
        assert sum(len(f) for f in whole_folder) == whole_num
        train_eval_idxs = []
        for ii in range(num_folder):
            train_idxs = []
            for jj in range(num_folder):
                if jj != ii:
                    train_idxs.extend(whole_folder[jj])
            train_eval_idxs.append([train_idxs, whole_folder[ii]])
        return train_eval_idxs

    def calculate_results(self, emo_probs=[], emo_labels=[], val_preds=[], val_labels=[]):
        emo_preds = np.argmax(emo_probs, 1)
        emo_accuracy = accuracy_score(emo_labels, emo_preds)
        emo_fscore = f1_score(emo_labels, emo_preds, average='weighted')
        val_mse = mean_squared_error(val_labels, val_preds)
        results = {'emoprobs': emo_probs, 'emolabels': emo_labels, 'emoacc': emo_accuracy, 'emofscore': emo_fscore,
                   'valpreds': val_preds, 'vallabels': val_labels, 'valmse': val_mse}
        return results, f'f1:{emo_fscore:.4f}_acc:{emo_accuracy:.4f}_val:{val_mse:.4f}'
