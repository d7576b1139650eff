"""MER2023 corpus: labels, the 5-fold cross-validation split, loaders and metrics.

Behaviour follows MERBench/toolkit/dataloader/mer2023.py:12-155 and is bit-exact against vectors produced by the reference's
own class on its own label file (tests/golden/index_paths.npz, mer2023_label-6way.npz): same label dicts (missing valence ->
-10), the same folds under a given `random.seed` (the reference shuffles a numpy index array with python's `random.shuffle`,
so that exact call is kept), the same metric values and result-string formatting.  The pieces are plain functions; the
`MER2023` class is the interface `main_release` (and MERBench's `get_dataloaders`) expects."""
import random

import numpy as np
from sklearn.metrics import accuracy_score, f1_score, mean_squared_error
from torch.utils.data import DataLoader
from torch.utils.data.sampler import SubsetRandomSampler

from ... import config
from ..data import get_datasets
from ..globals import emo2idx_mer

SPLITS = ('train', 'test1', 'test2', 'test3')
MISSING_VALENCE = -10          # reference :95-98: test3 carries no valence; the sentinel is still fed to the MSE


def parse_corpus(label_path, split, limit=None):
    """(names, [{'emo': class index, 'val': valence or -10}]) of one split of the label npz, in file order."""
    assert split in SPLITS
    corpus = np.load(label_path, allow_pickle=True)[f'{split}_corpus'].tolist()
    names = list(corpus)[:limit]
    labels = []
    for name in names:
        entry = corpus[name]
        val = entry.get('val', '')
        labels.append({'emo': emo2idx_mer[entry['emo']], 'val': MISSING_VALENCE if val == '' else val})
    return names, labels


def kfold_indices(n, k):
    """k (train_indices, eval_indices) pairs over range(n): python-RNG shuffle, k - 1 blocks of n // k and the rest in the
    last block; fold i evaluates on block i and trains on the other blocks in block order."""
    order = np.arange(n)
    random.shuffle(order)
    size = n // k
    cuts = [size * i for i in range(k)] + [n]
    blocks = [order[cuts[i]:cuts[i + 1]] for i in range(k)]
    assert sum(len(b) for b in blocks) == n
    return [[[int(j) for i, b in enumerate(blocks) if i != held for j in b], blocks[held]] for held in range(k)]


def emotion_valence_metrics(emo_probs, emo_labels, val_preds, val_labels):
    """(result dict, 'f1:…_acc:…_val:…') — weighted F1 / accuracy of argmax(probs), MSE of the valence."""
    pred = np.argmax(emo_probs, 1)
    acc = accuracy_score(emo_labels, pred)
    f1 = f1_score(emo_labels, pred, average='weighted')
    mse = mean_squared_error(val_labels, val_preds)
    res = dict(emoprobs=emo_probs, emolabels=emo_labels, emoacc=acc, emofscore=f1, valpreds=val_preds, vallabels=val_labels, valmse=mse)
    return res, f'f1:{f1:.4f}_acc:{acc:.4f}_val:{mse:.4f}'


class MER2023:
    num_folder = 5

    def __init__(self, args):
        assert args.dataset in ['MER2023']
        self.args, self.dataset, self.debug = args, args.dataset, args.debug
        self.batch_size, self.num_workers = args.batch_size, args.num_workers
        self.label_path = config.PATH_TO_LABEL[args.dataset]
        args.output_dim1, args.output_dim2, args.metric_name = 6, 1, 'emoval'

    def _loader(self, dataset, indices=None):
        sampler = SubsetRandomSampler(indices) if indices is not None else None      # eval folds are sampled too (reference :57)
        return DataLoader(dataset, batch_size=self.batch_size, sampler=sampler, num_workers=self.num_workers,
                          collate_fn=dataset.collater, shuffle=False if sampler is None else None, pin_memory=True)

    def get_loaders(self):
        names, labels = self.read_names_labels(self.label_path, 'train', debug=self.debug)
        print(f'train: sample number {len(names)}')
        train_set = get_datasets(self.args, names, labels)
        folds = self.random_split_indexes(len(names), self.num_folder)
        train_loaders = [self._loader(train_set, tr) for tr, _ in folds]
        eval_loaders = [self._loader(train_set, ev) for _, ev in folds]
        test_loaders = []
        for split in SPLITS[1:]:
            names, labels = self.read_names_labels(self.label_path, split, debug=self.debug)
            print(f'{split}: sample number {len(names)}')
            test_loaders.append(self._loader(get_datasets(self.args, names, labels)))
        return train_loaders, eval_loaders, test_loaders

    # the reference's method names
    def read_names_labels(self, label_path, data_type, debug=False):
        return parse_corpus(label_path, data_type, limit=100 if debug else None)

    def random_split_indexes(self, whole_num, num_folder):
        return kfold_indices(whole_num, num_folder)

    def calculate_results(self, emo_probs=[], emo_labels=[], val_preds=[], val_labels=[]):
        return emotion_valence_metrics(emo_probs, emo_labels, val_preds, val_labels)
