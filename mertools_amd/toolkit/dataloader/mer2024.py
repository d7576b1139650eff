"""MER2024 corpus (`--dataset MER2024`): emotion-only labels, 5-fold CV on `train`, one test set.

Behaviour follows MER2024/toolkit/dataloader/mer2024.py:12-146 and is checked against vectors made by the reference's own class
(tests/golden/gen_golden.py -> index_paths_mer2024.npz).  What differs from MER2023 and matters for a drop-in:
  * `args.output_dim2 = 0`, `args.metric_name = 'emo'`: there is no valence head, `main_release` skips the MSE term and model
    selection runs on the weighted F1 alone (:25-27);
  * every label dict carries the sentinel valence -10 whatever the file holds (:97-99);
  * splits are `train` and `test1` only (:90-92);
  * `calculate_results` reports accuracy / weighted F1 and the string 'f1:…_acc:…' — no MSE (:131-146);
  * the feature directory of a split is chosen by `args.snr` (train: `args.train_snr`, test: `args.test_snr`, :37,69), which
    `Data_Feat` turns into `<model>-<snr>-UTT` (MER2024/toolkit/data/feat_data.py:13-22).
The label parsing and the fold construction are the MER2023 functions (same arithmetic in the reference: :103-127 == mer2023.py's)."""
import numpy as np
from sklearn.metrics import accuracy_score, f1_score

from ... import config
from ..data import get_datasets
from ..globals import emo2idx_mer
from .mer2023 import MER2023, MISSING_VALENCE, kfold_indices

SPLITS = ('train', 'test1')


def parse_corpus(label_path, split, limit=None):
    """(names, [{'emo': class index, 'val': -10}]) of one split, in file order."""
    assert split in SPLITS
    corpus = np.load(label_path, allow_pickle=True)[f'{split}_corpus'].tolist()
    names = list(corpus)[:limit]
    return names, [{'emo': emo2idx_mer[corpus[n]['emo']], 'val': MISSING_VALENCE} for n in names]


def emotion_metrics(emo_probs, emo_labels):
    pred = np.argmax(emo_probs, 1)
    acc = accuracy_score(emo_labels, pred)
    f1 = f1_score(emo_labels, pred, average='weighted')
    return dict(emoprobs=emo_probs, emolabels=emo_labels, emoacc=acc, emofscore=f1), f'f1:{f1:.4f}_acc:{acc:.4f}'


class MER2024(MER2023):
    def __init__(self, args):
        assert args.dataset in ['MER2024']
        self.args, self.dataset, self.debug = args, args.dataset, args.debug
        self.batch_size, self.num_workers = args.batch_size, args.num_workers
        self.label_path = config.PATH_TO_LABEL[args.dataset]
        args.output_dim1, args.output_dim2, args.metric_name = 6, 0, 'emo'

    def get_loaders(self):
        names, labels = self.read_names_labels(self.label_path, 'train', debug=self.debug)
        print(f'train: sample number {len(names)}')
        self.args.snr = getattr(self.args, 'train_snr', None)
        train_set = get_datasets(self.args, names, labels)
        folds = self.random_split_indexes(len(names), self.num_folder)
        train_loaders = [self._loader(train_set, tr) for tr, _ in folds]
        eval_loaders = [self._loader(train_set, ev) for _, ev in folds]
        test_loaders = []
        for split in SPLITS[1:]:
            names, labels = self.read_names_labels(self.label_path, split, debug=self.debug)
            print(f'{split}: sample number {len(names)}')
            self.args.snr = getattr(self.args, 'test_snr', None)
            test_loaders.append(self._loader(get_datasets(self.args, names, labels)))
        return train_loaders, eval_loaders, test_loaders

    def read_names_labels(self, label_path, data_type, debug=False):
        return parse_corpus(label_path, data_type, limit=100 if debug else None)

    def random_split_indexes(self, whole_num, num_folder):
        return kfold_indices(whole_num, num_folder)

    def calculate_results(self, emo_probs=[], emo_labels=[], val_preds=[], val_labels=[]):
        return emotion_metrics(emo_probs, emo_labels)
