"""get_datasets(args, names, labels) — mirror of MERBench/toolkit/data/__init__.py:6-41."""
from torch.utils.data import Dataset

from .feat_data import Data_Feat
from .feat_data_topn import Data_Feat_TOPN

MODEL_DATASET_MAP = {k: Data_Feat for k in ['attention', 'lf_dnn', 'lmf', 'misa', 'mmim', 'tfn', 'mfn', 'graph_mfn',
                                            'ef_lstm', 'mfm', 'mctn', 'mult']}
MODEL_DATASET_MAP['attention_topn'] = Data_Feat_TOPN     # MER2024/toolkit/data/__init__.py:27: several feature sets per modality


class get_datasets(Dataset):
    def __init__(self, args, names, labels):
        self.dataset_class = MODEL_DATASET_MAP[args.model]
        self.dataset = self.dataset_class(args, names, labels)

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        return self.dataset[index]

    def collater(self, instances):
        return self.dataset.collater(instances)

    def get_featdim(self):
        return self.dataset.get_featdim()
