"""Data_Feat_TOPN — the dataset behind `--model attention_topn`: mirror of MER2024/toolkit/data/feat_data_topn.py:9-96.

For `--fusion_topn n --fusion_modality AVT|AV|AT|VT` it loads the n best-ranked utterance-level feature sets of each of three
modality slots (MER2024/toolkit/globals.py:218-231: the ranking of the MER2024 baseline paper, low to high), time-averages each
clip's feature and hands the model `feat0 .. feat{3n-1}`.  One deliberate difference: the reference decides between the
`<model>_UTT` and `<model>-UTT` directory spellings by probing the SIMS corpus' feature root whatever the dataset
(MER2024/toolkit/utils/functions.py:246-253); here the probe looks under the dataset's own feature root."""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from ... import config
from ..globals import AUDIO_RANK_LOW2HIGH, FEATURE_DIR_OF, IMAGE_RANK_LOW2HIGH, TEXT_RANK_LOW2HIGH
from ..utils.read_data import func_read_multiprocess

_SLOTS = {'AVT': ('A', 'T', 'V'), 'AT': ('A', 'T', 'T'), 'AV': ('A', 'V', 'V'), 'VT': ('T', 'T', 'V')}   # reference :17-24
_RANK = {'A': AUDIO_RANK_LOW2HIGH, 'T': TEXT_RANK_LOW2HIGH, 'V': IMAGE_RANK_LOW2HIGH}


def topn_feature_names(topn, modality, feat_root=None, suffix='UTT'):
    """Directory names of the 3 * topn feature sets, in model-input order."""
    assert topn is not None and modality in _SLOTS
    names = []
    for slot in _SLOTS[modality]:
        names.extend(_RANK[slot][-topn:])
    assert len(names) == topn * 3
    out = []
    for display in names:
        stem = FEATURE_DIR_OF[display]
        under = f'{stem}_{suffix}'
        out.append(under if feat_root is not None and os.path.exists(os.path.join(feat_root, under)) else f'{stem}-{suffix}')
    return out


class Data_Feat_TOPN(Dataset):
    def __init__(self, args, names, labels):
        self.names, self.labels = names, labels
        feat_root = config.PATH_TO_FEATURES[args.dataset]
        featnames = topn_feature_names(args.fusion_topn, args.fusion_modality, feat_root)
        print(f'feature number: {len(featnames)}')
        self.feat_type, self.feat_scale = args.feat_type, args.feat_scale
        assert self.feat_scale == 1 and self.feat_type == 'utt'
        self.whole_features, self.whole_dims = [], []
        for name in featnames:
            feats, dim = func_read_multiprocess(os.path.join(feat_root, name), self.names, read_type='feat')
            self.whole_features.append([np.mean(f, axis=0) for f in feats])
            self.whole_dims.append(dim)
        if args.debug:    # reference :52-58: one collate over random items as a smoke check
            self.collater([self[random.randint(0, len(self.names) - 1)] for _ in range(32)])

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        item = dict(emo=self.labels[index]['emo'], val=self.labels[index]['val'], name=self.names[index])
        for i, feats in enumerate(self.whole_features):
            item[f'feat{i}'] = feats[index]
        return item

    def collater(self, instances):
        batch = {f'feat{i}': torch.FloatTensor(np.array([x[f'feat{i}'] for x in instances])) for i in range(len(self.whole_features))}
        emos = torch.LongTensor([x['emo'] for x in instances])
        vals = torch.FloatTensor([x['val'] for x in instances])
        return batch, emos, vals, [x['name'] for x in instances]

    def get_featdim(self):
        print(f'topn feature dims: {self.whole_dims}')
        return self.whole_dims, self.whole_dims, self.whole_dims
