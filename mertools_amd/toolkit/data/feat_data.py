"""Data_Feat — mirror of MERBench/toolkit/data/feat_data.py:6-82: loads the three per-clip .npy feature sets
into RAM, applies feat_scale / utterance alignment, collates FloatTensor / LongTensor batches."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from ... import config
from ..utils.read_data import (align_to_text, align_to_utt, feature_scale_compress, func_read_multiprocess,
                               pad_to_maxlen_pre_modality)


class Data_Feat(Dataset):
    def __init__(self, args, names, labels):
        self.names, self.labels = names, labels
        feat_root = config.PATH_TO_FEATURES[args.dataset]
        # MER2024's noise-robustness runs (MER2024/toolkit/data/feat_data.py:13-22): `args.snr` re-points every modality at
        # <model><sep><snr><sep>UTT, <sep> being the separator the feature name already uses before its UTT suffix
        snr = getattr(args, 'snr', None)

        def root_of(name):
            return os.path.join(feat_root, name if snr is None else f'{name[:-4]}{name[-4]}{snr}{name[-4]}UTT')

        audio_root, text_root, video_root = root_of(args.audio_feature), root_of(args.text_feature), root_of(args.video_feature)
        print(f'audio feature root: {audio_root}')
        self.feat_type, self.feat_scale = args.feat_type, args.feat_scale
        assert self.feat_scale >= 1
        assert self.feat_type in ['utt', 'frm_align', 'frm_unalign']
        audios, self.adim = func_read_multiprocess(audio_root, self.names, read_type='feat')
        texts, self.tdim = func_read_multiprocess(text_root, self.names, read_type='feat')
        videos, self.vdim = func_read_multiprocess(video_root, self.names, read_type='feat')
        audios, texts, videos = feature_scale_compress(audios, texts, videos, self.feat_scale)
        if self.feat_type == 'utt':
            audios, texts, videos = align_to_utt(audios, texts, videos)
        elif self.feat_type == 'frm_align':
            audios, texts, videos = align_to_text(audios, texts, videos)
            audios, texts, videos = pad_to_maxlen_pre_modality(audios, texts, videos)
        else:
            audios, texts, videos = pad_to_maxlen_pre_modality(audios, texts, videos)
        self.audios, self.texts, self.videos = audios, texts, videos

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        return dict(audio=self.audios[index], text=self.texts[index], video=self.videos[index],
                    emo=self.labels[index]['emo'], val=self.labels[index]['val'], name=self.names[index])

    def collater(self, instances):
        batch = dict(audios=torch.FloatTensor(np.array([i['audio'] for i in instances])),
                     texts=torch.FloatTensor(np.array([i['text'] for i in instances])),
                     videos=torch.FloatTensor(np.array([i['video'] for i in instances])))
        emos = torch.LongTensor([i['emo'] for i in instances])
        vals = torch.FloatTensor([i['val'] for i in instances])
        names = [i['name'] for i in instances]
        return batch, emos, vals, names

    def get_featdim(self):
        print(f'audio dimension: {self.adim}; text dimension: {self.tdim}; video dimension: {self.vdim}')
        return self.adim, self.tdim, self.vdim
