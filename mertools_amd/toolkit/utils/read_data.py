""".npy feature readers and utterance/frame alignment — mirror of MERBench/toolkit/utils/read_data.py:15-125.
This is the on-disk contract between the extractors and the trainer: shapes, dtypes,
padding side and pooling windows are reproduced exactly; values are numpy float arithmetic on the host."""
import math
import multiprocessing
import os

import numpy as np


def func_read_one_feat(argv=None, feature_root=None, name=None, processor=None, model_name=None):
    feature_root, name, processor, model_name = argv
    feature_path = os.path.join(feature_root, name + '.npy')
    feature_dir = os.path.join(feature_root, name)
    feature = []
    if os.path.exists(feature_path):
        feature.append(np.load(feature_path).squeeze())  # [D] or [T, D]
    elif os.path.isdir(feature_dir):
        for facename in sorted(os.listdir(feature_dir)):
            feature.append(np.load(os.path.join(feature_dir, facename)))
    else:
        raise Exception('feature path or dir do not exist!')
    single_feature = np.array(feature).squeeze()
    if len(single_feature) == 0:
        print('feature has errors!!')
    elif len(single_feature.shape) == 1:
        single_feature = single_feature[np.newaxis, :]
    return single_feature


def func_read_multiprocess(feature_root, names, processor=None, read_type='feat', model_name=None, processes=8):
    params = [(feature_root, name, processor, model_name) for name in names]
    features = []
    if read_type == 'feat':
        if processes and processes > 1 and len(params) > 64:
            with multiprocessing.Pool(processes=processes) as pool:
                features = list(pool.imap(func_read_one_feat, params))
        else:
            features = [func_read_one_feat(p) for p in params]
    feature_shape = np.array(features[0]).shape
    print(f'Input feature {os.path.basename(feature_root)} ===> dim is {feature_shape}')
    assert len(names) == len(features), 'Error: len(names) != len(features)'
    return features, feature_shape[-1]


def func_mapping_feature(feature, dst_len):
    """(seqlen, D) -> (dst_len, D): zero-pad in FRONT when short; mean-pool ceil windows (front-padded) when long."""
    featlen, featdim = feature.shape
    if featlen == dst_len:
        return feature
    if featlen < dst_len:
        return np.concatenate((np.zeros((dst_len - featlen, featdim)), feature), axis=0)
    if featlen // dst_len == featlen / dst_len:
        pad_len, pool_size = 0, featlen // dst_len
    else:
        pad_len, pool_size = dst_len - featlen % dst_len, featlen // dst_len + 1
    feature = np.concatenate([np.zeros((pad_len, featdim)), feature]).reshape(dst_len, pool_size, featdim)
    return np.mean(feature, axis=1)


def align_to_utt(audios, texts, videos):
    for ii in range(len(audios)):
        audios[ii] = np.mean(audios[ii], axis=0)
        texts[ii] = np.mean(texts[ii], axis=0)
        videos[ii] = np.mean(videos[ii], axis=0)
    return audios, texts, videos


def feature_scale_compress(audios, texts, videos, scale_factor=1):
    for ii in range(len(audios)):
        audios[ii] = func_mapping_feature(audios[ii], math.ceil(len(audios[ii]) / scale_factor))
        texts[ii] = func_mapping_feature(texts[ii], math.ceil(len(texts[ii]) / scale_factor))
        videos[ii] = func_mapping_feature(videos[ii], math.ceil(len(videos[ii]) / scale_factor))
    return audios, texts, videos


def align_to_text(audios, texts, videos):
    for ii in range(len(audios)):
        dst_len = len(texts[ii])
        audios[ii] = func_mapping_feature(audios[ii], dst_len)
        texts[ii] = func_mapping_feature(texts[ii], dst_len)
        videos[ii] = func_mapping_feature(videos[ii], dst_len)
    return audios, texts, videos


def pad_to_maxlen_pre_modality(audios, texts, videos):
    amax, tmax, vmax = max(len(f) for f in audios), max(len(f) for f in texts), max(len(f) for f in videos)
    for ii in range(len(audios)):
        audios[ii] = func_mapping_feature(audios[ii], amax)
        texts[ii] = func_mapping_feature(texts[ii], tmax)
        videos[ii] = func_mapping_feature(videos[ii], vmax)
    return audios, texts, videos
