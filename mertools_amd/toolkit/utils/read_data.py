"""Per-clip feature files -> model-ready arrays: the on-disk contract between the extractors and the fusion trainer
(SURVEY.md §8 a16).  Behaviour follows MERBench/toolkit/utils/read_data.py:15-125 — shapes, dtypes (float64 as soon as
padding zeros are involved), FRONT padding, ceil-sized mean-pooling windows — and is checked bit for bit against vectors
produced by the reference's own functions (tests/golden/index_paths.npz); the code is organised around two primitives:

    load_rows(root, name)   one clip's [T, D] array from `<name>.npy` or a directory of per-face files
    fit_length(rows, n)     [T, D] -> [n, D]: zeros in front when short, mean over ceil(T / n)-row windows when long

and `regroup()`, which expresses every alignment mode of the reference (utterance pooling, feat_scale compression,
align-to-text, pad-to-longest) as a choice of target lengths per clip and modality.  The reference's function names are kept
as aliases at the bottom because MERBench's `feat_data.py` / user scripts import them.
"""
import math
import multiprocessing
import os

import numpy as np


def load_rows(root, name):
    """[T, D] features of clip `name` under `root` (reference :15-41: `<name>.npy`, else every file of directory `<name>` in
    sorted order; a 1-D utterance vector becomes one row)."""
    single = os.path.join(root, name + '.npy')
    folder = os.path.join(root, name)
    if os.path.exists(single):
        parts = [np.load(single).squeeze()]
    elif os.path.isdir(folder):
        parts = [np.load(os.path.join(folder, f)) for f in sorted(os.listdir(folder))]
    else:
        raise Exception('feature path or dir do not exist!')
    rows = np.array(parts).squeeze()
    if len(rows) == 0:
        print('feature has errors!!')
        return rows
    return rows[np.newaxis, :] if rows.ndim == 1 else rows


def _load_rows_packed(job):
    return load_rows(*job)


def load_all(root, names, processes=8):
    """([T_i, D] per clip, D).  Large corpora are read by a process pool (the reference's Pool(8), :44-68)."""
    jobs = [(root, name) for name in names]
    if processes and processes > 1 and len(jobs) > 64:
        with multiprocessing.Pool(processes=processes) as pool:
            clips = list(pool.imap(_load_rows_packed, jobs))
    else:
        clips = [load_rows(*j) for j in jobs]
    assert len(clips) == len(names), 'Error: len(names) != len(features)'
    shape = np.array(clips[0]).shape
    print(f'Input feature {os.path.basename(root)} ===> dim is {shape}')
    return clips, shape[-1]


def fit_length(rows, n):
    """[T, D] -> [n, D] (reference :72-89).  T == n: the array itself.  Otherwise zeros are put in FRONT until the length is
    n * w with w = ceil(T / n) (w = 1 when T < n) and every w consecutive rows are averaged — float64, because the padding
    zeros are."""
    t, d = rows.shape
    if t == n:
        return rows
    w = 1 if t < n else -(-t // n)
    padded = np.concatenate([np.zeros((n * w - t, d)), rows])
    return padded if w == 1 else padded.reshape(n, w, d).mean(axis=1)


def regroup(audios, texts, videos, mode, scale=1):
    """The reference's alignment pipeline (feat_data.py:30-44 over read_data.py:92-125) for three per-clip lists, in place:
         'compress'     every stream to ceil(T / scale) rows                                   (feature_scale_compress)
         'utt'          every stream to its mean row [D]                                       (align_to_utt)
         'text'         audio and video to the clip's text length                              (align_to_text)
         'longest'      every stream to the longest clip of its own modality                   (pad_to_maxlen_pre_modality)"""
    streams = (audios, texts, videos)
    if mode == 'utt':
        for s in streams:
            for i, rows in enumerate(s):
                s[i] = np.mean(rows, axis=0)
        return audios, texts, videos
    if mode == 'compress':
        target = lambda s, i: math.ceil(len(s[i]) / scale)           # noqa: E731
    elif mode == 'text':
        lens = [len(t) for t in texts]
        target = lambda s, i: lens[i]                                # noqa: E731
    elif mode == 'longest':
        longest = {id(s): max(len(r) for r in s) for s in streams}
        target = lambda s, i: longest[id(s)]                         # noqa: E731
    else:
        raise ValueError(mode)
    for i in range(len(audios)):
        for s in streams:
            s[i] = fit_length(s[i], target(s, i))
    return audios, texts, videos


# ---- the reference's names (MERBench/toolkit/utils/read_data.py) --------------------------------------------------------
func_mapping_feature = fit_length


def func_read_one_feat(argv=None, feature_root=None, name=None, processor=None, model_name=None):
    root, name = argv[0], argv[1]
    return load_rows(root, name)


def func_read_multiprocess(feature_root, names, processor=None, read_type='feat', model_name=None, processes=8):
    assert read_type == 'feat', 'only pre-extracted features are on the hot path (SURVEY.md §2: e2e readers are out of scope)'
    return load_all(feature_root, names, processes)


def align_to_utt(audios, texts, videos):
    return regroup(audios, texts, videos, 'utt')


def feature_scale_compress(audios, texts, videos, scale_factor=1):
    return regroup(audios, texts, videos, 'compress', scale_factor)


def align_to_text(audios, texts, videos):
    return regroup(audios, texts, videos, 'text')


def pad_to_maxlen_pre_modality(audios, texts, videos):
    return regroup(audios, texts, videos, 'longest')
