"""Path tables — mirror of MERBench/config.py:4-86 (same dict names and keys).  The reference hard-codes the
authors' cluster paths; here every dataset root hangs off MER_DATA_ROOT (default ./dataset) and can be
re-pointed at run time with set_data_root()."""
import os

_DATASETS = {'MER2023': 'mer2023-dataset-process', 'MER2024': 'mer2024-dataset-process'}
_LABELS = {'MER2023': 'label-6way.npz', 'MER2024': 'label-6way.npz'}

DATA_DIR, PATH_TO_RAW_AUDIO, PATH_TO_RAW_VIDEO, PATH_TO_RAW_FACE = {}, {}, {}, {}
PATH_TO_TRANSCRIPTIONS, PATH_TO_FEATURES, PATH_TO_LABEL = {}, {}, {}

PATH_TO_PRETRAINED_MODELS = './tools'
SAVED_ROOT = os.path.join('./saved')
MODEL_DIR = os.path.join(SAVED_ROOT, 'model')
LOG_DIR = os.path.join(SAVED_ROOT, 'log')


def set_data_root(root):
    for ds, sub in _DATASETS.items():
        d = os.path.join(root, sub)
        DATA_DIR[ds] = d
        PATH_TO_RAW_AUDIO[ds] = os.path.join(d, 'audio')
        PATH_TO_RAW_VIDEO[ds] = os.path.join(d, 'video')
        PATH_TO_RAW_FACE[ds] = os.path.join(d, 'openface_face')
        PATH_TO_TRANSCRIPTIONS[ds] = os.path.join(d, 'transcription-engchi-polish.csv')
        PATH_TO_FEATURES[ds] = os.path.join(d, 'features')
        PATH_TO_LABEL[ds] = os.path.join(d, _LABELS[ds])


set_data_root(os.environ.get('MER_DATA_ROOT', './dataset'))
