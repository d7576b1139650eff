import os
import sys

import pytest

# the numerics-study presets ("mean_all", "mixed", "a2f_conv3", ...: encoders._STUDY_PREC) resolve for the suite only
os.environ.setdefault("MER_STUDY_PRESETS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
