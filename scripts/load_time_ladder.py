"""How long a load takes with the self-check ladder (VERDICT r5 #6c): plain build, build + self-check (healthy checkpoint: the twin only), and the
worst case (outlier checkpoint: every rung up to `accurate`), HuBERT-base.  Usage: python scripts/load_time_ladder.py"""
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MER_STUDY_PRESETS", "0")
from mertools_amd import synthetic as W   # noqa: E402
from mertools_amd.encoders import HipHubertModel   # noqa: E402

dev = torch.device("cuda:0")
cfg = W.hubert_config("base")
sd = W.hubert_state_dict(cfg, 0)
sdo = W.ln_outliers(W.hubert_state_dict(cfg, 0))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = dict(seconds=round(dt, 2), precision=m.precision, result=getattr(m, "self_check_result", None))
    del m
    torch.cuda.empty_cache()
    return r


warnings.simplefilter("ignore")
timed(lambda: HipHubertModel(sd, cfg, device=dev, self_check=False))   # warm-up (library load, first launches)
out = {
    "plain build (self_check=False)": timed(lambda: HipHubertModel(sd, cfg, device=dev, self_check=False)),
    "build + self-check, healthy checkpoint (twin only)": timed(lambda: HipHubertModel(sd, cfg, device=dev, self_check=True)),
    "build + self-check, outlier checkpoint (every rung)": timed(lambda: HipHubertModel(sdo, cfg, device=dev, self_check=True)),
}
os.environ["MER_PLANE_CACHE"] = "0"
out["outlier checkpoint (every rung), planes NOT shared between the builds (round 5)"] = timed(lambda: HipHubertModel(sdo, cfg, device=dev, self_check=True))
os.environ["MER_PLANE_CACHE"] = "1"
os.environ["MER_SELF_CHECK_LADDER"] = "0"
out["outlier checkpoint, ladder off (twin only)"] = timed(lambda: HipHubertModel(sdo, cfg, device=dev, self_check=True))
print(json.dumps(out, indent=1, default=str))
