#!/usr/bin/env python3
"""CPU-only study: the batch-mean weight-residual correction on NON-STATIONARY audio (0.3 s loud / 0.3 s `quiet` x as loud), HuBERT.
Columns: no correction | exact second pass | batch-mean bias everywhere ("mean_all") | conv stack exact + bias behind LayerNorms (what
"mean" ships, with MX in place of exact) | conv stack with the bias scaled per row by the row's projection on the mean token
(alpha_r = <a_r, m> / <m, m>: a candidate for a one-pass, scale-equivariant conv-stack correction) | the same with the mean of the
NORMALISED rows as the token | two tokens (rows above / below a quarter of the RMS row norm), each row projected on their span.
UTT / FRAME relative errors against the fp32 oracle.  usage: mean_correction_dynamics.py [tiny|base]"""
import os, sys, torch, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, HERE)
import mean_correction as MC
from mertools_amd import synthetic as W
from oracle import encoders_ref as R
from mertools_amd.extract import audio
size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = W.hubert_config(size); sd = W.hubert_state_dict(cfg, 1)
rng = np.random.RandomState(0)
L = 32000
def clips(quiet):
    env = np.where((np.arange(L) // 4800) % 2 == 0, 1.0, quiet)
    return torch.cat([audio.wav2vec2_normalize(np.round(np.clip(rng.randn(L) * 0.1 * env, -1, 1 - 1/32768) * 32768) / 32768) for _ in range(2)], 0)
with torch.no_grad():
    for quiet in (1.0, 0.1, 0.01, 0.0):
        iv = clips(quiet)
        R.F.linear, R.F.conv1d = MC._lin, MC._conv
        ref = torch.stack(R.hubert_hidden_states(sd, vars(cfg), iv))[[-4, -3, -2, -1]].sum(0)
        row = []
        for name, cm, lm in (("none", "none", "none"), ("exact", "exact", "exact"), ("gmean all", "gmean", "gmean"), ("conv exact+gmean", "exact", "gmean"), ("conv smean+gmean", "smean", "gmean"), ("conv nmean+gmean", "nmean", "gmean"), ("conv smean2+gmean", "smean2", "gmean")):
            def lin2(x, w, b=None, lm=lm):
                MC.MODE["corr"] = lm; return MC.lin(x, w, b)
            def conv2(x, w, b=None, stride=1, padding=0, dilation=1, groups=1, cm=cm):
                MC.MODE["corr"] = cm; return MC.conv(x, w, b, stride, padding, dilation, groups)
            R.F.linear, R.F.conv1d = lin2, conv2
            f = torch.stack(R.hubert_hidden_states(sd, vars(cfg), iv))[[-4, -3, -2, -1]].sum(0)
            row.append(f"{name}: {MC.rel(f.mean(1), ref.mean(1)):.1e}/{MC.rel(f, ref):.1e}")
        print(f"quiet={quiet:<5}", " | ".join(row), flush=True)
R.F.linear, R.F.conv1d = MC._lin, MC._conv
