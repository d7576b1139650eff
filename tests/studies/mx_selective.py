#!/usr/bin/env python3
"""CPU-only study: which GEMMs need the weight-residual correction?  Emulates f16(a)·f16(w) [+ bf8(a)·mxfp4(w_lo)] per linear
layer class and reports the UTT / frame error of CLIP-B/16 features and HuBERT-base last-4 features when the correction is
dropped for some classes (q/k/v/out/fc1/fc2 by weight shape + call order)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mertools_amd import synthetic as W
from oracle import encoders_ref as R
from mx_numerics import q_mx

DROP = set()
_lin = F.linear
state = {"i": 0}

def kind(w):
    n, k = w.shape
    if n == 4 * k: return "fc1"
    if k == 4 * n: return "fc2"
    if n == k:
        j = state["i"] % 4          # oracle order inside _mhsa: q, k, v, out
        state["i"] += 1
        return ("q", "k", "v", "out")[j]
    return "other"

def lin(x, w, b=None):
    kd = kind(w)
    a16 = x.half().float(); wh = w.half().float()
    out = _lin(a16, wh)
    if kd not in DROP:
        out = out + _lin(q_mx(a16, "bf8"), q_mx(w - wh, "fp4"))
    return out if b is None else out + b

def main():
    torch.manual_seed(0)
    cc = W.clip_config("base16"); csd = W.clip_state_dict(cc, 0); px = W.synth_frames(8)
    ccfg = dict(vars(cc.vision_config), projection_dim=cc.projection_dim)
    ref_c = R.clip_image_features(csd, ccfg, px)
    hc = W.hubert_config("base"); hsd = W.hubert_state_dict(hc, 0); wav = W.synth_audio(2, 32000)
    hs = R.hubert_hidden_states(hsd, vars(hc), wav); ref_h = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    R.F.linear = lin
    for drop in [set(), {"q", "k"}, {"q", "k", "v"}, {"fc1"}, {"fc2"}, {"out"}, {"q", "k", "fc1"}, {"q", "k", "v", "out", "fc1", "fc2"}]:
        DROP.clear(); DROP.update(drop)
        state["i"] = 0
        oc = R.clip_image_features(csd, ccfg, px)
        state["i"] = 0
        oh = torch.stack(R.hubert_hidden_states(hsd, vars(hc), wav))[[-4, -3, -2, -1]].sum(0)
        e = lambda o, r: ((o - r).abs().max() / r.abs().max()).item()
        print(f"no correction for {sorted(drop) or '-'}: CLIP utt {e(oc.mean(0), ref_c.mean(0)):.2e} frames {e(oc, ref_c):.2e} | "
              f"HuBERT utt {e(oh.mean(1), ref_h.mean(1)):.2e} frame {e(oh, ref_h):.2e}", flush=True)

if __name__ == "__main__":
    main()
