#!/usr/bin/env python3
"""HuBERT-base, 8 clips of 5 s (the bench's kernels): parity of precision presets against the CPU oracle, normal and heavy-tailed weights.
usage: conv_precision_ab.py mx mxc1 ..."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mertools_amd import synthetic as W
from mertools_amd.encoders import HipHubertModel
from oracle import encoders_ref as R
from util import rel_err
dev = torch.device("cuda:0")
cfg = W.hubert_config("base")
B = 8
wav = W.synth_audio(B, 80000, seed=4321)
for heavy in (False, True):
    sd = W.hubert_state_dict(cfg, 0)
    if heavy:
        sd = W.heavy_tailed(sd)
    with torch.no_grad():
        hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    utt = feat.mean(1)
    for prec in sys.argv[1:]:
        m = HipHubertModel(sd, cfg, device=dev, precision=prec)
        hsd, fr, pooled = m.forward_raw(wav.to(dev), hidden_states=True, frames=True, seg_start=[b * 249 for b in range(B)], seg_len=[249] * B)
        torch.cuda.synchronize()
        print(f"hubert-base B={B} heavy={heavy} [{prec}]: hs0={rel_err(hsd[0].cpu(), hs[0])[0]:.2e} hs12={rel_err(hsd[-1].cpu(), hs[-1])[0]:.2e} "
              f"frame={rel_err(fr.cpu().view(B, 249, 768), feat)[0]:.2e} utt={rel_err(pooled.cpu(), utt)[0]:.2e} "
              f"utt_worst_clip={max(rel_err(pooled[b].cpu(), utt[b])[0] for b in range(B)):.2e}", flush=True)
        del m
