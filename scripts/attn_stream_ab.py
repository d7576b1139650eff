#!/usr/bin/env python3
"""A/B of the streaming attention variants (QS sub-tiles per wave x register prefetch) on the VideoMAE-L shape (B = 64, T = 1568, H = 16)
and the single-pass kernel on CLIP's (B = 512, T = 197, H = 12): microseconds per launch and algorithmic TFLOP/s."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.lib()

def run(B, T, H, reps=20):
    qkv = (torch.randn(B * T, 3 * H * 64, device=dev) * 0.5).half()
    for _ in range(3): ops.attention(qkv, B, T, H, 0.125)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = ops.attention(qkv, B, T, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 4.0 * B * H * T * T * 64 / us / 1e6, out[0]

ref = None
for qs in (1, 2):
    for pf in (0, 1):
        lib.mer_set_option(b"attn_stream_qs", qs); lib.mer_set_option(b"attn_stream_pf", pf)
        us, tf, out = run(64, 1568, 16)
        if ref is None: ref = out.clone()
        print(json.dumps({"kernel": "attn_stream", "QS": qs, "PF": pf, "us": round(us, 1), "TFLOPs": round(tf, 1), "max_abs_diff_vs_first": float((out.float() - ref.float()).abs().max())}), flush=True)
lib.mer_set_option(b"attn_stream_qs", 2); lib.mer_set_option(b"attn_stream_pf", 0)
for w in (4, 8):
    lib.mer_set_option(b"attn_waves", w)
    us, tf, _ = run(512, 197, 12)
    print(json.dumps({"kernel": "attn_sp T=197", "waves": w, "us": round(us, 1), "TFLOPs": round(tf, 1)}), flush=True)
lib.mer_set_option(b"attn_waves", 8)
# the other single-pass shapes of the bench: HuBERT (T = 249, 16 key tiles) and RoBERTa (T = 64, 4-wave kernel)
for shape in ((64, 249, 12), (64, 64, 12)):
    us, tf, _ = run(*shape, reps=50)
    print(json.dumps({"kernel": "attn_sp B=%d T=%d H=%d" % shape, "us": round(us, 1), "TFLOPs": round(tf, 1)}), flush=True)
