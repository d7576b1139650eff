#!/usr/bin/env python3
"""Which (row, 32-wide k-block) does the scale register of lane l scale?  Sets one lane's scale byte to 2.0 at a time."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mx_probe import A, B, Af, Bf, pack, run, kmap_contig, ref  # noqa
hits = {}
for which in ("a", "b"):
    for l0 in range(64):
        a, b, sa, sb = pack(kmap_contig, 127, 127)
        (sa if which == "a" else sb)[l0] = 128
        D = run(a, b, sa, sb)
        delta = D - ref
        found = []
        for idx in range(16):
            for blk in range(4):
                if which == "a":
                    cand = torch.zeros(16, 16); cand[idx] = Af[idx, 32 * blk:32 * blk + 32] @ Bf[32 * blk:32 * blk + 32, :]
                else:
                    cand = torch.zeros(16, 16); cand[:, idx] = Af[:, 32 * blk:32 * blk + 32] @ Bf[32 * blk:32 * blk + 32, idx]
                if (delta - cand).abs().max() < 1e-2 and cand.abs().max() > 0.1:
                    found.append((idx, blk))
        hits[(which, l0)] = found if found else ("none" if delta.abs().max() < 1e-3 else f"other(max {delta.abs().max():.2f})")
for which in ("a", "b"):
    print(which, "scale lane -> (row/col, kblock):", {l0: hits[(which, l0)] for l0 in range(64)})
# byte selection: put 2.0 in byte 1,2,3 of every lane's scale register (byte 0 = 1.0) with opsel 0
for byte in (1, 2):
    a, b, sa, sb = pack(kmap_contig, 127, 127)
    sa[:] = 127 | (128 << (8 * byte))
    D = run(a, b, sa, sb)
    print(f"scale 2.0 in byte {byte} (opsel 0): max|D-ref| = {(D - ref).abs().max().item():.3g}")
