"""Confirms: HW k index of lane (i,g) byte j = 64*(j//16) + 16*g + j%16; E8M0 scale of (row i, k-block b) sits in byte 0 of
lane i+16*b's scale VGPR (same for B with column n)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mx_probe import Af, Bf, pack, run, kmap_split16  # noqa
g = torch.Generator().manual_seed(1)
ea = torch.randint(124, 131, (16, 4), generator=g)          # [row, block]
eb = torch.randint(124, 131, (16, 4), generator=g)          # [col, block]
a, b, sa, sb = pack(kmap_split16, 127, 127)
for l in range(64):
    sa[l] = int(ea[l & 15, l >> 4]); sb[l] = int(eb[l & 15, l >> 4])
D = run(a, b, sa, sb)
As = Af.clone(); Bs = Bf.clone()
for blk in range(4):
    As[:, 32 * blk:32 * blk + 32] *= (2.0 ** (ea[:, blk].float() - 127)).unsqueeze(1)
    Bs[32 * blk:32 * blk + 32, :] *= (2.0 ** (eb[:, blk].float() - 127)).unsqueeze(0)
ref = As @ Bs
print(f"random per-block scales, split16 layout: max|D-ref| = {(D - ref).abs().max().item():.4g} (ref max {ref.abs().max().item():.3g})")
