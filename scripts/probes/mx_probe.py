#!/usr/bin/env python3
"""Finds the lane<->(row,k) layout and scale semantics of the MX-scaled 16x16x128 fp8 MFMA by brute force over a few
hypotheses.  Build: hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/probes/libmxprobe.so scripts/probes/mx_probe.hip"""
import ctypes, itertools, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmxprobe.so"))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
A = (torch.randn(16, 128, generator=g)).to(torch.float8_e4m3fn)       # [i, k]
B = (torch.randn(128, 16, generator=g)).to(torch.float8_e4m3fn)       # [k, n]
Af, Bf = A.float(), B.float()

def kmap_contig(g_, j): return g_ * 32 + j
def kmap_split16(g_, j): return (g_ * 16 + j) if j < 16 else (64 + g_ * 16 + (j - 16))
def kmap_split8(g_, j): return (j // 8) * 32 + g_ * 8 + (j % 8)
KM = {"contig32": kmap_contig, "split16": kmap_split16, "split8x4": kmap_split8}

def pack(kmap, sa_exp, sb_exp):
    a = torch.zeros(64, 32, dtype=torch.uint8); b = torch.zeros(64, 32, dtype=torch.uint8)
    Ab, Bb = A.view(torch.uint8), B.view(torch.uint8)
    for l in range(64):
        i, g_ = l & 15, l >> 4
        for j in range(32):
            k = kmap(g_, j)
            a[l, j] = Ab[i, k]; b[l, j] = Bb[k, i]
    sa = torch.full((64,), sa_exp, dtype=torch.int32); sb = torch.full((64,), sb_exp, dtype=torch.int32)
    return a, b, sa, sb

def run(a, b, sa, sb, fmt=0):
    out = torch.zeros(64, 4, device=dev)
    ad, bd, sad, sbd = a.to(dev), b.to(dev), sa.to(dev), sb.to(dev)
    rc = lib.run_probe(ctypes.c_void_p(ad.data_ptr()), ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(sad.data_ptr()),
                       ctypes.c_void_p(sbd.data_ptr()), ctypes.c_void_p(out.data_ptr()), fmt, None)
    torch.cuda.synchronize()
    assert rc == 0, rc
    D = torch.zeros(16, 16)
    o = out.cpu()
    for l in range(64):
        for r in range(4):
            D[(l >> 4) * 4 + r, l & 15] = o[l, r]
    return D

ref = Af @ Bf
for name, km in KM.items():
    D = run(*pack(km, 127, 127))
    print(f"layout {name:9s}: max|D-ref| = {(D - ref).abs().max().item():.4g}  (ref max {ref.abs().max().item():.3g})")
# scale semantics with the contiguous layout: scale byte = E8M0 exponent (127 -> 1.0)
for (ea, eb) in [(128, 127), (127, 129), (126, 126)]:
    D = run(*pack(kmap_contig, ea, eb))
    print(f"scales ({ea},{eb}): D/ref median ratio = {(D / ref).median().item():.4g}  expected {2.0 ** (ea - 127 + eb - 127):.4g}")
# per-lane (per 32-block) scale: only k-group 1 of A scaled by 2
a, b, sa, sb = pack(kmap_contig, 127, 127)
for l in range(64):
    if (l >> 4) == 1: sa[l] = 128
D = run(a, b, sa, sb)
ref2 = ref + Af[:, 32:64] @ Bf[32:64, :]
print(f"per-block scale (A k-group 1 x2): max|D-ref2| = {(D - ref2).abs().max().item():.4g}")
