#!/usr/bin/env python3
"""GEMM microbenchmark (GPU): the encoder GEMM shapes, every pass count, loader variants A/B in one process.
Loader variants: r = register-staged, d = LDS-DMA double-buffered, m = LDS-DMA multi-stage counted-vmcnt (default); X = 256x256 tile (else 128x128).
Prints algorithmic TFLOP/s (2*M*N*K / time) and checks each variant against torch on the same fp16 operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # (name, M, N, K)
    ("clip qkv", 100864, 2304, 768), ("clip out", 100864, 768, 768), ("clip fc1", 100864, 3072, 768), ("clip fc2", 100864, 768, 3072),
    ("hubert qkv", 15936, 2304, 768), ("hubert fc1", 15936, 3072, 768), ("hubert fc2", 15936, 768, 3072),
    ("hubert conv1", 64 * 7999, 512, 1536), ("roberta fc1", 4096, 3072, 768),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        ah, al = ops.split16(a, "f16")
        wh, wl = ops.split16(w, "f16")
        bias = torch.randn(N, device=dev)
        ref = None
        if M * N <= 100864 * 768:
            ref = ah.float() @ wh.float().T
        line = f"{name:13s} M={M:<7d} N={N:<5d} K={K:<5d}"
        for passes in (1, 2, 3):
            for glds, tile in ((0, 1), (2, 1), (1, 1), (1, 3)):
                lib.mer_set_option(b"gemm_glds", glds)
                kw = dict(a_lo=al if passes == 3 else None, w_lo=wl if passes >= 2 else None, passes=passes, dtype="f16", tile=tile)
                fn = lambda: ops.gemm16(ah, wh, bias=bias, act="gelu", out16=True, **kw)  # noqa: E731
                t = timeit(fn)
                line += f" | p{passes}{'rmd'[glds]}{'' if tile == 1 else 'X'} {2.0 * M * N * K / t / 1e12:6.1f}"
                if ref is not None and passes == 1:
                    c32, _, _ = ops.gemm16(ah, wh, out32=True, **kw)
                    err = ((c32 - ref).abs().max() / ref.abs().max()).item()
                    assert err < 1e-4, (name, passes, glds, err)
        print(line, flush=True)
    lib.mer_set_option(b"gemm_glds", 1)


if __name__ == "__main__":
    main()
