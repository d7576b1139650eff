#!/usr/bin/env python3
"""1-pass / 2-pass (f16 lo plane) / MX-corrected (passes=4) GEMM on the 256x256 tile, encoder shapes; algorithmic TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import ops  # noqa: E402
from bench_gemm import SHAPES, timeit  # noqa: E402

dev = torch.device("cuda:0")


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, generator=g).half().to(dev)
        w = torch.randn(N, K, generator=g) * 0.05
        wh, wl = ops.split16_host(w, "f16")
        mx = ops.mx_pack(w - wh.float()).to(dev)
        wh, wl = wh.to(dev), wl.to(dev)
        bias = torch.randn(N, device=dev)
        line = f"{name:13s} M={M:<7d} N={N:<5d} K={K:<5d}"
        outs = {}
        for label, kw in (("p1", dict(passes=1)), ("p2", dict(passes=2, w_lo=wl)), ("mx", dict(passes=4, w_lo=wl, w_mx=mx))):
            fn = lambda: ops.gemm16(a, wh, bias=bias, act="gelu", out16=True, tile=3, **kw)  # noqa: E731
            t = timeit(fn)
            line += f" | {label} {2.0 * M * N * K / t / 1e12:6.1f} TF ({t * 1e6:7.1f} us)"
            if M * N <= 16000 * 3072:
                outs[label] = ops.gemm16(a, wh, out32=True, tile=3, **kw)[0]
        if outs:
            ref = a.double() @ w.to(dev).double().T
            line += " | err " + " ".join(f"{k} {((v.double() - ref).abs().max() / ref.abs().max()).item():.1e}" for k, v in outs.items())
        print(line, flush=True)


if __name__ == "__main__":
    main()
