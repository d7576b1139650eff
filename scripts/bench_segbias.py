#!/usr/bin/env python3
"""What the per-segment bias table costs in each GEMM epilogue, and what its two helper kernels cost (mer_seg_mean16, the table GEMM)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def timed(fn, reps=30):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for name, M, N, K, seg, act, res in [("clip V", 100864, 768, 768, 197, None, False), ("clip out-proj", 100864, 768, 768, 197, None, True),
                                     ("hubert fc1", 15936, 3072, 768, 249, "gelu", False), ("hubert fc2", 15936, 768, 3072, 249, None, True),
                                     ("hubert V", 15936, 768, 768, 249, None, False)]:
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).half().to(dev)
    wb = ops.w_block_pack(w)
    bias = torch.randn(N, generator=g).to(dev)
    nseg = (M + seg - 1) // seg
    table = torch.randn(nseg, N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev) if res else None
    kw = dict(w_hi_blk=wb, act=act, residual=resid, out32=res, out16=not res, passes=1)
    t_vec = timed(lambda: ops.gemm16(a, w, bias=bias, **kw))
    t_tab = timed(lambda: ops.gemm16(a, w, bias=table, bias_seg_rows=seg, **kw))
    t_mean = timed(lambda: ops.seg_mean16(a, seg, stride=8))
    m16 = ops.seg_mean16(a, seg, stride=8)
    t_tg = timed(lambda: ops.gemm16(m16, w, bias=bias, out32=True, passes=1))
    fl = 2.0 * M * N * K
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "us_bias_vector": round(t_vec, 1), "TF_bias_vector": round(fl / t_vec / 1e6), "us_bias_table": round(t_tab, 1),
                      "TF_bias_table": round(fl / t_tab / 1e6), "us_seg_mean16": round(t_mean, 1), "us_table_gemm": round(t_tg, 1)}), flush=True)
