#!/usr/bin/env python3
"""Per-workgroup phase timing of the 256x256 GEMM (s_memtime stamps): prologue / K loop / epilogue cycles."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
SKIP = int(os.environ.get("SKIP", "0"))
lib.mer_set_option(b"gemm_dbg_skip", SKIP)
for (name, M, N, K, act, passes) in [("fc1 gelu p2", 100864, 3072, 768, "gelu", 2), ("qkv none p2", 100864, 2304, 768, None, 2),
                                     ("fc2 res p2", 100864, 768, 3072, None, 2), ("fc1 gelu p1", 100864, 3072, 768, "gelu", 1),
                                     ("fc1 qgelu p2", 100864, 3072, 768, "quick_gelu", 2), ("qkv none p1", 100864, 2304, 768, None, 1),
                                     ("qkv none mx", 100864, 2304, 768, None, 4), ("fc2 res mx", 100864, 768, 3072, None, 4),
                                     ("hub qkv p2", 15936, 2304, 768, None, 2), ("hub qkv mx", 15936, 2304, 768, None, 4)]:
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    ah, _ = ops.split16(a, "f16", lo=False); wh, wl = ops.split16(w, "f16")
    mx = ops.mx_pack(w.cpu() - wh.cpu().float()).to(dev) if passes == 4 else None
    bias = torch.randn(N, device=dev)
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    STAMP = int(os.environ.get("STAMP", "0"))
    buf = torch.zeros(nblk * 21, dtype=torch.int64, device=dev)   # 4 stamps + 16 phase counters + the CU id per workgroup
    kw = dict(w_lo=wl if passes >= 2 else None, w_mx=mx, passes=passes, dtype="f16", tile=3, bias=bias, act=act, out16=True)
    ops.gemm16(ah, wh, **kw); torch.cuda.synchronize()
    WARM = int(os.environ.get("WARM", "0"))   # > 0: stamp a launch that follows WARM back-to-back launches (sustained clocks)
    wall_us = float("nan")
    if WARM:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(WARM): ops.gemm16(ah, wh, **kw)
        e0.record()
        for _ in range(WARM): ops.gemm16(ah, wh, **kw)
        e1.record()
    lib.mer_set_debug_buffer(buf.data_ptr())
    lib.mer_set_option(b"gemm_stamp", STAMP)
    ops.gemm16(ah, wh, **kw); torch.cuda.synchronize()
    lib.mer_set_option(b"gemm_stamp", 0)
    lib.mer_set_debug_buffer(None)
    if WARM: wall_us = e0.elapsed_time(e1) / WARM * 1e3
    if STAMP:
        ph = buf[nblk * 4:].view(nblk, 2, 8).cpu().double()
        nsl = K // 32
        for gi in range(2):
            m = ph[:, gi].median(0).values
            n_mx = nsl // 4 if passes == 4 else 0
            n_other = nsl - n_mx
            line = f"   group {gi}: per plain slab  LOAD {m[0] / n_other:.0f} | wait {m[1] / n_other:.0f} | MATH {m[2] / n_other:.0f} | wait {m[3] / n_other:.0f}"
            if n_mx:
                line += f"   || per MX slab  LOAD {m[4] / n_mx:.0f} | wait {m[5] / n_mx:.0f} | MATH {m[6] / n_mx:.0f} | wait {m[7] / n_mx:.0f}"
            print(line)
    t = buf[:nblk * 4].view(nblk, 4).cpu().double()
    pro, loop, epi, tot = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2]), (t[:, 3] - t[:, 0])
    span = (t[:, 3].max() - t[:, 0].min()).item()
    if os.environ.get("WALL_ONLY"):
        print(f"{name:13s} wall {wall_us:.1f} us/launch = {2.0 * M * N * K / wall_us / 1e6:.0f} TF"); continue
    print(f"{name:13s} blocks={nblk} median cycles/ticks: prologue {pro.median():.0f}  kloop {loop.median():.0f}  epilogue {epi.median():.0f}  total {tot.median():.0f}"
          f" | wall {wall_us:.1f} us/launch -> eff. clock {tot.sum().item() / 256 / wall_us / 1e3:.2f} GHz (sum of tile cycles / 256 CUs / wall)"
          f" | per-slab {loop.median() / (K / 32):.0f} | kernel span {span:.0f} ticks; sum(total)/256/span = {tot.sum().item() / 256 / span:.2f}")
