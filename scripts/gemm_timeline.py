#!/usr/bin/env python3
"""Per-CU timeline of one 256x256 GEMM launch from the kernel's s_memtime stamps + hardware CU id: for every CU the sequence of
(start, first slab ready, K loop done, end) of the workgroups it ran.  Answers: how long is the gap between a workgroup's
end and the next one's start on the same CU (dispatch + store drain), do the CUs run in lockstep, where does the tile time go.
    WARM=30 python scripts/gemm_timeline.py            (MER_OPTIONS=... to switch kernel options)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
WARM = int(os.environ.get("WARM", "30"))
for (name, M, N, K, act, passes, res) in [("clip Q|K p1", 100864, 1536, 768, None, 1, False), ("clip fc1 qgelu p1", 100864, 3072, 768, "quick_gelu", 1, False),
                                          ("clip out-proj mx", 100864, 768, 768, None, 4, True), ("clip fc2 p1", 100864, 768, 3072, None, 1, True)]:
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    a = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    ah, _ = ops.split16(a, "f16", lo=False); wh, wl = ops.split16(w, "f16")
    mx = ops.mx_pack(w.cpu() - wh.cpu().float()).to(dev) if passes == 4 else None
    wb = ops.w_block_pack(wh)
    bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if res else None
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(nblk * 21, dtype=torch.int64, device=dev)
    kw = dict(w_lo=None, w_mx=mx, w_hi_blk=wb, passes=passes, dtype="f16", tile=3, bias=bias, act=act, out16=not res, out32=res, residual=resid)
    for _ in range(WARM): ops.gemm16(ah, wh, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(WARM): ops.gemm16(ah, wh, **kw)
    e1.record()
    lib.mer_set_debug_buffer(buf.data_ptr())
    ops.gemm16(ah, wh, **kw); torch.cuda.synchronize()
    lib.mer_set_debug_buffer(None)
    wall_us = e0.elapsed_time(e1) / WARM * 1e3
    t = buf[:nblk * 4].view(nblk, 4).cpu().double()
    ids = buf[nblk * 20:nblk * 21].cpu()
    cu = ((ids >> 32) << 16) | (ids & 0xFF00)          # xcc | se, sh, cu
    t -= t[:, 0].min()
    pro, loop, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    span = t[:, 3].max().item()
    gaps, first_start, per_cu = [], [], {}
    for c in cu.unique().tolist():
        idx = (cu == c).nonzero().flatten()
        order = idx[t[idx, 0].argsort()]
        per_cu[c] = len(order)
        first_start.append(t[order[0], 0].item())
        gaps += (t[order[1:], 0] - t[order[:-1], 3]).tolist()
    gaps = torch.tensor(gaps)
    q = lambda x, p: x.quantile(p).item()  # noqa: E731
    # lockstep: spread of workgroup START phases, modulo the median tile period, late in the launch
    period = (pro + loop + epi).median().item() + (gaps.median().item() if len(gaps) else 0)
    late = t[:, 0] > 0.5 * span
    ph = (t[late, 0] % period) / period
    hist = torch.histc(ph.float(), bins=8, min=0, max=1).int().tolist()
    # does the hardware hand workgroup L to XCD L % 8 (what the kernel's tile map assumes), and do the column tiles of one A row tile
    # run on ONE XCD (sharing its L2) close together in time?
    xcc = (ids >> 32).long()
    L = torch.arange(nblk)
    tiles_n = (N + 255) // 256
    q8, r8 = nblk >> 3, nblk & 7
    xa, loc = L & 7, L >> 3
    swz = torch.where(xa < r8, xa * (q8 + 1), r8 * (q8 + 1) + (xa - r8) * q8) + loc       # the kernel's tile_of()
    tm = swz // tiles_n
    nx = [len(xcc[tm == r].unique()) for r in tm.unique().tolist()[::7]]   # (start times are not compared: s_memtime bases differ between XCDs)
    print(f"{name:18s} XCDs seen {len(xcc.unique())}; workgroups with xcc == L % 8: {(xcc == (L & 7)).float().mean():.3f}; "
          f"xcc == (L % 8 + c) % 8 for the best c: {max(((xcc == ((L + c) & 7)).float().mean().item(), c) for c in range(8))}; "
          f"XCDs per A row tile ({tiles_n} column tiles): {sum(nx) / len(nx):.2f}")
    print(f"{name:18s} wall {wall_us:.1f} us = {2.0 * M * N * K / wall_us / 1e6:.0f} TF | CUs seen {len(per_cu)} tiles/CU {min(per_cu.values())}-{max(per_cu.values())} | "
          f"cycles: prologue {pro.median():.0f} kloop {loop.median():.0f} epilogue {epi.median():.0f} (p10 {q(epi, .1):.0f} p90 {q(epi, .9):.0f}) | "
          f"gap end->next start on the CU: median {gaps.median():.0f} p10 {q(gaps, .1):.0f} p90 {q(gaps, .9):.0f} | span {span:.0f} | "
          f"busy = sum(tile)/CUs/span {((t[:, 3] - t[:, 0]).sum().item() / max(len(per_cu), 1) / span):.2f} | start-phase histogram (2nd half) {hist}")
