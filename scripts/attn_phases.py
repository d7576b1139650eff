#!/usr/bin/env python3
"""Per-workgroup phase timing of the single-pass attention kernel (s_memtime stamps): staging / compute cycles."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.lib()
for (B, T, H) in [(512, 197, 12), (64, 249, 12), (64, 64, 12)]:
    D = H * 64
    qkv = torch.randn(B * T, 3 * D, device=dev).half()
    buf = torch.zeros(B * H * 4, dtype=torch.int64, device=dev)
    ops.attention(qkv, B, T, H, 0.125); torch.cuda.synchronize()
    lib.mer_set_debug_buffer(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attention(qkv, B, T, H, 0.125); e1.record(); torch.cuda.synchronize()
    lib.mer_set_debug_buffer(None)
    t = buf.view(B * H, 4).cpu().double()
    st, cp = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
    print(f"B={B} T={T} H={H}: kernel {e0.elapsed_time(e1)*1e3:.0f} us; per workgroup median ticks: staging {st.median():.0f} compute {cp.median():.0f}; "
          f"sum(total)/(256 CUs * 2 wg) = {(st+cp).sum().item()/512:.0f} ticks")
