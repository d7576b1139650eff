import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.lib()
B, T, H = 1024, 64, 12
qkv = torch.randn(B * T, 3 * H * 64, device=dev).half()
for nkt, lds in [(0, 17920), (14, 61440), (18, 78848), (32, 139776)]:
    lib.mer_set_option(b"attn_force_nkt", nkt)
    ops.attention(qkv, B, T, H, 0.125); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.attention(qkv, B, T, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    print(f"NKT={nkt or 4:2d} LDS={lds:6d} B  -> {e0.elapsed_time(e1)/5*1e3:8.1f} us for {B*H} workgroups")
lib.mer_set_option(b"attn_force_nkt", 0)
