"""Which OPERATOR gives a row different bits in a batch of 64 x 249 rows than in a batch of 8 x 249?  (GPU only.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))


def main():
    from mertools_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(7)
    T, D, F = 249, 768, 3072
    Mb, Ms = 64 * T, 8 * T
    for name, N, K, act, o16, res in (("QKV", 2304, 768, None, True, False), ("out-proj", 768, 768, None, False, True),
                                      ("fc1", 3072, 768, "gelu", True, False), ("fc2", 768, 3072, None, False, True)):
        a = torch.randn(Mb, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.03)
        ah, _ = ops.split16(a, "f16", lo=False)
        wh, wl = ops.split16_host(w, "f16")
        wh, wl = wh.to(dev), wl.to(dev)
        hb, hp, hq = ops.w_block_pack(wh), ops.w_block_pack_p(wh, 0), ops.w_block_pack_p(wh, 1)
        lb = ops.w_block_pack(wl)
        bias = torch.randn(N, generator=g).to(dev)
        r = torch.randn(Mb, N, generator=g).to(dev) if res else None
        for passes in (1, 2):
            kw = dict(bias=bias, act=act, out16=o16, out32=not o16, passes=passes, w_lo=wl if passes == 2 else None, w_hi_blk=hb,
                      w_lo_blk=lb if passes == 2 else None, w_hi_blkp=hp, w_hi_blkq=hq)
            f32, f16, _ = ops.gemm16(ah, wh, residual=r, **kw)
            s32, s16, _ = ops.gemm16(ah[:Ms].contiguous(), wh, residual=r[:Ms].contiguous() if res else None, **kw)
            o32, o16_, _ = ops.gemm16(ah[:T].contiguous(), wh, residual=r[:T].contiguous() if res else None, **kw)
            torch.cuda.synchronize()
            full, small, one = (f16, s16, o16_) if o16 else (f32, s32, o32)
            print(f"gemm {name:8s} passes={passes}: rows of 8 clips equal in batch 64 and batch 8: {torch.equal(full[:Ms], small)}   batch 64 vs batch 1: {torch.equal(full[:T], one)}"
                  f"   max diff {(full[:Ms].float() - small.float()).abs().max().item():.3e}")
    x = torch.randn(Mb, D, generator=g).to(dev) * 3
    gm, bt = torch.randn(D, generator=g).to(dev), torch.randn(D, generator=g).to(dev)
    f = ops.layernorm(x, gm, bt, 1e-5, out16=True)
    s = ops.layernorm(x[:Ms].contiguous(), gm, bt, 1e-5, out16=True)
    o = ops.layernorm(x[:T].contiguous(), gm, bt, 1e-5, out16=True)
    torch.cuda.synchronize()
    print("layernorm fp32 equal 64 vs 8:", torch.equal(f[0][:Ms], s[0]), " 64 vs 1:", torch.equal(f[0][:T], o[0]), " 16-bit:", torch.equal(f[1][:Ms], s[1]), torch.equal(f[1][:T], o[1]))
    qkv = (torch.randn(Mb, 3 * D, generator=g)).to(dev).half()
    f, _ = ops.attention(qkv, 64, T, 12, 0.125)
    s, _ = ops.attention(qkv[:Ms].contiguous(), 8, T, 12, 0.125)
    o, _ = ops.attention(qkv[:T].contiguous(), 1, T, 12, 0.125)
    torch.cuda.synchronize()
    print("attention equal 64 vs 8:", torch.equal(f[:Ms], s), " 64 vs 1:", torch.equal(f[:T], o))


if __name__ == "__main__":
    main()
