"""Which operator of the `accurate` (3-pass) path loses the outlier channels of the activation-outlier HuBERT-base?
The first transformer block, operator by operator through the C ABI (mer_split16 / mer_gemm16 passes = 3 / mer_attention /
mer_layernorm), every intermediate against the same operator in fp64 on the CPU fed with the ORACLE's input of that operator — so an
error is charged to the operator that made it, not to its inputs.  GPU + CPU oracle (test infrastructure).
Run: python tests/studies/outlier_block_bisect_gpu.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import encoders_ref as R  # noqa: E402
from oracle import weights as W  # noqa: E402


def report(name, got, ref, big):
    d = (got.double().cpu() - ref).abs()
    per = d.amax(0)
    c = int(per.argmax())
    rb = ref.abs().amax(0)
    print(f"  {name:34s} max|ref| {float(ref.abs().max()):9.2f}  max err {float(d.max()):.3e} (rel {float(d.max() / ref.abs().max()):.2e})  worst column {c:4d} "
          f"(|ref| there {float(rb[c]):8.2f})  err on the 3 outlier columns {float(per[big].max()) if per.numel() > max(big) else float('nan'):.3e}  "
          f"err elsewhere {float(per[[i for i in range(per.numel()) if i not in big]].max()):.3e}")


def main():
    from mertools_amd import ops
    dev = torch.device("cuda:0")
    cfg = W.hubert_config("base")
    sd = W.ln_outliers(W.hubert_state_dict(cfg, 0))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # 8: the batch of test_activation_outliers_post_ln (M = 1992 rows: the large-M kernels)
    wav = W.synth_audio(B, 80000 if B > 1 else 32000, seed=4321)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    T, D = hs[0].shape[1:]
    x = hs[0].reshape(B * T, D).double()                    # [B T, D]: the block's input, from the oracle
    H = cfg.num_attention_heads
    big = [int(i) for i in hs[1].abs().amax((0, 1)).topk(3).indices]
    print(f"B = {B}, T = {T}, outlier columns {big}, |x| there {[round(float(x[:, i].abs().max()), 1) for i in big]}")
    p = "encoder.layers.0."
    g = lambda k: sd[p + k].double()
    wqkv = torch.cat([g("attention.q_proj.weight"), g("attention.k_proj.weight"), g("attention.v_proj.weight")])
    bqkv = torch.cat([g("attention.q_proj.bias"), g("attention.k_proj.bias"), g("attention.v_proj.bias")])

    def planes(t):          # fp32 device tensor -> hi, lo planes (the product's own split kernel)
        return ops.split16(t.float().to(dev), "f16", lo=True)

    def wplanes(w):
        hi, lo = ops.split16_host(w.float(), "f16")
        return hi.to(dev), lo.to(dev)

    def gemm3(a32, w, b, **kw):
        ah, al = planes(a32)
        wh, wl = wplanes(w)
        return ops.gemm16(ah, wh, a_lo=al, w_lo=wl, bias=b.float().to(dev), passes=3, **kw)

    # --- QKV
    ref_qkv = x @ wqkv.T + bqkv
    c32, _, _ = gemm3(x, wqkv, bqkv, out32=True)
    report("QKV (3 passes, fp32 out)", c32, ref_qkv, big)
    # --- attention on the oracle's q | k | v (rounded once to f16, as the product does)
    qkv16 = ref_qkv.float().half().to(dev).contiguous()
    ctx16, _ = ops.attention(qkv16, B, T, H, 0.125)
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2) for t in ref_qkv.split(D, 1)]
    ref_ctx = (torch.softmax(q @ k.transpose(2, 3) * 0.125, -1) @ v).transpose(1, 2).reshape(B * T, D)
    report("attention (f16 q | k | v)", ctx16, ref_ctx, big)
    ch, cl = ops.attention_f32(ref_qkv.float().to(dev).contiguous(), B, T, H, 0.125)
    report("attention_f32 (hi + lo planes)", ch.float() + cl.float(), ref_ctx, big)
    # --- attention output + residual
    wo, bo = g("attention.out_proj.weight"), g("attention.out_proj.bias")
    ref_t1 = ref_ctx @ wo.T + bo + x
    c32, _, _ = gemm3(ref_ctx, wo, bo, residual=x.float().to(dev), out32=True)
    report("out-proj + residual", c32, ref_t1, big)
    # --- LayerNorm 1
    g1, b1 = g("layer_norm.weight"), g("layer_norm.bias")
    ref_h1 = F.layer_norm(ref_t1, (D,), g1, b1, cfg.layer_norm_eps)
    o32, oh, ol = ops.layernorm(ref_t1.float().to(dev), g1.float().to(dev), b1.float().to(dev), cfg.layer_norm_eps, out32=True, out16=True, out16_lo=True)
    report("LayerNorm 1 (fp32 out)", o32, ref_h1, big)
    report("LayerNorm 1 (hi + lo planes)", oh.float() + ol.float(), ref_h1, big)
    # --- fc1 + GELU
    w1, bb1 = g("feed_forward.intermediate_dense.weight"), g("feed_forward.intermediate_dense.bias")
    ref_f = F.gelu(ref_h1 @ w1.T + bb1)
    _, fh, fl = gemm3(ref_h1, w1, bb1, act="gelu", out16=True, out16_lo=True)
    report("fc1 + GELU (hi + lo planes)", fh.float() + fl.float(), ref_f, [0, 1, 2])
    # --- fc2 + residual
    w2, bb2 = g("feed_forward.output_dense.weight"), g("feed_forward.output_dense.bias")
    ref_t2 = ref_f @ w2.T + bb2 + ref_h1
    c32, _, _ = gemm3(ref_f, w2, bb2, residual=ref_h1.float().to(dev), out32=True)
    report("fc2 + residual", c32, ref_t2, big)
    # --- LayerNorm 2
    g2, b2 = g("final_layer_norm.weight"), g("final_layer_norm.bias")
    ref_y = F.layer_norm(ref_t2, (D,), g2, b2, cfg.layer_norm_eps)
    o32, _, _ = ops.layernorm(ref_t2.float().to(dev), g2.float().to(dev), b2.float().to(dev), cfg.layer_norm_eps, out32=True)
    report("LayerNorm 2 (fp32 out)", o32, ref_y, big)
    report("  (fp64 chain vs the oracle's hs[1])", ref_y, hs[1].reshape(B * T, D).double(), big)
    # the same LayerNorm on an input that is off by one fp32 ulp in the outlier columns: how much an honest input error costs
    t2p = ref_t2.float().clone()
    for c in big:
        t2p[:, c] = torch.nextafter(t2p[:, c], torch.full_like(t2p[:, c], float("inf")))
    report("  (fp64 LN of input + 1 ulp(fp32))", F.layer_norm(t2p.double(), (D,), g2, b2, cfg.layer_norm_eps), ref_y, big)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
