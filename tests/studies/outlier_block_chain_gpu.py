"""Follow-up of outlier_block_bisect_gpu.py: the FIRST block of the activation-outlier HuBERT-base under `accurate`, three ways on the
same input (the GPU model's own hs[0]): the model's hs[1]; the same operators chained through the C ABI from Python; fp64 on the CPU.
Tells whether the model's 0.1-0.3 error on the outlier channel of hs[1] is made by an operator (then the chain shows it too) or by what
the encoder does around the operators (weight preparation, planes, workspace).  Run: python tests/studies/outlier_block_chain_gpu.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import weights as W  # noqa: E402


def main():
    from mertools_amd import ops
    from mertools_amd.encoders import HipHubertModel
    dev = torch.device("cuda:0")
    cfg = W.hubert_config("base")
    sd = W.ln_outliers(W.hubert_state_dict(cfg, 0))
    B = 8
    wav = W.synth_audio(B, 80000, seed=4321)
    m = HipHubertModel(sd, cfg, device=dev, precision="accurate", self_check=False)
    out, _, _ = m.forward_raw(wav.to(dev), hidden_states=True)
    torch.cuda.synchronize()
    T, D = out.shape[2:]
    H = cfg.num_attention_heads
    x32 = out[0].reshape(B * T, D).contiguous()                 # the model's own block input (device fp32)
    x = x32.double().cpu()
    model_y = out[1].reshape(B * T, D).double().cpu()
    p = "encoder.layers.0."
    g = lambda k: sd[p + k].double()
    wqkv = torch.cat([g("attention.q_proj.weight"), g("attention.k_proj.weight"), g("attention.v_proj.weight")])
    bqkv = torch.cat([g("attention.q_proj.bias"), g("attention.k_proj.bias"), g("attention.v_proj.bias")])
    # ---- fp64 on the CPU
    qkv = x @ wqkv.T + bqkv
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2) for t in qkv.split(D, 1)]
    ctx = (torch.softmax(q @ k.transpose(2, 3) * 0.125, -1) @ v).transpose(1, 2).reshape(B * T, D)
    t1 = ctx @ g("attention.out_proj.weight").T + g("attention.out_proj.bias") + x
    h1 = F.layer_norm(t1, (D,), g("layer_norm.weight"), g("layer_norm.bias"), cfg.layer_norm_eps)
    f = F.gelu(h1 @ g("feed_forward.intermediate_dense.weight").T + g("feed_forward.intermediate_dense.bias"))
    t2 = f @ g("feed_forward.output_dense.weight").T + g("feed_forward.output_dense.bias") + h1
    y64 = F.layer_norm(t2, (D,), g("final_layer_norm.weight"), g("final_layer_norm.bias"), cfg.layer_norm_eps)
    big = [int(i) for i in y64.abs().amax(0).topk(3).indices]

    # ---- the same operators through the C ABI, chained (each one fed with the previous one's device output)
    def wplanes(w):
        hi, lo = ops.split16_host(w.float(), "f16")
        return hi.to(dev), lo.to(dev)

    def gemm3(ah, al, w, b, **kw):
        wh, wl = wplanes(w)
        return ops.gemm16(ah, wh, a_lo=al, w_lo=wl, bias=b.float().to(dev), passes=3, **kw)
    xh, xl = ops.split16(x32, "f16", lo=True)
    qkv32, _, _ = gemm3(xh, xl, wqkv, bqkv, out32=True)
    ch, cl = ops.attention_f32(qkv32, B, T, H, 0.125)
    t1d, _, _ = gemm3(ch, cl, g("attention.out_proj.weight"), g("attention.out_proj.bias"), residual=x32, out32=True)
    h1d, hh, hl = ops.layernorm(t1d, g("layer_norm.weight").float().to(dev), g("layer_norm.bias").float().to(dev), cfg.layer_norm_eps, out32=True, out16=True, out16_lo=True)
    _, fh, fl = gemm3(hh, hl, g("feed_forward.intermediate_dense.weight"), g("feed_forward.intermediate_dense.bias"), act="gelu", out16=True, out16_lo=True)
    t2d, _, _ = gemm3(fh, fl, g("feed_forward.output_dense.weight"), g("feed_forward.output_dense.bias"), residual=h1d, out32=True)
    yd, _, _ = ops.layernorm(t2d, g("final_layer_norm.weight").float().to(dev), g("final_layer_norm.bias").float().to(dev), cfg.layer_norm_eps, out32=True)
    torch.cuda.synchronize()

    def line(name, a, b):
        d = (a.double().cpu() - b.double().cpu()).abs().amax(0)
        rest = [i for i in range(D) if i not in big]
        print(f"  {name:46s} outlier columns {big}: " + " ".join(f"{float(d[c]):.3e}" for c in big) + f"   elsewhere {float(d[rest].max()):.3e}")
    print(f"max|y| on the outlier columns: {[round(float(y64[:, c].abs().max()), 1) for c in big]}")
    line("model hs[1]      vs fp64 chain from its hs[0]", model_y, y64)
    line("C-ABI chain      vs fp64 chain from its hs[0]", yd, y64)
    line("model hs[1]      vs C-ABI chain", model_y, yd)
    for name, a, b in (("t1 (out-proj + residual)", t1d, t1), ("h1 (LayerNorm 1)", h1d, h1), ("t2 (fc2 + residual)", t2d, t2)):
        line("  chain " + name + " vs fp64", a, b)


if __name__ == "__main__":
    main()
