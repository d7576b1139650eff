"""CPU only (oracle = test infrastructure): how ill-conditioned is the activation-outlier HuBERT-base (synthetic.ln_outliers)?  The
first transformer block in fp64, fed with the oracle's hs[0] and with the same tensor perturbed at the level of ONE fp32 rounding
(relative 6e-8) and at the level the GPU's `accurate` front end reaches (3e-6 of the row maximum, 1e-5 absolute on the ordinary
channels): the change of hs[1] on the outlier channel, in fp64, with no kernel involved.
Run: python tests/studies/outlier_conditioning.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import encoders_ref as R  # noqa: E402
from oracle import weights as W  # noqa: E402


def block(sd, cfg, x, B, T, layer=0):
    D, H = x.shape[1], cfg.num_attention_heads
    p = f"encoder.layers.{layer}."
    g = lambda k: sd[p + k].double()
    wqkv = torch.cat([g("attention.q_proj.weight"), g("attention.k_proj.weight"), g("attention.v_proj.weight")])
    bqkv = torch.cat([g("attention.q_proj.bias"), g("attention.k_proj.bias"), g("attention.v_proj.bias")])
    qkv = x @ wqkv.T + bqkv
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2) for t in qkv.split(D, 1)]
    ctx = (torch.softmax(q @ k.transpose(2, 3) * 0.125, -1) @ v).transpose(1, 2).reshape(B * T, D)
    t1 = ctx @ g("attention.out_proj.weight").T + g("attention.out_proj.bias") + x
    h1 = F.layer_norm(t1, (D,), g("layer_norm.weight"), g("layer_norm.bias"), cfg.layer_norm_eps)
    f = F.gelu(h1 @ g("feed_forward.intermediate_dense.weight").T + g("feed_forward.intermediate_dense.bias"))
    t2 = f @ g("feed_forward.output_dense.weight").T + g("feed_forward.output_dense.bias") + h1
    return F.layer_norm(t2, (D,), g("final_layer_norm.weight"), g("final_layer_norm.bias"), cfg.layer_norm_eps)


def main():
    cfg = W.hubert_config("base")
    base = W.hubert_state_dict(cfg, 0)
    B = 2
    wav = W.synth_audio(B, 80000, seed=4321)
    for name, sd in (("outlier checkpoint (ln_outliers)", W.ln_outliers(base)), ("plain checkpoint", base)):
        hs = R.hubert_hidden_states(sd, vars(cfg), wav)
        T, D = hs[0].shape[1:]
        x = hs[0].reshape(B * T, D).double()
        y = block(sd, cfg, x, B, T)
        a = int(y.abs().amax(0).argmax())
        print(f"{name}: max|hs0| {float(x.abs().max()):.1f}, max|hs1| {float(y.abs().max()):.1f} (channel {a})")
        g = torch.Generator().manual_seed(1)
        for label, rel_big, abs_small in (("one fp32 rounding of hs[0]", 6e-8, None), ("the accurate front end's error (3e-6 of the row max, 1e-5 abs elsewhere)", 3e-6, 1e-5)):
            noise = torch.randn(x.shape, generator=g, dtype=torch.float64)
            if abs_small is None:
                dx = x.abs() * rel_big * noise
            else:
                dx = torch.where(x.abs() > 5, x.abs() * rel_big, torch.full_like(x, abs_small)) * noise
            dy = (block(sd, cfg, x + dx, B, T) - y).abs()
            print(f"   perturbation = {label}: max|dx| {float(dx.abs().max()):.2e} -> max|dy| {float(dy.max()):.3e} on channel {int(dy.amax(0).argmax())} "
                  f"(= {float(dy.max() / y.abs().max()):.1e} of max|hs1|); elsewhere (|y| < 5) {float(dy[:, y.abs().amax(0) < 5].max()):.2e}")


def frame_sensitivity(sd, cfg, wav, trials=2, seed=2):
    """Per clip: how far the FRAME feature (sum of the last four hidden states, relative to its batch maximum) moves, in fp64 and with
    no kernel involved, when hs[0] is perturbed by noise of the size of the `accurate` front end's error (3e-6 of the large channels,
    1e-5 absolute elsewhere) — the network's own conditioning, which any arithmetic's rounding is multiplied by.  Max over `trials`
    noise draws.  Used by tests/test_encoders_gpu.py::test_activation_outliers_post_ln to bound the HIP path's error per clip."""
    hs_all = R.hubert_hidden_states(sd, vars(cfg), wav)
    hs0 = hs_all[0]
    B, T, D = hs0.shape
    x = hs0.reshape(B * T, D).double()
    g = torch.Generator().manual_seed(seed)

    def run(x0):
        hs, h = [], x0
        for l in range(cfg.num_hidden_layers):
            h = block(sd, cfg, h, B, T, l)
            hs.append(h)
        return torch.stack(hs[-4:]).sum(0).view(B, T, D)
    ref = run(x)
    sens = [0.0] * B
    for _ in range(trials):
        noise = torch.randn(x.shape, generator=g, dtype=torch.float64)
        dx = torch.where(x.abs() > 5, x.abs() * 3e-6, torch.full_like(x, 1e-5)) * noise
        got = run(x + dx)
        for b in range(B):
            sens[b] = max(sens[b], float((got[b] - ref[b]).abs().max() / ref.abs().max()))
    return sens


def deep():
    """The same question through all 12 blocks, on the batch of test_activation_outliers_post_ln: fp64 from the oracle's hs[0] and from
    hs[0] + noise of the size of the accurate front end's error -> the FRAME feature (sum of the last four hidden states), per clip."""
    cfg = W.hubert_config("base")
    sd = W.ln_outliers(W.hubert_state_dict(cfg, 0))
    B = 8
    wav = W.synth_audio(B, 80000, seed=4321)
    hs_all = R.hubert_hidden_states(sd, vars(cfg), wav)
    hs0 = hs_all[0]
    T, D = hs0.shape[1:]
    x = hs0.reshape(B * T, D).double()
    g = torch.Generator().manual_seed(2)

    def run(x0):
        hs, h = [], x0
        for l in range(cfg.num_hidden_layers):
            h = block(sd, cfg, h, B, T, l)
            hs.append(h)
        return torch.stack(hs[-4:]).sum(0).view(B, T, D)
    ref = run(x)
    orc = torch.stack(hs_all)[[-4, -3, -2, -1]].sum(0).double()
    per = [float((orc[b] - ref[b]).abs().max() / ref.abs().max()) for b in range(B)]
    print("the fp32 ORACLE's own FRAME feature against the fp64 blocks from the same hs[0] (what fp32 rounding alone does), per clip: " + " ".join(f"{e:.1e}" for e in per))
    for trial in range(2):
        noise = torch.randn(x.shape, generator=g, dtype=torch.float64)
        dx = torch.where(x.abs() > 5, x.abs() * 3e-6, torch.full_like(x, 1e-5)) * noise
        got = run(x + dx)
        per = [float((got[b] - ref[b]).abs().max() / ref.abs().max()) for b in range(B)]
        print(f"12 blocks in fp64, hs[0] perturbed by max {float(dx.abs().max()):.1e}: FRAME feature error / max|ref| per clip: " + " ".join(f"{e:.1e}" for e in per))


if __name__ == "__main__":
    main()
    deep()
