"""Which operand keeps `accurate` (3-pass GEMMs, hi + lo planes) at FRAME 4.6e-3 on the activation-outlier HuBERT-base
(profiles/r03_activation_outlier_stress.txt)?  CPU emulation in the oracle: fp64 reference; fp32 oracle; fp32 with the attention
operands (q, k, v, P) rounded to f16 as the kernels do; GEMM inputs as hi + lo f16 planes (22 bits) with and without that rounding.
Run: python tests/studies/outlier_attention.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import encoders_ref as R  # noqa: E402
from oracle import weights as W  # noqa: E402
from util import rel_err  # noqa: E402

torch.set_num_threads(8)
cfg = W.hubert_config("base")
sd = W.ln_outliers(W.hubert_state_dict(cfg, 0))
wav = W.synth_audio(2, 80000, seed=4321)


def feats(hs):
    f = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    return f, f.mean(1)


sd64 = {k: v.double() for k, v in sd.items()}
ref_f, ref_u = feats(R.hubert_hidden_states(sd64, vars(cfg), wav.double()))
f32_f, f32_u = feats(R.hubert_hidden_states(sd, vars(cfg), wav))
print(f"fp32 oracle vs fp64: frame={rel_err(f32_f, ref_f)[0]:.2e} utt={rel_err(f32_u, ref_u)[0]:.2e}")

orig_matmul, orig_softmax, orig_linear = torch.matmul, torch.softmax, F.linear


def h16(x):
    return x.half().float()


def planes22(x):
    hi = x.half().float()
    return hi + (x - hi).half().float()


def run(attn16, planes):
    def mm(a, b):
        if attn16 and a.dim() == 4:
            a, b = h16(a), h16(b)
        return orig_matmul(a, b)

    def sm(x, dim=-1):
        return orig_softmax(x, dim=dim)

    def lin(x, w, b=None):
        if planes:
            x, w = planes22(x), planes22(w)
        return orig_linear(x, w, b)
    torch.matmul, torch.softmax, F.linear = mm, sm, lin
    try:
        return feats(R.hubert_hidden_states(sd, vars(cfg), wav))
    finally:
        torch.matmul, torch.softmax, F.linear = orig_matmul, orig_softmax, orig_linear


for name, a16, pl in (("attention operands f16", True, False), ("GEMM inputs hi+lo", False, True), ("both (= accurate)", True, True)):
    f, u = run(a16, pl)
    print(f"{name:28s} vs fp64: frame={rel_err(f, ref_f)[0]:.2e} utt={rel_err(u, ref_u)[0]:.2e}   vs fp32 oracle: frame={rel_err(f, f32_f)[0]:.2e} utt={rel_err(u, f32_u)[0]:.2e}")


def flush16(x):     # f16 with subnormals flushed to zero (what an MFMA that ignores denormal inputs would see)
    h = x.half()
    return torch.where(h.abs() < 6.1035e-5, torch.zeros_like(h), h).float()


def planes22_ftz(x):
    hi = flush16(x)
    return hi + flush16(x - hi)


planes22 = planes22_ftz
f, u = run(True, True)
print(f"accurate, f16 subnormals flushed vs fp64: frame={rel_err(f, ref_f)[0]:.2e} utt={rel_err(u, ref_u)[0]:.2e}")


# ---- channel equalisation: one f16 plane per GEMM operand (the one-pass presets), with and without power-of-two per-channel scales
def run1(equalise):
    def lin(x, w, b=None):
        if equalise:
            ax = x.reshape(-1, x.shape[-1]).abs().amax(0).clamp_min(1e-20)
            aw = w.abs().amax(0).clamp_min(1e-20)
            s = torch.exp2(torch.round(0.5 * (torch.log2(ax) - torch.log2(aw))))        # a_k / s_k and w_k * s_k of equal size
            s = s / torch.exp2(torch.round(torch.log2(s.median())))
            return orig_linear(h16(x / s), h16(w * s), b)
        return orig_linear(h16(x), h16(w), b)

    def mm(a, b):
        return orig_matmul(h16(a), h16(b)) if a.dim() == 4 else orig_matmul(a, b)
    torch.matmul, F.linear = mm, lin
    try:
        return feats(R.hubert_hidden_states(sd, vars(cfg), wav))
    finally:
        torch.matmul, F.linear = orig_matmul, orig_linear


for eq in (False, True):
    f, u = run1(eq)
    print(f"one f16 plane per operand, equalised={eq}: frame={rel_err(f, ref_f)[0]:.2e} utt={rel_err(u, ref_u)[0]:.2e}")
sd_plain = W.hubert_state_dict(cfg, 0)
sd0, sd = sd, sd_plain
ref0_f, ref0_u = feats(R.hubert_hidden_states({k: v.double() for k, v in sd_plain.items()}, vars(cfg), wav.double()))
f, u = run1(False)
print(f"(the same network without outliers, one f16 plane: frame={rel_err(f, ref0_f)[0]:.2e} utt={rel_err(u, ref0_u)[0]:.2e})")
