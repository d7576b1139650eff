#!/usr/bin/env python3
"""CPU-only feasibility study for an MX-corrected GEMM: out = f16(a)·f16(w) + Qa(f16(a))·Qw(w − f16(w)) where Q is an OCP-MX
block-32 quantiser (fp4 e2m1 or fp8 e4m3 with a shared E8M0 scale).  Emulated in fp32 by patching F.linear / F.conv1d of
the oracle; prints the UTT / FRAME relative error of hubert-base last-4-sum features for each scheme."""
import os, sys, math, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mertools_amd import synthetic as syn
from oracle import encoders_ref as ref

FP4 = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6.])

def q_mx(x, fmt):
    """block-32 along the last dim, E8M0 scale = 2^(floor(log2(amax)) - emax_elem)"""
    if fmt is None: return x
    if fmt == "zero": return torch.zeros_like(x)
    if fmt == "bf8": return x.to(torch.float8_e5m2).float()                 # unscaled e5m2 (same exponent range as f16), RNE
    if fmt == "bf8t": return (x.half().view(torch.int16) & -256).view(torch.half).float()   # truncation = high byte of the f16
    K = x.shape[-1]; pad = (-K) % 32
    xp = F.pad(x, (0, pad)).reshape(*x.shape[:-1], -1, 32)
    amax = xp.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    emax = {"fp4": 2, "fp8": 8}[fmt]
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = xp / scale
    if fmt == "fp4":
        mag = y.abs().clamp(max=6.0)
        idx = (mag.unsqueeze(-1) - FP4).abs().argmin(-1)     # nearest (ties -> lower index; fine for a study)
        y = FP4[idx] * y.sign()
    else:
        y = y.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    return (y * scale).reshape(*x.shape[:-1], -1)[..., :K]

MODE = dict(qa=None, qw=None, passes=2)
_lin, _conv = F.linear, F.conv1d

def lin(x, w, b=None):
    a16 = x.half().float(); wh = w.half().float(); wl = w - wh
    out = _lin(a16, wh)
    if MODE["passes"] == 2:
        out = out + _lin(q_mx(a16, MODE["qa"]), q_mx(wl, MODE["qw"]))
    return out if b is None else out + b

def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if w.shape[1] == 1:                      # conv0 is fp32 VALU in the product
        return _conv(x, w, b, stride, padding, dilation, groups)
    a16 = x.half().float(); wh = w.half().float(); wl = w - wh
    out = _conv(a16, wh, None, stride, padding, dilation, groups)
    if MODE["passes"] == 2:                  # weights quantised along the flattened (k, cin) reduction; activations left f16
        Co, Ci, Kk = wl.shape
        wq = q_mx(wl.permute(0, 2, 1).reshape(Co, Kk * Ci), MODE["qw"]).reshape(Co, Kk, Ci).permute(0, 2, 1)
        out = out + _conv(a16, wq, None, stride, padding, dilation, groups)
    return out if b is None else out + b[None, :, None]

def feats(sd, cfg, wav):
    hs = ref.hubert_hidden_states(sd, cfg, wav)
    f = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)      # [B,T,D]
    return f, f.mean(1)

def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "base"
    torch.manual_seed(0)
    c = syn.hubert_config(size); cfg = vars(c); sd = syn.hubert_state_dict(c)
    wav = 0.1 * torch.randn(2, 32000); wav = (wav - wav.mean(1, keepdim=True)) / wav.std(1, keepdim=True)
    fr0, ut0 = feats(sd, cfg, wav)
    F.linear, F.conv1d = lin, conv
    ref.F.linear, ref.F.conv1d = lin, conv
    for name, m in [("1-pass f16", dict(passes=1)), ("2-pass f16 (now)", dict(passes=2, qa=None, qw=None)),
                    ("w_lo fp4, a bf8", dict(passes=2, qa="bf8", qw="fp4")), ("w_lo fp4, a bf8 trunc", dict(passes=2, qa="bf8t", qw="fp4")),
                    ("w_lo fp8, a bf8", dict(passes=2, qa="bf8", qw="fp8")),
                    ("w_lo fp8, a fp8", dict(passes=2, qa="fp8", qw="fp8")), ("w_lo fp4, a fp8", dict(passes=2, qa="fp8", qw="fp4")),
                    ("w_lo fp4, a fp4", dict(passes=2, qa="fp4", qw="fp4")), ("w_lo fp8, a fp4", dict(passes=2, qa="fp4", qw="fp8"))]:
        MODE.update(dict(qa=None, qw=None)); MODE.update(m)
        fr, ut = feats(sd, cfg, wav)
        print(f"{name:18s}: UTT rel {((ut - ut0).abs().max() / ut0.abs().max()).item():.3g}   FRAME rel {((fr - fr0).abs().max() / fr0.abs().max()).item():.3g}", flush=True)

if __name__ == "__main__":
    main()
