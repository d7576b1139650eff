#!/usr/bin/env python3
"""CPU-only study: can the weight-rounding correction be applied to the per-clip MEAN activation only?
out = f16(a) f16(w)^T + corr, with  corr = f16(a) w_lo^T (the 2-pass / MX schemes: +50-100 % MFMA work)  or
corr = mean_t(f16(a)) w_lo^T per clip (one GEMV per clip and GEMM: < 1 % extra work, added like a per-clip bias).
The rounding error of the weights is the same perturbation for every token, so what survives the utterance mean is
mean_t(a) w_lo^T exactly; what is left per token is (a_t - mean) w_lo^T.  Prints UTT / FRAME errors of HuBERT-base, CLIP-B/16
and RoBERTa-base features for: one pass, exact 2-pass, mean-corrected one pass (all GEMMs), mean-corrected with Q/K uncorrected."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mertools_amd import synthetic as W
from oracle import encoders_ref as R

MODE = {"corr": "none"}
CAL = {"rec": None, "i": 0}   # "cal" mode: per-call-site mean tokens recorded on a calibration batch, replayed on the test batch
_lin, _conv = F.linear, F.conv1d


def lin(x, w, b=None):
    a16 = x.half().float(); wh = w.half().float(); wl = w - wh
    out = _lin(a16, wh)
    if MODE["corr"] == "exact":
        out = out + _lin(a16, wl)
    elif MODE["corr"] == "mean":
        m = a16.mean(dim=-2, keepdim=True) if a16.dim() >= 3 else a16      # [B, 1, K]: the clip's (frame's) mean token
        out = out + _lin(m, wl)
    elif MODE["corr"] == "gmean":                                           # ONE mean token for the whole batch (all clips, all tokens)
        m = a16.reshape(-1, a16.shape[-1]).mean(dim=0, keepdim=True)
        out = out + _lin(m, wl)
    elif MODE["corr"] == "smean":                                           # batch mean token, scaled per row by the row's projection on it
        m = a16.reshape(-1, a16.shape[-1]).mean(dim=0, keepdim=True)
        alpha = (a16 * m).sum(-1, keepdim=True) / (m * m).sum().clamp_min(1e-30)
        out = out + alpha * _lin(m, wl)
    elif MODE["corr"] == "calrec":
        CAL["rec"].append(a16.reshape(-1, a16.shape[-1]).mean(dim=0, keepdim=True))
    elif MODE["corr"] == "cal":
        m = CAL["rec"][CAL["i"]]; CAL["i"] += 1
        out = out + _lin(m, wl)
    elif MODE["corr"].startswith("sub"):                                    # mean over every s-th token only
        st = int(MODE["corr"][3:])
        m = a16[..., ::st, :].mean(dim=-2, keepdim=True) if a16.dim() >= 3 else a16
        out = out + _lin(m, wl)
    return out if b is None else out + b


def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if w.shape[1] == 1 or groups > 1:        # conv0 is fp32 VALU in the product; the grouped positional conv keeps its 2-pass form
        if groups > 1:
            a16 = x.half().float()
            return _conv(a16, w, b, stride, padding, dilation, groups)
        return _conv(x, w, b, stride, padding, dilation, groups)
    a16 = x.half().float(); wh = w.half().float(); wl = w - wh
    out = _conv(a16, wh, None, stride, padding, dilation, groups)
    if MODE["corr"] == "exact":
        out = out + _conv(a16, wl, None, stride, padding, dilation, groups)
    elif MODE["corr"] == "gmean":
        full = _conv(a16, wl, None, stride, padding, dilation, groups)
        out = out + full.mean(dim=(0, 2), keepdim=True)
    elif MODE["corr"] in ("smean2", "nmean"):
        k = w.shape[-1]
        cols = a16.unfold(2, k, stride)
        cols = cols.permute(0, 2, 1, 3).reshape(a16.shape[0], cols.shape[2], -1)
        flat = cols.reshape(-1, cols.shape[-1])
        wl2 = wl.reshape(wl.shape[0], -1)
        if MODE["corr"] == "nmean":
            # mean of the NORMALISED rows (direction only), scaled back per row by the row's own norm-projection: one token, scale-free
            nrm = flat.norm(dim=1, keepdim=True).clamp_min(1e-30)
            u = (flat / nrm).mean(0)
            alpha = (flat @ u) / (u @ u).clamp_min(1e-30)
            corr = alpha[:, None] * (wl2 @ u)[None, :]
        else:
            # two mean tokens: rows above / below the batch's RMS row norm; each row projected on the span of both (least squares)
            nrm = flat.norm(dim=1)
            thr = nrm.pow(2).mean().sqrt() * 0.25
            hi, lo = flat[nrm >= thr], flat[nrm < thr]
            m1 = hi.mean(0)
            m2 = lo.mean(0) if len(lo) else torch.zeros_like(m1)
            Mx = torch.stack([m1, m2], 1)                                   # [K, 2]
            G = Mx.T @ Mx + 1e-12 * torch.eye(2)
            coef = torch.linalg.solve(G, Mx.T @ flat.T).T                   # [rows, 2]
            corr = coef @ (wl2 @ Mx).T                                      # [rows, Co]
        corr = corr.reshape(cols.shape[0], cols.shape[1], -1).permute(0, 2, 1)
        out = out + corr
    elif MODE["corr"] == "smean":
        # im2col rows r = (b, t): m = mean row, alpha_r = <a_r, m> / <m, m>, correction alpha_r * (m w_lo^T)
        k = w.shape[-1]
        cols = a16.unfold(2, k, stride)                                  # [B, Ci, T, k]
        cols = cols.permute(0, 2, 1, 3).reshape(a16.shape[0], cols.shape[2], -1)   # [B, T, Ci*k]  (ci-major, like w.reshape(Co, -1))
        m = cols.reshape(-1, cols.shape[-1]).mean(0)                      # [Ci*k]
        alpha = (cols @ m) / (m @ m).clamp_min(1e-30)                     # [B, T]
        c = wl.reshape(wl.shape[0], -1) @ m                               # [Co]
        out = out + alpha[:, None, :] * c[None, :, None]
    elif MODE["corr"] == "calrec":
        CAL["rec"].append(_conv(a16, wl, None, stride, padding, dilation, groups).mean(dim=(0, 2), keepdim=True))
    elif MODE["corr"] == "cal":
        m = CAL["rec"][CAL["i"]]; CAL["i"] += 1
        out = out + m
    elif MODE["corr"] == "mean" or MODE["corr"].startswith("sub"):
        # mean over the output positions of the im2col rows == conv of the correction evaluated on the mean window
        full = _conv(a16, wl, None, stride, padding, dilation, groups)       # [B, Co, T]
        st = int(MODE["corr"][3:]) if MODE["corr"].startswith("sub") else 1
        out = out + full[..., ::st].mean(dim=-1, keepdim=True)
    return out if b is None else out + b[None, :, None]


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def main():
    torch.manual_seed(0)
    hc = W.hubert_config("base"); hsd = W.hubert_state_dict(hc, 0); wav = W.synth_audio(2, 48000)
    cc = W.clip_config("base16"); csd = W.clip_state_dict(cc, 0); px = W.synth_frames(8)
    ccfg = dict(vars(cc.vision_config), projection_dim=cc.projection_dim)
    bc = W.bert_config("roberta-base"); bsd = W.bert_state_dict(bc, 0); ids = W.synth_tokens(4)
    heavy = "--heavy" in sys.argv
    if heavy:
        hsd, csd, bsd = W.heavy_tailed(hsd), W.heavy_tailed(csd), W.heavy_tailed(bsd)
    if "--outliers" in sys.argv:   # 3 LayerNorm-gamma channels x 30-100 (activation outliers)
        hsd, csd, bsd = W.ln_outliers(hsd), W.ln_outliers(csd), W.ln_outliers(bsd)
    def run():
        h = torch.stack(R.hubert_hidden_states(hsd, vars(hc), wav))[[-4, -3, -2, -1]].sum(0)
        c = R.clip_image_features(csd, ccfg, px)
        t = torch.stack(R.bert_hidden_states(bsd, dict(vars(bc), roberta=True), ids, torch.ones_like(ids)))[[-4, -3, -2, -1]].sum(0)[:, 1:-1]
        return h, c, t
    with torch.no_grad():
        h0, c0, t0 = run()
        R.F.linear, R.F.conv1d = lin, conv
        calkind = next((a[6:] for a in sys.argv[1:] if a.startswith("--cal=")), None)
        if calkind:   # record the per-call-site means on a calibration batch
            if calkind == "seed":      # same distribution, other samples
                cw, cp, ci = W.synth_audio(2, 48000, seed=99), W.synth_frames(8, seed=98), W.synth_tokens(4, seed=97)
            elif calkind == "odd":     # a deliberately different distribution: a 440 Hz tone, flat grey frames, one repeated token
                tt = torch.arange(48000) / 16000.0
                cw = torch.sin(2 * 3.14159265 * 440 * tt)[None].repeat(2, 1); cw = (cw - cw.mean(1, keepdim=True)) / cw.std(1, keepdim=True)
                cp = torch.zeros(8, 3, 224, 224); ci = torch.full((4, 64), 1000, dtype=torch.long); ci[:, 0] = 0; ci[:, -1] = 2
            CAL["rec"] = []; MODE["corr"] = "calrec"
            torch.stack(R.hubert_hidden_states(hsd, vars(hc), cw)); R.clip_image_features(csd, ccfg, cp)
            R.bert_hidden_states(bsd, dict(vars(bc), roberta=True), ci, torch.ones_like(ci))
        for mode in (sys.argv[1:] and [a for a in sys.argv[1:] if not a.startswith("--")] or ["none", "exact", "mean"]):
            MODE["corr"] = mode
            CAL["i"] = 0
            h, c, t = run()
            print(f"{'heavy ' if heavy else ''}corr={mode:6s}: HuBERT utt {rel(h.mean(1), h0.mean(1)):.2e} frame {rel(h, h0):.2e} | "
                  f"CLIP utt {rel(c.mean(0), c0.mean(0)):.2e} frames {rel(c, c0):.2e} | RoBERTa utt {rel(t.mean(1), t0.mean(1)):.2e} frame {rel(t, t0):.2e}", flush=True)


if __name__ == "__main__":
    main()
