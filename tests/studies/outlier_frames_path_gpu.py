"""Is the FRAME error of the activation-outlier HuBERT-base under `accurate` (3.5e-3) in the hidden states or in the
frames path (hidden-state ring + fused last-4 sum)?  Same batch as tests/test_encoders_gpu.py::test_activation_outliers_post_ln.
Run: python tests/studies/outlier_frames_path_gpu.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import encoders_ref as R  # noqa: E402
from oracle import weights as W  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def main():
    from mertools_amd.encoders import HipHubertModel
    cfg = W.hubert_config("base")
    sd = W.ln_outliers(W.hubert_state_dict(cfg, 0))
    B = 8
    x = W.synth_audio(B, 80000, seed=4321)
    hs = torch.stack(R.hubert_hidden_states(sd, vars(cfg), x))            # [13, B, T, D]
    feat = hs[[-4, -3, -2, -1]].sum(0)
    m = HipHubertModel(sd, cfg, device="cuda:0", precision="accurate", self_check=False)
    _, fr, pooled = m.forward_raw(x.cuda(), frames=True, seg_start=[b * 249 for b in range(B)], seg_len=[249] * B)
    out, _, _ = m.forward_raw(x.cuda(), hidden_states=True)
    torch.cuda.synchronize()
    fr = fr.cpu().view(B, 249, -1)
    out = out.cpu()
    gsum = out[[-4, -3, -2, -1]].sum(0)
    print(f"frames path vs oracle: {rel(fr, feat):.2e}   sum of the GPU's own last four hidden states vs oracle: {rel(gsum, feat):.2e}   frames path vs that sum: {rel(fr, gsum):.2e}")
    for l in range(13):
        d = (out[l] - hs[l]).abs()
        per_clip = [float(d[b].max() / hs[l][b].abs().max()) for b in range(B)]
        c = int(d.amax((0, 1)).argmax())
        print(f"  layer {l:2d}: max|ref| {float(hs[l].abs().max()):8.1f}  err/max {float(d.max() / hs[l].abs().max()):.2e}  worst channel {c} (|ref| {float(hs[l][..., c].abs().max()):.1f})  per clip: " + " ".join(f"{e:.1e}" for e in per_clip))
    d = (fr - feat).abs()
    c = int(d.amax((0, 1)).argmax())
    b, t = divmod(int(d[..., c].argmax()), 249)
    print(f"worst frame-feature element: clip {b} frame {t} channel {c}: ref {float(feat[b, t, c]):.4f} got {float(fr[b, t, c]):.4f}; per-layer ref / got there: " +
          "  ".join(f"{float(hs[l][b, t, c]):.3f}/{float(out[l][b, t, c]):.3f}" for l in (9, 10, 11, 12)))


if __name__ == "__main__":
    main()
