"""Where does the activation-outlier HuBERT-base go wrong on the GPU?  Per layer and per channel class (the 3 x 12 outlier channels of
synthetic.ln_outliers vs the others): max |hs_gpu - hs_oracle| for a few presets.  GPU + CPU oracle (test infrastructure).
Run: python tests/studies/outlier_layers_gpu.py [preset ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import encoders_ref as R  # noqa: E402
from oracle import weights as W  # noqa: E402


def main(presets):
    from mertools_amd.encoders import HipHubertModel
    cfg = W.hubert_config("base")
    sd = W.ln_outliers(W.hubert_state_dict(cfg, 0))
    wav = W.synth_audio(2, 80000, seed=4321)
    hs = torch.stack(R.hubert_hidden_states(sd, vars(cfg), wav))            # [13, B, T, D]
    big = hs.abs().amax((1, 2))                                            # [13, D] per-layer channel maxima
    for prec in presets:
        m = HipHubertModel(sd, cfg, device="cuda:0", precision=prec)
        out, _, _ = m.forward_raw(wav.cuda(), hidden_states=True)
        torch.cuda.synchronize()
        out = out.cpu()
        print(f"[{prec}] layer: max|ref|  err(all)/max|ref|   worst channel (|ref| max there, abs err)   err on channels with max|ref| < 5 (abs)")
        for l in range(hs.shape[0]):
            d = (out[l] - hs[l]).abs().amax((0, 1))                        # per channel
            c = int(d.argmax())
            small = big[l] < 5
            print(f"   {l:2d}: {big[l].max():9.1f}  {d.max() / big[l].max():.2e}   ch {c:3d} ({big[l][c]:8.1f}, {d[c]:.3e})   {d[small].max():.3e}  (n outlier-ish channels: {int((~small).sum())})")
        del m


if __name__ == "__main__":
    main(sys.argv[1:] or ["accurate", "balanced"])
