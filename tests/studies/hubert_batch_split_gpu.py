"""Which stage of HuBERT makes a clip's features depend on the batch size?  Hidden states of the same clips from a batch of 64 and from
batches of 8, layer by layer (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import weights as W  # noqa: E402


def main(prec):
    from mertools_amd.encoders import HipHubertModel
    cfg = W.hubert_config("base")
    m = HipHubertModel(W.hubert_state_dict(cfg, 0), cfg, device="cuda:0", precision=prec)
    x = W.synth_audio(64, 80000, seed=5005).cuda()
    big, _, _ = m.forward_raw(x, hidden_states=True)
    small, _, _ = m.forward_raw(x[:8].contiguous(), hidden_states=True)
    one, _, _ = m.forward_raw(x[:1].contiguous(), hidden_states=True)
    torch.cuda.synchronize()
    for l in range(big.shape[0]):
        d8 = (big[l, :8] - small[l]).abs().max().item()
        d1 = (big[l, :1] - one[l]).abs().max().item()
        print(f"[{prec}] layer {l:2d}: max|hs(64) - hs(8)| = {d8:.3e}   max|hs(64) - hs(1)| = {d1:.3e}   (max|hs| {big[l, :8].abs().max().item():.2f})")


if __name__ == "__main__":
    for p in sys.argv[1:] or ["mean"]:
        main(p)
