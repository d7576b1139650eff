"""Where does the FRAME error of the HF-initialised data2vec-audio module come from?  Per hidden state, per preset."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, transformers as tr
from test_from_hf_gpu import _perturb
from oracle import weights as W
from mertools_amd.encoders import HipHubertModel
dev = torch.device("cuda:0")
hf = _perturb(tr.Data2VecAudioModel(tr.Data2VecAudioConfig(num_hidden_layers=4, mask_time_prob=0.0, attn_implementation="eager")), 4)
wav = W.synth_audio(2, 48000, seed=31)
with torch.no_grad():
    ref = hf(wav, output_hidden_states=True).hidden_states
for prec in ("mean", "mx", "balanced", "mean_a2", "mixed", "accurate"):
    m = HipHubertModel.from_hf(hf, device=dev, precision=prec, self_check=False)
    hs = m(wav.to(dev), output_hidden_states=True).hidden_states
    torch.cuda.synchronize()
    errs = [float((h.cpu() - r).abs().max() / r.abs().max()) for h, r in zip(hs, ref)]
    feat = torch.stack([h.cpu() for h in hs])[[-4, -3, -2, -1]].sum(0); rfeat = torch.stack(ref)[[-4, -3, -2, -1]].sum(0)
    print(f"d2v-audio HF-init [{prec}]: hs " + " ".join(f"{e:.2e}" for e in errs) + f"  frame={float((feat - rfeat).abs().max() / rfeat.abs().max()):.2e} utt={float((feat.mean(1) - rfeat.mean(1)).abs().max() / rfeat.mean(1).abs().max()):.2e}"
          f"  |hs| max {[round(float(r.abs().max()), 2) for r in ref]}")
    del m
