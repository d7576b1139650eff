#!/usr/bin/env python3
"""Fusion-classifier training step rate: Attention model, H=128, dims 768/768/512, batch 32 —
the reference's shapes.  Compares (a) the HIP kernels driven step by step through autograd + torch.optim.Adam (what
main_release.py does, incl. the reference's per-step host syncs), (b) the same without host syncs, (c) the whole step
replayed from one HIP graph (FusionGraphTrainer), and (d) the CPU oracle of the reference arithmetic."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mertools_amd.fusion_trainer import FusionGraphTrainer  # noqa: E402
from mertools_amd.toolkit.models import get_models  # noqa: E402
from mertools_amd.toolkit.utils.loss import CELoss, MSELoss  # noqa: E402

dev = torch.device("cuda:0")
args = argparse.Namespace(model="attention", text_dim=768, audio_dim=768, video_dim=512, output_dim1=6, output_dim2=1, dropout=0.3,
                          hidden_dim=128, grad_clip=-1.0, feat_type="utt")
B, STEPS = 32, 300
torch.manual_seed(0)
batch = dict(audios=torch.randn(B, 768, device=dev), texts=torch.randn(B, 768, device=dev), videos=torch.randn(B, 512, device=dev))
emos, vals = torch.randint(0, 6, (B,), device=dev), torch.randn(B, device=dev)


def timed(fn, steps=STEPS):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


res = {}
m = get_models(args).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
cls, reg = CELoss(), MSELoss()


def eager(sync):
    opt.zero_grad()
    f, e, v, il = m(batch)
    loss = il + cls(e, emos) + reg(v, vals)
    if sync:  # the reference pulls probs / labels / loss to the host every step (main-release.py:53-59)
        e.data.cpu().numpy(); emos.data.cpu().numpy(); v.data.cpu().numpy(); loss.data.cpu().numpy()
    loss.backward()
    opt.step()


res["hip_eager_with_reference_syncs"] = timed(lambda: eager(True))
res["hip_eager_no_sync"] = timed(lambda: eager(False))
m2 = get_models(args).to(dev).train()
tr = FusionGraphTrainer(m2, lr=1e-3, weight_decay=1e-5)
res["hip_graph_replay"] = timed(lambda: tr.train_step(batch, emos, vals), steps=2000)

# CPU oracle of the reference arithmetic (plain torch modules on the host cores)
from oracle import fusion_ref as FR  # noqa: E402  (baseline leg only)
torch.set_num_threads(min(os.cpu_count() or 1, 8))
sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.model.state_dict().items()}
cb = {k: v.cpu() for k, v in batch.items()}
copt = torch.optim.Adam(list(sd.values()), lr=1e-3, weight_decay=1e-5)
ce, cv = emos.cpu(), vals.cpu()


def cpu_step():
    copt.zero_grad()
    f, e, v = FR.attention_forward(sd, cb)
    (FR.ce_loss(e, ce) + FR.mse_loss(v, cv)).backward()
    copt.step()


for _ in range(10):
    cpu_step()
t0 = time.perf_counter()
for _ in range(200):
    cpu_step()
res["cpu_oracle_torch"] = 200 / (time.perf_counter() - t0)
print(json.dumps({"metric": "fusion train steps/sec (Attention H=128, 768/768/512, batch 32)", "unit": "steps/s",
                  **{k: round(v, 1) for k, v in res.items()}}))
