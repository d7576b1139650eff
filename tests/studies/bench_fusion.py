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

# (e) / (f): the same model driven through main_release's two epoch loops over a DataLoader of pinned host minibatches —
# the reference's loop (train_or_eval_model: 3-5 host round trips per minibatch) and the graph loop (train_or_eval_graph)
from mertools_amd import main_release as MR  # noqa: E402


class _DS(torch.utils.data.Dataset):
    def __init__(self, n):
        g = torch.Generator().manual_seed(1)
        self.a, self.t, self.v = torch.randn(n, 768, generator=g), torch.randn(n, 768, generator=g), torch.randn(n, 512, generator=g)
        self.e, self.val = torch.randint(0, 6, (n,), generator=g), torch.randn(n, generator=g)

    def __len__(self):
        return len(self.e)

    def __getitem__(self, i):
        return i

    def collater(self, idx):
        idx = torch.tensor(idx)
        return dict(audios=self.a[idx], texts=self.t[idx], videos=self.v[idx]), self.e[idx], self.val[idx], [f"c{int(i)}" for i in idx]


class _Results:
    @staticmethod
    def calculate_results(emo_probs=[], emo_labels=[], val_preds=[], val_labels=[]):
        return {}, ""


ds = _DS(32 * 200)
loader = torch.utils.data.DataLoader(ds, batch_size=32, collate_fn=ds.collater, pin_memory=True)
margs = argparse.Namespace(output_dim1=6, output_dim2=1, print_iters=1e8)
m3 = get_models(args).to(dev)
opt3 = torch.optim.Adam(m3.parameters(), lr=1e-3, weight_decay=1e-5)
MR.train_or_eval_model(margs, m3, MSELoss(), CELoss(), loader, 0, opt3, True, dataloader_class=_Results)
torch.cuda.synchronize()
t0 = time.perf_counter()
MR.train_or_eval_model(margs, m3, MSELoss(), CELoss(), loader, 0, opt3, True, dataloader_class=_Results)
torch.cuda.synchronize()
res["main_release_eager_epoch"] = len(loader) / (time.perf_counter() - t0)
m4 = get_models(args).to(dev)
tr4 = FusionGraphTrainer(m4, lr=1e-3, weight_decay=1e-5)
MR.train_or_eval_graph(margs, tr4, loader, 0, True, dataloader_class=_Results)
torch.cuda.synchronize()
t0 = time.perf_counter()
MR.train_or_eval_graph(margs, tr4, loader, 0, True, dataloader_class=_Results)
torch.cuda.synchronize()
res["main_release_graph_epoch"] = len(loader) / (time.perf_counter() - t0)

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
