"""Fusion classifier on the HIP kernels (GPU) against golden vectors produced by the reference's own modules
(tests/golden/fusion_*.npz, losses.npz): forward, gradients, 5 Adam steps, the loss functions.
Tolerance 2e-5 (fp32 kernels; only the summation order differs from torch's CPU GEMM)."""
import argparse
import os

import numpy as np
import pytest
import torch

from util import assert_close, rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5


def _args(model="attention"):
    return argparse.Namespace(model=model, text_dim=80, audio_dim=96, video_dim=64, output_dim1=6, output_dim2=1, dropout=0.0,
                              hidden_dim=64, grad_clip=-1.0, feat_type="utt")


def _load(m, g, prefix):
    sd = {k: torch.from_numpy(g[prefix + k]) for k in m.state_dict()}
    m.load_state_dict(sd)


def test_attention_forward_matches_reference(dev):
    from mertools_amd.toolkit.models import get_models
    g = np.load(os.path.join(G, "fusion_attention.npz"))
    m = get_models(_args()).to(dev).eval()
    _load(m.model, g, "init_")
    batch = {k: torch.from_numpy(g["x_" + k][0]).to(dev) for k in ("audios", "texts", "videos")}
    with torch.no_grad():
        f, e, v, il = m(batch)
    torch.cuda.synchronize()
    assert_close(f.cpu(), torch.from_numpy(g["out_features"]), TOL, "features")
    assert_close(e.cpu(), torch.from_numpy(g["out_emos_out"]), TOL, "emos_out")
    assert_close(v.cpu(), torch.from_numpy(g["out_vals_out"]), TOL, "vals_out")
    assert il.dtype == torch.int64 and il.ndim == 0 and int(il) == 0 and il.is_cuda


@pytest.mark.parametrize("opt", ["torch_adam", "hip_adam"])
def test_attention_training_steps_match_reference(dev, opt):
    """main-release.py:50-66 for 5 steps: per-step loss, first-step gradients and final parameters."""
    from mertools_amd.fusion_ops import HipAdam
    from mertools_amd.toolkit.models import get_models
    from mertools_amd.toolkit.utils.loss import CELoss, MSELoss
    g = np.load(os.path.join(G, "fusion_attention.npz"))
    m = get_models(_args()).to(dev).train()
    _load(m.model, g, "init_")
    cls_loss, reg_loss = CELoss(), MSELoss()
    o = (HipAdam if opt == "hip_adam" else torch.optim.Adam)(m.parameters(), lr=1e-3, weight_decay=1e-5)
    steps = len(g["losses"])
    for s in range(steps):
        o.zero_grad()
        batch = {k: torch.from_numpy(g["x_" + k][s]).to(dev) for k in ("audios", "texts", "videos")}
        f, e, v, il = m(batch)
        loss = il + cls_loss(e, torch.from_numpy(g["emos"][s]).to(dev)) + reg_loss(v, torch.from_numpy(g["vals"][s]).to(dev))
        loss.backward()
        if s == 0:
            for k, p in m.model.named_parameters():
                assert_close(p.grad.cpu(), torch.from_numpy(g["grad0_" + k]), 5e-5, f"grad {k}")
        o.step()
        assert abs(loss.item() - g["losses"][s]) <= 5e-5 * abs(g["losses"][s]), (s, loss.item(), g["losses"][s])
    torch.cuda.synchronize()
    for k, p in m.model.state_dict().items():
        assert_close(p.cpu(), torch.from_numpy(g["final_" + k]), 1e-4, f"final {k}")


def test_losses_match_reference(dev):
    from mertools_amd.toolkit.utils.loss import CELoss, MSELoss
    g = np.load(os.path.join(G, "losses.npz"))
    ce = CELoss()(torch.from_numpy(g["pred"]).to(dev), torch.from_numpy(g["tgt"]).to(dev)).item()
    mse = MSELoss()(torch.from_numpy(g["vp"]).to(dev), torch.from_numpy(g["vt"]).to(dev)).item()
    assert abs(ce - float(g["ce"])) < 2e-6 * max(1, abs(float(g["ce"]))) and abs(mse - float(g["mse"])) < 2e-6 * max(1, abs(float(g["mse"])))


def test_lf_dnn_and_mer2023_models_match_reference(dev):
    from mertools_amd.toolkit.models import get_models
    from mertools_amd.toolkit.models.mer2023 import MLP, Attention
    ga = np.load(os.path.join(G, "fusion_attention.npz"))
    batch = {k: torch.from_numpy(ga["x_" + k][0]).to(dev) for k in ("audios", "texts", "videos")}
    g = np.load(os.path.join(G, "fusion_lf_dnn.npz"))
    m = get_models(_args("lf_dnn")).to(dev).eval()
    _load(m.model, g, "init_")
    with torch.no_grad():
        f, e, v, _ = m(batch)
    assert_close(f.cpu(), torch.from_numpy(g["features"]), TOL, "lf_dnn features")
    assert_close(e.cpu(), torch.from_numpy(g["emos_out"]), TOL, "lf_dnn emos")
    assert_close(v.cpu(), torch.from_numpy(g["vals_out"]), TOL, "lf_dnn vals")
    g = np.load(os.path.join(G, "fusion_mer2023.npz"))
    mlp = MLP(96 + 80 + 64, 6, 1, layers="64,32", dropout=0.0).to(dev).eval()
    _load(mlp, g, "mlp_")
    att = Attention(96, 80, 64, 6, 1, layers="64,32", dropout=0.0).to(dev).eval()
    _load(att, g, "att_")
    with torch.no_grad():
        mf, me, mv = mlp(torch.cat([batch["audios"], batch["texts"], batch["videos"]], dim=1))
        af, ae, av = att(batch["audios"], batch["texts"], batch["videos"])
    for o, k in [(mf, "mlp_features"), (me, "mlp_emos"), (mv, "mlp_vals"), (af, "att_features"), (ae, "att_emos"), (av, "att_vals")]:
        assert_close(o.cpu(), torch.from_numpy(g[k]), TOL, k)


def test_dropout_is_inverted_and_seeded(dev):
    from mertools_amd.fusion_ops import dropout
    x = torch.ones(64, 256, device=dev, requires_grad=True)
    torch.manual_seed(5)
    y = dropout(x, 0.25, True)
    torch.manual_seed(5)
    y2 = dropout(x, 0.25, True)
    assert torch.equal(y, y2)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.03 and abs(y.max().item() - 1 / 0.75) < 1e-6
    y.sum().backward()
    assert torch.equal((x.grad != 0), (y != 0))
    assert dropout(x, 0.25, False) is x


@pytest.mark.parametrize("use_graph", [False, True])
def test_graph_trainer_matches_reference(dev, use_graph):
    """FusionGraphTrainer (flat parameters, device-side Adam step counter, hipGraph replay) reproduces the reference's
    5 training steps; graph replay and eager execution of the same kernels agree bit for bit."""
    from mertools_amd.fusion_trainer import FusionGraphTrainer
    from mertools_amd.toolkit.models import get_models
    g = np.load(os.path.join(G, "fusion_attention.npz"))
    m = get_models(_args()).to(dev)
    _load(m.model, g, "init_")
    tr = FusionGraphTrainer(m, lr=1e-3, weight_decay=1e-5, grad_clip=-1.0, use_graph=use_graph)
    losses = []
    for s in range(len(g["losses"])):
        batch = {k: torch.from_numpy(g["x_" + k][s]).to(dev) for k in ("audios", "texts", "videos")}
        loss, e, v = tr.train_step(batch, torch.from_numpy(g["emos"][s]).to(dev), torch.from_numpy(g["vals"][s]).to(dev))
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-5)
    for k, p in m.model.state_dict().items():
        assert_close(p.cpu(), torch.from_numpy(g["final_" + k]), 1e-4, f"final {k} (graph={use_graph})")
    assert sorted(m.model.state_dict()) == sorted(k[5:] for k in g.files if k.startswith("init_"))
