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


# ---- frame-level fusion: LSTMEncoder on the HIP LSTM kernels ----
@pytest.mark.parametrize("B,T,D,H", [(5, 7, 24, 16), (32, 40, 768, 128), (3, 130, 100, 256)])
def test_lstm_last_matches_torch_lstm(dev, B, T, D, H):
    """mer_lstm_fwd / mer_lstm_bwd + the GEMMs around them against torch.nn.LSTM (the reference's own op) on the CPU:
    final hidden state and the gradients of every parameter and of the input."""
    import torch.nn as nn
    from mertools_amd.fusion_ops import lstm_last
    torch.manual_seed(0)
    ref = nn.LSTM(D, H, num_layers=1, batch_first=True)
    x = torch.randn(B, T, D)
    x[: B // 2, : T // 3] = 0.0                        # front padding, as func_mapping_feature produces
    xr = x.clone().requires_grad_(True)
    _, (hn, _) = ref(xr)
    wgt = torch.randn(B, H)
    (hn[0] * wgt).sum().backward()
    dev_rnn = nn.LSTM(D, H, num_layers=1, batch_first=True)
    dev_rnn.load_state_dict(ref.state_dict())
    dev_rnn = dev_rnn.to(dev)
    xd = x.clone().to(dev).requires_grad_(True)
    h = lstm_last(xd, dev_rnn)
    (h * wgt.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert_close(h.detach().cpu(), hn[0].detach(), 2e-5, "lstm h_T")
    assert_close(xd.grad.cpu(), xr.grad, 2e-4, "lstm dX")
    for name in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        assert_close(getattr(dev_rnn, name).grad.cpu(), getattr(ref, name).grad, 2e-4, f"lstm d{name}")


def test_attention_model_frame_level_matches_reference(dev):
    """toolkit.models.Attention with feat_type='frm_align' (LSTMEncoder x3 on mer_lstm_fwd / mer_lstm_bwd) against vectors made by
    RUNNING the reference's own Attention(feat_type='frm_align') / LSTMEncoder (attention.py:22-57, modules/encoder.py:45-72;
    tests/golden/gen_golden.py:fusion_frame_goldens): first-step outputs and every gradient, then 3 Adam steps."""
    from types import SimpleNamespace
    from mertools_amd.toolkit.models import get_models
    from mertools_amd.toolkit.utils.loss import CELoss, MSELoss
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fusion_attention_frm_align.npz"))
    args = SimpleNamespace(model="attention", feat_type="frm_align", audio_dim=48, text_dim=40, video_dim=32, output_dim1=6, output_dim2=1,
                           dropout=0.0, hidden_dim=64, grad_clip=-1.0)
    model = get_models(args)
    keys = sorted(k[len("init_"):] for k in g.files if k.startswith("init_"))
    assert sorted(k[len("model."):] for k in model.state_dict()) == keys          # same parameter names as the reference module
    model.load_state_dict({"model." + k: torch.from_numpy(g["init_" + k]) for k in keys})
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    cls_loss, reg_loss = CELoss(), MSELoss()
    xs = {k: torch.from_numpy(g["x_" + k]).to(dev) for k in ("audios", "texts", "videos")}
    emos, vals = torch.from_numpy(g["emos"]).to(dev), torch.from_numpy(g["vals"]).to(dev)
    losses = []
    for s in range(emos.shape[0]):
        opt.zero_grad()
        f, e, v, il = model({k: x[s] for k, x in xs.items()})
        loss = il + cls_loss(e, emos[s]) + reg_loss(v, vals[s])
        loss.backward()
        if s == 0:
            assert_close(f.detach().cpu(), torch.from_numpy(g["out_features"]), 2e-5, "frame-level fused features")
            assert_close(e.detach().cpu(), torch.from_numpy(g["out_emos_out"]), 2e-5, "frame-level emos_out")
            assert_close(v.detach().cpu(), torch.from_numpy(g["out_vals_out"]), 2e-5, "frame-level vals_out")
            for k, p in model.named_parameters():
                assert_close(p.grad.cpu(), torch.from_numpy(g["grad0_" + k[len("model."):]]), 5e-4, f"grad {k}")
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    assert np.allclose(losses, g["losses"], rtol=2e-4), (losses, g["losses"])
    for k, p in model.state_dict().items():
        assert_close(p.cpu(), torch.from_numpy(g["final_" + k[len("model."):]]), 2e-3, f"final {k}")


@pytest.mark.parametrize("feat_type", ["utt", "frm_align", "frm_unalign"])
def test_main_release_end_to_end_on_synthetic_features(dev, tmp_path, feat_type):
    """python -m mertools_amd.main_release on a small synthetic MER2023-shaped corpus (label npz + per-clip FRAME .npy files),
    all three feature types of main-release.py:157-166: reads, pools / pads (read_data.py), trains 2 epochs x 5 folds on the
    GPU (MLPEncoder or LSTMEncoder), writes the result files; losses must be finite and the run deterministic under --seed."""
    from mertools_amd import main_release
    root = tmp_path / "data" / "mer2023-dataset-process"
    rng = np.random.RandomState(0)
    names = {"train": [f"tr_{i:03d}" for i in range(40)], "test1": [f"t1_{i:02d}" for i in range(8)],
             "test2": [f"t2_{i:02d}" for i in range(8)], "test3": [f"t3_{i:02d}" for i in range(8)]}
    emos = ['neutral', 'angry', 'happy', 'sad', 'worried', 'surprise']
    corp = {}
    for split, ns in names.items():
        corp[f"{split}_corpus"] = {n: {"emo": emos[rng.randint(6)], "val": float(rng.uniform(-3, 3))} for n in ns}
        if split == "test3":
            for n in ns:
                del corp[f"{split}_corpus"][n]["val"]            # test3 has no valence labels (mer2023.py:95-98 -> -10 sentinel)
    os.makedirs(root / "features", exist_ok=True)
    np.savez_compressed(root / "label-6way.npz", **{k: np.array(v, dtype=object) for k, v in corp.items()})
    dims = {"audio-FRA": 24, "text-FRA": 16, "video-FRA": 20}
    for feat, d in dims.items():
        os.makedirs(root / "features" / feat, exist_ok=True)
        for ns in names.values():
            for n in ns:
                np.save(root / "features" / feat / f"{n}.npy", rng.randn(rng.randint(3, 30), d).astype(np.float32))
    argv = ["--model", "attention", "--feat_type", feat_type, "--dataset", "MER2023", "--audio_feature", "audio-FRA", "--text_feature", "text-FRA",
            "--video_feature", "video-FRA", "--epochs", "2", "--batch_size", "16", "--gpu", "0", "--seed", "7", "--data_root", str(tmp_path / "data"),
            "--save_root", str(tmp_path / f"saved-{feat_type}")]
    res1 = main_release.main(argv)
    res2 = main_release.main(argv)
    assert len(res1) == 5
    for a, b in zip(res1, res2):
        for k in a:
            if isinstance(a[k], np.ndarray) and a[k].dtype.kind == "f":
                assert np.isfinite(a[k]).all(), k
                assert np.array_equal(a[k], b[k]), f"{feat_type}: {k} differs between two seeded runs"
    saved = os.listdir(tmp_path / f"saved-{feat_type}-trimodal" / "result")
    assert sum(f.startswith("cv_") for f in saved) == 2 and sum(f.startswith("test1_") for f in saved) == 2


def test_main_release_graph_loop_equals_eager_loop(dev, tmp_path):
    """main_release's default loop (one captured HIP graph per minibatch, per-epoch result copies: train_or_eval_graph) against
    --eager --hip_adam (the reference's loop, main-release.py:17-87, with its per-step host syncs): same folds, same epochs,
    same stored probabilities / predictions (to fp32 rounding of the Adam bias correction) and the same result-file names."""
    from mertools_amd import main_release
    root = tmp_path / "data" / "mer2023-dataset-process"
    rng = np.random.RandomState(1)
    names = {"train": [f"tr_{i:03d}" for i in range(50)], "test1": [f"t1_{i:02d}" for i in range(9)],
             "test2": [f"t2_{i:02d}" for i in range(7)], "test3": [f"t3_{i:02d}" for i in range(5)]}
    emos = ['neutral', 'angry', 'happy', 'sad', 'worried', 'surprise']
    corp = {}
    for split, ns in names.items():
        corp[f"{split}_corpus"] = {n: {"emo": emos[rng.randint(6)], "val": float(rng.uniform(-3, 3))} for n in ns}
        if split == "test3":
            for n in ns:
                del corp[f"{split}_corpus"][n]["val"]
    os.makedirs(root / "features", exist_ok=True)
    np.savez_compressed(root / "label-6way.npz", **{k: np.array(v, dtype=object) for k, v in corp.items()})
    for feat, d in {"audio-UTT": 24, "text-UTT": 16, "video-UTT": 20}.items():
        os.makedirs(root / "features" / feat, exist_ok=True)
        for ns in names.values():
            for n in ns:
                np.save(root / "features" / feat / f"{n}.npy", rng.randn(d).astype(np.float32))
    hyper = tmp_path / "tune.yaml"
    hyper.write_text("attention:\n  hidden_dim: 64\n  dropout: 0.0\n  grad_clip: 1.0\n  lr: 1.0e-3\n")   # (dropout draws come from different RNG paths in the two loops)
    base = ["--model", "attention", "--feat_type", "utt", "--dataset", "MER2023", "--audio_feature", "audio-UTT", "--text_feature", "text-UTT",
            "--video_feature", "video-UTT", "--epochs", "3", "--batch_size", "16", "--gpu", "0", "--seed", "3", "--data_root", str(tmp_path / "data"),
            "--hyper_path", str(hyper)]
    res_g = main_release.main(base + ["--save_root", str(tmp_path / "graph")])
    res_e = main_release.main(base + ["--save_root", str(tmp_path / "eager"), "--eager", "--hip_adam"])
    assert len(res_g) == len(res_e) == 5
    for a, b in zip(res_g, res_e):
        assert set(a) == set(b)
        for k in a:
            if isinstance(a[k], np.ndarray) and a[k].dtype.kind == "f":
                assert a[k].shape == b[k].shape and np.allclose(a[k], b[k], rtol=2e-4, atol=2e-5), k
            elif isinstance(a[k], list):
                assert a[k] == b[k], k     # clip names: same order
    strip = lambda f: f.rsplit("_", 1)[0]   # noqa: E731  (drop the time stamp)
    fg = sorted(strip(f) for f in os.listdir(tmp_path / "graph-trimodal" / "result"))
    fe = sorted(strip(f) for f in os.listdir(tmp_path / "eager-trimodal" / "result"))
    assert fg == fe, (fg, fe)


def test_attention_topn_forward_matches_reference(dev):
    """Attention_TOPN (MER2024/toolkit/models/attention_topn.py:7-89) on the HIP kernels against the reference class's own outputs
    (tests/golden/fusion_attention_topn.npz: 6 streams, emotion-only head output_dim2 = 0), same weights and inputs."""
    from mertools_amd.toolkit.models import get_models
    gold = np.load(os.path.join(G, "fusion_attention_topn.npz"))
    dims = gold["dims"].tolist()
    args = argparse.Namespace(model="attention_topn", audio_dim=dims, output_dim1=6, output_dim2=0, dropout=0.0, hidden_dim=64, grad_clip=-1.0,
                              feat_type="utt")
    m = get_models(args).to(dev).eval()
    m.model.load_state_dict({k[len("init_"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("init_")})
    batch = {f"feat{i}": torch.from_numpy(gold[f"x_feat{i}"]).to(dev) for i in range(len(dims))}
    with torch.no_grad():
        f, e, v, il = m(batch)
    torch.cuda.synchronize()
    assert_close(f.cpu(), torch.from_numpy(gold["features"]), 2e-6, "attention_topn features")
    assert_close(e.cpu(), torch.from_numpy(gold["emos_out"]), 2e-6, "attention_topn emos_out")
    assert tuple(v.shape) == tuple(gold["vals_out"].shape) == (16, 0) and int(il) == int(gold["interloss"])


@pytest.mark.parametrize("model", ["attention", "attention_topn"])
def test_main_release_mer2024(dev, tmp_path, model):
    """`--dataset MER2024` (north_star: "drops into MERBench / MER2024 unchanged"): emotion-only labels (output_dim2 = 0, no MSE term,
    model selection on the weighted F1), train / test1 only, the test npz carries the fold-averaged probabilities
    (MER2024/main-release.py:280-290); with `--model attention` and with `--model attention_topn --fusion_topn 2` (six feature sets)."""
    from mertools_amd import main_release
    from mertools_amd.toolkit.data.feat_data_topn import topn_feature_names
    root = tmp_path / "data" / "mer2024-dataset-process"
    rng = np.random.RandomState(3)
    names = {"train": [f"tr_{i:03d}" for i in range(40)], "test1": [f"t1_{i:02d}" for i in range(9)]}
    emos = ['neutral', 'angry', 'happy', 'sad', 'worried', 'surprise']
    corp = {f"{split}_corpus": {n: {"emo": emos[rng.randint(6)]} for n in ns} for split, ns in names.items()}
    os.makedirs(root / "features", exist_ok=True)
    np.savez_compressed(root / "label-6way.npz", **{k: np.array(v, dtype=object) for k, v in corp.items()})
    feats = {"audio-UTT": 24, "text-UTT": 16, "video-UTT": 20}
    feats.update({n: 8 + 4 * i for i, n in enumerate(topn_feature_names(2, "AVT"))})
    for feat, d in feats.items():
        os.makedirs(root / "features" / feat, exist_ok=True)
        for ns in names.values():
            for n in ns:
                np.save(root / "features" / feat / f"{n}.npy", rng.randn(d).astype(np.float32))
    argv = ["--model", model, "--feat_type", "utt", "--dataset", "MER2024", "--epochs", "2", "--batch_size", "16", "--gpu", "0", "--seed", "5",
            "--data_root", str(tmp_path / "data"), "--save_root", str(tmp_path / "saved")]
    if model == "attention":
        argv += ["--audio_feature", "audio-UTT", "--text_feature", "text-UTT", "--video_feature", "video-UTT"]
    else:
        argv += ["--fusion_topn", "2", "--fusion_modality", "AVT"]
    res = main_release.main(argv)
    assert len(res) == 5
    for fold in res:
        assert "eval_emofscore" in fold and "eval_valmse" not in fold and "test1_emoprobs" in fold and "test2_emoprobs" not in fold
        assert np.isfinite(fold["test1_emoprobs"]).all() and fold["test1_emoprobs"].shape == (9, 6)
    out = tmp_path / ("saved-trimodal" if model == "attention" else "saved-others-multitop") / "result"
    saved = sorted(os.listdir(out))
    assert sum(f.startswith("cv_") for f in saved) == 1 and sum(f.startswith("test1_") for f in saved) == 1, saved
    t1 = [f for f in saved if f.startswith("test1_")][0]
    assert "_f1:" in t1 and "_val:" not in t1 and ("fusiontopn:2_modality:AVT" in t1) == (model == "attention_topn")
    z = np.load(out / t1, allow_pickle=True)
    assert np.asarray(z["emo_probs"]).shape == (9, 6)
