"""DINOv2 branch (extract_vision_huggingface.py:133-144): oracle pinned against the live HF class, host
preprocessing pinned against HF's BitImageProcessor (CPU); HIP encoder against the oracle (GPU)."""
import warnings

import numpy as np
import pytest
import torch

from mertools_amd import synthetic as W
from oracle import encoders_ref as R


def _hf_model(c):
    from transformers import Dinov2Config, Dinov2Model
    hc = Dinov2Config(hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                      mlp_ratio=c.mlp_ratio, image_size=c.image_size, patch_size=c.patch_size, layer_norm_eps=c.layer_norm_eps,
                      attn_implementation="eager")
    return Dinov2Model(hc).eval()


def test_oracle_matches_hf_dinov2():
    c = W.dinov2_config("tiny")
    sd = W.dinov2_state_dict(c, 0)
    m = _hf_model(c)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    g = torch.Generator().manual_seed(1)
    for size in (70, 42):              # native grid (no interpolation) and a smaller input (bicubic position interpolation)
        px = torch.randn(2, 3, size, size, generator=g)
        with torch.no_grad():
            hs = m(px, output_hidden_states=True).hidden_states
        ours = R.dinov2_hidden_states(sd, vars(c), px)
        assert len(hs) == len(ours) == c.num_hidden_layers + 1
        for a, b in zip(ours, hs):
            assert torch.allclose(a, b, rtol=0, atol=1e-5)
        assert torch.allclose(R.dinov2_frame_features(sd, vars(c), px), torch.stack(hs)[-1].sum(dim=1), rtol=0, atol=1e-4)


def test_oracle_matches_hf_dinov2_swiglu():
    """dinov2-giant wiring (Dinov2SwiGLUFFN) against the live HF class."""
    from transformers import Dinov2Config, Dinov2Model
    c = W.dinov2_config("tiny", use_swiglu_ffn=True)
    sd = W.dinov2_state_dict(c, 0)
    m = Dinov2Model(Dinov2Config(hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                                 mlp_ratio=c.mlp_ratio, image_size=c.image_size, patch_size=c.patch_size, layer_norm_eps=c.layer_norm_eps,
                                 use_swiglu_ffn=True, attn_implementation="eager")).eval()
    m.load_state_dict(sd, strict=True)
    px = torch.randn(2, 3, 42, 42, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        hs = m(px, output_hidden_states=True).hidden_states
    for a, b in zip(R.dinov2_hidden_states(sd, vars(c), px), hs):
        assert torch.allclose(a, b, rtol=0, atol=1e-5)


def test_dinov2_preprocess_matches_hf_processor():
    from PIL import Image
    from mertools_amd.extract.visual import dinov2_preprocess
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from transformers import BitImageProcessor
        # facebook/dinov2-*: preprocessor_config.json
        proc = BitImageProcessor(size={"shortest_edge": 256}, crop_size={"height": 224, "width": 224}, resample=3, do_convert_rgb=True,
                                 image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225])
    rng = np.random.default_rng(0)
    for (h, w) in [(200, 260), (300, 224), (224, 224), (97, 131)]:
        fr = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        ref = proc(images=[Image.fromarray(f[:, :, ::-1].copy()) for f in fr], return_tensors="pt")["pixel_values"]
        assert torch.allclose(dinov2_preprocess(fr), ref, rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("accurate", 3e-4), ("mx", 1e-3), ("mean", 1e-3)])
def test_dinov2_tiny(dev, precision, tol):
    from mertools_amd.encoders import HipDinov2Model
    from util import assert_close
    c = W.dinov2_config("tiny")
    sd = W.dinov2_state_dict(c, 0)
    g = torch.Generator().manual_seed(2)
    px = torch.randn(6, 3, 42, 42, generator=g)          # 3x3 patches + CLS; position table interpolated 5x5 -> 3x3
    m = HipDinov2Model(sd, c, device=dev, precision=precision, input_size=42)
    hs = R.dinov2_hidden_states(sd, vars(c), px)
    out = m(px.to(dev), output_hidden_states=True).hidden_states
    feats = m.extract_frames(px.to(dev))
    utt = m.extract_utterance(px.to(dev), [4, 2])
    torch.cuda.synchronize()
    assert_close(torch.stack(out)[-1].cpu(), hs[-1], tol, f"dinov2-tiny[{precision}] last hidden state")
    ref = hs[-1].sum(dim=1)
    assert_close(feats.cpu(), ref, tol, f"dinov2-tiny[{precision}] token-sum features")
    assert_close(utt.cpu(), torch.stack([ref[:4].mean(0), ref[4:].mean(0)]), tol, f"dinov2-tiny[{precision}] UTT")


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("accurate", 3e-4), ("mx", 1e-3), ("mean", 1e-3)])
def test_dinov2_swiglu_tiny(dev, precision, tol):
    """dinov2-giant's SwiGLU feed-forward (mer_swiglu between the two FFN GEMMs)."""
    from mertools_amd.encoders import HipDinov2Model
    from util import assert_close
    c = W.dinov2_config("tiny", use_swiglu_ffn=True)
    sd = W.dinov2_state_dict(c, 0)
    px = torch.randn(6, 3, 42, 42, generator=torch.Generator().manual_seed(6))
    ref = R.dinov2_hidden_states(sd, vars(c), px)[-1]
    m = HipDinov2Model(sd, c, device=dev, precision=precision, input_size=42)
    out = m(px.to(dev), output_hidden_states=True).hidden_states[-1]
    feats = m.extract_frames(px.to(dev))
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, tol, f"dinov2-swiglu-tiny[{precision}] last hidden state")
    assert_close(feats.cpu(), ref.sum(dim=1), tol, f"dinov2-swiglu-tiny[{precision}] token-sum features")


@pytest.mark.gpu
def test_dinov2_base_224(dev):
    """dinov2-base architecture (768/12/12, patch 14, 37x37 trained grid) on 224x224 crops: 257 tokens, 6 frames."""
    from mertools_amd.encoders import HipDinov2Model
    from util import rel_err
    c = W.dinov2_config("base")                             # full depth (12 blocks), as the f2 row of SURVEY §8 names it
    sd = W.dinov2_state_dict(c, 0)
    px = W.synth_frames(6)
    ref = R.dinov2_frame_features(sd, vars(c), px)
    for prec in ("mean", "mx", "accurate"):     # "mean" = the constructor's default
        m = HipDinov2Model(sd, c, device=dev, precision=prec)
        out = m.extract_frames(px.to(dev))
        utt = m.extract_utterance(px.to(dev), [6])
        torch.cuda.synchronize()
        e, eu = rel_err(out.cpu(), ref)[0], rel_err(utt.cpu(), ref.mean(0, keepdim=True))[0]
        print(f"dinov2-base[{prec}]: frames={e:.2e} utt={eu:.2e}")
        assert eu <= 1e-3 and e <= 1e-3
        del m


@pytest.mark.gpu
def test_dinov2_extract_files(dev, tmp_path):
    from mertools_amd.encoders import HipDinov2Model
    from mertools_amd.extract import visual
    c = W.dinov2_config("tiny")
    sd = W.dinov2_state_dict(c, 0)
    m = HipDinov2Model(sd, c, device=dev, input_size=42)
    rng = np.random.default_rng(3)
    vids = {"v1": rng.integers(0, 256, (5, 60, 50, 3), dtype=np.uint8), "v2": rng.integers(0, 256, (70, 48, 48, 3), dtype=np.uint8),
            "v3": np.zeros((0, 48, 48, 3), dtype=np.uint8)}
    real_pre = visual.dinov2_preprocess
    visual.dinov2_preprocess = lambda fr, size=224, resize_to=256: real_pre(fr, 42, 48)
    try:
        for level, sub in (("UTTERANCE", "utt"), ("FRAME", "fra")):
            visual.extract_dinov2(m, "unused", str(tmp_path / sub), level, vids=list(vids), reader=lambda d, v: vids[v], nframe=8)
    finally:
        visual.dinov2_preprocess = real_pre
    for vid, fr in vids.items():
        u, f = np.load(tmp_path / "utt" / f"{vid}.npy"), np.load(tmp_path / "fra" / f"{vid}.npy")
        if len(fr) == 0:
            assert f.shape[0] == 1 and not f.any() and not u.any()
            continue
        px = real_pre(visual.resample_frames_uniform(fr, 8), 42, 48)
        ref = R.dinov2_frame_features(sd, vars(c), px).numpy()
        assert f.shape == (8, c.hidden_size) and u.shape == (c.hidden_size,) and f.dtype == np.float32
        assert np.abs(f - ref).max() / np.abs(ref).max() < 1e-3
        assert np.abs(u - ref.mean(0)).max() / np.abs(ref.mean(0)).max() < 1e-3


# ---- data2vec-vision / BEiT (extract_vision_huggingface.py:123-131) on the same engine variant ----
def test_oracle_matches_hf_data2vec_vision():
    from transformers import Data2VecVisionConfig, Data2VecVisionModel
    for over in (dict(), dict(use_relative_position_bias=True, use_shared_relative_position_bias=False),
                 dict(use_relative_position_bias=True, use_absolute_position_embeddings=True, layer_scale_init_value=0.0)):
        c = W.data2vec_vision_config("tiny", **over)
        sd = {k: v for k, v in W.data2vec_vision_state_dict(c, 0).items() if not k.startswith("pooler.")}
        hc = Data2VecVisionConfig(hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                                  intermediate_size=c.intermediate_size, image_size=c.image_size, patch_size=c.patch_size,
                                  layer_norm_eps=c.layer_norm_eps, use_absolute_position_embeddings=c.use_absolute_position_embeddings,
                                  use_relative_position_bias=c.use_relative_position_bias,
                                  use_shared_relative_position_bias=c.use_shared_relative_position_bias,
                                  layer_scale_init_value=c.layer_scale_init_value, use_mean_pooling=True, use_mask_token=False,
                                  drop_path_rate=0.0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager")
        m = Data2VecVisionModel(hc).eval()
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not [k for k in missing if not k.startswith("pooler.")] and not unexpected
        px = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            hs = m(px, output_hidden_states=True).hidden_states
        ours = R.data2vec_vision_hidden_states(sd, vars(c), px)
        assert len(hs) == len(ours)
        for a, b in zip(ours, hs):
            assert torch.allclose(a, b, rtol=0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("over", [dict(), dict(use_relative_position_bias=True, use_shared_relative_position_bias=False),
                                  dict(use_relative_position_bias=True, use_absolute_position_embeddings=True, layer_scale_init_value=0.0)],
                         ids=["shared-bias", "per-layer-bias", "both-abs-pos-no-scale"])
def test_data2vec_vision_tiny(dev, over):
    from mertools_amd.encoders import HipData2VecVisionModel
    from util import assert_close
    c = W.data2vec_vision_config("tiny", **over)
    sd = W.data2vec_vision_state_dict(c, 0)
    px = torch.randn(5, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    hs = R.data2vec_vision_hidden_states(sd, vars(c), px)
    ref = hs[-1].sum(dim=1)
    for prec, tol in (("accurate", 5e-4), ("mx", 1e-3), ("mean", 1e-3)):   # "accurate" is bounded by attention rounding q/k/v/P to f16 once
        m = HipData2VecVisionModel(sd, c, device=dev, precision=prec)
        out = m(px.to(dev), output_hidden_states=True).hidden_states
        feats = m.extract_frames(px.to(dev))
        torch.cuda.synchronize()
        assert_close(torch.stack(out)[-1].cpu(), hs[-1], tol, f"data2vec-vision-tiny[{prec}] last hidden state")
        assert_close(feats.cpu(), ref, tol, f"data2vec-vision-tiny[{prec}] token-sum features")
        del m


@pytest.mark.gpu
def test_data2vec_vision_base_224(dev):
    """data2vec-vision-base architecture (768/12/12, patch 16, 197 tokens, shared relative position bias), full depth."""
    from mertools_amd.encoders import HipData2VecVisionModel
    from util import rel_err
    c = W.data2vec_vision_config("base")
    sd = W.data2vec_vision_state_dict(c, 0)
    px = W.synth_frames(6)
    ref = R.data2vec_vision_frame_features(sd, vars(c), px)
    for prec in ("mean", "mx", "accurate"):     # "mean" = the constructor's default
        m = HipData2VecVisionModel(sd, c, device=dev, precision=prec)
        out = m.extract_frames(px.to(dev))
        utt = m.extract_utterance(px.to(dev), [6])
        torch.cuda.synchronize()
        e, eu = rel_err(out.cpu(), ref)[0], rel_err(utt.cpu(), ref.mean(0, keepdim=True))[0]
        print(f"data2vec-vision-base[{prec}]: frames={e:.2e} utt={eu:.2e}")
        assert eu <= 1e-3 and e <= 1e-3
        del m


def test_data2vec_vision_preprocess_matches_hf_processor():
    """BeitImageProcessor as configured by facebook/data2vec-vision-base: square bicubic resize, no crop, mean = std = 0.5."""
    from PIL import Image
    from mertools_amd.extract.visual import data2vec_vision_preprocess
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from transformers import BeitImageProcessor
        proc = BeitImageProcessor(do_resize=True, size={"height": 224, "width": 224}, resample=3, do_center_crop=False, do_normalize=True,
                                  image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    rng = np.random.default_rng(0)
    for (h, w) in [(200, 260), (224, 224), (97, 131)]:
        fr = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        ref = proc(images=[Image.fromarray(f[:, :, ::-1].copy()) for f in fr], return_tensors="pt")["pixel_values"]
        assert torch.allclose(data2vec_vision_preprocess(fr), ref, rtol=0, atol=2e-6)
