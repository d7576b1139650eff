"""AffectGPT front-ends: registry semantics, the joint-normalisation quirk pinned against HF's own
feature extractor (CPU), and the two HIP encoders against the oracle (GPU)."""
import numpy as np
import pytest
import torch


def test_registry_semantics():
    from mertools_amd.affectgpt import registry
    assert registry.get_visual_encoder_class("HIP_CLIP_VIT_LARGE").__name__ == "HIP_CLIP_VIT_LARGE"
    assert registry.get_acoustic_encoder_class("HIP_HUBERT_LARGE").__name__ == "HIP_HUBERT_LARGE"
    assert registry.get_visual_encoder_class("nope") is None and registry.get_acoustic_encoder_class("nope") is None
    with pytest.raises(KeyError):
        registry.register_visual_encoder("HIP_CLIP_VIT_LARGE")(object)
    assert "HIP_HUBERT_LARGE" in registry.list_acoustic_encoders()


def test_joint_normalisation_matches_hf_feature_extractor():
    """encoder.py:425-427 hands the extractor a 2-D TENSOR and takes input_values[0]: one 'utterance', global statistics."""
    from transformers import Wav2Vec2FeatureExtractor
    from mertools_amd.affectgpt.encoder import joint_zero_mean_unit_var
    g = torch.Generator().manual_seed(0)
    raw = torch.randn(6, 3200, generator=g) * torch.tensor([0.1, 1.0, 3.0, 0.5, 0.01, 2.0])[:, None] + 0.3
    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True)
    ref = fe(raw, sampling_rate=16000, return_tensors="pt").input_values[0]
    out = joint_zero_mean_unit_var(raw)
    assert ref.shape == out.shape == (6, 3200)
    assert torch.allclose(out, ref, rtol=0, atol=2e-6), (out - ref).abs().max()
    per_row = (raw - raw.mean(1, keepdim=True)) / raw.var(1, unbiased=False, keepdim=True).add(1e-7).sqrt()
    assert (per_row - ref).abs().max() > 0.1      # the per-row normalisation one would expect is NOT what the reference computes


@pytest.mark.gpu
def test_hip_hubert_large_frontend(dev):
    from mertools_amd import synthetic as W
    from mertools_amd.affectgpt.encoder import HIP_HUBERT_LARGE, joint_zero_mean_unit_var
    from mertools_amd.encoders import HipHubertModel
    from oracle import encoders_ref as R
    from util import assert_close
    cfg = W.hubert_config("tiny", feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)   # large-style wiring
    sd = W.hubert_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(5)
    raw_audio = torch.randn(2, 3, 1, 32000, generator=g) * 0.05 + 0.01      # [b, t, 1, 32000]: three 2-s chunks per sample
    enc = HIP_HUBERT_LARGE(model=HipHubertModel(sd, cfg, device=dev), device=dev)
    out = enc(None, raw_audio)
    torch.cuda.synchronize()
    x = joint_zero_mean_unit_var(raw_audio[:, :, 0, :].reshape(6, 32000))
    hs = R.hubert_hidden_states(sd, vars(cfg), x)
    ref = torch.stack(hs)[[-4, -3, -2, -1]].mean(0).mean(1).view(2, 3, -1)
    assert out.shape == (2, 3, cfg.hidden_size) and enc.hidden_size == cfg.hidden_size
    assert_close(out.cpu(), ref, 1e-3, "HIP_HUBERT_LARGE")


@pytest.mark.gpu
def test_hip_clip_vit_large_frontend(dev):
    from mertools_amd import synthetic as W
    from mertools_amd.affectgpt.encoder import HIP_CLIP_VIT_LARGE
    from mertools_amd.encoders import HipCLIPModel
    from mertools_amd.extract.visual import clip_preprocess
    from oracle import encoders_ref as R
    from util import assert_close
    cfg = W.clip_config("tiny", patch_size=14, image_size=224)
    sd = W.clip_state_dict(cfg, 4)
    g = torch.Generator().manual_seed(6)
    raw = torch.randint(0, 256, (2, 3, 2, 240, 300), generator=g, dtype=torch.uint8)      # [b, c, t, h, w] RGB, not yet 224x224
    enc = HIP_CLIP_VIT_LARGE(model=HipCLIPModel(sd, cfg, device=dev), device=dev)
    out = enc(None, raw)
    torch.cuda.synchronize()
    frames = raw.permute(0, 2, 3, 4, 1).reshape(4, 240, 300, 3).numpy()
    px = clip_preprocess(np.ascontiguousarray(frames[:, :, :, ::-1]), 224)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px).view(2, 2, -1)
    assert out.shape == (2, 2, cfg.projection_dim) and enc.hidden_size == cfg.projection_dim
    assert_close(out.cpu(), ref, 1e-3, "HIP_CLIP_VIT_LARGE")
