"""CPU checks of round 3's host-side pieces: the activation-outlier re-parametrisation (exact for the pre-LN CLIP oracle), the PCM16 reader
of the audio driver, the writer interface of extract.pipeline on a machine without a GPU."""
import os
import wave

import numpy as np
import torch

from oracle import encoders_ref as R
from oracle import weights as W
from mertools_amd import synthetic as S


def test_ln_outliers_is_an_exact_reparametrisation_for_the_clip_oracle():
    cfg = W.clip_config("tiny")
    sd = W.clip_state_dict(cfg, 3)
    out = W.ln_outliers(sd)
    vcfg = dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim)
    px = W.synth_frames(3, 64, seed=7)
    a, b = R.clip_image_features(sd, vcfg, px), R.clip_image_features(out, vcfg, px)
    assert ((a - b).abs().max() / a.abs().max()).item() < 1e-5
    # ... while the LayerNorm parameters and the consumer columns did change, by the same 30-100x factor on the same three channels
    k = "vision_model.encoder.layers.0.layer_norm1.weight"
    ratio = out[k] / sd[k]
    idx = (ratio - 1).abs() > 1e-3
    assert int(idx.sum()) == 3 and float(ratio[idx].min()) >= 30 and float(ratio[idx].max()) <= 100
    wq0, wq1 = sd["vision_model.encoder.layers.0.self_attn.q_proj.weight"], out["vision_model.encoder.layers.0.self_attn.q_proj.weight"]
    assert torch.allclose(wq1[:, idx] * ratio[idx], wq0[:, idx], rtol=1e-6) and torch.equal(wq1[:, ~idx], wq0[:, ~idx])


def test_ln_outliers_edges_cover_the_post_ln_encoders_but_not_the_saved_states():
    hc = W.hubert_config("tiny")
    edges = dict(S._ln_linear_edges(W.hubert_state_dict(hc, 1)))
    L = hc.num_hidden_layers
    assert "encoder.layer_norm" in edges and f"encoder.layers.{L - 1}.layer_norm" in edges
    assert all(f"encoder.layers.{i}.final_layer_norm" not in edges for i in range(max(L - 4, 0), L))   # last-4 hidden states are saved features
    bc = W.bert_config("tiny")
    bedges = dict(S._ln_linear_edges(W.bert_state_dict(bc, 4)))
    assert "embeddings.LayerNorm" in bedges and "encoder.layer.0.attention.output.LayerNorm" in bedges
    # the perturbed post-LN network is a different, still well-conditioned function: finite and not degenerate
    sd = W.ln_outliers(W.hubert_state_dict(hc, 1))
    hs = R.hubert_hidden_states(sd, vars(hc), W.synth_audio(1, 4000, seed=2))
    assert all(torch.isfinite(h).all() for h in hs) and hs[-1].std() > 0


def test_read_pcm16_and_the_exact_pcm_test(tmp_path):
    from mertools_amd.extract import audio
    x = (np.random.RandomState(0).randn(5000) * 0.1).clip(-1, 1 - 1 / 32768)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((x * 32768).astype("<i2").tobytes())
    pcm, sr = audio.read_pcm16(p)
    f64, sr2 = audio.read_audio(p)
    assert sr == sr2 == 16000 and pcm.dtype == np.int16 and pcm.flags.writeable
    assert np.array_equal(pcm.astype(np.float64) / 32768.0, f64)                     # the same samples the float reader returns
    assert np.array_equal(audio.to_pcm16_or_f32(f64), pcm)                           # exactly-representable samples go up as int16 ...
    y = audio.to_pcm16_or_f32(f64 * 0.3333)
    assert y.dtype == np.float32                                                     # ... anything else as fp32
    stereo = str(tmp_path / "s.wav")
    with wave.open(stereo, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.zeros(200, "<i2").tobytes())
    assert audio.read_pcm16(stereo) is None                                          # multi-channel files keep the reference's float path


def test_writer_on_cpu_is_the_blocking_reference_path(tmp_path):
    from mertools_amd.extract.pipeline import SyncWriter, writer
    seen = []
    with writer("cpu") as out:
        assert isinstance(out, SyncWriter)
        out.submit(torch.arange(6.0).view(2, 3), lambda arr: seen.append(arr.copy()))
        out.submit([torch.ones(2), torch.zeros(3)], lambda a, b: seen.append((a.sum(), b.sum())))
    assert np.array_equal(seen[0], np.arange(6.0).reshape(2, 3)) and seen[1] == (2.0, 0.0)
