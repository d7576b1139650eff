"""Whisper branch of the audio extractor (extract_audio_huggingface.py:79-89: two decoder start tokens -> (2, D) per clip):
oracle pinned against the live HF WhisperModel and the log-mel front end against WhisperFeatureExtractor (CPU); the HIP path
(implicit-im2col conv GEMMs, pre-LN encoder, fp32 decoder with mer_small_attention) against the oracle (GPU)."""
import os
import wave

import numpy as np
import pytest
import torch

from mertools_amd import synthetic as W
from oracle import encoders_ref as R
from util import assert_close


def _inputs(c, B, seed=1):
    g = torch.Generator().manual_seed(seed)
    mel = torch.randn(B, c.num_mel_bins, 2 * c.max_source_positions, generator=g) * 0.5
    ids = torch.full((B, 2), c.decoder_start_token_id, dtype=torch.long)
    return mel, ids


def test_oracle_matches_hf_whisper():
    from transformers import WhisperConfig, WhisperModel
    c = W.whisper_config("tiny")
    sd = W.whisper_state_dict(c, 0)
    hc = WhisperConfig(vocab_size=c.vocab_size, num_mel_bins=c.num_mel_bins, encoder_layers=c.encoder_layers, decoder_layers=c.decoder_layers,
                       encoder_attention_heads=c.encoder_attention_heads, decoder_attention_heads=c.decoder_attention_heads,
                       encoder_ffn_dim=c.encoder_ffn_dim, decoder_ffn_dim=c.decoder_ffn_dim, d_model=c.d_model,
                       max_source_positions=c.max_source_positions, max_target_positions=c.max_target_positions,
                       decoder_start_token_id=c.decoder_start_token_id, pad_token_id=c.pad_token_id, bos_token_id=1, eos_token_id=2,
                       attn_implementation="eager")
    m = WhisperModel(hc).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    mel, ids = _inputs(c, 2)
    with torch.no_grad():
        out = m(mel, decoder_input_ids=ids)
    assert torch.allclose(R.whisper_encoder(sd, vars(c), mel), out.encoder_last_hidden_state, rtol=0, atol=1e-5)
    assert torch.allclose(R.whisper_last_hidden_state(sd, vars(c), mel, ids), out.last_hidden_state, rtol=0, atol=1e-5)


def test_whisper_log_mel_matches_hf_feature_extractor():
    from transformers import WhisperFeatureExtractor
    from mertools_amd.extract.audio import whisper_log_mel, whisper_mel_filters
    fe = WhisperFeatureExtractor()
    assert np.abs(fe.mel_filters - whisper_mel_filters()).max() < 1e-12
    rng = np.random.default_rng(0)
    for secs in (0.4, 7.0, 31.0):                    # short, typical, longer than the 30 s window (cut)
        n = int(16000 * secs)
        x = rng.standard_normal(n) * 0.1 * np.sin(np.arange(n) / 300.0)
        ref = fe(x, sampling_rate=16000, return_tensors="pt").input_features
        out = whisper_log_mel(x)
        assert out.shape == ref.shape == (1, 80, 3000) and out.dtype == torch.float32
        assert (out - ref).abs().max().item() < 5e-5


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("Tq,Tk,causal", [(2, 2, True), (2, 100, False), (2, 1500, False), (5, 5, True)])
def test_small_attention(Tq, Tk, causal):
    from mertools_amd import ops
    g = torch.Generator().manual_seed(Tk)
    B, H = 3, 2
    q, k, v = torch.randn(B * Tq, H * 64, generator=g), torch.randn(B * Tk, H * 64, generator=g), torch.randn(B * Tk, H * 64, generator=g)
    out = ops.small_attention(q.cuda(), k.cuda(), v.cuda(), B, Tq, Tk, H, 0.125, causal).cpu()
    qh, kh, vh = (t.view(B, -1, H, 64).transpose(1, 2).double() for t in (q, k, v))
    s = qh @ kh.transpose(2, 3) * 0.125
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), diagonal=1), float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Tq, H * 64)
    assert_close(out, ref, 2e-6, f"small_attention {Tq}x{Tk}")


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("mx", 1e-3), ("balanced", 1e-3), ("accurate", 3e-4)])
def test_whisper_tiny_matches_oracle(precision, tol):
    from mertools_amd.whisper import HipWhisperModel
    c = W.whisper_config("tiny")
    sd = W.whisper_state_dict(c, 0)
    mel, ids = _inputs(c, 3)
    m = HipWhisperModel(sd, c, precision=precision)
    enc = m.encode(mel)[0].cpu().view(3, c.max_source_positions, c.d_model)
    e = assert_close(enc, R.whisper_encoder(sd, vars(c), mel), 2 * tol, f"whisper encoder states ({precision})")
    out = m(mel, decoder_input_ids=ids).last_hidden_state.cpu()
    d = assert_close(out, R.whisper_last_hidden_state(sd, vars(c), mel, ids), tol, f"whisper decoder states ({precision})")
    print(f"whisper tiny {precision}: encoder {e:.2e} decoder {d:.2e}")
    assert torch.equal(m.extract_utterance(mel).cpu(), out)


@pytest.mark.gpu
def test_whisper_base_shape_matches_oracle():
    """whisper-base widths with 2+2 layers and the real 1500-frame window: streaming attention, MX conv2 / block GEMMs, Tk = 1500."""
    from mertools_amd.whisper import HipWhisperModel
    c = W.whisper_config("base", encoder_layers=2, decoder_layers=2, vocab_size=128, decoder_start_token_id=5, pad_token_id=4)
    sd = W.whisper_state_dict(c, 3)
    mel, ids = _inputs(c, 2, seed=7)
    out = HipWhisperModel(sd, c).extract_utterance(mel).cpu()
    assert out.shape == (2, 2, 512)
    assert_close(out, R.whisper_last_hidden_state(sd, vars(c), mel, ids), 1e-3, "whisper-base-shaped decoder states")


@pytest.mark.gpu
def test_extract_whisper_writes_reference_layout(tmp_path):
    from mertools_amd.extract.audio import extract_whisper, whisper_log_mel
    from mertools_amd.whisper import HipWhisperModel
    c = W.whisper_config("tiny")
    sd = W.whisper_state_dict(c, 0)
    m = HipWhisperModel(sd, c)
    rng = np.random.default_rng(0)
    files = []
    for i, n in enumerate((9000, 16000, 40000)):      # the tiny model's window is 2 s: shorter, exact fit and cut clips
        p = tmp_path / f"clip{i}.wav"
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((rng.standard_normal(n) * 3000).astype("<i2").tobytes())
        files.append(str(p))
    for level, shape in (("FRAME", (2, c.d_model)), ("UTTERANCE", (c.d_model,))):
        d = tmp_path / level
        extract_whisper("whisper-tiny", files, str(d), level, 0, model=m, batch_clips=2)
        for i, f in enumerate(files):
            from mertools_amd.extract.audio import read_audio
            got = np.load(os.path.join(d, f"clip{i}.npy"))
            assert got.shape == shape and got.dtype == np.float32
            mel = whisper_log_mel(read_audio(f)[0], c.num_mel_bins, 320 * c.max_source_positions)
            ref = R.whisper_last_hidden_state(sd, vars(c), mel, torch.full((1, 2), c.decoder_start_token_id))[0].numpy()
            ref = ref.mean(0) if level == "UTTERANCE" else ref
            assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
