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


def test_read_pcm16_walks_the_riff_chunks_like_the_wave_module(tmp_path):
    """Extra chunks before `data`, an odd-sized chunk (padded to even), a data chunk that claims more bytes than the file holds."""
    import struct
    from mertools_amd.extract import audio
    pcm = (np.random.RandomState(1).randn(3001) * 3000).astype("<i2")
    fmt = struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)

    def riff(chunks):
        body = b"WAVE" + b"".join(t + struct.pack("<I", n) + d + (b"\0" if len(d) & 1 else b"") for t, n, d in chunks)
        return b"RIFF" + struct.pack("<I", len(body)) + body
    cases = {
        "list.wav": riff([(b"fmt ", 16, fmt), (b"LIST", 7, b"INFOabc"), (b"data", pcm.nbytes, pcm.tobytes())]),
        "fmt18.wav": riff([(b"fmt ", 18, fmt + b"\0\0"), (b"data", pcm.nbytes, pcm.tobytes())]),
        "short.wav": riff([(b"fmt ", 16, fmt), (b"data", pcm.nbytes + 4000, pcm.tobytes())]),
    }
    for name, blob in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        got, sr = audio.read_pcm16(p)
        with wave.open(p, "rb") as w:
            ref = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
            assert sr == w.getframerate()
        assert got.dtype == np.int16 and got.flags.writeable and np.array_equal(got, ref) and np.array_equal(got, pcm), name
    for name, blob in {"float.wav": riff([(b"fmt ", 16, struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32)), (b"data", 8, b"\0" * 8)]),
                       "nofmt.wav": riff([(b"data", 8, b"\0" * 8)]), "junk.wav": b"not a wav file at all"}.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        assert audio.read_pcm16(p) is None, name
    assert audio.read_pcm16(str(tmp_path / "missing.wav")) is None


def test_npy_save_writes_np_saves_bytes(tmp_path):
    from mertools_amd.extract.pipeline import npy_save
    rs = np.random.RandomState(0)
    big = rs.randn(6, 768).astype(np.float32)
    arrays = [big[0], big[2:5], big[1:2], np.zeros((1, 768)), np.zeros((768,)), np.zeros((0,)), np.float32(3.0), rs.randn(5, 4)[:, ::2],
              np.arange(6).reshape(2, 3).T, big[3], big.astype(np.float16)]   # (same shape twice: the cached header)
    for i, a in enumerate(arrays):
        npy_save(str(tmp_path / f"mine{i}"), a)            # no suffix: np.save appends .npy
        np.save(str(tmp_path / f"ref{i}.npy"), a)
        assert (tmp_path / f"mine{i}.npy").read_bytes() == (tmp_path / f"ref{i}.npy").read_bytes(), i
    npy_save(tmp_path / "pathlike.npy", big[0])
    assert (tmp_path / "pathlike.npy").read_bytes() == (tmp_path / "ref0.npy").read_bytes()


def test_read_into_pinned_is_np_load(tmp_path):
    from mertools_amd.extract.pipeline import read_into_pinned
    rs = np.random.RandomState(0)
    frames = rs.randint(0, 256, (5, 32, 32, 3)).astype(np.uint8)
    np.save(tmp_path / "f.npy", frames)
    t = read_into_pinned(str(tmp_path / "f.npy"), pin=False)
    assert t.dtype == torch.uint8 and np.array_equal(t.numpy(), frames)
    np.save(tmp_path / "g.npy", rs.randn(4, 3).astype(np.float32))
    assert np.array_equal(read_into_pinned(str(tmp_path / "g.npy"), pin=False).numpy(), np.load(tmp_path / "g.npy"))
    np.save(tmp_path / "fortran.npy", np.asfortranarray(rs.randn(4, 3)))
    np.save(tmp_path / "obj.npy", np.array([{"a": 1}], dtype=object), allow_pickle=True)
    open(tmp_path / "cut.npy", "wb").write((tmp_path / "f.npy").read_bytes()[:-100])
    for bad in ("fortran.npy", "obj.npy", "cut.npy", "nope.npy"):
        assert read_into_pinned(str(tmp_path / bad), pin=False) is None, bad   # the caller falls back to np.load


def test_text_driver_column_pass_equals_the_reference_row_loop(tmp_path):
    """extract_embedding reads the transcription csv as columns and tokenises the kept sentences in one call; the reference walks
    df.iterrows() and tokenises sentence by sentence (extract_text_huggingface.py:216-233).  Same sentences kept, same ids per
    sentence, same files — checked with a stand-in encoder whose output is a function of the ids it is handed."""
    import contextlib
    import io
    import pandas as pd
    import pytest
    tr = pytest.importorskip("transformers")
    from mertools_amd.extract import text
    chars = list("今天气真好你我他是的很高兴见到们吗不")
    (tmp_path / "vocab.txt").write_text("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + chars), encoding="utf-8")
    tok = tr.BertTokenizer(str(tmp_path / "vocab.txt"))
    rs = np.random.RandomState(3)
    sents = ["".join(chars[j] for j in rs.randint(0, len(chars), n)) for n in (5, 1, 17, 9, 9, 30, 2)]
    rows = [dict(name=f"c{i}", chinese=s, english="x") for i, s in enumerate(sents)]
    rows.insert(2, dict(name="empty", chinese=None, english="x"))
    rows.append(dict(name=123, chinese="?", english="x"))    # out-of-vocabulary character, numeric name
    csv = str(tmp_path / "t.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)

    class Model:
        device = torch.device("cpu")

        class config:
            pad_token_id = 0

        def __call__(self, **kw):
            out = type("O", (), {})()
            out.hidden_states = [torch.zeros(1, kw["input_ids"].shape[1], 8)]
            return out

        def extract_utterance(self, batch, lens, start, end):   # mean over the kept tokens of (id, id^2, position, ...)
            feats = []
            for r, n in enumerate(lens):
                ids = batch[r, start:n + (end or 0)].double()
                pos = torch.arange(len(ids)).double()
                feats.append(torch.stack([ids, ids * ids, pos, ids * pos] * 2, 1).mean(0) if len(ids) else torch.zeros(8).double())
            return torch.stack(feats).float()

    with contextlib.redirect_stdout(io.StringIO()):
        text.extract_embedding("stub", csv, str(tmp_path / "out"), "UTTERANCE", model=Model(), tokenizer=tok, batch_size=4, rank=0, world=1)
        start, end = text.find_start_end_pos(tok)
    out = tmp_path / "out" / "stub-UTT"
    assert sorted(os.listdir(out)) == sorted(f"{r['name']}.npy" for r in rows)
    df = pd.read_csv(csv)
    for _, row in df.iterrows():   # the reference's loop
        s = row["chinese"]
        got = np.load(out / f"{row['name']}.npy")
        if pd.isna(s) == False and len(s) > 0:  # noqa: E712
            ids = tok(s, return_tensors="pt")["input_ids"]
            want = Model().extract_utterance(ids, [ids.shape[1]], start, end)[0].numpy()
            assert got.dtype == np.float32 and np.array_equal(got, want), row["name"]
        else:
            assert got.dtype == np.float64 and got.shape == (8,) and not got.any()


def test_batch_mean_bias_is_unsound_in_huberts_conv_stack():
    """Why the "mean" preset keeps the batch-mean bias behind LayerNorms only.  Emulated on the CPU (tests/studies/mean_correction.py:
    f16 operands, fp32 accumulation, the correction as c = mean_rows(a) w_lo^T): speech-like audio — 0.3 s loud, 0.3 s 40 dB down —
    through the tiny HuBERT.  With the bias in the conv stack the quiet passages' rows (20-50x smaller than the batch mean, inputs are
    un-normalised GELU outputs) take an absolute offset they cannot absorb and the feature projection's LayerNorm magnifies it: 1e-2.
    With the conv stack corrected per row (two passes here; the MX plane in the product) the same clips are at 2e-4."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "studies"))
    import mean_correction as MC
    from mertools_amd.extract import audio
    cfg = S.hubert_config("tiny")
    sd = S.hubert_state_dict(cfg, 1)
    rng = np.random.RandomState(0)
    L = 32000
    env = np.where((np.arange(L) // 4800) % 2 == 0, 1.0, 0.01)
    iv = torch.cat([audio.wav2vec2_normalize(np.round(np.clip(rng.randn(L) * 0.1 * env, -1, 1 - 1 / 32768) * 32768) / 32768) for _ in range(2)], 0)

    def feats():
        return torch.stack(R.hubert_hidden_states(sd, vars(cfg), iv))[[-4, -3, -2, -1]].sum(0)

    def rel(a, b):
        return ((a - b).abs().max() / b.abs().max()).item()
    res = {}
    try:
        with torch.no_grad():
            ref = feats()
            for name, cm, lm in (("mean_all", "gmean", "gmean"), ("mean", "exact", "gmean"), ("fast", "none", "none")):
                def lin(x, w, b=None, lm=lm):
                    MC.MODE["corr"] = lm
                    return MC.lin(x, w, b)

                def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1, cm=cm):
                    MC.MODE["corr"] = cm
                    return MC.conv(x, w, b, stride, padding, dilation, groups)
                R.F.linear, R.F.conv1d = lin, conv
                f = feats()
                res[name] = (rel(f.mean(1), ref.mean(1)), rel(f, ref))
    finally:
        R.F.linear, R.F.conv1d = MC._lin, MC._conv   # (R.F is torch.nn.functional)
    assert res["mean_all"][0] > 5e-3, res                          # the batch-mean bias in the conv stack: an order of magnitude out
    assert res["mean"][0] < 5e-4 and res["mean"][1] < 1e-3, res    # per-row correction there, batch-mean bias behind the LayerNorms
    assert res["fast"][0] < 1e-3, res                              # (no correction at all is scale-equivariant: 6e-4)
