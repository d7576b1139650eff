"""The path a real checkpoint takes (VERDICT r4 missing #1): a LIVE HuggingFace module — the object
`AutoModel.from_pretrained(model_file)` returns at the reference's call sites
    MERBench/feature_extraction/audio/extract_audio_huggingface.py:63-69,97
    MERBench/feature_extraction/visual/extract_vision_huggingface.py:83-101,121
    MERBench/feature_extraction/text/extract_text_huggingface.py:189-199,225
— handed to `Hip*Model.from_hf(hf)` with its live config object (HubertConfig, CLIPConfig with .vision_config, RobertaConfig ...,
transformers-5 parametrised weight-norm keys, buffers, the text tower CLIP carries along) and compared with THAT module's own fp32
CPU forward in the scripts' post-processing; then the drivers BY NAME: `save_pretrained` into a temporary
PATH_TO_PRETRAINED_MODELS, `extract.audio.extract(model_name=...)` / `extract.text.extract_embedding(model_name=...)` /
`extract.visual.extract_by_name(...)` load the checkpoint themselves and write .npy files that are compared with the HF forward of
each file.  Default preset everywhere (what `from_hf` gives a user), the load-time self-check running as it does on a real checkpoint.
Tolerance: north_star's 1e-3, utterance AND frame level, max|x - ref| / max|ref|."""
import os
import wave

import numpy as np
import pytest
import torch

from oracle import hf_live as H
from oracle import weights as W
from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _perturb(m, seed):
    """HF's default init leaves every bias at 0 and every LayerNorm at identity: give the 1-D parameters non-trivial values (the
    matrices keep HF's own initialisation), so that a dropped bias or a swapped gamma / beta shows."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1 and p.numel() > 1:
                is_gain = ("norm" in n.lower() or "ln" in n.lower()) and n.endswith("weight") or "lambda" in n or "layer_scale" in n
                noise = torch.randn(p.shape, generator=g)
                p.copy_(1.0 + 0.1 * noise if is_gain else 0.05 * noise)
    return m.eval()


def _checked(m, name, may_escalate_to=()):
    """Every from_hf() object has gone through the load-time comparison with its accurate twin; these ordinary checkpoints keep the
    default preset — but for the one whose calibration batch exceeds the bar (data2vec-audio as HF initialises it: a "layer"-norm
    conv stack + post-LN blocks with small weights, FRAME 1.5e-3 under "mean"), which must have moved up the ladder BY ITSELF."""
    r = getattr(m, "self_check_result", None)
    assert r is not None, f"{name}: from_hf() did not run the load-time self-check"
    print(f"{name}: self-check {r} -> running '{m.precision}'")
    if m.escalated is None:
        assert m.precision == "mean", (name, r)
    else:
        assert m.precision in may_escalate_to, (name, r, m.escalated)


# ---- the base trio BASELINE.json names, as live HF modules carrying the synthetic base checkpoints -------------------------------
def test_from_hf_base_trio(dev):
    from mertools_amd.encoders import HipBertModel, HipCLIPModel, HipHubertModel
    hub, clip, rob = H.build_base_trio(W)
    wav, px, ids = W.synth_audio(2, 80000), W.synth_frames(8), W.synth_tokens(4, 64)
    # audio (extract_audio_huggingface.py:97-108)
    m = HipHubertModel.from_hf(hub, device=dev)
    _checked(m, "hubert-base")
    with torch.no_grad():
        feat = torch.stack(hub(wav, output_hidden_states=True).hidden_states)[[-4, -3, -2, -1]].sum(0)
    T = feat.shape[1]
    _, fr, pooled = m.forward_raw(wav.to(dev), frames=True, seg_start=[0, T], seg_len=[T, T])
    ea, eaf = rel_err(pooled.cpu(), feat.mean(1))[0], rel_err(fr.cpu().view(2, T, -1), feat)[0]
    # ... and the HF call signature itself: hidden_states tuple of L+1 [B, T, D] tensors
    hs = m(wav.to(dev), output_hidden_states=True).hidden_states
    assert len(hs) == hub.config.num_hidden_layers + 1 and tuple(hs[-1].shape) == tuple(feat.shape)
    del m
    # visual (extract_vision_huggingface.py:118-122)
    m = HipCLIPModel.from_hf(clip, device=dev)
    _checked(m, "clip-B/16")
    with torch.no_grad():
        ref = clip.get_image_features(px)
        ref = ref if torch.is_tensor(ref) else ref.pooler_output
    out = m.get_image_features(px.to(dev))
    ev, evu = rel_err(out.cpu(), ref)[0], rel_err(out.cpu().mean(0), ref.mean(0))[0]
    del m
    # text (extract_text_huggingface.py:225-231)
    m = HipBertModel.from_hf(rob, device=dev)
    _checked(m, "roberta-base")
    with torch.no_grad():
        tfeat = torch.stack(rob(input_ids=ids, attention_mask=torch.ones_like(ids), output_hidden_states=True).hidden_states)[[-4, -3, -2, -1]].sum(0)
    o = m(input_ids=ids.to(dev), attention_mask=torch.ones_like(ids), output_hidden_states=True).hidden_states
    tf = torch.stack([h.cpu() for h in o])[[-4, -3, -2, -1]].sum(0)
    et, etf = rel_err(tf[:, 1:-1].mean(1), tfeat[:, 1:-1].mean(1))[0], rel_err(tf, tfeat)[0]
    print(f"from_hf vs the live HF module: hubert-base utt={ea:.2e} frame={eaf:.2e}  clip-B/16 utt={evu:.2e} frames={ev:.2e}  "
          f"roberta-base utt={et:.2e} frame={etf:.2e}")
    assert max(ea, eaf, ev, evu, et, etf) <= TOL


# ---- every other architecture the constructors accept, as a live module of real width; the audio encoders at their real depth too
# (their saved feature is the sum of the LAST FOUR of 12 / 24 hidden states: a 4-block truncation would sum the four states right behind
# the conv stack instead, whose one-plane rounding (hs[0]: 1e-3 on a "layer"-norm front end, profiles/r05_d2v_audio_hs_errors.txt) has
# not yet been forgotten by the post-LN blocks); vision / text: 2 - 4 blocks keep the CPU forward at seconds ----
def _audio_pair(hf, dev, name, may_escalate_to=()):
    import warnings
    from mertools_amd.encoders import HipHubertModel
    wav = W.synth_audio(2, 48000, seed=31)
    with torch.no_grad():
        feat = torch.stack(hf(wav, output_hidden_states=True).hidden_states)[[-4, -3, -2, -1]].sum(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # (an escalation warns)
        m = HipHubertModel.from_hf(hf, device=dev)
    _checked(m, name, may_escalate_to)
    T = feat.shape[1]
    _, fr, pooled = m.forward_raw(wav.to(dev), frames=True, seg_start=[0, T], seg_len=[T, T])
    return rel_err(pooled.cpu(), feat.mean(1))[0], rel_err(fr.cpu().view(2, T, -1), feat)[0]


def _text_pair(hf, dev, name, vocab):
    from mertools_amd.encoders import HipBertModel
    ids = W.synth_tokens(4, 40, vocab=vocab, seed=32, bos=2, eos=3)
    with torch.no_grad():
        feat = torch.stack(hf(input_ids=ids, attention_mask=torch.ones_like(ids), output_hidden_states=True).hidden_states)[[-4, -3, -2, -1]].sum(0)
    m = HipBertModel.from_hf(hf, device=dev)
    _checked(m, name)
    _, fr, pooled = m.forward_raw(ids.to(dev), lengths=[40] * 4, frames=True, seg_start=[b * 40 + 1 for b in range(4)], seg_len=[38] * 4)
    return rel_err(pooled.cpu(), feat[:, 1:-1].mean(1))[0], rel_err(fr.cpu().view(4, 40, -1), feat)[0]


@pytest.mark.parametrize("kind", ["wav2vec2", "wav2vec2-large-style", "wavlm", "data2vec-audio", "bert", "electra", "albert",
                                  "videomae", "dinov2", "data2vec-vision", "clip-L14-style"])
def test_from_hf_other_architectures(dev, kind):
    import transformers as tr
    eager = dict(attn_implementation="eager")
    torch.manual_seed(20260926)      # HF initialises the matrices from torch's global generator: the same module on every run and box
    audio_rungs = ("mean_conv3", "mean_a2", "a2_conv3")   # an HF-initialised audio module may sit close enough to the bar to climb a rung
    if kind == "wav2vec2":
        e = _audio_pair(_perturb(tr.Wav2Vec2Model(tr.Wav2Vec2Config(mask_time_prob=0.0, **eager)), 1), dev, kind, audio_rungs)
    elif kind == "wav2vec2-large-style":   # "layer" front end with conv bias, pre-LN blocks, hidden 1024
        c = tr.Wav2Vec2Config(num_hidden_layers=24, hidden_size=1024, num_attention_heads=16, intermediate_size=4096, feat_extract_norm="layer",
                              conv_bias=True, do_stable_layer_norm=True, mask_time_prob=0.0, **eager)
        e = _audio_pair(_perturb(tr.Wav2Vec2Model(c), 2), dev, kind, audio_rungs)
    elif kind == "wavlm":
        e = _audio_pair(_perturb(tr.WavLMModel(tr.WavLMConfig(mask_time_prob=0.0)), 3), dev, kind, audio_rungs)
    elif kind == "data2vec-audio":
        e = _audio_pair(_perturb(tr.Data2VecAudioModel(tr.Data2VecAudioConfig(mask_time_prob=0.0, **eager)), 4), dev, kind,
                        may_escalate_to=audio_rungs)
    elif kind == "bert":
        e = _text_pair(_perturb(tr.BertModel(tr.BertConfig(num_hidden_layers=4, vocab_size=2000, **eager), add_pooling_layer=False), 5), dev, kind, 2000)
    elif kind == "electra":
        c = tr.ElectraConfig(num_hidden_layers=4, vocab_size=2000, embedding_size=128, hidden_size=256, num_attention_heads=4, intermediate_size=1024, **eager)
        e = _text_pair(_perturb(tr.ElectraModel(c), 6), dev, kind, 2000)
    elif kind == "albert":
        c = tr.AlbertConfig(num_hidden_layers=4, vocab_size=2000, hidden_size=768, num_attention_heads=12, intermediate_size=3072, **eager)
        e = _text_pair(_perturb(tr.AlbertModel(c, add_pooling_layer=False), 7), dev, kind, 2000)
    elif kind == "videomae":
        from mertools_amd.encoders import HipVideoMAEModel
        hf = _perturb(tr.VideoMAEModel(tr.VideoMAEConfig(num_hidden_layers=2, **eager)), 8)
        px = W.synth_video(1)
        with torch.no_grad():
            ref = hf(px).last_hidden_state
        m = HipVideoMAEModel.from_hf(hf, device=dev)
        _checked(m, kind)
        out = m(px.to(dev)).last_hidden_state
        seg = m.extract_segments(px.to(dev))
        e = (rel_err(seg.cpu(), ref.view(8, 196, -1).mean(1))[0], rel_err(out.cpu(), ref)[0])
    elif kind == "dinov2":
        from mertools_amd.encoders import HipDinov2Model
        hf = _perturb(tr.Dinov2Model(tr.Dinov2Config(num_hidden_layers=2, image_size=518, patch_size=14, **eager)), 9)
        px = W.synth_frames(4, 224, seed=33)
        with torch.no_grad():
            ref = torch.stack(hf(px, output_hidden_states=True).hidden_states)[-1].sum(dim=1)   # extract_vision_huggingface.py:141-142
        m = HipDinov2Model.from_hf(hf, device=dev)
        _checked(m, kind)
        out = m.extract_frames(px.to(dev))
        e = (rel_err(out.cpu().mean(0), ref.mean(0))[0], rel_err(out.cpu(), ref)[0])
    elif kind == "data2vec-vision":
        from mertools_amd.encoders import HipData2VecVisionModel
        c = tr.Data2VecVisionConfig(num_hidden_layers=2, use_relative_position_bias=True, use_absolute_position_embeddings=False,
                                    layer_scale_init_value=0.1, **eager)
        hf = _perturb(tr.Data2VecVisionModel(c, add_pooling_layer=False), 10)
        px = W.synth_frames(4, 224, seed=34)
        with torch.no_grad():
            ref = torch.stack(hf(px, output_hidden_states=True).hidden_states)[-1].sum(dim=1)   # extract_vision_huggingface.py:130-131
        m = HipData2VecVisionModel.from_hf(hf, device=dev)
        _checked(m, kind)
        out = m.extract_frames(px.to(dev))
        e = (rel_err(out.cpu().mean(0), ref.mean(0))[0], rel_err(out.cpu(), ref)[0])
    else:   # CLIP-L/14 wiring: patch 14 (588 -> 592 padded columns), 257 tokens, 1024 -> 768 projection
        from mertools_amd.encoders import HipCLIPModel
        c = tr.CLIPConfig(vision_config=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=3, num_attention_heads=16, patch_size=14,
                                             image_size=224), text_config=dict(num_hidden_layers=1), projection_dim=768, **eager)
        hf = _perturb(tr.CLIPModel(c), 11)
        px = W.synth_frames(4, 224, seed=35)
        with torch.no_grad():
            ref = hf.get_image_features(px)
            ref = ref if torch.is_tensor(ref) else ref.pooler_output
        m = HipCLIPModel.from_hf(hf, device=dev)
        _checked(m, kind)
        out = m.get_image_features(px.to(dev))
        e = (rel_err(out.cpu().mean(0), ref.mean(0))[0], rel_err(out.cpu(), ref)[0])
    torch.cuda.synchronize()
    print(f"from_hf[{kind}] vs the live HF module: utt={e[0]:.2e} frame={e[1]:.2e}")
    assert e[0] <= TOL and e[1] <= TOL, (kind, e)


# ---- the drivers by name: save_pretrained -> PATH_TO_PRETRAINED_MODELS/transformers/<name> -> extract(model_name=...) -> .npy ----
def _write_wav(path, x):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes((np.clip(x, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())


@pytest.fixture
def pretrained_root(tmp_path, monkeypatch):
    from mertools_amd import config
    root = tmp_path / "tools"
    (root / "transformers").mkdir(parents=True)
    monkeypatch.setattr(config, "PATH_TO_PRETRAINED_MODELS", str(root))
    return root


@pytest.mark.parametrize("level", ["UTTERANCE", "FRAME"])
def test_audio_driver_by_name(dev, tmp_path, pretrained_root, level):
    """`extract(model_name, audio_files, save_dir, feature_level, gpu)` exactly as extract_audio_huggingface.py:52-113 is called:
    the driver reads the checkpoint and the feature extractor's config from disk itself."""
    torch.manual_seed(20260927)
    import transformers as tr
    from mertools_amd.extract import audio
    name = "chinese-hubert-base"
    hf = _perturb(tr.HubertModel(tr.HubertConfig(mask_time_prob=0.0, attn_implementation="eager")), 21)
    d = str(pretrained_root / "transformers" / name)
    hf.save_pretrained(d)
    fe = tr.Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True, return_attention_mask=False)
    fe.save_pretrained(d)
    rng = np.random.RandomState(3)
    files = []
    for i, L in enumerate([24000, 30500, 16000, 24000]):
        p = str(tmp_path / f"clip{i}.wav")
        _write_wav(p, rng.randn(L) * 0.1 if i < 3 else np.zeros(L))      # the last clip is digital silence: constant rows go through the accurate twin
        files.append(p)
    save_dir = str(tmp_path / f"{name}-{level[:3]}")
    audio.extract(name, files, save_dir, level, 0)
    hf = tr.AutoModel.from_pretrained(d).eval()        # what the reference runs (:63-69)
    worst = 0.0
    for i, p in enumerate(files):
        samples, sr = audio.read_audio(p)
        iv = fe(samples, sampling_rate=sr, return_tensors="pt").input_values       # :94
        with torch.no_grad():
            feat = torch.stack(hf(iv, output_hidden_states=True).hidden_states)[[-4, -3, -2, -1]].sum(0).view(-1, 768).numpy()   # :97-100
        ref = feat.mean(0) if level == "UTTERANCE" else feat
        out = np.load(os.path.join(save_dir, f"clip{i}.npy"))
        assert out.shape == ref.shape and out.dtype == np.float32, (out.shape, ref.shape)
        e = rel_err(torch.from_numpy(out), torch.from_numpy(ref))[0]
        print(f"audio driver by name [{level}] clip{i} ({len(samples)} samples{', silent' if i == 3 else ''}): {e:.2e}")
        worst = max(worst, e)
    print(f"audio driver by name [{level}]: worst clip {worst:.2e}")
    assert worst <= TOL


def test_text_driver_by_name(dev, tmp_path, pretrained_root):
    """`extract_embedding(model_name, trans_dir, save_dir, feature_level, gpu)` as extract_text_huggingface.py:139-252 is called:
    AutoModel + AutoTokenizer(use_fast=False) from the checkpoint directory, special-token probing, per-sentence .npy."""
    torch.manual_seed(20260927)
    import pandas as pd
    import transformers as tr
    from mertools_amd.extract import text
    name = "chinese-roberta-wwm-ext"                   # a BERT-architecture checkpoint in the reference's list (:24)
    chars = list(dict.fromkeys("今天气真好我很开心难过生惊讶害怕的了是不你他她们这那有没在和也都就要会可以说看想"))
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + chars
    d = pretrained_root / "transformers" / name
    d.mkdir()
    (d / "vocab.txt").write_text("\n".join(vocab), encoding="utf-8")
    tr.BertTokenizer(str(d / "vocab.txt")).save_pretrained(str(d))
    hf = _perturb(tr.BertModel(tr.BertConfig(num_hidden_layers=4, vocab_size=len(vocab), attn_implementation="eager")), 22)
    hf.save_pretrained(str(d))
    rng = np.random.RandomState(4)
    names = [f"s{i}" for i in range(7)]
    sents = ["".join(rng.choice(chars, size=n)) for n in (5, 12, 30, 8, 19, 3, 25)]
    csv = str(tmp_path / "trans.csv")
    pd.DataFrame({"name": names, "chinese": sents, "english": ["x"] * 7}).to_csv(csv, index=False)
    tok = tr.AutoTokenizer.from_pretrained(str(d), use_fast=False)
    ref_model = tr.AutoModel.from_pretrained(str(d)).eval()
    for level in ("UTTERANCE", "FRAME"):
        text.extract_embedding(name, csv, str(tmp_path / "feat"), level, gpu=0)
        save_dir = str(tmp_path / "feat" / f"{name}-{level[:3]}")
        worst = 0.0
        for n, s in zip(names, sents):
            enc = tok(s, return_tensors="pt")
            with torch.no_grad():
                feat = torch.stack(ref_model(**enc, output_hidden_states=True).hidden_states)[[-4, -3, -2, -1]].sum(0)[0, 1:-1].numpy()   # :225-231
            ref = feat.mean(0) if level == "UTTERANCE" else feat
            out = np.load(os.path.join(save_dir, f"{n}.npy"))
            assert out.shape == ref.shape, (out.shape, ref.shape)
            worst = max(worst, rel_err(torch.from_numpy(out).float(), torch.from_numpy(ref))[0])
        print(f"text driver by name [{level}]: worst sentence {worst:.2e}")
        assert worst <= TOL


def test_visual_driver_by_name(dev, tmp_path, pretrained_root):
    """`--model_name clip-vit-base-patch32` (extract_vision_huggingface.py:18,83-122): the checkpoint is found by name, its
    architecture picks the branch, one .npy per video."""
    torch.manual_seed(20260927)
    import transformers as tr
    from mertools_amd.extract import visual
    c = tr.CLIPConfig(vision_config=dict(num_hidden_layers=4, patch_size=32, image_size=224), text_config=dict(num_hidden_layers=1),
                      projection_dim=512, attn_implementation="eager")
    hf = _perturb(tr.CLIPModel(c), 23)
    d = str(pretrained_root / "transformers" / visual.CLIP_VIT_BASE)
    hf.save_pretrained(d)
    rng = np.random.RandomState(5)
    face_dir = tmp_path / "openface_face"
    counts = {"v0": 4, "v1": 1, "v2": 6}
    vids = {}
    for vid, n in counts.items():
        (face_dir / vid).mkdir(parents=True)
        vids[vid] = rng.randint(0, 256, (n, 112, 96, 3)).astype(np.uint8)       # BGR frames, not the model's resolution
        np.save(str(face_dir / vid / f"{vid}.npy"), vids[vid])
    proc = tr.CLIPImageProcessor()                                                 # the CLIP checkpoints' preprocessor_config
    ref_model = tr.AutoModel.from_pretrained(d).eval()
    for level in ("UTTERANCE", "FRAME"):
        save_dir = str(tmp_path / f"clip-{level[:3]}")
        visual.extract_by_name(visual.CLIP_VIT_BASE, str(face_dir), save_dir, level, gpu=0)
        worst = 0.0
        for vid, n in counts.items():
            from PIL import Image
            frames = [Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])) for f in vids[vid]]      # :29-31 (BGR -> RGB)
            px = proc(images=frames, return_tensors="pt")["pixel_values"]
            with torch.no_grad():
                ref = ref_model.get_image_features(px)
                ref = (ref if torch.is_tensor(ref) else ref.pooler_output).numpy()
            if level == "UTTERANCE":
                ref = ref.mean(0) if n > 1 else ref.squeeze()
            out = np.load(os.path.join(save_dir, f"{vid}.npy"))
            worst = max(worst, rel_err(torch.from_numpy(out), torch.from_numpy(ref).view(out.shape))[0])
        print(f"visual driver by name [{level}]: worst video {worst:.2e}")
        assert worst <= TOL
