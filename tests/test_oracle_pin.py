"""Pins the CPU oracle (oracle/) before anything trusts it:
  * against golden vectors produced by the reference's own code / the HF classes it calls (tests/golden/*.npz);
  * against the live HuggingFace classes when `transformers` is importable (same weights, same inputs)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import encoders_ref as R
from oracle import fusion_ref as FR
from oracle import host_ref as HR
from oracle import weights as W
from util import rel_err

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g(name):
    return np.load(os.path.join(G, name), allow_pickle=True)


def test_encoder_oracle_matches_hf_goldens():
    g = _g("encoders_tiny_hf.npz")
    cfg = W.hubert_config("tiny")
    hs = R.hubert_hidden_states(W.hubert_state_dict(cfg, 1), vars(cfg), W.synth_audio(3, 8000, seed=5))
    assert rel_err(hs[0], torch.from_numpy(g["hubert_hs0"]))[0] < 2e-6
    assert rel_err(hs[-1], torch.from_numpy(g["hubert_hs_last"]))[0] < 2e-6
    utt = torch.stack(hs)[[-4, -3, -2, -1]].sum(0).mean(1)
    assert rel_err(utt, torch.from_numpy(g["hubert_utt"]))[0] < 2e-6
    cc = W.clip_config("tiny")
    feats = R.clip_image_features(W.clip_state_dict(cc, 3), dict(vars(cc.vision_config), projection_dim=cc.projection_dim), W.synth_frames(5, 64, seed=7))
    assert rel_err(feats, torch.from_numpy(g["clip_feats"]))[0] < 2e-6
    bc = W.bert_config("tiny")
    ids = W.synth_tokens(4, 24, vocab=300, seed=8, bos=3, eos=4)
    hs = R.bert_hidden_states(W.bert_state_dict(bc, 4), dict(vars(bc), roberta=True), ids, torch.ones_like(ids))
    f = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    assert rel_err(f, torch.from_numpy(g["roberta_frame"]))[0] < 2e-6
    assert rel_err(f[:, 1:-1].mean(1), torch.from_numpy(g["roberta_utt"]))[0] < 2e-6


def test_encoder_oracle_matches_live_hf_stable_layer_norm():
    """HuBERT-large style front end / encoder (LayerNorm convs with bias, pre-LN blocks) against the live HF class."""
    tr = pytest.importorskip("transformers")
    cfg = W.hubert_config("tiny", feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True)
    sd = W.hubert_state_dict(cfg, 2)
    hc = tr.HubertConfig(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, conv_dim=(64,) * 7,
                         num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, attn_implementation="eager",
                         feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True)
    m = tr.HubertModel(hc).eval()
    m.load_state_dict(sd, strict=False)
    wav = W.synth_audio(2, 6000, seed=9)
    with torch.no_grad():
        hs = m(wav, output_hidden_states=True).hidden_states
    ref = R.hubert_hidden_states(sd, vars(cfg), wav)
    for a, b in zip(ref, hs):
        assert rel_err(a, b)[0] < 2e-6


def test_wav2vec2_shares_the_hubert_restatement():
    """Wav2Vec2Model (chinese-wav2vec2-base/large in the reference's model list) has HuBERT's forward and key names."""
    tr = pytest.importorskip("transformers")
    cfg = W.hubert_config("tiny")
    sd = W.hubert_state_dict(cfg, 7)
    wc = tr.Wav2Vec2Config(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256, conv_dim=(64,) * 7,
                           num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, attn_implementation="eager",
                           feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False)
    m = tr.Wav2Vec2Model(wc).eval()
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and set(r.missing_keys) <= {"masked_spec_embed"}, r
    wav = W.synth_audio(2, 6000, seed=13)
    with torch.no_grad():
        hs = m(wav, output_hidden_states=True).hidden_states
    for a, b in zip(R.hubert_hidden_states(sd, vars(cfg), wav), hs):
        assert rel_err(a, b)[0] < 2e-6


def test_videomae_oracle_matches_live_hf():
    tr = pytest.importorskip("transformers")
    cfg = W.videomae_config("tiny")
    sd = W.videomae_state_dict(cfg, 5)
    hc = tr.VideoMAEConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, image_size=96,
                           num_frames=8, use_mean_pooling=False, attn_implementation="eager")
    m = tr.VideoMAEModel(hc).eval()
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    px = W.synth_video(2, 8, 96, seed=9)
    with torch.no_grad():
        ref = m(px).last_hidden_state
    assert rel_err(R.videomae_last_hidden_state(sd, vars(cfg), px), ref)[0] < 2e-6


def test_fusion_oracle_matches_reference_goldens():
    g = _g("fusion_attention.npz")
    sd = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init_")}
    xs = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("x_")}
    f, e, v = FR.attention_forward(sd, {k: x[0] for k, x in xs.items()})
    assert rel_err(f, torch.from_numpy(g["out_features"]))[0] < 1e-6
    assert rel_err(e, torch.from_numpy(g["out_emos_out"]))[0] < 1e-6
    assert rel_err(v, torch.from_numpy(g["out_vals_out"]))[0] < 1e-6
    assert g["out_interloss"].dtype == np.int64 and int(g["out_interloss"]) == 0
    losses, final, grads0 = FR.train_steps(sd, xs, torch.from_numpy(g["emos"]), torch.from_numpy(g["vals"]), len(g["losses"]))
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-6)
    for k in sd:
        assert rel_err(grads0[k], torch.from_numpy(g["grad0_" + k]))[0] < 1e-5, k
        assert rel_err(final[k], torch.from_numpy(g["final_" + k]))[0] < 1e-5, k
    gl = _g("losses.npz")
    assert abs(FR.ce_loss(torch.from_numpy(gl["pred"]), torch.from_numpy(gl["tgt"])).item() - float(gl["ce"])) < 1e-6
    assert abs(FR.mse_loss(torch.from_numpy(gl["vp"]), torch.from_numpy(gl["vt"])).item() - float(gl["mse"])) < 1e-6


def test_host_oracle_matches_reference_goldens_bit_exact():
    g = _g("index_paths.npz")
    for k in [k for k in g.files if k.startswith("resample_")]:
        _, vlen, n = k.split("_")
        assert HR.resample_indices(int(vlen), int(n)) == g[k].tolist(), k
    assert HR.visual_batches(70, 32) == g["vsplit_70_32"].tolist()
    x = np.arange(1, 26, dtype=np.float32)[None]
    assert np.array_equal(HR.audio_split(x, 10), g["asplit_25_10"])
    assert np.array_equal(HR.audio_split(x, 30), g["asplit_25_30"])
    assert np.array_equal(HR.audio_split(x[:, :20], 10), g["asplit_20_10"])
    for k in [k for k in g.files if k.startswith("mapfeat_in_")]:
        _, _, L, dst = k.split("_")
        out = HR.mapping_feature(g[k].copy(), int(dst))
        ref = g[f"mapfeat_out_{L}_{dst}"]
        assert out.shape == ref.shape and out.dtype == ref.dtype and np.array_equal(out, ref), k
    lab = os.path.join(G, "mer2023_label-6way.npz")
    for split in ["train", "test1", "test2", "test3"]:
        names, labels = HR.read_names_labels(lab, split)
        assert len(names) == int(g[f"labels_{split}_n"]) and names[:5] == g[f"labels_{split}_first_names"].tolist()
        assert [l["emo"] for l in labels] == g[f"labels_{split}_emo"].tolist()
        assert [float(l["val"]) for l in labels] == g[f"labels_{split}_val"].tolist()
    random.seed(2023)
    folds = HR.random_split_indexes(3373, 5)
    for i, (tr, ev) in enumerate(folds):
        assert np.array_equal(np.array(tr), g[f"fold{i}_train"]) and np.array_equal(np.array(ev), g[f"fold{i}_eval"])


def test_data2vec_audio_oracle_matches_hf():
    """data2vec-audio branch (extract_audio_huggingface.py:22-23): 'layer'-norm conv stack + 5 x [grouped conv k=19 -> LayerNorm
    without affine -> GELU] positional stack, post-LN blocks — the restatement against the live HF class on a tiny checkpoint."""
    import torch
    from transformers import Data2VecAudioConfig, Data2VecAudioModel
    from mertools_amd import synthetic as W
    from oracle import encoders_ref as R
    c = W.data2vec_audio_config("tiny")
    sd = W.hubert_state_dict(c, 0)
    hc = Data2VecAudioConfig(hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                             intermediate_size=c.intermediate_size, conv_dim=c.conv_dim, conv_kernel=c.conv_kernel, conv_stride=c.conv_stride,
                             conv_bias=False, num_conv_pos_embeddings=5, conv_pos_kernel_size=19,
                             num_conv_pos_embedding_groups=c.num_conv_pos_embedding_groups, layer_norm_eps=c.layer_norm_eps, hidden_dropout=0.0,
                             attention_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0, attn_implementation="eager")
    m = Data2VecAudioModel(hc).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    wav = W.synth_audio(2, 16000)
    with torch.no_grad():
        hs = m(wav, output_hidden_states=True).hidden_states
    ours = R.hubert_hidden_states(sd, vars(c), wav)
    assert len(hs) == len(ours) == c.num_hidden_layers + 1
    for a, b in zip(ours, hs):
        assert torch.allclose(a, b, rtol=0, atol=2e-5)


def test_wavlm_oracle_matches_hf():
    """WavLM branch (extract_audio_huggingface.py:33-34): gated relative position bias — bucket table, per-layer gate, additive
    score bias — in both the post-LN (base) and stable-LayerNorm (large) wirings, against the live HF class."""
    import torch
    from transformers import WavLMConfig, WavLMModel
    from mertools_amd import synthetic as W
    from oracle import encoders_ref as R
    for over in ({}, dict(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)):
        c = W.wavlm_config("tiny", **over)
        sd = W.hubert_state_dict(c, 0)
        hc = WavLMConfig(hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                         intermediate_size=c.intermediate_size, conv_dim=c.conv_dim, conv_kernel=c.conv_kernel, conv_stride=c.conv_stride,
                         conv_bias=c.conv_bias, feat_extract_norm=c.feat_extract_norm, do_stable_layer_norm=c.do_stable_layer_norm,
                         num_conv_pos_embeddings=c.num_conv_pos_embeddings, num_conv_pos_embedding_groups=c.num_conv_pos_embedding_groups,
                         layer_norm_eps=c.layer_norm_eps, hidden_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
                         mask_time_prob=0.0, num_buckets=320, max_bucket_distance=800)
        m = WavLMModel(hc).eval()
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not missing and not unexpected
        wav = W.synth_audio(2, 16000)
        with torch.no_grad():
            hs = m(wav, output_hidden_states=True).hidden_states
        ours = R.hubert_hidden_states(sd, vars(c), wav)
        assert len(hs) == len(ours)
        for a, b in zip(ours, hs):
            assert torch.allclose(a, b, rtol=0, atol=2e-5)
    # the bucket table itself, at a length that reaches the logarithmic buckets (|distance| >= 80)
    emb = torch.randn(320, 4)
    att = m.encoder.layers[0].attention
    att.rel_attn_embed.weight.data = emb[:, :att.num_heads].clone()
    ref = att.compute_bias(499, 499)
    assert torch.equal(R.wavlm_position_bias(emb[:, :att.num_heads], 499), ref)


def test_electra_and_albert_oracle_match_hf():
    """ELECTRA with embedding_size != hidden_size (`embeddings_project`) and ALBERT (factorised embeddings, one shared block,
    gelu_new) — extract_text_huggingface.py:22-28,46-48 — against the live HF classes."""
    import torch
    from transformers import AlbertConfig, AlbertModel, ElectraConfig, ElectraModel
    from mertools_amd import synthetic as W
    from oracle import encoders_ref as R
    common = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=0, attn_implementation="eager")
    c = W.electra_config("tiny")
    sd = W.bert_state_dict(c, 0)
    m = ElectraModel(ElectraConfig(vocab_size=c.vocab_size, embedding_size=c.embedding_size, hidden_size=c.hidden_size,
                                   num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                                   intermediate_size=c.intermediate_size, max_position_embeddings=c.max_position_embeddings,
                                   type_vocab_size=c.type_vocab_size, layer_norm_eps=c.layer_norm_eps, **common)).eval()
    m.load_state_dict(sd, strict=True)
    ids = W.synth_tokens(3, 20) % c.vocab_size
    with torch.no_grad():
        hs = m(ids, attention_mask=torch.ones_like(ids), output_hidden_states=True).hidden_states
    for a, b in zip(R.bert_hidden_states(sd, vars(c), ids, torch.ones_like(ids)), hs):
        assert torch.allclose(a, b, rtol=0, atol=1e-5)
    c = W.albert_config("tiny")
    sd = W.albert_state_dict(c, 0)
    m = AlbertModel(AlbertConfig(vocab_size=c.vocab_size, embedding_size=c.embedding_size, hidden_size=c.hidden_size,
                                 num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                                 intermediate_size=c.intermediate_size, max_position_embeddings=c.max_position_embeddings,
                                 type_vocab_size=c.type_vocab_size, layer_norm_eps=c.layer_norm_eps, hidden_act="gelu_new", **common),
                    add_pooling_layer=False).eval()
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        hs = m(ids, attention_mask=torch.ones_like(ids), output_hidden_states=True).hidden_states
    ours = R.bert_hidden_states(sd, vars(c), ids, torch.ones_like(ids))
    assert len(ours) == len(hs) == c.num_hidden_layers + 1
    for a, b in zip(ours, hs):
        assert torch.allclose(a, b, rtol=0, atol=1e-5)


def test_pil_bicubic_restatement_matches_pillow():
    """oracle.host_ref.pil_resize_bicubic_u8 (Pillow's 8-bit two-pass fixed-point resampler, restated) against the installed
    Pillow itself: byte-identical for down- and up-scaling, one- and two-axis resizes, tiny and large inputs."""
    from PIL import Image
    from oracle.host_ref import pil_resize_bicubic_u8
    rng = np.random.RandomState(0)
    for (h, w, nh, nw) in [(256, 320, 224, 280), (320, 256, 280, 224), (112, 112, 224, 224), (500, 375, 298, 224), (224, 224, 224, 224),
                           (97, 131, 224, 302), (480, 640, 224, 298), (224, 300, 224, 300), (30, 40, 224, 298), (64, 64, 17, 23)]:
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        img[: h // 2] = np.linspace(0, 255, w)[None, :, None].astype(np.uint8)      # a smooth half next to a noisy half
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), resample=Image.BICUBIC))
        assert np.array_equal(pil_resize_bicubic_u8(img, nw, nh), ref), (h, w, nh, nw)


def test_device_resize_tables_and_geometry():
    """The product's per-axis tables (mertools_amd/extract/resize.py, what the GPU kernels consume) equal the oracle's, and its
    shortest-edge / centre-crop geometry reproduces the host pre-processing (clip_preprocess = PIL + the processor's crop)."""
    from oracle.host_ref import pil_resample_coeffs, pil_resize_bicubic_u8
    from mertools_amd.extract.resize import pil_coeffs, shortest_edge_geometry
    from mertools_amd.extract.visual import CLIP_MEAN, CLIP_STD, clip_preprocess
    for i, o in [(320, 280), (256, 224), (112, 224), (500, 298), (224, 224), (97, 224), (640, 298), (1080, 224), (1920, 398), (225, 224), (30, 224)]:
        b, k = pil_resample_coeffs(i, o)
        b2, k2, ks = pil_coeffs(i, o)
        assert np.array_equal(b, b2) and np.array_equal(k, k2) and k.shape[1] == ks, (i, o)
    rng = np.random.RandomState(1)
    mean = np.array(CLIP_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(CLIP_STD, dtype=np.float32)[:, None, None]
    for h, w in [(256, 320), (300, 200), (100, 100), (224, 224), (231, 517)]:
        f = rng.randint(0, 256, (2, h, w, 3), dtype=np.uint8)       # BGR frames as the reader returns them
        nw, nh, left, top, crop = shortest_edge_geometry(h, w, 224)
        ref = clip_preprocess(f, 224).numpy()
        for n in range(2):
            r = pil_resize_bicubic_u8(f[n], nw, nh)[top:top + crop, left:left + crop]
            rgb = r[:, :, ::-1].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
            assert np.array_equal((rgb - mean) / std, ref[n]), (h, w)


def test_pil_bilinear_restatement_matches_pillow():
    from PIL import Image
    from oracle.host_ref import pil_resample_coeffs, pil_resize_u8
    from mertools_amd.extract.resize import pil_coeffs
    rng = np.random.RandomState(2)
    for (h, w, nh, nw) in [(256, 320, 224, 280), (112, 112, 224, 224), (500, 375, 298, 224), (97, 131, 224, 302), (480, 640, 224, 298), (64, 64, 17, 23)]:
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), resample=Image.BILINEAR))
        assert np.array_equal(pil_resize_u8(img, nw, nh, "bilinear"), ref), (h, w, nh, nw)
        for i, o in ((h, nh), (w, nw)):
            b, k = pil_resample_coeffs(i, o, "bilinear")
            b2, k2, _ = pil_coeffs(i, o, "bilinear")
            assert np.array_equal(b, b2) and np.array_equal(k, k2)


@pytest.mark.parametrize("recipe", ["clip", "videomae", "dinov2", "data2vec-vision"])
def test_device_preprocess_recipes_reproduce_the_host_processors(recipe):
    """extract.resize.recipe_geometry (what the GPU path executes: resize geometry, filter, crop window, mean / std) applied with
    the oracle's Pillow restatement == the branch's host pre-processing (PIL + processor arithmetic, pinned against HF)."""
    from oracle.host_ref import pil_resize_u8
    from mertools_amd.extract import visual
    from mertools_amd.extract.resize import recipe_geometry
    host = {"clip": lambda f: visual.clip_preprocess(f, 224), "videomae": lambda f: visual.videomae_preprocess(f, 224)[0],
            "dinov2": lambda f: visual.dinov2_preprocess(f, 224, 256), "data2vec-vision": lambda f: visual.data2vec_vision_preprocess(f, 224)}[recipe]
    rng = np.random.RandomState(3)
    for h, w in [(256, 320), (300, 200), (224, 224), (231, 517)]:
        f = rng.randint(0, 256, (2, h, w, 3), dtype=np.uint8)
        g = recipe_geometry(recipe, h, w)
        mean = np.array(g["mean"], dtype=np.float32)[:, None, None]
        std = np.array(g["std"], dtype=np.float32)[:, None, None]
        ref = host(f).numpy()
        for n in range(2):
            r = pil_resize_u8(f[n], g["new_w"], g["new_h"], g["filt"])[g["top"]:g["top"] + g["crop"], g["left"]:g["left"] + g["crop"]]
            rgb = r[:, :, ::-1].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
            assert np.array_equal((rgb - mean) / std, ref[n]), (recipe, h, w)


def test_pil_restatement_random_sweep():
    """Edge cases by brute force: sizes from 1 pixel up, extreme up- and down-scaling, both filters — the oracle against Pillow and the
    product's tables against the oracle's."""
    from PIL import Image
    from oracle.host_ref import pil_resample_coeffs, pil_resize_u8
    from mertools_amd.extract.resize import pil_coeffs
    rnd = random.Random(0)
    rng = np.random.RandomState(0)
    for _ in range(150):
        i = rnd.choice([1, 2, 3, 5, 7, 16, 31, 64, 97, 224, 225, 300, 1000])
        o = rnd.choice([1, 2, 3, 4, 9, 32, 64, 100, 224, 256, 299, 640])
        for f in ("bicubic", "bilinear"):
            b, k = pil_resample_coeffs(i, o, f)
            b2, k2, _ = pil_coeffs(i, o, f)
            assert np.array_equal(b, b2) and np.array_equal(k, k2), (i, o, f)
    for _ in range(60):
        h, w, nh, nw = rnd.randint(1, 70), rnd.randint(1, 70), rnd.randint(1, 90), rnd.randint(1, 90)
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        for f, res in (("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)):
            ref = np.asarray(Image.fromarray(img).resize((nw, nh), resample=res))
            assert np.array_equal(pil_resize_u8(img, nw, nh, f), ref), (h, w, nh, nw, f)


def test_oracle_matches_live_hf_at_base_size():
    """The oracle restatement against the live HF classes at the FULL base architectures of BASELINE.json (HuBERT-base 5 s,
    CLIP-B/16 8 frames, RoBERTa-base 64 tokens) — the same modules bench.py's cpu_baseline leg times (oracle/hf_live.py)."""
    pytest.importorskip("transformers")
    from oracle import hf_live as H
    hub, clip, rob = H.build_base_trio(W)
    wav, px, ids = W.synth_audio(1), W.synth_frames(8), W.synth_tokens(1)
    hc, cc, bc = W.hubert_config("base"), W.clip_config("base16"), W.bert_config("roberta-base")
    ra = torch.stack(R.hubert_hidden_states(W.hubert_state_dict(hc, 0), vars(hc), wav))[[-4, -3, -2, -1]].sum(0).mean(1)
    rv = R.clip_image_features(W.clip_state_dict(cc, 0), dict(vars(cc.vision_config), projection_dim=cc.projection_dim), px).mean(0, keepdim=True)
    rt = torch.stack(R.bert_hidden_states(W.bert_state_dict(bc, 0), dict(vars(bc), roberta=True), ids, torch.ones_like(ids)))[[-4, -3, -2, -1]].sum(0)[:, 1:-1].mean(1)
    assert rel_err(ra, H.audio_utt(hub, wav))[0] < 2e-6
    assert rel_err(rv, H.visual_utt(clip, px))[0] < 2e-6
    assert rel_err(rt, H.text_utt(rob, ids))[0] < 2e-6
