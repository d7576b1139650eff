"""Encoder-level parity (GPU): the HIP path through the C-ABI against the CPU oracle, same seeded
weights and inputs.  Tolerance: north_star's 1e-3 relative on the saved feature (max-norm relative:
max|x-ref| / max|ref|).  The "accurate" preset is held to 4e-4 wherever it is parametrised (tiny models, the encoders whose attention
carries a score bias and therefore keeps the f16 attention kernel) and to 1e-4 FRAME / 2e-5 UTT on the base-size HuBERT / CLIP /
RoBERTa: three-pass GEMMs (test_gemm16_three_pass_is_fp32_grade), hi + lo activation planes and, since round 4, attention on fp32
q | k | v (mer_attention_f32)."""
import os

import pytest
import torch

from oracle import encoders_ref as R
from oracle import weights as W
from util import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3
X3 = 4e-4
L2_TOL = 5e-4      # RMS error / RMS feature (norm-free; the bench line's parity_detail.rms_rel): measured, default preset, UTT / FRAME:
                   # HuBERT-base 2.7e-4 / 3.8e-4, CLIP-B/16 2.5e-4 / 4.4e-4, RoBERTa-base 2.3e-4 / 3.8e-4
DIM_TOL = 2e-2     # the WORST single feature dimension's RMS error over ITS OWN RMS across the frames (util.dim_rel; what a max-norm figure
                   # cannot see: the small-magnitude dimensions).  Measured, default preset: HuBERT-base 1.1e-2, RoBERTa-base 5.4e-3,
                   # CLIP-B/16 3.3e-3 (`accurate`: 1.4e-4 / 4.6e-5): a dimension whose RMS is 1 % of the vector's maximum carries the
                   # same absolute error as the others.  A regression guard at ~2x the measurement, not a parity claim.


def _hf_like_cfg(cfg):
    return cfg


@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("balanced", TOL), ("mx", TOL), ("mixed", 2e-3), ("fast", 2e-3)])
def test_hubert_tiny_hidden_states(dev, precision, tol):
    from mertools_amd.encoders import HipHubertModel
    cfg = W.hubert_config("tiny")
    sd = W.hubert_state_dict(cfg, 1)
    wav = W.synth_audio(3, 8000, seed=5)
    ref = R.hubert_hidden_states(sd, vars(cfg), wav)
    m = HipHubertModel(sd, cfg, device=dev, precision=precision)
    out = m(wav.to(dev), output_hidden_states=True)
    torch.cuda.synchronize()
    assert len(out.hidden_states) == cfg.num_hidden_layers + 1
    for i, (o, r) in enumerate(zip(out.hidden_states, ref)):
        assert_close(o.cpu(), r, tol, f"hubert-tiny[{precision}] hidden_states[{i}]")
    # fused last-4 sum + utterance mean == the reference post-processing (extract_audio_huggingface.py:98-108)
    feat = torch.stack(ref)[[-4, -3, -2, -1]].sum(0)
    pooled = m.extract_utterance(wav.to(dev))
    torch.cuda.synchronize()
    assert_close(pooled.cpu(), feat.mean(1), tol, f"hubert-tiny[{precision}] UTT feature")


@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("balanced", TOL)])
def test_hubert_tiny_large_style(dev, precision, tol):
    """HuBERT-large / wav2vec2-large structure: LayerNorm after every conv (with conv bias), pre-LN blocks, final
    LayerNorm only on the last state (HF:hubert/modeling_hubert.py:127-151,504-624)."""
    from mertools_amd.encoders import HipHubertModel
    cfg = W.hubert_config("tiny", feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True)
    sd = W.hubert_state_dict(cfg, 3)
    wav = W.synth_audio(2, 7000, seed=11)
    ref = R.hubert_hidden_states(sd, vars(cfg), wav)
    m = HipHubertModel(sd, cfg, device=dev, precision=precision)
    out = m(wav.to(dev), output_hidden_states=True)
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(out.hidden_states, ref)):
        # raw pre-LN residual streams of a 128-wide toy model: hold them to 2x the feature tolerance in "balanced"
        assert_close(o.cpu(), r, tol if precision == "accurate" else 2 * tol, f"hubert-tiny-large-style[{precision}] hidden_states[{i}]")
    pooled = m.extract_utterance(wav.to(dev))
    torch.cuda.synchronize()
    assert_close(pooled.cpu(), torch.stack(ref)[[-4, -3, -2, -1]].sum(0).mean(1), tol, "large-style UTT feature")


def test_clip_patch14_tiny(dev):
    """patch 14 (CLIP-L/14): C*P*P = 588 is not a multiple of 8 -> zero-padded GEMM rows."""
    from mertools_amd.encoders import HipCLIPModel
    cfg = W.clip_config("tiny", patch_size=14, image_size=56)
    sd = W.clip_state_dict(cfg, 6)
    px = W.synth_frames(3, 56, seed=12)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px)
    out = HipCLIPModel(sd, cfg, device=dev, precision="accurate").get_image_features(px.to(dev))
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, X3, "clip patch-14 image features")


def test_hubert_tiny_chunked_clip(dev):
    """>10 s clips become several batch rows whose frames are pooled together (extract_audio_huggingface.py:40-50,104-108)."""
    from mertools_amd.encoders import HipHubertModel
    cfg = W.hubert_config("tiny")
    sd = W.hubert_state_dict(cfg, 2)
    wav = W.synth_audio(5, 4000, seed=6)
    ref = torch.stack(R.hubert_hidden_states(sd, vars(cfg), wav))[[-4, -3, -2, -1]].sum(0)  # [5,T,D]
    m = HipHubertModel(sd, cfg, device=dev, precision="accurate")
    pooled = m.extract_utterance(wav.to(dev), clip_chunks=[2, 3])
    torch.cuda.synchronize()
    exp = torch.stack([ref[0:2].reshape(-1, ref.shape[-1]).mean(0), ref[2:5].reshape(-1, ref.shape[-1]).mean(0)])
    assert_close(pooled.cpu(), exp, X3, "chunked clip pooling")


@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("balanced", TOL), ("mx", TOL), ("fast", 2e-3)])
def test_clip_tiny_image_features(dev, precision, tol):
    from mertools_amd.encoders import HipCLIPModel
    cfg = W.clip_config("tiny")
    sd = W.clip_state_dict(cfg, 3)
    px = W.synth_frames(5, 64, seed=7)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px)
    m = HipCLIPModel(sd, cfg, device=dev, precision=precision)
    out = m.get_image_features(px.to(dev))
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, tol, f"clip-tiny[{precision}] image features")
    pooled = m.extract_utterance(px.to(dev), [2, 3])
    torch.cuda.synchronize()
    assert_close(pooled.cpu(), torch.stack([ref[:2].mean(0), ref[2:].mean(0)]), tol, "clip-tiny frame mean")


@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("balanced", TOL), ("mx", TOL), ("fast", 2e-3)])
@pytest.mark.parametrize("kind", ["tiny", "tiny-bert"])
def test_bert_tiny_hidden_states(dev, precision, tol, kind):
    from mertools_amd.encoders import HipBertModel
    if kind == "tiny":
        cfg = W.bert_config("tiny")
    else:
        cfg = W.bert_config("tiny", model_type="bert", pad_token_id=0, type_vocab_size=2, layer_norm_eps=1e-12)
    sd = W.bert_state_dict(cfg, 4)
    pad = cfg.pad_token_id
    ids = W.synth_tokens(4, 24, vocab=300, seed=8, bos=3, eos=4)
    lens = [24, 9, 17, 24]
    for b, n in enumerate(lens):
        ids[b, n:] = pad
    mask = (torch.arange(24)[None] < torch.tensor(lens)[:, None]).long()
    rcfg = dict(vars(cfg), roberta=(cfg.model_type == "roberta"))
    ref = R.bert_hidden_states(sd, rcfg, ids, mask)
    m = HipBertModel(sd, cfg, device=dev, precision=precision)
    out = m(input_ids=ids.to(dev), attention_mask=mask, output_hidden_states=True)
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(out.hidden_states, ref)):
        for b, n in enumerate(lens):  # padded positions are unspecified in both implementations
            assert_close(o[b, :n].cpu(), r[b, :n], tol, f"bert-{kind}[{precision}] hs[{i}] row {b}")
    # fused: last-4 sum, strip [start:end] = [1:-1], mean (extract_text_huggingface.py:226-249)
    feat = torch.stack(ref)[[-4, -3, -2, -1]].sum(0)
    exp = torch.stack([feat[b, 1:n - 1].mean(0) for b, n in enumerate(lens)])
    pooled = m.extract_utterance(ids.to(dev), lens, 1, -1)
    torch.cuda.synchronize()
    assert_close(pooled.cpu(), exp, tol, f"bert-{kind}[{precision}] UTT feature")


@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("balanced", TOL)])
def test_videomae_tiny(dev, precision, tol):
    """VideoMAE branch (extract_vision_huggingface.py:147-159): last_hidden_state and the per-segment patch mean.
    8 frames x 96^2 -> 4 x 36 = 144 tokens (single-pass attention); the 1568-token streaming path is covered by
    test_videomae_base_16frames and test_attention[...1568...]."""
    from mertools_amd.encoders import HipVideoMAEModel
    cfg = W.videomae_config("tiny")
    sd = W.videomae_state_dict(cfg, 5)
    px = W.synth_video(2, 8, 96, seed=9)
    ref = R.videomae_last_hidden_state(sd, vars(cfg), px)
    m = HipVideoMAEModel(sd, cfg, device=dev, precision=precision)
    out = m(px.to(dev)).last_hidden_state
    seg = m.extract_segments(px.to(dev))
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, tol, f"videomae-tiny[{precision}] last_hidden_state")
    nseg, per = cfg.num_frames // cfg.tubelet_size, (cfg.image_size // cfg.patch_size) ** 2
    exp = ref.view(2 * nseg, per, -1).mean(1)
    assert_close(seg.cpu(), exp, tol, f"videomae-tiny[{precision}] segment means")


def test_videomae_base_16frames(dev):
    from mertools_amd.encoders import HipVideoMAEModel
    from util import rel_err
    cfg = W.videomae_config("base", num_hidden_layers=4)   # 4 of 12 blocks keeps the CPU oracle at a few seconds; T = 1568
    sd = W.videomae_state_dict(cfg, 0)
    px = W.synth_video(1)
    ref = R.videomae_last_hidden_state(sd, vars(cfg), px)
    exp = ref.view(8, 196, -1).mean(1)
    for prec in ("mean", "balanced", "mx", "accurate"):
        m = HipVideoMAEModel(sd, cfg, device=dev, precision=prec)
        out, seg = m(px.to(dev)).last_hidden_state, m.extract_segments(px.to(dev))
        torch.cuda.synchronize()
        e, es = rel_err(out.cpu(), ref)[0], rel_err(seg.cpu(), exp)[0]
        print(f"videomae-base(4 layers)[{prec}]: last_hidden_state={e:.2e} segment-mean={es:.2e}")
        assert out.shape == (1, 1568, 768) and seg.shape == (8, 768)
        assert es <= TOL and e <= TOL    # the saved feature AND the raw last_hidden_state (measured 2-6e-4 with one-plane presets)
        del m


# ---- full-size architectures (BASELINE.json configs 2/3 and the text leg), small batch so the CPU oracle takes seconds ----
# UTT features (what the benchmark extracts and the reference saves by default) must meet 1e-3 at the
# benchmark precision; FRAME features and raw hidden states must meet it at precision="x3".
def _report(name, errs):
    print(name + ": " + "  ".join(f"{k}={v:.2e}" for k, v in errs.items()))


def test_hubert_base_5s(dev):
    from mertools_amd.encoders import HipHubertModel
    from util import dim_rel, rel_err
    cfg = W.hubert_config("base")
    sd = W.hubert_state_dict(cfg, 0)
    wav = W.synth_audio(2, 80000)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    utt = feat.mean(1)
    res = {}
    for prec in ("fast", "mixed", "balanced", "mx", "mean", "mean_a2", "balanced3", "accurate"):
        m = HipHubertModel(sd, cfg, device=dev, precision=prec)
        assert m.out_frames(80000) == 249
        hsd, fr, pooled = m.forward_raw(wav.to(dev), hidden_states=True, frames=True, seg_start=[0, 249], seg_len=[249, 249])
        torch.cuda.synchronize()
        res[prec] = dict(hs0=rel_err(hsd[0].cpu(), hs[0])[0], hs12=rel_err(hsd[-1].cpu(), hs[-1])[0],
                         frame=rel_err(fr.cpu().view(2, 249, 768), feat)[0], utt=rel_err(pooled.cpu(), utt)[0],
                         frame_l2=rel_err(fr.cpu().view(2, 249, 768), feat)[1], utt_l2=rel_err(pooled.cpu(), utt)[1],
                         frame_dim=dim_rel(fr.cpu().view(2 * 249, 768), feat))
        _report(f"hubert-base[{prec}]", res[prec])
        del m
    assert res["balanced"]["utt"] <= TOL, res
    assert res["mx"]["utt"] <= TOL and res["mx"]["frame"] <= TOL, res     # the MX-corrected kernel whatever the row count (round 4)
    assert res["mean"]["utt"] <= TOL and res["mean"]["frame"] <= TOL, res  # the default preset: one pass + per-sequence correction table
    # norm-free companions (VERDICT r4 #2c): RMS error over RMS feature, and the worst single feature dimension's over its own RMS
    assert res["mean"]["utt_l2"] <= L2_TOL and res["mean"]["frame_l2"] <= L2_TOL and res["mean"]["frame_dim"] <= DIM_TOL, res
    # round 5: hi + lo activation planes, two passes + the table, f16 attention: the first rung of the self-check's ladder
    assert res["mean_a2"]["utt"] <= TOL and res["mean_a2"]["frame"] <= TOL and res["mean_a2"]["frame"] <= res["mean"]["frame"], res
    # three passes + fp32 attention (mer_attention_f32, round 4): no operand of a block is a single 16-bit plane any more
    assert res["accurate"]["utt"] <= 2e-5 and res["accurate"]["frame"] <= 1e-4 and res["accurate"]["hs12"] <= 1e-4, res


def test_clip_base16_8frames(dev):
    from mertools_amd.encoders import HipCLIPModel
    from util import dim_rel, rel_err
    cfg = W.clip_config("base16")
    sd = W.clip_state_dict(cfg, 0)
    px = W.synth_frames(8)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px)
    res = {}
    for prec in ("fast", "balanced", "mx", "mean", "mean_a2", "accurate"):
        m = HipCLIPModel(sd, cfg, device=dev, precision=prec)
        out = m.get_image_features(px.to(dev))
        pooled = m.extract_utterance(px.to(dev), [8])
        torch.cuda.synchronize()
        res[prec] = dict(frames=rel_err(out.cpu(), ref)[0], utt=rel_err(pooled.cpu(), ref.mean(0, keepdim=True))[0],
                         frames_l2=rel_err(out.cpu(), ref)[1], utt_l2=rel_err(pooled.cpu(), ref.mean(0, keepdim=True))[1], frames_dim=dim_rel(out.cpu(), ref))
        _report(f"clip-B/16[{prec}]", res[prec])
        del m
    assert res["balanced"]["utt"] <= TOL, res
    assert res["mx"]["utt"] <= TOL and res["mx"]["frames"] <= TOL, res   # 1576 rows: every block GEMM runs the MX kernel
    assert res["mean"]["utt"] <= TOL and res["mean"]["frames"] <= TOL, res  # one pass + per-frame mean-token correction
    assert res["mean"]["utt_l2"] <= L2_TOL and res["mean"]["frames_l2"] <= L2_TOL and res["mean"]["frames_dim"] <= DIM_TOL, res
    assert res["mean_a2"]["utt"] <= TOL and res["mean_a2"]["frames"] <= TOL, res
    assert res["accurate"]["frames"] <= 1e-4 and res["accurate"]["utt"] <= 2e-5, res


def test_roberta_base_64tok(dev):
    from mertools_amd.encoders import HipBertModel
    from util import dim_rel, rel_err
    cfg = W.bert_config("roberta-base")
    sd = W.bert_state_dict(cfg, 0)
    ids = W.synth_tokens(4, 64)
    ref = R.bert_hidden_states(sd, dict(vars(cfg), roberta=True), ids, torch.ones_like(ids))
    feat = torch.stack(ref)[[-4, -3, -2, -1]].sum(0)
    res = {}
    for prec in ("fast", "balanced", "mx", "mean", "mean_a2", "accurate"):
        m = HipBertModel(sd, cfg, device=dev, precision=prec)
        hs, fr, pooled = m.forward_raw(ids.to(dev), lengths=[64] * 4, hidden_states=True, frames=True,
                                       seg_start=[b * 64 + 1 for b in range(4)], seg_len=[62] * 4)
        torch.cuda.synchronize()
        res[prec] = dict(hs12=rel_err(hs[-1].cpu(), ref[-1])[0], frame=rel_err(fr.cpu().view(4, 64, 768), feat)[0],
                         utt=rel_err(pooled.cpu(), feat[:, 1:-1].mean(1))[0], frame_l2=rel_err(fr.cpu().view(4, 64, 768), feat)[1],
                         utt_l2=rel_err(pooled.cpu(), feat[:, 1:-1].mean(1))[1], frame_dim=dim_rel(fr.cpu().view(256, 768), feat))
        _report(f"roberta-base[{prec}]", res[prec])
        del m
    assert res["balanced"]["utt"] <= TOL, res
    assert res["mx"]["utt"] <= TOL and res["mx"]["frame"] <= TOL, res      # 256 rows: still the MX-corrected kernel (a clip alone == its row of 64)
    assert res["mean"]["utt"] <= TOL and res["mean"]["frame"] <= TOL, res  # the default preset
    assert res["mean"]["utt_l2"] <= L2_TOL and res["mean"]["frame_l2"] <= L2_TOL and res["mean"]["frame_dim"] <= DIM_TOL, res
    assert res["mean_a2"]["utt"] <= TOL and res["mean_a2"]["frame"] <= TOL, res
    assert res["accurate"]["frame"] <= 1e-4 and res["accurate"]["utt"] <= 2e-5, res


# ---- the kernel selection bench.py times (B = 64): M >= 1024 rows puts every block GEMM on the 256x256 tiles — one-pass
# [Q|K], MX-corrected V / attention output / post-LN FFN, pre-blocked weight planes.  Batches just large enough to cross
# that threshold keep the CPU oracle at seconds; a clip's features do not depend on its batch mates, so each row is
# compared with the oracle's row.  `heavy`: log-normal outlier weights (synthetic.heavy_tailed) — what a pretrained
# checkpoint's outlier channels do to 16-bit weight rounding and the fp4 residual plane.
@pytest.mark.parametrize("heavy", [False, True])
def test_hubert_base_bench_tiles(dev, heavy):
    from mertools_amd.encoders import HipHubertModel
    from util import rel_err
    cfg = W.hubert_config("base")
    sd = W.hubert_state_dict(cfg, 0)
    if heavy:   # "outliers": 3 LayerNorm-gamma channels x 30-100 (activation outliers); True: log-normal outlier weights
        sd = W.ln_outliers(sd) if heavy == "outliers" else W.heavy_tailed(sd)
    B = 8                                   # M = 8 * 249 = 1992 rows
    wav = W.synth_audio(B, 80000, seed=4321)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    utt = feat.mean(1)
    res = {}
    for prec in ("mx", "mean", "balanced"):
        m = HipHubertModel(sd, cfg, device=dev, precision=prec)
        _, fr, pooled = m.forward_raw(wav.to(dev), frames=True, seg_start=[b * 249 for b in range(B)], seg_len=[249] * B)
        torch.cuda.synchronize()
        res[prec] = dict(frame=rel_err(fr.cpu().view(B, 249, 768), feat)[0], utt=rel_err(pooled.cpu(), utt)[0],
                         utt_worst_clip=max(rel_err(pooled[b].cpu(), utt[b])[0] for b in range(B)))
        _report(f"hubert-base B={B} heavy={heavy} [{prec}]", res[prec])
        del m
    for prec in res:
        assert res[prec]["utt"] <= TOL and res[prec]["utt_worst_clip"] <= TOL, res
        assert res[prec]["frame"] <= TOL, res     # FRAME features (feature_level=FRAME saves them): the same 1e-3 bar


@pytest.mark.parametrize("heavy", [False, True])
def test_roberta_base_bench_tiles(dev, heavy):
    from mertools_amd.encoders import HipBertModel
    from util import rel_err
    cfg = W.bert_config("roberta-base")
    sd = W.bert_state_dict(cfg, 0)
    if heavy:
        sd = W.ln_outliers(sd) if heavy == "outliers" else W.heavy_tailed(sd)
    B = 16                                  # M = 16 * 64 = 1024 rows
    ids = W.synth_tokens(B, 64, seed=4322)
    ref = R.bert_hidden_states(sd, dict(vars(cfg), roberta=True), ids, torch.ones_like(ids))
    feat = torch.stack(ref)[[-4, -3, -2, -1]].sum(0)
    utt = feat[:, 1:-1].mean(1)
    res = {}
    for prec in ("mx", "mean", "balanced"):
        m = HipBertModel(sd, cfg, device=dev, precision=prec)
        _, fr, pooled = m.forward_raw(ids.to(dev), lengths=[64] * B, frames=True, seg_start=[b * 64 + 1 for b in range(B)], seg_len=[62] * B)
        torch.cuda.synchronize()
        res[prec] = dict(frame=rel_err(fr.cpu().view(B, 64, 768), feat)[0], utt=rel_err(pooled.cpu(), utt)[0],
                         utt_worst_clip=max(rel_err(pooled[b].cpu(), utt[b])[0] for b in range(B)))
        _report(f"roberta-base B={B} heavy={heavy} [{prec}]", res[prec])
        del m
    for prec in res:
        assert res[prec]["utt"] <= TOL and res[prec]["utt_worst_clip"] <= TOL, res
        assert res[prec]["frame"] <= TOL, res


def test_clip_base16_heavy_tailed(dev):
    """CLIP-B/16, 8 frames (1576 rows: the bench's kernels) with outlier weights."""
    from mertools_amd.encoders import HipCLIPModel
    from util import rel_err
    cfg = W.clip_config("base16")
    sd = W.heavy_tailed(W.clip_state_dict(cfg, 0))
    px = W.synth_frames(8, seed=4323)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px)
    for prec in ("mx", "mean"):
        m = HipCLIPModel(sd, cfg, device=dev, precision=prec)
        out = m.get_image_features(px.to(dev))
        pooled = m.extract_utterance(px.to(dev), [8])
        torch.cuda.synchronize()
        e, eu = rel_err(out.cpu(), ref)[0], rel_err(pooled.cpu(), ref.mean(0, keepdim=True))[0]
        print(f"clip-B/16 heavy-tailed [{prec}]: frames={e:.2e} utt={eu:.2e}")
        assert eu <= TOL and e <= TOL
        del m


def test_clip_base16_48frames_full_tile_selection(dev):
    """48 frames = 9456 rows: EVERY block GEMM takes the 256x256 kernel (37 x 3 = 111 tiles for the N = 768 ones: above the 96-tile
    switch to 128x128), i.e. the bench's kernel selection for the CLIP tower including the fp32 + residual epilogues."""
    from mertools_amd.encoders import HipCLIPModel
    from util import rel_err
    cfg = W.clip_config("base16")
    sd = W.clip_state_dict(cfg, 0)
    px = W.synth_frames(48, seed=4325)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px)
    m = HipCLIPModel(sd, cfg, device=dev)
    out = m.get_image_features(px.to(dev))
    pooled = m.extract_utterance(px.to(dev), [8] * 6)
    torch.cuda.synchronize()
    e, eu = rel_err(out.cpu(), ref)[0], rel_err(pooled.cpu(), ref.view(6, 8, -1).mean(1))[0]
    print(f"clip-B/16 48 frames [mean]: frames={e:.2e} utt={eu:.2e}")
    assert eu <= TOL and e <= TOL


def test_clip_base32_frames(dev):
    """CLIP-ViT-B/32 (extract_vision_huggingface.py:64-72 model list): 50 tokens per frame -> the short-T attention
    instantiation; 32 frames = 1600 rows so the block GEMMs take the 256x256 kernels."""
    from mertools_amd.encoders import HipCLIPModel
    from util import rel_err
    cfg = W.clip_config("base16", patch_size=32)
    sd = W.clip_state_dict(cfg, 0)
    px = W.synth_frames(32, seed=4324)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px)
    for prec in ("mx", "accurate"):
        m = HipCLIPModel(sd, cfg, device=dev, precision=prec)
        out = m.get_image_features(px.to(dev))
        pooled = m.extract_utterance(px.to(dev), [8] * 4)
        torch.cuda.synchronize()
        e, eu = rel_err(out.cpu(), ref)[0], rel_err(pooled.cpu(), ref.view(4, 8, -1).mean(1))[0]
        print(f"clip-B/32[{prec}]: frames={e:.2e} utt={eu:.2e}")
        assert out.shape == (32, 512)
        assert eu <= (TOL if prec == "mx" else X3) and e <= TOL
        del m


# ---- speech-like dynamics: loud and quiet passages in one clip.  Stationary noise (every other audio test here) cannot see a
# correction that is not scale-equivariant; this can (tests/test_round3_cpu.py::test_batch_mean_bias_is_unsound_in_huberts_conv_stack).
@pytest.mark.parametrize("size", ["tiny", "base"])
def test_hubert_loud_and_quiet_passages(dev, size):
    from mertools_amd.encoders import HipHubertModel
    from util import rel_err
    cfg = W.hubert_config(size)
    sd = W.hubert_state_dict(cfg, 1)
    g = torch.Generator().manual_seed(11)
    B, L = 2, 32000
    env = torch.where((torch.arange(L) // 4800) % 2 == 0, 1.0, 0.01)       # 0.3 s loud, 0.3 s 40 dB down
    wav = torch.randn(B, L, generator=g) * 0.1 * env
    wav = (wav - wav.mean(1, keepdim=True)) / torch.sqrt(wav.var(1, unbiased=False, keepdim=True) + 1e-7)
    feat = torch.stack(R.hubert_hidden_states(sd, vars(cfg), wav))[[-4, -3, -2, -1]].sum(0)
    T = feat.shape[1]
    res = {}
    for prec in ("mean", "mx", "mean_all"):
        m = HipHubertModel(sd, cfg, device=dev, precision=prec)
        _, fr, pooled = m.forward_raw(wav.to(dev), frames=True, seg_start=[b * T for b in range(B)], seg_len=[T] * B)
        torch.cuda.synchronize()
        res[prec] = (rel_err(pooled.cpu(), feat.mean(1))[0], rel_err(fr.cpu().view(B, T, -1), feat)[0])
        print(f"hubert-{size} loud/quiet [{prec}]: utt={res[prec][0]:.2e} frame={res[prec][1]:.2e}")
        del m
    for prec in ("mean", "mx"):
        assert res[prec][0] <= TOL, res
        assert res[prec][1] <= (TOL if size == "base" else 1.5e-3), res   # (tiny model, one-plane activations: the per-frame maximum sits at 1e-3)
    assert res["mean_all"][0] > 2 * res["mean"][0], res   # the batch-mean bias in the conv stack is what this test is here to keep out


# ---- ragged audio batches: clips of different lengths in ONE batch, each equal to its batch-of-one forward (the reference
# runs batch 1 and never pads or masks audio: extract_audio_huggingface.py:93-100) ----
@pytest.mark.parametrize("style", ["tiny-base", "tiny-large", "tiny-wavlm", "base", "large"])
def test_hubert_ragged_batch(dev, style):
    from mertools_amd.encoders import HipHubertModel
    from util import rel_err
    if style == "tiny-base":
        cfg, lens = W.hubert_config("tiny"), [8000, 3001, 5555, 7999, 4000, 6123, 2000, 7000]
    elif style == "tiny-large":
        cfg, lens = W.hubert_config("tiny", feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True), [8000, 3001, 5555, 7999, 4000, 6123, 2000, 7000]
    elif style == "tiny-wavlm":
        cfg, lens = W.wavlm_config("tiny"), [8000, 3001, 5555, 7999]
    elif style == "base":   # 8 clips of 8 different lengths, 2.1 .. 5 s: 1992 rows -> the 256x256 kernels
        cfg, lens = W.hubert_config("base"), [80000, 33791, 52345, 79999, 41000, 66123, 71717, 60000]
    else:
        cfg, lens = W.hubert_config("large", num_hidden_layers=6), [48000, 20011, 33333, 40960]
    sd = W.hubert_state_dict(cfg, 7)
    B, L = len(lens), max(lens)
    g = torch.Generator().manual_seed(77)
    clips = []
    for n in lens:   # each clip normalised on its own, as the feature extractor does per utterance
        w = 0.1 * torch.randn(1, n, generator=g)
        clips.append((w - w.mean()) / torch.sqrt(w.var(unbiased=False) + 1e-7))
    batch = torch.zeros(B, L)
    for b, w in enumerate(clips):
        batch[b, :lens[b]] = w[0]
    m = HipHubertModel(sd, cfg, device=dev, precision=os.environ.get("MER_TEST_PRECISION", "mean") if style in ("base", "large") else "accurate")
    tol = TOL if style in ("base", "large") else X3
    T = m.out_frames(L)
    starts, seglens = m.clip_segments(L, [1] * B, lens)
    _, fr, pooled = m.forward_raw(batch.to(dev), frames=True, seg_start=starts, seg_len=seglens, valid_samples=lens)
    torch.cuda.synchronize()
    fr = fr.cpu().view(B, T, -1)
    worst = {"utt": 0.0, "frame": 0.0}
    for b, w in enumerate(clips):
        hs = R.hubert_hidden_states(sd, vars(cfg), w)          # the reference's batch-of-one forward of this clip
        feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)[0]       # [T_b, D]
        Tb = feat.shape[0]
        assert Tb == seglens[b] == m.out_frames(lens[b])
        worst["utt"] = max(worst["utt"], rel_err(pooled[b].cpu(), feat.mean(0))[0])
        worst["frame"] = max(worst["frame"], rel_err(fr[b, :Tb], feat)[0])
    print(f"hubert ragged [{style}]: worst clip utt={worst['utt']:.2e} frame={worst['frame']:.2e}")
    assert worst["utt"] <= tol, worst
    assert worst["frame"] <= (TOL if style in ("base", "large") else tol), worst
    # padding must not leak: the same clips in a batch padded 1000 samples further give the same features
    batch2 = torch.zeros(B, L + 1000)
    batch2[:, :L] = batch
    starts2, seglens2 = m.clip_segments(L + 1000, [1] * B, lens)
    _, _, pooled2 = m.forward_raw(batch2.to(dev), seg_start=starts2, seg_len=seglens2, valid_samples=lens)
    torch.cuda.synchronize()
    assert seglens2 == seglens
    assert rel_err(pooled2.cpu(), pooled.cpu())[0] <= (1e-5 if style.startswith("tiny") else 2e-4)


# ---- BASELINE.json configs[4]: the large trio (HuBERT-large, VideoMAE-L, RoBERTa-large) — full depth, batch 1 ----
def test_large_trio(dev):
    from mertools_amd.encoders import HipBertModel, HipHubertModel, HipVideoMAEModel
    from util import rel_err
    res = {}
    cfg = W.hubert_config("large")
    sd = W.hubert_state_dict(cfg, 0)
    wav = W.synth_audio(1, 80000)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    utt = torch.stack(hs)[[-4, -3, -2, -1]].sum(0).mean(1)
    for prec, dtype in (("balanced", "f16"), ("mx", "f16"), ("mean", "f16"), ("accurate", "bf16")):
        m = HipHubertModel(sd, cfg, device=dev, precision=prec, dtype=dtype)
        res[f"hubert-large[{prec},{dtype}]"] = rel_err(m.extract_utterance(wav.to(dev)).cpu(), utt)[0]
        del m
    del sd, hs
    cfg = W.videomae_config("large")
    sd = W.videomae_state_dict(cfg, 0)
    px = W.synth_video(1)
    exp = R.videomae_last_hidden_state(sd, vars(cfg), px).view(8, 196, -1).mean(1)
    for prec, dtype in (("balanced", "f16"), ("mx", "f16"), ("mean", "f16"), ("accurate", "bf16")):
        m = HipVideoMAEModel(sd, cfg, device=dev, precision=prec, dtype=dtype)
        res[f"videomae-large[{prec},{dtype}]"] = rel_err(m.extract_segments(px.to(dev)).cpu(), exp)[0]
        del m
    del sd
    cfg = W.bert_config("roberta-large")
    sd = W.bert_state_dict(cfg, 0)
    B = 16                                  # M = 1024 rows: the 256x256 kernels the bench times
    ids = W.synth_tokens(B, 64)
    ref = torch.stack(R.bert_hidden_states(sd, dict(vars(cfg), roberta=True), ids, torch.ones_like(ids)))[[-4, -3, -2, -1]].sum(0)[:, 1:-1].mean(1)
    for prec, dtype in (("balanced", "f16"), ("mx", "f16"), ("mean", "f16")):
        m = HipBertModel(sd, cfg, device=dev, precision=prec, dtype=dtype)
        res[f"roberta-large[{prec},{dtype}]"] = rel_err(m.extract_utterance(ids.to(dev), [64] * B, 1, -1).cpu(), ref)[0]
        del m
    torch.cuda.synchronize()
    print("large trio: " + "  ".join(f"{k}={v:.2e}" for k, v in res.items()))
    # configs[4] is served with f16 MFMA operands (same MFMA rate as bf16 on gfx950, three more mantissa bits).
    # bf16 planes stay available: 3-pass bf16 ("accurate", attention on fp32 q | k | v) meets the bar for all three large encoders
    # (hubert-large 9e-6, videomae-large 6e-6 above, roberta-large 2e-5 in test_roberta_large_bf16_accurate).
    for k, v in res.items():
        assert v <= TOL, (k, v)


def test_roberta_large_bf16_accurate(dev):
    """north_star names bf16 for the large trio.  Until round 4 this was an expected failure (1.2e-3: the attention kernel's bf16
    q | k | v | P planes, 8 bits of mantissa); with attention on fp32 operands (mer_attention_f32) the three-pass bf16 preset holds
    the bar by two orders of magnitude.  (configs[4] is still SERVED with f16 operands under the one-pass preset: DESIGN.md §4.)"""
    from mertools_amd.encoders import HipBertModel
    from util import rel_err
    cfg = W.bert_config("roberta-large")
    sd = W.bert_state_dict(cfg, 0)
    ids = W.synth_tokens(2, 64)
    ref = torch.stack(R.bert_hidden_states(sd, dict(vars(cfg), roberta=True), ids, torch.ones_like(ids)))[[-4, -3, -2, -1]].sum(0)[:, 1:-1].mean(1)
    m = HipBertModel(sd, cfg, device=dev, precision="accurate", dtype="bf16")
    e = rel_err(m.extract_utterance(ids.to(dev), [64, 64], 1, -1).cpu(), ref)[0]
    print(f"roberta-large[accurate,bf16]={e:.2e}")
    assert e <= 1e-4


# ---- data2vec-audio: 5-layer positional conv stack on the HuBERT engine ----
@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("mx", TOL), ("mean", TOL)])
def test_data2vec_audio_tiny(dev, precision, tol):
    from mertools_amd.encoders import HipData2VecAudioModel
    cfg = W.data2vec_audio_config("tiny")
    sd = W.hubert_state_dict(cfg, 1)
    wav = W.synth_audio(3, 16000)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    m = HipData2VecAudioModel(sd, cfg, device=dev, precision=precision)
    out = m(wav.to(dev), output_hidden_states=True).hidden_states
    pooled = m.extract_utterance(wav.to(dev))
    torch.cuda.synchronize()
    assert len(out) == len(hs)
    for i, (o, r) in enumerate(zip(out, hs)):
        # raw hidden states of a 128-wide toy model after 5 conv+LayerNorm layers: 2x the feature tolerance outside "accurate"
        assert_close(o.cpu(), r, tol if precision == "accurate" else 2 * tol, f"data2vec-audio-tiny[{precision}] hidden_states[{i}]")
    assert_close(pooled.cpu(), torch.stack(hs)[[-4, -3, -2, -1]].sum(0).mean(1), tol, f"data2vec-audio-tiny[{precision}] UTT feature")


def test_data2vec_audio_base_5s(dev):
    from mertools_amd.encoders import HipData2VecAudioModel
    from util import rel_err
    cfg = W.data2vec_audio_config("base")
    sd = W.hubert_state_dict(cfg, 0)
    wav = W.synth_audio(2, 80000)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    utt = feat.mean(1)
    T = feat.shape[1]
    for prec in ("mean", "mx", "accurate"):     # "mean" = the constructors' default, what from_hf() and the drivers give a user
        m = HipData2VecAudioModel(sd, cfg, device=dev, precision=prec)
        _, fr, pooled = m.forward_raw(wav.to(dev), frames=True, seg_start=[0, T], seg_len=[T, T])
        torch.cuda.synchronize()
        e, ef = rel_err(pooled.cpu(), utt)[0], rel_err(fr.cpu().view(2, T, -1), feat)[0]
        print(f"data2vec-audio-base[{prec}]: utt={e:.2e} frame={ef:.2e}")
        assert e <= (X3 if prec == "accurate" else TOL) and ef <= (X3 if prec == "accurate" else TOL)
        del m


# ---- WavLM: gated relative position bias in every attention ----
@pytest.mark.parametrize("style", ["base", "large"])
@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("mx", TOL), ("mean", TOL)])
def test_wavlm_tiny(dev, precision, tol, style):
    from mertools_amd.encoders import HipWavLMModel
    over = {} if style == "base" else dict(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
    cfg = W.wavlm_config("tiny", **over)
    sd = W.hubert_state_dict(cfg, 2)
    wav = W.synth_audio(3, 16000)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    m = HipWavLMModel(sd, cfg, device=dev, precision=precision)
    out = m(wav.to(dev), output_hidden_states=True).hidden_states
    pooled = m.extract_utterance(wav.to(dev))
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(out, hs)):
        assert_close(o.cpu(), r, tol if precision == "accurate" else 2 * tol, f"wavlm-tiny-{style}[{precision}] hidden_states[{i}]")
    assert_close(pooled.cpu(), torch.stack(hs)[[-4, -3, -2, -1]].sum(0).mean(1), tol, f"wavlm-tiny-{style}[{precision}] UTT feature")


def test_wavlm_base_5s(dev):
    from mertools_amd.encoders import HipWavLMModel
    from util import rel_err
    cfg = W.wavlm_config("base")
    sd = W.hubert_state_dict(cfg, 0)
    wav = W.synth_audio(2, 80000)
    hs = R.hubert_hidden_states(sd, vars(cfg), wav)
    feat = torch.stack(hs)[[-4, -3, -2, -1]].sum(0)
    utt = feat.mean(1)
    T = feat.shape[1]
    for prec in ("mean", "mx", "accurate"):     # "mean" = the constructors' default, what from_hf() and the drivers give a user
        m = HipWavLMModel(sd, cfg, device=dev, precision=prec)
        _, fr, pooled = m.forward_raw(wav.to(dev), frames=True, seg_start=[0, T], seg_len=[T, T])
        torch.cuda.synchronize()
        e, ef = rel_err(pooled.cpu(), utt)[0], rel_err(fr.cpu().view(2, T, -1), feat)[0]
        print(f"wavlm-base[{prec}]: utt={e:.2e} frame={ef:.2e}")
        assert e <= (X3 if prec == "accurate" else TOL) and ef <= (X3 if prec == "accurate" else TOL)
        del m


# ---- ELECTRA (embedding projection) and ALBERT (shared block, gelu_new) on the BERT engine ----
@pytest.mark.parametrize("kind", ["electra", "albert"])
@pytest.mark.parametrize("precision,tol", [("accurate", X3), ("mx", TOL), ("mean", TOL)])
def test_electra_albert_tiny(dev, precision, tol, kind):
    from mertools_amd.encoders import HipBertModel
    if kind == "electra":
        cfg = W.electra_config("tiny")
        sd = W.bert_state_dict(cfg, 5)
    else:
        cfg = W.albert_config("tiny")
        sd = W.albert_state_dict(cfg, 5)
    ids = W.synth_tokens(4, 24) % cfg.vocab_size
    ref = R.bert_hidden_states(sd, vars(cfg), ids, torch.ones_like(ids))
    m = HipBertModel(sd, cfg, device=dev, precision=precision)
    out = m(input_ids=ids.to(dev), attention_mask=torch.ones_like(ids).to(dev), output_hidden_states=True).hidden_states
    pooled = m.extract_utterance(ids.to(dev), [24] * 4, 1, -1)
    torch.cuda.synchronize()
    assert len(out) == len(ref)
    for i, (o, r) in enumerate(zip(out, ref)):
        assert_close(o.cpu(), r, tol if precision == "accurate" else 2 * tol, f"{kind}-tiny[{precision}] hs[{i}]")
    assert_close(pooled.cpu(), torch.stack(ref)[[-4, -3, -2, -1]].sum(0)[:, 1:-1].mean(1), tol, f"{kind}-tiny[{precision}] UTT feature")


def test_clip_large14_frames(dev):
    """CLIP-ViT-L/14 (the AffectGPT visual front-end): 24 pre-LN layers, patch 14 (588 -> 592 padded columns), 257 tokens,
    1024 -> 768 projection; the default preset runs Q/K and the FFN without the weight-residual correction here."""
    from mertools_amd.encoders import HipCLIPModel
    from util import rel_err
    cfg = W.clip_config("large14")
    sd = W.clip_state_dict(cfg, 0)
    px = W.synth_frames(5)
    ref = R.clip_image_features(sd, dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim), px)
    for prec in ("mean", "mx", "accurate"):
        m = HipCLIPModel(sd, cfg, device=dev, precision=prec)
        out = m.get_image_features(px.to(dev))
        torch.cuda.synchronize()
        e, eu = rel_err(out.cpu(), ref)[0], rel_err(out.cpu().mean(0), ref.mean(0))[0]
        print(f"clip-L/14[{prec}]: frames={e:.2e} utt={eu:.2e}")
        assert out.shape == (5, 768)
        assert eu <= TOL and e <= TOL
        del m


# ---- activation outliers ("a few LayerNorm-gamma / residual channels 30-100x the rest") --------------------------------
# synthetic.ln_outliers scales 3 channels of every block LayerNorm's gamma / beta by 30-100x and divides the matching input columns of
# the Linear layers that read it: massive activation channels against tiny weight columns, the Linear outputs of the unperturbed
# network.  In the pre-LN CLIP tower that is an exact re-parametrisation (the oracle's features do not move), so it isolates what the
# kernels do with such planes; in the post-LN encoders the outlier channels also ride the residual stream (as massive activations do).
def test_activation_outliers_clip(dev):
    from mertools_amd.encoders import HipCLIPModel
    from util import rel_err
    cfg = W.clip_config("base16")
    sd0 = W.clip_state_dict(cfg, 0)
    sd = W.ln_outliers(sd0)
    px = W.synth_frames(8, seed=4324)
    vcfg = dict(vars(cfg.vision_config), projection_dim=cfg.projection_dim)
    ref = R.clip_image_features(sd, vcfg, px)
    assert rel_err(ref, R.clip_image_features(sd0, vcfg, px))[0] < 1e-5     # the re-parametrisation is exact for the fp32 oracle
    for prec in ("mean", "mx", "balanced"):
        m = HipCLIPModel(sd, cfg, device=dev, precision=prec, self_check=False)
        out = m.get_image_features(px.to(dev))
        pooled = m.extract_utterance(px.to(dev), [8])
        torch.cuda.synchronize()
        e, eu = rel_err(out.cpu(), ref)[0], rel_err(pooled.cpu(), ref.mean(0, keepdim=True))[0]
        print(f"clip-B/16 activation outliers [{prec}]: frames={e:.2e} utt={eu:.2e}")
        assert eu <= TOL and e <= TOL, (prec, e, eu)
        del m
    # the default constructor: the load-time self-check sees the outlier LayerNorm channels, compares with `accurate` on its calibration
    # batch, finds the pre-LN tower unharmed and keeps the fast preset
    m = HipCLIPModel(sd, cfg, device=dev)
    assert m.escalated is None and m.self_check_result["ln_outlier_ratio"] > 8 and m.self_check_result["utt"] <= 1e-3, m.self_check_result
    print(f"clip-B/16 activation outliers, default constructor: self-check {m.self_check_result}")


@pytest.mark.parametrize("kind", ["hubert", "roberta"])
def test_activation_outliers_post_ln(dev, kind):
    """Post-LN encoders: the re-parametrisation is not exact there (the outlier channels ride the residual stream into the next
    LayerNorm, whose rows they then dominate: hidden states of ~2.5e3 in three channels against ~5e-2 in the others).  RoBERTa-base
    holds the bar with every preset.  HuBERT-base does NOT with one 16-bit activation plane — mx / mean / balanced all land at
    utt ~3e-3..5e-3 (frame 0.3..0.4) on the MI355X, printed below — and does with `accurate` (hi + lo activation planes, three MFMA
    passes, fp32 attention): asserted.  What pretrained HuBERT checkpoints' outlier channels actually look like cannot be checked
    offline; the default constructor's self-check escalates to "accurate" by itself when it sees them."""
    from mertools_amd.encoders import HipBertModel, HipHubertModel
    from util import rel_err
    if kind == "hubert":
        cfg = W.hubert_config("base")
        sd = W.ln_outliers(W.hubert_state_dict(cfg, 0))
        B = 8
        x = W.synth_audio(B, 80000, seed=4321)
        feat = torch.stack(R.hubert_hidden_states(sd, vars(cfg), x))[[-4, -3, -2, -1]].sum(0)
        utt = feat.mean(1)
    else:
        cfg = W.bert_config("roberta-base")
        sd = W.ln_outliers(W.bert_state_dict(cfg, 0))
        B = 16
        x = W.synth_tokens(B, 64, seed=4322)
        feat = torch.stack(R.bert_hidden_states(sd, dict(vars(cfg), roberta=True), x, torch.ones_like(x)))[[-4, -3, -2, -1]].sum(0)
        utt = feat[:, 1:-1].mean(1)
    res, clipwise, perclip = {}, {}, {}
    study = ("mean_a2f", "a2_conv2", "a2_conv3", "a2f_conv3", "mean_conv3", "x3_conv4") if kind == "hubert" else ()   # (printed, not asserted: where the error enters)
    for prec in ("mean", "mx", "balanced", "mean_a2", "accurate", "mean_blocks", "mean_conv") + study + (None,):
        kw = dict(precision=prec, self_check=False) if prec else {}     # None: the default constructor, self-check on
        if kind == "hubert":
            m = HipHubertModel(sd, cfg, device=dev, **kw)
            _, fr, pooled = m.forward_raw(x.to(dev), frames=True, seg_start=[b * 249 for b in range(B)], seg_len=[249] * B)
            ef = rel_err(fr.cpu().view(B, 249, -1), feat)[0]
            perclip[prec] = [float((fr.cpu().view(B, 249, -1)[b] - feat[b]).abs().max() / feat.abs().max()) for b in range(B)]
            clipwise[prec] = sorted(perclip[prec])
        else:
            m = HipBertModel(sd, cfg, device=dev, **kw)
            _, fr, pooled = m.forward_raw(x.to(dev), lengths=[64] * B, frames=True, seg_start=[b * 64 + 1 for b in range(B)], seg_len=[62] * B)
            ef = rel_err(fr.cpu().view(B, 64, -1), feat)[0]
        torch.cuda.synchronize()
        eu = rel_err(pooled.cpu(), utt)[0]
        tag = prec or f"default constructor -> {'accurate (escalated)' if m.escalated else 'kept'}; self-check {m.self_check_result}"
        print(f"{kind}-base activation outliers [{tag}]: frame={ef:.2e} utt={eu:.2e}")
        res[prec] = (eu, ef, bool(getattr(m, "escalated", None)))
        del m
    # Every preset is asserted where it holds the bar.  HuBERT-base (post-LN, outlier channels riding the residual stream): only the
    # three-pass arithmetic does — and the DEFAULT constructor gets there by itself: its load-time self-check sees the outlier LayerNorm
    # channels, measures 'mean' against 'accurate' on its calibration batch and switches.  UTT then holds 1e-3 with a wide margin.
    # FRAME: six of the eight clips sit at ~1e-5; two (the same two in every arithmetic) are frames on which THIS synthetic network is
    # 10^2 - 10^3 times more sensitive than elsewhere — in fp64, a perturbation of hs[0] of the size of fp32 rounding noise moves exactly
    # those clips (tests/studies/outlier_conditioning.py) — and reach 4.5e-4 / 3.3e-3 although every operator is exact to 3e-7 of its row
    # maximum on exact inputs and the model's block equals the fp64 block on the model's own input to 2e-7
    # (tests/studies/outlier_block_bisect_gpu.py, outlier_block_chain_gpu.py).  Asserted: the worst clip at 5e-3, the typical clip at 1e-4.
    if kind == "hubert":
        assert res["accurate"][0] <= TOL and res["accurate"][1] <= 5e-3, res
        assert res[None][2] and res[None][0] <= TOL and res[None][1] <= 5e-3, res
        assert clipwise["accurate"][B // 2] <= 1e-4 and clipwise[None][B // 2] <= 1e-4, clipwise     # the median clip
        # ... and the conditioning argument as an assertion instead of a flat bar: per clip, the HIP path's FRAME error is
        # bounded by a fixed multiple of what fp64 arithmetic itself does to that clip when hs[0] moves by the accurate front end's error
        # (tests/studies/outlier_conditioning.py: 1e-7 .. 4e-6 on six clips, 4e-5 and 1.4 - 3.2e-4 on the two ill-conditioned ones)
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "studies"))
        from outlier_conditioning import frame_sensitivity
        sens = frame_sensitivity(sd, cfg, x)
        for prec in ("accurate", None):
            for b in range(B):
                bound = min(5e-3, 1e-4 + 25.0 * sens[b])
                assert perclip[prec][b] <= bound, (prec, b, perclip[prec][b], sens[b], bound)
        print("hubert-base activation outliers, per clip under accurate: error / fp64 sensitivity = " +
              "  ".join(f"{perclip['accurate'][b]:.1e}/{sens[b]:.1e}" for b in range(B)))
    else:
        for prec in ("mean", "mx", "accurate", None):
            assert res[prec][0] <= TOL and res[prec][1] <= TOL, (kind, res)
        assert not res[None][2], res     # RoBERTa holds the bar as built: the self-check measured it and kept the preset
