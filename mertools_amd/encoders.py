"""Drop-in encoder objects for the three HuggingFace call sites of the MERTools extractors.

    HipHubertModel(input_values, output_hidden_states=True).hidden_states
        -> MERBench/feature_extraction/audio/extract_audio_huggingface.py:97
    HipCLIPModel.get_image_features(pixel_values)
        -> MERBench/feature_extraction/visual/extract_vision_huggingface.py:121
    HipBertModel(input_ids=..., attention_mask=..., output_hidden_states=True).hidden_states
        -> MERBench/feature_extraction/text/extract_text_huggingface.py:225

Each object is built from a HuggingFace state_dict (the checkpoint format the reference loads with
AutoModel.from_pretrained), converts the weights once into the layouts the HIP kernels want
(16-bit hi/lo planes, fused QKV, im2col-ordered conv weights, folded weight-norm) and afterwards
only passes device pointers to libmer_hip.so.  torch provides device memory and the stream; there
is no torch compute on the forward path and no CPU fallback.

precision (see DESIGN.md §numerics for the measured parity of each):
    "fast"      every GEMM one fp16 MFMA pass (fp32 accumulate): ~1e-3 on the saved features
    "mean"      (default) One fp16 pass per GEMM + the weight-rounding residual applied through each SEQUENCE's mean activation:
                table[s] = bias + mean_rows(a[sequence s]) (w - f16(w))^T (mer_seq_bias: exact sampled column means per clip / frame /
                sentence + a small MFMA product on the f16 residual plane, two tiny launches); the one-pass GEMM adds row (m / T) of
                that table where it would add the bias.  The rounding error of the weights is the same perturbation for every token,
                so almost all of what reaches the features goes through the mean activation: same parity as "balanced" / "mx"
                (tests/studies/mean_correction.py) — and, the mean being the sequence's own, a clip's features do not
                depend on its batch mates (bit for bit: tests/test_parity_hardening_gpu.py).  The argument needs rows of comparable
                size — true behind a LayerNorm (every GEMM of the transformer blocks, the patch embedding, HuBERT's projection), NOT
                in HuBERT's conv stack: its inputs are un-normalised GELU outputs, a quiet passage's rows are 20-50x smaller than a
                loud one's, a mean-token offset is an absolute error they cannot absorb, and the feature projection's LayerNorm then
                magnifies it (1e-2 on speech-like loud / quiet audio, tests/test_round3_cpu.py).  So HuBERT's conv stack runs the
                "mx" scheme under this preset — a per-row correction — and the table is used behind LayerNorms only.
    "mx"        One fp16 pass + the weight-rounding residual w - f16(w) as an MX-fp4 plane (e2m1 + E8M0 per 32 k)
                applied through v_mfma_scale_f32_16x16x128_f8f6f4 against bf8 copies of the activations: removes the
                weight-rounding error (coherent across tokens, so it survives the utterance mean) like "balanced",
                at 1/2 f16-pass of extra MFMA work instead of a whole pass.  GEMMs the MX kernel does not cover
                (< 1024 rows, K % 128 != 0, batched, bf16) run the "balanced" path.
    "balanced"  weights carried as hi+lo fp16 planes, two MFMA passes (a*w_hi + a*w_lo)
    "accurate"  both operands split, three passes, attention on fp32 q | k | v (mer_attention_f32): fp32-grade, 2-4e-6 on the base
                encoders, 0.40x the default's throughput
    "mean_a2"   "mean" with every GEMM input of the transformer blocks carried as hi + lo 16-bit planes: a_hi*w_hi + a_lo*w_hi (two
                MFMA passes, mer_gemm16 passes = 6) + the per-sequence table for the weight residual, f16 attention: FRAME error a
                third of "mean"'s on the base trio (1.9e-4 / 2.6e-4 / 2.0e-4), 0.61x its throughput
    "mean_conv3"  HuBERT family: "mean" with the conv stack / projection / positional conv on hi + lo planes (three passes) — for a
                "layer"-norm front end (a LayerNorm behind every conv: wav2vec2-large, data2vec-audio, WavLM-large), whose single
                activation planes put ~1e-3 on hidden_states[0]
    "a2_conv3"  both of the above
                The load-time self-check (below) climbs mean -> mean_conv3 -> mean_a2 -> a2_conv3 -> accurate by itself.
These eight names are the public `precision=` surface.  The numerics studies of rounds 3-5 (which operand an error enters through)
used further pass combinations — "mixed", "balanced3", "mean_all", "mean_blocks", "mean_conv", "mean_a2f", "a2_conv2", "a2f_conv3",
"x3_conv4" (`_STUDY_PREC`) — that no deployment should pick ("mean_all" is known-unsound on speech); they resolve only under
MER_STUDY_PRESETS=1, which tests/conftest.py sets for the suite.
"""
import ctypes as C
import functools
import os

import torch

from . import _lib
from .ops import torch16
from ._lib import (BertConfig, BertWeights, HubertConfig, HubertWeights, MER_ACT_GELU, MER_ACT_GELU_TANH, MER_ACT_QUICK_GELU, MER_MAX_CONV,
                   TfConfig, TfLayer, VideoMAEConfig, VideoMAEWeights, VitConfig, VitWeights, W16)
from .ops import dt_code, split16_host, stream


class EncoderOutput:
    """Minimal stand-in for transformers' BaseModelOutput at the reference call sites."""

    def __init__(self, last_hidden_state=None, hidden_states=None, pooler_output=None):
        self.last_hidden_state = last_hidden_state
        self.hidden_states = hidden_states
        self.pooler_output = pooler_output

    def __getitem__(self, i):
        return (self.last_hidden_state, self.hidden_states)[i]


def _sd_of(model_or_sd):
    sd = model_or_sd.state_dict() if hasattr(model_or_sd, "state_dict") else model_or_sd
    return {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}


_WBLK = os.environ.get("MER_WBLK", "1") != "0"   # pre-blocked weight planes (tuning / A-B switch)
_WBLKP = os.environ.get("MER_WBLKP", "1") != "0"   # ... and the row-permuted copy for the persistent one-pass kernel


# ---- weight planes shared between the builds of ONE load (VERDICT r5 #6c) ----
# from_hf() / a self-checking constructor builds the object, its `accurate` twin and — when the preset fails the check — up to three
# rungs in between; each used to split, upload and pre-block every weight again (a worst-case load packed the checkpoint five times).
# While a load is in progress the device planes are kept in a cache keyed by the weight's content fingerprint + plane kind: a later
# build of the same load takes what an earlier one made and only adds the planes it alone needs (the twin's `lo` planes, a rung's MX
# plane).  The cache dies with the load; planes a discarded candidate alone referenced are freed with it.
_PLANES = None


class _plane_cache:
    def __enter__(self):
        global _PLANES
        self.owner = _PLANES is None and os.environ.get("MER_PLANE_CACHE", "1") != "0"   # (0: every build packs its own planes — A/B, scripts/load_time_ladder.py)
        if self.owner:
            _PLANES = {}
        return self

    def __exit__(self, *a):
        global _PLANES
        if self.owner:
            _PLANES = None


def _fingerprint(t):
    """Content key of a host weight tensor: shape, the fp64 sum (one pass) and 64 samples spread over it."""
    f = t.detach().reshape(-1)
    n = f.numel()
    idx = (torch.arange(64, dtype=torch.long) * max(n - 1, 0)) // 63
    return (tuple(t.shape), str(t.dtype), float(f.sum(dtype=torch.float64)), tuple(f[idx].tolist()))


class _Holder:
    """Owns the device copies of the weights and hands out raw pointers."""

    def __init__(self, device, dtype):
        self.device = torch.device(device)
        self.dtype = dtype
        self.keep = []

    def _plane(self, key, make):
        """The device tensor of plane `key` — from the load's cache when an earlier build made it — kept alive by this holder."""
        cache = _PLANES      # (read once: another thread's load may end — and drop the cache — while this build is under way)
        if cache is not None and key is not None:
            d = cache.get(key)
            if d is None:
                d = make()
                if d is not None:
                    cache[key] = d
        else:
            d = make()
        if d is not None:
            self.keep.append(d)
        return d

    def f32(self, t):
        if t is None:
            return None
        key = ("f32", _fingerprint(t), str(self.device)) if _PLANES is not None else None
        return self._plane(key, lambda: t.detach().to(torch.float32).contiguous().to(self.device)).data_ptr()

    def w16(self, t, lo, mx=False, out="f16"):
        """out: what the GEMM reading this weight writes — "f16" (one 16-bit plane: QKV, fc1, conv stack), "f32" (fp32 (+ residual): attention
        output, fc2, projections) or "both": selects the row permutation of the persistent kernel's pre-blocked plane."""
        t = t.contiguous()
        base = (_fingerprint(t), self.dtype, str(self.device)) if _PLANES is not None else None
        key = (lambda kind: (kind,) + base) if base is not None else (lambda kind: None)
        halves = {}

        def split(which):
            if not halves:
                h, l = split16_host(t, self.dtype, lo)
                halves["hi"], halves["lo"] = h.contiguous(), (l.contiguous() if l is not None else None)
            return halves[which]
        hi = self._plane(key("hi"), lambda: split("hi").to(self.device))
        w = W16()
        w.hi = hi.data_ptr()
        w.lo = None
        w.mx = None
        w.hi_blk = None
        w.lo_blk = None
        w.hi_blkp = None
        w.hi_blkq = None
        lo_t = None
        if lo:
            lo_t = self._plane(key("lo"), lambda: split("lo").to(self.device))
            w.lo = lo_t.data_ptr()
        if t.dim() == 2 and t.shape[0] >= 192 and t.shape[1] % 32 == 0 and self.device.type == "cuda" and _WBLK:
            # pre-blocked copies for the 256-wide LDS-DMA kernels (1 KiB contiguous DMA pieces); the row-major planes stay
            # for the small-batch tiles.  Costs a second copy of the weights in HBM (a few hundred MB at most).
            from .ops import w_block_pack, w_block_pack_p
            with torch.cuda.device(self.device):
                hb = self._plane(key("hb"), lambda: w_block_pack(hi))
                lb = self._plane(key("lb"), lambda: w_block_pack(lo_t)) if lo and hb is not None else None
                # the persistent one-pass kernel's planes (rows permuted per 128: layout 0 for 16-bit outputs, 1 for fp32 outputs)
                hp = self._plane(key("hp"), lambda: w_block_pack_p(hi, 0)) if _WBLKP and out in ("f16", "both") else None
                hq = self._plane(key("hq"), lambda: w_block_pack_p(hi, 1)) if _WBLKP and out in ("f32", "both") else None
                torch.cuda.current_stream().synchronize()   # the forwards may run on other streams
            if hp is not None:
                w.hi_blkp = hp.data_ptr()
            if hq is not None:
                w.hi_blkq = hq.data_ptr()
            if hb is not None:
                w.hi_blk = hb.data_ptr()
                if lb is not None:
                    w.lo_blk = lb.data_ptr()
        if mx and torch16(self.dtype) == torch.float16 and t.dim() == 2:
            # MX-fp4 plane of the rounding residual for the passes=4 GEMM (shapes without one keep using `lo`)
            from .ops import mx_pack

            def make_mx():
                packed = mx_pack(t.to(torch.float32) - hi.to("cpu", torch.float32))
                return packed.to(self.device) if packed is not None else None
            packed = self._plane(key("mx"), make_mx)
            if packed is not None:
                w.mx = packed.data_ptr()
        return w


def _tf_layer(hold, lo, wq, bq, wk, bk, wv, bv, wo, bo, ln1, w1, b1, w2, b2, ln2, mx=False):
    D = wq.shape[0]
    z = torch.zeros(D)
    L = TfLayer()
    L.wqkv = hold.w16(torch.cat([wq, wk, wv], 0), lo, mx)
    L.bqkv = hold.f32(torch.cat([bq if bq is not None else z, bk if bk is not None else z, bv if bv is not None else z], 0))
    L.wo = hold.w16(wo, lo, mx, out="f32")
    L.bo = hold.f32(bo)
    L.ln1_g, L.ln1_b = hold.f32(ln1[0]), hold.f32(ln1[1])
    L.w1 = hold.w16(w1, lo, mx)
    L.b1 = hold.f32(b1)
    L.w2 = hold.w16(w2, lo, mx, out="f32")
    L.b2 = hold.f32(b2)
    L.ln2_g, L.ln2_b = hold.f32(ln2[0]), hold.f32(ln2[1])
    return L


def _tf_config(hidden, heads, ffn, layers, pre_ln, act, eps, dtype, passes, mx_skip=None, attn_f32=0):
    """mx_skip (passes == 4): which block GEMMs run without the weight-residual correction (bit 0: Q/K, bit 1: fc1, bit 2: fc2).
    From the emulated encoders (tests/studies/mx_selective.py) and the GPU parity tests: dropping it for Q/K changes nothing
    anywhere (their rounding only perturbs softmax logits); in PRE-LN blocks (CLIP, VideoMAE, DINOv2, data2vec-vision, the
    stable-LayerNorm HuBERT / WavLM large) the FFN weights do not need it either (CLIP-B/16 UTT 2.4e-4 / frames 4.1e-4 with
    or without, large models 1.8e-4 - 3.3e-4) — only V and the attention output projection carry the error that reaches the
    features; in POST-LN blocks (HuBERT / wav2vec2 / WavLM base, BERT family) dropping it for the FFN triples the error
    (HuBERT-base UTT 6.3e-4), so those keep everything but Q/K corrected.  Default: 7 for pre-LN, 1 for post-LN; the
    MER_MX_SKIP environment variable overrides it (tuning)."""
    c = TfConfig()
    if mx_skip is None:
        mx_skip = int(os.environ.get("MER_MX_SKIP", "-1"))
        if mx_skip < 0:
            mx_skip = 7 if pre_ln else 1
    c.mx_skip = mx_skip
    c.hidden, c.heads, c.ffn, c.layers, c.pre_ln = hidden, heads, ffn, layers, int(pre_ln)
    c.act, c.ln_eps, c.dtype, c.passes = act, eps, dt_code(dtype), passes
    c.attn_f32 = int(attn_f32)
    return c


# precision preset -> (GEMM passes in the HuBERT conv stack, GEMM passes in the transformer blocks)
# (a third entry: attention on fp32 q | k | v under tf passes == 6)
_PREC = {"fast": (1, 1), "balanced": (2, 2), "mx": (4, 4), "mean": (4, 5), "mean_conv3": (3, 5), "mean_a2": (4, 6), "a2_conv3": (3, 6), "accurate": (3, 3)}
# study presets (tests / numerics studies only, MER_STUDY_PRESETS=1): "mixed" / "balanced3" conv stack three passes with one- / two-pass
# blocks; "mean_all" the mean-token table in HuBERT's conv stack too (what "mean" was before the quiet-passage case was tested: unsound);
# "mean_blocks" / "mean_conv" one side two passes; "...f" = fp32 attention under passes = 6; "x3_conv4" MX conv stack under three-pass blocks
_STUDY_PREC = {"mixed": (3, 1), "balanced3": (3, 2), "mean_all": (5, 5), "mean_blocks": (2, 5), "mean_conv": (5, 2), "mean_a2f": (4, 6, 1),
               "a2_conv2": (2, 6), "a2f_conv3": (3, 6, 1), "x3_conv4": (4, 3), "f16": (1, 1), "x3": (3, 3)}


def _prec(precision):
    """(conv-stack passes, transformer-block passes, fp32 attention under passes == 6) of a preset name."""
    t = _PREC.get(precision)
    if t is None:
        if precision not in _STUDY_PREC:
            raise _lib.MerError(f"unknown precision {precision!r}: one of {sorted(_PREC)}")
        if os.environ.get("MER_STUDY_PRESETS", "0") != "1":
            raise _lib.MerError(f"precision {precision!r} is a numerics-study preset (set MER_STUDY_PRESETS=1 to resolve it); deployments use one of {sorted(_PREC)}")
        t = _STUDY_PREC[precision]
    return t[0], t[1], (t[2] if len(t) > 2 else 0)


def _planes(tf_passes):
    """(carry the f16 residual plane `lo`, carry the MX-fp4 residual plane) for a transformer-block pass code."""
    return tf_passes >= 2, tf_passes == 4   # `lo` also backs the MX preset's 2-pass fallback (shapes the MX kernel does not cover)


# ---- load-time precision self-check (VERDICT r3 #1c / ADVICE r3): the reference is fp32 and needs no precision choice ----
# A one-plane preset ("mean", "mx", "balanced", "fast") carries every GEMM input as ONE 16-bit plane.  Checkpoints whose LayerNorms
# have a few massive channels (gamma / beta tens of times the rest: pretrained HuBERT / RoBERTa are reported to) can put activations
# next to each other that 11 bits of mantissa do not hold (profiles/r03_activation_outlier_stress.txt: post-LN HuBERT-base, UTT 5e-3
# / FRAME 0.4).  So the constructor looks at the checkpoint — does any LayerNorm carry such channels? — and, only then, runs a small
# built-in calibration batch through the model AND through its `accurate` twin (both operands as hi + lo planes, three passes) on
# the GPU: if they disagree by more than the parity bar allows, the object becomes the twin (`model.escalated` says why) and a
# warning is raised.  `self_check=False` (or MER_SELF_CHECK=0) skips it, `self_check=True` always runs the comparison.
_ONE_PLANE = ("fast", "f16", "mean", "mx", "balanced", "mean_all", "mean_blocks", "mean_conv")
# preset vs accurate on the calibration batch: north_star's bar (1e-3, UTT and FRAME) with a 20 % margin — the calibration batch is two
# clips, a user's corpus is not: an HF-initialised data2vec-audio module measured 8.8e-4 on it and 1.14e-3 on other audio (round 5).
# A healthy checkpoint sits at 1-5e-4 / 2-7e-4.
SELF_CHECK_UTT, SELF_CHECK_FRAME = 0.8e-3, 0.8e-3
# What is tried, in order, when the preset fails it — each a cheaper arithmetic than `accurate` (three passes + fp32 attention, 0.4x):
#   mean_conv3   the HuBERT family only: conv stack / projection / positional conv on hi + lo planes (three passes), blocks as "mean" —
#                a "layer"-norm front end (wav2vec2-large, data2vec-audio, WavLM-large: a LayerNorm behind every conv) carries ~1e-3 at
#                hidden_states[0] on single 16-bit planes (profiles/r05_d2v_audio_hs_errors.txt), and post-LN blocks with small weights
#                hand it on undamped
#   mean_a2      blocks on hi + lo ACTIVATION planes (two passes + the per-sequence table), conv stack as "mean"
#   a2_conv3     both
_LADDER_AUDIO = ("mean_conv3", "mean_a2", "a2_conv3")
_LADDER = ("mean_a2",)


def ln_outlier_ratio(sd):
    """How far the checkpoint's LayerNorm channels stick out: max over the norm layers of max|gamma| / median|gamma| and of
    max|beta| / max(median|beta|, median|gamma|) (1-D tensors whose key names a norm layer).  A bias vector is measured against the
    layer's GAIN scale as well as its own median: the median of a pretrained LayerNorm bias sits near zero, and max / median of the
    bias alone would flag every real checkpoint — what matters is a channel whose affine output is many times the typical
    channel's, i.e. |beta_c| or |gamma_c| against the typical |gamma|."""
    worst = 1.0
    gains = {}
    for k, v in sd.items():
        if torch.is_tensor(v) and v.dim() == 1 and v.numel() >= 64 and ("norm" in k.lower() or "ln_" in k.lower() or ".ln" in k.lower()):
            if k.endswith("weight"):
                gains[k[:-len("weight")]] = float(v.detach().abs().float().median())
    for k, v in sd.items():
        if torch.is_tensor(v) and v.dim() == 1 and v.numel() >= 64 and ("norm" in k.lower() or "ln_" in k.lower() or ".ln" in k.lower()):
            a = v.detach().abs().float()
            med = float(a.median())
            if k.endswith("bias"):
                med = max(med, gains.get(k[:-len("bias")], 0.0))
            if med > 0:
                worst = max(worst, float(a.max()) / med)
    return worst


def _self_check_may_run(state_dict, mode):
    """Whether a constructor called with self_check=`mode` may go on to build an `accurate` twin (cheap: 1-D tensors only)."""
    if mode is False or os.environ.get("MER_SELF_CHECK", "1") == "0":
        return False
    if mode == "auto":
        sd = state_dict.state_dict() if hasattr(state_dict, "state_dict") else state_dict
        return ln_outlier_ratio(sd) >= 8.0
    return True


def _self_check(model, state_dict, config, args, kwargs, mode):
    import inspect
    import warnings
    if mode is False or os.environ.get("MER_SELF_CHECK", "1") == "0" or not hasattr(model, "_probe_features"):
        return
    bound = inspect.signature(type(model).__init__.__wrapped__).bind(model, state_dict, config, *args, **kwargs)
    bound.apply_defaults()
    prec = bound.arguments.get("precision", "mean")
    if prec not in _ONE_PLANE or model.device.type != "cuda":
        return
    sd = state_dict.state_dict() if hasattr(state_dict, "state_dict") else state_dict
    ratio = ln_outlier_ratio(sd)
    if mode == "auto" and ratio < 8.0:
        return
    kw = {k: v for k, v in bound.arguments.items() if k not in ("self", "state_dict", "config", "precision")}
    twin = type(model)(state_dict, config, precision="accurate", self_check=False, **kw)
    with torch.cuda.device(model.device):
        got, ref = model._probe_features(), twin._probe_features()
        torch.cuda.synchronize()

    def rel(a, b):
        """max over the calibration clips of max|a - b| / max|b| of THAT clip: the norm the parity bar is stated in is per saved file
        (a quiet or short clip must not hide behind a loud one's maximum)."""
        n = ref[0].shape[0]
        a, b = a.double().reshape(n, -1), b.double().reshape(n, -1)
        return float(((a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-30)).max())
    du, df = rel(got[0], ref[0]), rel(got[1], ref[1])
    model.self_check_result = dict(ln_outlier_ratio=ratio, utt=du, frame=df, precision=prec)
    if du > SELF_CHECK_UTT or df > SELF_CHECK_FRAME:
        keep = model.self_check_result
        # the rungs between this preset and `accurate`, cheapest first: the first that agrees with the twin on the calibration batch wins
        target, why = twin, "accurate"
        ladder = _LADDER_AUDIO if isinstance(model, HipHubertModel) else _LADDER
        if os.environ.get("MER_SELF_CHECK_LADDER", "1") != "0":
            for rung in ladder:
                if rung == prec:
                    continue
                cand = type(model)(state_dict, config, precision=rung, self_check=False, **kw)
                with torch.cuda.device(model.device):
                    got2 = cand._probe_features()
                    torch.cuda.synchronize()
                du2, df2 = rel(got2[0], ref[0]), rel(got2[1], ref[1])
                keep[rung] = (du2, df2)
                if du2 <= SELF_CHECK_UTT and df2 <= SELF_CHECK_FRAME:
                    target, why = cand, rung
                    break
                del cand
        model.__dict__, target.__dict__ = target.__dict__, model.__dict__        # the object becomes its better-conditioned twin
        model.self_check_result = keep
        model.escalated = (f"precision '{prec}' disagrees with 'accurate' on the calibration batch (utt {du:.1e}, frame {df:.1e}; LayerNorm "
                           f"outlier ratio {ratio:.0f}): running '{why}'")
        warnings.warn(f"{type(model).__name__}: {model.escalated}")
        del target
    del twin


class _HipModule:
    """Shared plumbing: handle lifetime, cached workspace, nn.Module-ish no-ops the scripts call."""

    _destroy = None

    def __init__(self):
        self._handle = C.c_void_p()
        self._ws = None

    def __init_subclass__(cls, **kw):
        """Every forward runs with the model's own GPU current: the C ABI launches on the current HIP device and ops.stream()
        returns the current device's stream, while weights / workspaces / outputs live on self.device — a model built on
        cuda:1 and called while cuda:0 is current would otherwise launch on GPU 0 with GPU 1 pointers."""
        super().__init_subclass__(**kw)
        init = cls.__dict__.get("__init__")
        if init is not None:
            @functools.wraps(init)
            def checked_init(self, state_dict, config, *a, self_check="auto", **k):
                # (constructors nest — a subclass's __init__ calls its parent's, both wrapped: the OUTERMOST call, whichever class it
                #  belongs to, runs the self-check once the object is complete; a subclass without an __init__ of its own is covered too)
                outermost = "_mer_constructing" not in self.__dict__
                if outermost and "_mer_cache_scope" not in self.__dict__ and _PLANES is None and _self_check_may_run(state_dict, self_check):
                    # a load that may build a twin and rungs: they share the weight planes this build makes (_plane_cache)
                    self.__dict__["_mer_cache_scope"] = True
                    try:
                        with _plane_cache():
                            return checked_init(self, state_dict, config, *a, self_check=self_check, **k)
                    finally:
                        self.__dict__.pop("_mer_cache_scope", None)
                self.__dict__["_mer_constructing"] = True
                try:
                    init(self, state_dict, config, *a, **k)
                finally:
                    if outermost:
                        self.__dict__.pop("_mer_constructing", None)
                if not outermost:
                    return
                self.escalated = None
                if "precision" not in self.__dict__:      # the preset this object runs (after an escalation: its accurate twin's)
                    import inspect
                    sig = inspect.signature(init)
                    if "precision" in sig.parameters:
                        ba = sig.bind(self, state_dict, config, *a, **k)
                        ba.apply_defaults()
                        self.precision = ba.arguments["precision"]
                _self_check(self, state_dict, config, a, k, self_check)
            cls.__init__ = checked_init
        fr = cls.__dict__.get("forward_raw")
        if fr is not None:
            @functools.wraps(fr)
            def guarded(self, *a, **k):
                if self.device.type != "cuda":
                    raise _lib.MerError("mertools_amd encoders run on a GPU (there is no CPU path)")
                with torch.cuda.device(self.device):
                    return fr(self, *a, **k)
            cls.forward_raw = guarded

    def _workspace(self, nbytes):
        # one arena per HIP stream: forwards issued on different streams (half-batches overlapping each other's kernel
        # tails, bench.py --split) must not share scratch memory
        key = torch.cuda.current_stream(self.device).cuda_stream
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws.pop(key, None)
            ws = self._ws[key] = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
        off = (-ws.data_ptr()) % 256
        return ws.data_ptr() + off, ws.numel() - off

    def _ints(self, values):
        """Device int32 copy of a small host list (segment tables, sequence lengths), cached by content: a pageable H2D copy
        blocks the host until the current stream has drained, which would stop the host from queueing the next batch while
        this one runs — and the drivers / bench pass the same tables batch after batch.  The first copy of a key is a pinned
        non-blocking copy on the stream that is current; an event recorded behind it is what every LATER user of the cached table —
        possibly on another stream (bench --split, the per-modality streams) — waits for before its kernels read the table."""
        key = tuple(int(v) for v in values)
        cache = self.__dict__.setdefault("_int_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 256:
                # the kernels read these tables through raw pointers, so a tensor reference does not keep a table alive for a stream
                # that still has launches queued: drain the device before the memory goes back to the allocator (once per 256
                # distinct tables: ragged audio batches make a new one each)
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                cache.clear()
                self.__dict__.get("_int_pins", {}).clear()
            host = torch.tensor(key, dtype=torch.int32)
            if self.device.type == "cuda":   # pinned staging + async copy: a table that changes every batch (ragged audio) must not stall the host either
                host = host.pin_memory()
                t = host.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.__dict__.setdefault("_int_pins", {})[key] = host   # alive as long as the cached device copy
                cache[key] = (t, ev, torch.cuda.current_stream(self.device).cuda_stream)
            else:
                t = host
                cache[key] = (t, None, None)
            return t
        t, ev, st = hit
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            if cur.cuda_stream != st and not ev.query():
                cur.wait_event(ev)
        return t

    def _seg(self, seg_start, seg_len):
        if seg_start is None:
            return None, None, 0
        ss, sl = self._ints(seg_start), self._ints(seg_len)
        return ss, sl, ss.numel()

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self

    def half(self):
        raise _lib.MerError("precision is chosen at construction (precision=...), .half() is not supported")

    def __del__(self):
        try:
            if self._handle and self._destroy:
                getattr(_lib.lib(), self._destroy)(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass


# =================================================================================================
class HipHubertModel(_HipModule):
    _destroy = "mer_hubert_destroy"

    def __init__(self, state_dict, config, device="cuda:0", dtype="f16", precision="mean"):
        super().__init__()
        sd = _sd_of(state_dict)
        self.config = config
        self.device = torch.device(device)
        self.precision = precision
        # what the `accurate` twin for degenerate (constant) clips is built from, on first need (forward_raw, constant_rows); a reference,
        # not a copy — drop_source() lets it go
        self._source = (state_dict, config, dict(device=device, dtype=dtype))
        conv_passes, tf_passes, a32 = _prec(precision)
        hold = self._hold = _Holder(device, dtype)
        n_conv = len(config.conv_kernel)
        assert n_conv <= MER_MAX_CONV
        Cc = config.conv_dim[0]
        assert all(c == Cc for c in config.conv_dim), "conv_dim must be uniform"
        D = config.hidden_size
        cfg = HubertConfig()
        # live HF config objects differ per model type: Data2VecAudioConfig has no `do_stable_layer_norm` (its blocks are post-LN,
        # HF:data2vec/modeling_data2vec_audio.py), HubertConfig may carry `conv_pos_batch_norm` (a BatchNorm positional conv: not built)
        stable_ln = bool(getattr(config, "do_stable_layer_norm", False))
        if getattr(config, "conv_pos_batch_norm", False):
            raise _lib.MerError("conv_pos_batch_norm=True (BatchNorm positional convolution) is not supported")
        if getattr(config, "hidden_act", "gelu") != "gelu" or getattr(config, "feat_extract_activation", "gelu") != "gelu":
            raise _lib.MerError("only GELU activations are supported in the HuBERT / wav2vec2 engine")
        cfg.tf = _tf_config(D, config.num_attention_heads, config.intermediate_size, config.num_hidden_layers,
                            stable_ln, MER_ACT_GELU, config.layer_norm_eps, dtype, tf_passes, attn_f32=a32)
        cfg.n_conv, cfg.conv_dim = n_conv, Cc
        for i in range(n_conv):
            cfg.conv_kernel[i], cfg.conv_stride[i] = config.conv_kernel[i], config.conv_stride[i]
        cfg.feat_norm_group = 1 if getattr(config, "feat_extract_norm", "layer") == "group" else 0
        cfg.conv_bias = int(getattr(config, "conv_bias", False))
        cfg.feat_proj_layer_norm = int(getattr(config, "feat_proj_layer_norm", True))
        # data2vec-audio (HF:data2vec/modeling_data2vec_audio.py): num_conv_pos_embeddings is the NUMBER of positional conv
        # layers (5) and conv_pos_kernel_size their kernel (19); HuBERT / wav2vec2: one conv of kernel num_conv_pos_embeddings
        d2v = "encoder.pos_conv_embed.layers.0.conv.weight" in sd
        cfg.pos_k = config.conv_pos_kernel_size if d2v else config.num_conv_pos_embeddings
        cfg.pos_groups = config.num_conv_pos_embedding_groups
        cfg.pos_layers = config.num_conv_pos_embeddings if d2v else 0
        cfg.stable_layer_norm = int(stable_ln)
        cfg.conv_passes = conv_passes
        wavlm = "encoder.layers.0.attention.rel_attn_embed.weight" in sd
        cfg.tf.gated_rel_pos = int(wavlm)
        clo, cmx = conv_passes >= 2, conv_passes == 4
        w = HubertWeights()
        fe = "feature_extractor.conv_layers."
        w.conv0_w = hold.f32(sd[fe + "0.conv.weight"].reshape(Cc, -1))
        for i in range(n_conv):
            if fe + f"{i}.layer_norm.weight" in sd:
                w.conv_norm_g[i] = hold.f32(sd[fe + f"{i}.layer_norm.weight"])
                w.conv_norm_b[i] = hold.f32(sd[fe + f"{i}.layer_norm.bias"])
            if cfg.conv_bias:
                w.conv_b[i] = hold.f32(sd[fe + f"{i}.conv.bias"])
            if i >= 1:  # [Cout, Cin, k] -> [Cout, k*Cin] (column kk*Cin + ci == one contiguous im2col row)
                wt = sd[fe + f"{i}.conv.weight"]
                # what the conv GEMM writes (mer_hubert_forward): 16-bit planes for the next conv; fp32 for a LayerNorm behind it
                # ("layer" front end: every conv; "group": the last one, whose output the feature projection's LayerNorm reads)
                last = i == n_conv - 1
                to32 = (not cfg.feat_norm_group) or (last and cfg.feat_proj_layer_norm)
                w.conv_w[i] = hold.w16(wt.permute(0, 2, 1).reshape(Cc, -1), clo, cmx, out="f32" if to32 else "f16")
        if cfg.feat_proj_layer_norm:
            w.fp_ln_g, w.fp_ln_b = hold.f32(sd["feature_projection.layer_norm.weight"]), hold.f32(sd["feature_projection.layer_norm.bias"])
        w.fp_w = hold.w16(sd["feature_projection.projection.weight"], clo, cmx, out="f32")
        w.fp_b = hold.f32(sd["feature_projection.projection.bias"])
        # positional conv: fold weight-norm (dim=2), then [D, Dg, K] -> [G, Dg, K*Dg] with column kk*Dg + ci
        p = "encoder.pos_conv_embed.conv."
        G, K = cfg.pos_groups, cfg.pos_k
        Dg = D // G

        def pos_layout(pw):
            return pw.reshape(G, Dg, Dg, K).permute(0, 1, 3, 2).reshape(G * Dg, K * Dg)

        if d2v:
            for i in range(cfg.pos_layers):
                q = f"encoder.pos_conv_embed.layers.{i}.conv."
                w.pos_ws[i] = hold.w16(pos_layout(sd[q + "weight"]), clo)
                w.pos_bs[i] = hold.f32(sd[q + "bias"])
        else:
            if p + "weight" in sd:
                pw = sd[p + "weight"]
            else:
                if p + "parametrizations.weight.original0" in sd:
                    g, v = sd[p + "parametrizations.weight.original0"], sd[p + "parametrizations.weight.original1"]
                else:
                    g, v = sd[p + "weight_g"], sd[p + "weight_v"]
                pw = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
            w.pos_w = hold.w16(pos_layout(pw), clo)
            w.pos_b = hold.f32(sd[p + "bias"])
        w.enc_ln_g, w.enc_ln_b = hold.f32(sd["encoder.layer_norm.weight"]), hold.f32(sd["encoder.layer_norm.bias"])
        tlo, tmx = _planes(tf_passes)
        layers = (TfLayer * config.num_hidden_layers)()
        for l in range(config.num_hidden_layers):
            q = f"encoder.layers.{l}."
            a = q + "attention."
            layers[l] = _tf_layer(
                hold, tlo, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"], sd[a + "k_proj.weight"], sd[a + "k_proj.bias"],
                sd[a + "v_proj.weight"], sd[a + "v_proj.bias"], sd[a + "out_proj.weight"], sd[a + "out_proj.bias"],
                (sd[q + "layer_norm.weight"], sd[q + "layer_norm.bias"]),
                sd[q + "feed_forward.intermediate_dense.weight"], sd[q + "feed_forward.intermediate_dense.bias"],
                sd[q + "feed_forward.output_dense.weight"], sd[q + "feed_forward.output_dense.bias"],
                (sd[q + "final_layer_norm.weight"], sd[q + "final_layer_norm.bias"]), mx=tmx)
            if wavlm:   # gated relative position bias (HF:wavlm/modeling_wavlm.py WavLMAttention)
                layers[l].gru_w = hold.f32(sd[a + "gru_rel_pos_linear.weight"])
                layers[l].gru_b = hold.f32(sd[a + "gru_rel_pos_linear.bias"])
                layers[l].gru_const = hold.f32(sd[a + "gru_rel_pos_const"].reshape(-1))
        self._rel_embed = sd["encoder.layers.0.attention.rel_attn_embed.weight"] if wavlm else None
        self._pos_bias = {}
        w.layers = C.cast(layers, C.POINTER(TfLayer))
        self._layers = layers
        _lib.check(_lib.lib().mer_hubert_create(C.byref(cfg), C.byref(w), C.byref(self._handle)), "mer_hubert_create")
        self._cfg = cfg

    @classmethod
    def from_hf(cls, hf_model, **kw):
        """The drop-in call for a loaded HuggingFace module (AutoModel.from_pretrained(...) at the reference's call sites).  A real
        checkpoint always gets the load-time comparison against the `accurate` twin (self_check=True; the LayerNorm-ratio gate of
        "auto" is for hand-built state_dicts): `model.self_check_result` / `model.escalated` say what it found."""
        kw.setdefault("self_check", True)
        return cls(hf_model.state_dict(), hf_model.config, **kw)

    def _probe_features(self):
        """(utterance, frame) features of the built-in calibration batch (load-time self-check), three clips of 2 s: noise whose loudness
        ramps over 20 dB with a tone under it; a tone burst; and a speech-like clip — 0.3 s passages 40 dB apart, the dynamics that
        exposed the conv stack's mean-token bias in round 3 and that stationary noise cannot show (VERDICT r5 #6b)."""
        g = torch.Generator().manual_seed(20260926)
        L = 32000
        t = torch.arange(L, dtype=torch.float32) / 16000.0
        ramp = torch.logspace(-1, 0, L)
        a = 0.1 * torch.randn(L, generator=g) * ramp + 0.05 * torch.sin(2 * 3.14159265 * 220.0 * t)
        b = 0.2 * torch.sin(2 * 3.14159265 * 440.0 * t) * (t % 0.5 < 0.25) + 0.01 * torch.randn(L, generator=g)
        c = 0.1 * torch.randn(L, generator=g) * torch.where((torch.arange(L) // 4800) % 2 == 0, 1.0, 0.01)
        wav = torch.stack([a, b, c])
        wav = (wav - wav.mean(1, keepdim=True)) / torch.sqrt(wav.var(1, unbiased=False, keepdim=True) + 1e-7)
        T = self.out_frames(L)
        _, fr, pooled = self.forward_raw(wav.to(self.device), frames=True, seg_start=[0, T, 2 * T], seg_len=[T, T, T])
        return pooled, fr

    def out_frames(self, L):
        return _lib.lib().mer_hubert_out_frames(self._handle, int(L))

    def position_bias(self, T):
        """WavLM: the [H, T, ceil4(T)] relative position bias table for T frames (WavLMAttention.compute_bias, built once per
        T with the reference's own torch ops on the host — bucket edges are float-log comparisons — and cached on the device)."""
        if self._rel_embed is None:
            return None
        if T not in self._pos_bias:
            import math
            nbk, maxd = getattr(self.config, "num_buckets", 320), getattr(self.config, "max_bucket_distance", 800)
            rel = torch.arange(T, dtype=torch.long)[None, :] - torch.arange(T, dtype=torch.long)[:, None]
            nb = nbk // 2
            buckets = (rel > 0).to(torch.long) * nb
            rel = torch.abs(rel)
            max_exact = nb // 2
            large = torch.log(rel.float() / max_exact) / math.log(maxd / max_exact) * (nb - max_exact)
            large = torch.min((max_exact + large).to(torch.long), torch.full_like(rel, nb - 1))
            buckets = buckets + torch.where(rel < max_exact, rel, large)
            table = torch.nn.functional.embedding(buckets, self._rel_embed).permute(2, 0, 1)     # [H, T, T]
            ldb = (T + 3) // 4 * 4
            padded = torch.zeros(table.shape[0], T, ldb)
            padded[:, :, :T] = table
            self._pos_bias[T] = padded.contiguous().to(self.device)
        return self._pos_bias[T]

    def drop_source(self):
        """Lets go of the checkpoint reference kept for the constant-clip twin (constant rows then stay on this object's preset)."""
        self._source = None

    def build_constant_twin(self):
        """Builds the `accurate` twin that constant rows (digital silence, DC clips) are routed through NOW instead of on the first such
        row: a second set of weight planes (~3x this object's: hi + lo planes of both operands) is reserved at load time, not in the middle
        of an extraction.  -> True when a twin exists afterwards.  `drop_source()` afterwards releases the checkpoint
        reference the lazy path keeps alive."""
        return self._escalation_twin() is not None

    def _escalation_twin(self):
        tw = self.__dict__.get("_twin")
        if tw is None and self.__dict__.get("_source") is not None:
            sd, cfg, kw = self._source
            tw = self._twin = type(self)(sd, cfg, precision="accurate", self_check=False, **kw)
        return tw

    @staticmethod
    def constant_rows_of(x, valid_samples=None):
        """Rows of a HOST batch [B, L] that are constant over their valid samples (digital silence, a DC clip, an all-zero chunk).  Such
        a row is the degenerate input of this architecture: conv0 sees no variation, GroupNorm hands every frame the same beta (a
        "layer" front end: the same LayerNorm output), all frames of the row are identical — and so are their rounding errors, which
        then neither average out over frames nor inside attention."""
        B, L = x.shape
        if valid_samples is None:      # O(B) pre-check on four samples per row: ordinary clips stop here, in one vectorised comparison
            probe = x[:, [0, min(1, L - 1), L // 2, L - 1]]
            cand = (probe == probe[:, :1]).all(1).nonzero().flatten().tolist()
        else:
            cand = [r for r in range(B) if bool(x[r, 0] == x[r, max(int(valid_samples[r]) - 1, 0)])]
        rows = []
        for r in cand:
            n = L if valid_samples is None else int(valid_samples[r])
            if bool((x[r, :n] == x[r, 0]).all()):
                rows.append(r)
        return rows

    def forward_raw(self, input_values, *, hidden_states=False, frames=False, seg_start=None, seg_len=None, valid_samples=None,
                    constant_rows=None):
        """valid_samples: per-row sample counts of a RAGGED batch (rows = clips of different lengths, zero-padded to the
        common L).  Each row's valid frames then equal its batch-of-one forward — the reference never pads or masks audio
        (extract_audio_huggingface.py:93-100) — and the frames past a row's length are unspecified.
        constant_rows: indices of rows known to be constant over their valid samples (constant_rows_of).  A one-plane preset keeps the
        1e-3 bar on such rows only by running them through the `accurate` twin (built on first need from the retained checkpoint):
        their outputs are replaced by the twin's.  None: detected here when the batch arrives as a HOST tensor (it is looked at before
        the upload); a DEVICE tensor is not inspected (that would stall the stream) — the audio driver passes the rows it found on the
        host."""
        x = input_values
        if constant_rows is None:
            constant_rows = self.constant_rows_of(x, valid_samples) if (not x.is_cuda and x.dtype == torch.float32) else []
        # (every preset whose blocks carry ONE activation plane: the constant row's coherent rounding error is a property of the blocks)
        rows = [int(r) for r in constant_rows] if (self.precision in _ONE_PLANE or self.precision in ("mean_conv3", "mixed", "balanced3")) else []
        if rows and self._escalation_twin() is None:
            rows = []
        if not x.is_cuda:
            x = x.to(self.device)
        x = x.to(torch.float32).contiguous()
        B, L = x.shape
        T, D, nl = self.out_frames(L), self.config.hidden_size, self.config.num_hidden_layers
        ss, sl, nseg = self._seg(seg_start, seg_len)
        want_fr = frames or (bool(rows) and nseg > 0)      # patched rows: the pooled features are re-taken from the patched frames
        hs = torch.empty((nl + 1, B, T, D), dtype=torch.float32, device=self.device) if hidden_states else None
        fr = torch.empty((B * T, D), dtype=torch.float32, device=self.device) if want_fr else None
        pooled = torch.empty((nseg, D), dtype=torch.float32, device=self.device) if nseg else None
        nbytes = _lib.lib().mer_hubert_workspace_bytes(self._handle, B, L, int(hidden_states))
        wp, wn = self._workspace(nbytes)
        pb = self.position_bias(T)
        vs = None
        if valid_samples is not None:
            valid_samples = [int(v) for v in valid_samples]
            if len(valid_samples) != B or min(valid_samples) < 1 or max(valid_samples) > L:
                raise _lib.MerError(f"valid_samples must hold one count in [1, {L}] per row (got {valid_samples})")
            if min(self.out_frames(v) for v in set(valid_samples)) < 1:
                raise _lib.MerError("a clip is shorter than the conv stack's receptive field")
            if any(v != L for v in valid_samples):
                vs = self._ints(valid_samples)
        _lib.check(_lib.lib().mer_hubert_forward_ragged(
            self._handle, x.data_ptr(), B, L, vs.data_ptr() if vs is not None else None, wp, wn,
            hs.data_ptr() if hs is not None else None,
            fr.data_ptr() if fr is not None else None, ss.data_ptr() if nseg else None, sl.data_ptr() if nseg else None, nseg,
            pooled.data_ptr() if nseg else None, pb.data_ptr() if pb is not None else None, pb.shape[2] if pb is not None else 0,
            stream()), "mer_hubert_forward")
        if rows:
            # the constant rows again, as a side batch through the accurate twin (same stream: ordered behind the launch above);
            # torch only moves rows here (index_select / index_copy), the arithmetic is the twin's
            tw = self._escalation_twin()
            idx = torch.tensor(rows, dtype=torch.int64, device=self.device)
            sub_valid = [valid_samples[r] for r in rows] if valid_samples is not None else None
            hs2, fr2, _ = tw.forward_raw(x.index_select(0, idx), hidden_states=hs is not None, frames=fr is not None,
                                         valid_samples=sub_valid, constant_rows=[])
            if hs is not None:
                hs.index_copy_(1, idx, hs2)
            if fr is not None:
                fr.view(B, T, D).index_copy_(0, idx, fr2.view(len(rows), T, D))
            if nseg:
                from .ops import sum_pool
                pooled = sum_pool([fr], ss, sl)[1]      # the same pooling kernel over the patched frames (fr = the stored last-4 sums)
            if not frames:
                fr = None
        return hs, fr, pooled

    def __call__(self, input_values, attention_mask=None, output_hidden_states=False, **_):
        if attention_mask is not None:
            raise _lib.MerError("the reference passes no attention_mask for audio (extract_audio_huggingface.py:97)")
        hs, _, _ = self.forward_raw(input_values, hidden_states=True)
        return EncoderOutput(last_hidden_state=hs[-1], hidden_states=tuple(hs[i] for i in range(hs.shape[0])) if output_hidden_states else None)

    def clip_segments(self, L, clip_chunks, valid_samples=None):
        """(seg_start, seg_len) over the flattened [B*T] frame axis: clip i = clip_chunks[i] consecutive rows.  A one-row clip
        of a ragged batch covers the frames its own samples produce; a chunked clip (> 10 s) covers all frames of its
        rows, zero-padded tail included, as in the reference (extract_audio_huggingface.py:46-49,99-107)."""
        T = self.out_frames(L)
        starts, lens, r = [], [], 0
        for n in clip_chunks:
            starts.append(r * T)
            lens.append(n * T if (n > 1 or valid_samples is None) else self.out_frames(int(valid_samples[r])))
            r += n
        return starts, lens

    def extract_utterance(self, input_values, clip_chunks=None, valid_samples=None, constant_rows=None):
        """Fused path: last-4 sum + mean over all frames of each clip -> [nclip, D].
        clip_chunks[i] = number of consecutive batch rows belonging to clip i (default 1 each)."""
        B, L = input_values.shape
        starts, lens = self.clip_segments(L, clip_chunks or [1] * B, valid_samples)
        _, _, pooled = self.forward_raw(input_values, seg_start=starts, seg_len=lens, valid_samples=valid_samples, constant_rows=constant_rows)
        return pooled


# wav2vec 2.0 shares HuBERT's architecture and HF parameter names (feature_extractor.conv_layers.*,
# feature_projection.*, encoder.pos_conv_embed.*, encoder.layers.*): the reference's audio script treats them alike
# (extract_audio_huggingface.py:92-100), and so does this class.
HipWav2Vec2Model = HipHubertModel
# data2vec-audio: same conv stack ("layer" norm) and post-LN blocks, a 5-layer positional conv stack instead of one
# weight-normed conv (handled by pos_layers); the reference treats it like the others (extract_audio_huggingface.py:22-23,93-100)
HipData2VecAudioModel = HipHubertModel
# WavLM: HuBERT wiring + a gated relative position bias in every attention (state-dict keys encoder.layers.*.attention.
# gru_rel_pos_* / rel_attn_embed switch it on); extract_audio_huggingface.py:33-34
HipWavLMModel = HipHubertModel


# =================================================================================================
class HipCLIPModel(_HipModule):
    _destroy = "mer_vit_destroy"

    def __init__(self, state_dict, config, device="cuda:0", dtype="f16", precision="mean"):
        super().__init__()
        sd = _sd_of(state_dict)
        self.config = config
        vc = config.vision_config
        self.device = torch.device(device)
        _, tf_passes, a32 = _prec(precision)
        lo, tmx = _planes(tf_passes)
        hold = self._hold = _Holder(device, dtype)
        act = MER_ACT_QUICK_GELU if vc.hidden_act == "quick_gelu" else MER_ACT_GELU
        cfg = VitConfig()
        cfg.tf = _tf_config(vc.hidden_size, vc.num_attention_heads, vc.intermediate_size, vc.num_hidden_layers, True, act,
                            vc.layer_norm_eps, dtype, tf_passes, attn_f32=a32)
        cfg.image_size, cfg.patch_size, cfg.channels, cfg.proj_dim = vc.image_size, vc.patch_size, vc.num_channels, config.projection_dim
        v = "vision_model."
        w = VitWeights()
        pw = sd[v + "embeddings.patch_embedding.weight"].reshape(vc.hidden_size, -1)
        pad = (-pw.shape[1]) % 8   # CLIP-L/14: 588 -> 592 zero columns (16-byte rows for the MFMA GEMM)
        if pad:
            pw = torch.cat([pw, torch.zeros(pw.shape[0], pad)], 1)
        w.patch_w = hold.w16(pw, lo, tmx, out="f32")
        w.cls = hold.f32(sd[v + "embeddings.class_embedding"])
        w.pos = hold.f32(sd[v + "embeddings.position_embedding.weight"])
        w.pre_ln_g, w.pre_ln_b = hold.f32(sd[v + "pre_layrnorm.weight"]), hold.f32(sd[v + "pre_layrnorm.bias"])
        w.post_ln_g, w.post_ln_b = hold.f32(sd[v + "post_layernorm.weight"]), hold.f32(sd[v + "post_layernorm.bias"])
        w.proj_w = hold.w16(sd["visual_projection.weight"], lo, tmx)
        layers = (TfLayer * vc.num_hidden_layers)()
        for l in range(vc.num_hidden_layers):
            q = f"{v}encoder.layers.{l}."
            a = q + "self_attn."
            layers[l] = _tf_layer(
                hold, lo, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"], sd[a + "k_proj.weight"], sd[a + "k_proj.bias"],
                sd[a + "v_proj.weight"], sd[a + "v_proj.bias"], sd[a + "out_proj.weight"], sd[a + "out_proj.bias"],
                (sd[q + "layer_norm1.weight"], sd[q + "layer_norm1.bias"]), sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"],
                sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"], (sd[q + "layer_norm2.weight"], sd[q + "layer_norm2.bias"]), mx=tmx)
        w.layers = C.cast(layers, C.POINTER(TfLayer))
        self._layers = layers
        _lib.check(_lib.lib().mer_vit_create(C.byref(cfg), C.byref(w), C.byref(self._handle)), "mer_vit_create")
        self._cfg = cfg

    @classmethod
    def from_hf(cls, hf_model, **kw):
        """The drop-in call for a loaded HuggingFace module (AutoModel.from_pretrained(...) at the reference's call sites).  A real
        checkpoint always gets the load-time comparison against the `accurate` twin (self_check=True; the LayerNorm-ratio gate of
        "auto" is for hand-built state_dicts): `model.self_check_result` / `model.escalated` say what it found."""
        kw.setdefault("self_check", True)
        return cls(hf_model.state_dict(), hf_model.config, **kw)

    def forward_raw(self, pixel_values, *, features=True, seg_start=None, seg_len=None):
        x = pixel_values
        if not x.is_cuda:
            x = x.to(self.device)
        x = x.to(torch.float32).contiguous()
        N = x.shape[0]
        P = self.config.projection_dim
        assert x.shape[1:] == (self._cfg.channels, self._cfg.image_size, self._cfg.image_size), x.shape
        feats = torch.empty((N, P), dtype=torch.float32, device=self.device) if features else None
        ss, sl, nseg = self._seg(seg_start, seg_len)
        pooled = torch.empty((nseg, P), dtype=torch.float32, device=self.device) if nseg else None
        wp, wn = self._workspace(_lib.lib().mer_vit_workspace_bytes(self._handle, N))
        _lib.check(_lib.lib().mer_vit_forward(
            self._handle, x.data_ptr(), N, wp, wn, feats.data_ptr() if features else None,
            ss.data_ptr() if nseg else None, sl.data_ptr() if nseg else None, nseg, pooled.data_ptr() if nseg else None,
            stream()), "mer_vit_forward")
        return feats, pooled

    def _probe_features(self):
        """(clip, frame) features of the built-in calibration batch (load-time self-check): 4 frames of blocky noise at CLIP's statistics
        and a flat frame (one colour: every patch row identical — the visual tower's degenerate input, VERDICT r5 #6b)."""
        g = torch.Generator().manual_seed(20260926)
        S = self._cfg.image_size
        px = torch.rand((4, self._cfg.channels, S // 8, S // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :S, :S]
        px = (px + 0.1 * torch.rand((4, self._cfg.channels, S, S), generator=g) - 0.45) / 0.27
        px = torch.cat([px, torch.full((1, self._cfg.channels, S, S), (0.6 - 0.45) / 0.27)], 0)
        feats = self.forward_raw(px.contiguous().to(self.device))[0]
        return feats.mean(0, keepdim=True), feats

    def get_image_features(self, pixel_values=None, **_):
        """transformers-4.28 semantics: returns the projected embeddings tensor [N, projection_dim]."""
        return self.forward_raw(pixel_values)[0]

    def extract_utterance(self, pixel_values, frames_per_clip):
        """Fused path: per-frame features + mean over each clip's frames -> [nclip, projection_dim]."""
        starts, lens, r = [], [], 0
        for n in frames_per_clip:
            starts.append(r)
            lens.append(n)
            r += n
        return self.forward_raw(pixel_values, features=False, seg_start=starts, seg_len=lens)[1]


# =================================================================================================
class HipDinov2Model(_HipModule):
    """DINOv2 branch of extract_vision_huggingface.py:133-144: `model(batch, output_hidden_states=True).hidden_states`
    whose last entry is token-summed.  HF:dinov2/modeling_dinov2.py.  Built for ONE input resolution (`input_size`,
    224 = the processor's crop): the position table is interpolated to that grid once at load time, and the layer-scale
    vectors are folded into the attention-output / fc2 weights and biases (y = x + lambda * (h W^T + b) == x + h (lambda*W)^T
    + lambda*b), so the blocks are the plain pre-LN blocks of the ViT engine.  dinov2-giant's SwiGLU feed-forward is supported (`use_swiglu_ffn`)."""

    def __init__(self, state_dict, config, device="cuda:0", dtype="f16", precision="mean", input_size=224):
        super().__init__()
        sd = _sd_of(state_dict)
        self.config = config
        self.device = torch.device(device)
        swiglu = bool(getattr(config, "use_swiglu_ffn", False))   # dinov2-giant
        _, tf_passes, a32 = _prec(precision)
        lo, tmx = _planes(tf_passes)
        hold = self._hold = _Holder(device, dtype)
        D, Pz = config.hidden_size, config.patch_size
        ffn = int(D * config.mlp_ratio)
        if swiglu:
            ffn = (int(ffn * 2 / 3) + 7) // 8 * 8          # Dinov2SwiGLUFFN: width after the gate
        assert input_size % Pz == 0, "input size must be a multiple of the patch size"
        cfg = VitConfig()
        cfg.tf = _tf_config(D, config.num_attention_heads, ffn, config.num_hidden_layers, True, MER_ACT_GELU, config.layer_norm_eps, dtype, tf_passes, attn_f32=a32)
        cfg.tf.ffn_swiglu = int(swiglu)
        cfg.image_size, cfg.patch_size, cfg.channels, cfg.proj_dim, cfg.variant = input_size, Pz, config.num_channels, D, 1
        w = VitWeights()
        pw = sd["embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
        pad = (-pw.shape[1]) % 8   # patch 14: 588 -> 592 zero columns
        if pad:
            pw = torch.cat([pw, torch.zeros(pw.shape[0], pad)], 1)
        w.patch_w = hold.w16(pw, lo, tmx, out="f32")
        w.patch_b = hold.f32(sd["embeddings.patch_embeddings.projection.bias"])
        w.cls = hold.f32(sd["embeddings.cls_token"].reshape(D))
        w.pos = hold.f32(self.interpolate_pos_encoding(sd["embeddings.position_embeddings"], input_size // Pz))
        layers = (TfLayer * config.num_hidden_layers)()
        for l in range(config.num_hidden_layers):
            q = f"encoder.layer.{l}."
            a = q + "attention.attention."
            l1, l2 = sd[q + "layer_scale1.lambda1"], sd[q + "layer_scale2.lambda1"]
            layers[l] = _tf_layer(
                hold, lo, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], sd[a + "key.bias"],
                sd[a + "value.weight"], sd[a + "value.bias"],
                sd[q + "attention.output.dense.weight"] * l1[:, None], sd[q + "attention.output.dense.bias"] * l1,
                (sd[q + "norm1.weight"], sd[q + "norm1.bias"]),
                sd[q + ("mlp.weights_in.weight" if swiglu else "mlp.fc1.weight")], sd[q + ("mlp.weights_in.bias" if swiglu else "mlp.fc1.bias")],
                sd[q + ("mlp.weights_out.weight" if swiglu else "mlp.fc2.weight")] * l2[:, None],
                sd[q + ("mlp.weights_out.bias" if swiglu else "mlp.fc2.bias")] * l2,
                (sd[q + "norm2.weight"], sd[q + "norm2.bias"]), mx=tmx)
        w.layers = C.cast(layers, C.POINTER(TfLayer))
        self._layers = layers
        _lib.check(_lib.lib().mer_vit_create(C.byref(cfg), C.byref(w), C.byref(self._handle)), "mer_vit_create")
        self._cfg = cfg
        self.tokens = (input_size // Pz) ** 2 + 1

    @staticmethod
    def interpolate_pos_encoding(pos, grid):
        """Dinov2Embeddings.interpolate_pos_encoding for a square grid x grid input: [1, 1+n, D] -> [1+grid^2, D]."""
        n = pos.shape[1] - 1
        D = pos.shape[-1]
        if grid * grid == n:
            return pos[0]
        s0 = int(n ** 0.5)
        pp = pos[:, 1:].reshape(1, s0, s0, D).permute(0, 3, 1, 2).float()
        pp = torch.nn.functional.interpolate(pp, size=(grid, grid), mode="bicubic", align_corners=False)
        return torch.cat([pos[0, :1], pp.permute(0, 2, 3, 1).reshape(-1, D)], dim=0)

    @classmethod
    def from_hf(cls, hf_model, **kw):
        """The drop-in call for a loaded HuggingFace module (AutoModel.from_pretrained(...) at the reference's call sites).  A real
        checkpoint always gets the load-time comparison against the `accurate` twin (self_check=True; the LayerNorm-ratio gate of
        "auto" is for hand-built state_dicts): `model.self_check_result` / `model.escalated` say what it found."""
        kw.setdefault("self_check", True)
        return cls(hf_model.state_dict(), hf_model.config, **kw)

    def _probe_features(self):
        """(clip, frame) features of the built-in calibration batch (load-time self-check): 4 frames of blocky noise at ImageNet statistics."""
        g = torch.Generator().manual_seed(20260926)
        S, Cn = self._cfg.image_size, self._cfg.channels
        px = torch.rand((4, Cn, (S + 7) // 8, (S + 7) // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :S, :S]
        px = (px + 0.1 * torch.rand((4, Cn, S, S), generator=g) - 0.45) / 0.225
        feats = self.forward_raw(px.contiguous().to(self.device))[0]
        return feats.mean(0, keepdim=True), feats

    def forward_raw(self, pixel_values, *, features=True, tokens=False, seg_start=None, seg_len=None):
        """-> (frame features [N, D] = token sums, last residual stream [N, 1+P, D] or None, pooled [nseg, D] or None)."""
        x = pixel_values
        if not x.is_cuda:
            x = x.to(self.device)
        x = x.to(torch.float32).contiguous()
        N, D = x.shape[0], self.config.hidden_size
        assert x.shape[1:] == (self._cfg.channels, self._cfg.image_size, self._cfg.image_size), x.shape
        feats = torch.empty((N, D), dtype=torch.float32, device=self.device) if features else None
        tok = torch.empty((N, self.tokens, D), dtype=torch.float32, device=self.device) if tokens else None
        ss, sl, nseg = self._seg(seg_start, seg_len)
        pooled = torch.empty((nseg, D), dtype=torch.float32, device=self.device) if nseg else None
        wp, wn = self._workspace(_lib.lib().mer_vit_workspace_bytes(self._handle, N))
        _lib.check(_lib.lib().mer_vit_forward_tokens(
            self._handle, x.data_ptr(), N, wp, wn, feats.data_ptr() if features else None,
            ss.data_ptr() if nseg else None, sl.data_ptr() if nseg else None, nseg, pooled.data_ptr() if nseg else None,
            tok.data_ptr() if tokens else None, stream()), "mer_vit_forward_tokens")
        return feats, tok, pooled

    def __call__(self, pixel_values=None, output_hidden_states=False, **_):
        """Drop-in for the reference's call: `.hidden_states` is a 1-tuple holding the LAST hidden state (the only entry the
        script reads: `torch.stack(hidden_states)[-1]`), not all L+1 of them."""
        _, tok, _ = self.forward_raw(pixel_values, features=False, tokens=True)
        return EncoderOutput(last_hidden_state=None, hidden_states=(tok,) if output_hidden_states else None)

    def extract_frames(self, pixel_values):
        """Fused `torch.stack(hidden_states)[-1].sum(dim=1)` -> [N, D]."""
        return self.forward_raw(pixel_values)[0]

    def extract_utterance(self, pixel_values, frames_per_clip):
        starts, lens, r = [], [], 0
        for n in frames_per_clip:
            starts.append(r)
            lens.append(n)
            r += n
        return self.forward_raw(pixel_values, features=False, seg_start=starts, seg_len=lens)[2]


# =================================================================================================
class HipData2VecVisionModel(HipDinov2Model):
    """data2vec-vision (BEiT wiring) branch of extract_vision_huggingface.py:123-131: hidden_states[-1] token-summed.
    HF:data2vec/modeling_data2vec_vision.py.  Same engine variant as DINOv2 (patch bias, no embedding LN, token-sum output);
    differences handled at load time: optional absolute position table (zeros when the checkpoint has none), key projection
    without bias, lambda_1 / lambda_2 folded into the output projections, and the relative position bias — the per-layer
    table and/or the shared one, expanded to [H, T, ceil4(T)] and baked into each layer (`mer_tf_layer.attn_bias`), added to
    the scores by mer_attention_bias.  Native resolution only (config.image_size)."""

    def __init__(self, state_dict, config, device="cuda:0", dtype="f16", precision="mean"):
        _HipModule.__init__(self)
        sd = _sd_of(state_dict)
        self.config = config
        self.device = torch.device(device)
        _, tf_passes, a32 = _prec(precision)
        lo, tmx = _planes(tf_passes)
        hold = self._hold = _Holder(device, dtype)
        D, Pz, Hn = config.hidden_size, config.patch_size, config.num_attention_heads
        win = config.image_size // Pz
        T = win * win + 1
        cfg = VitConfig()
        cfg.tf = _tf_config(D, Hn, config.intermediate_size, config.num_hidden_layers, True, MER_ACT_GELU, config.layer_norm_eps, dtype, tf_passes, attn_f32=a32)
        cfg.image_size, cfg.patch_size, cfg.channels, cfg.proj_dim, cfg.variant = config.image_size, Pz, config.num_channels, D, 1
        w = VitWeights()
        pw = sd["embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
        pad = (-pw.shape[1]) % 8
        if pad:
            pw = torch.cat([pw, torch.zeros(pw.shape[0], pad)], 1)
        w.patch_w = hold.w16(pw, lo, tmx, out="f32")
        w.patch_b = hold.f32(sd["embeddings.patch_embeddings.projection.bias"])
        w.cls = hold.f32(sd["embeddings.cls_token"].reshape(D))
        pos = sd.get("embeddings.position_embeddings")
        w.pos = hold.f32(pos[0] if pos is not None else torch.zeros(T, D))
        shared = sd.get("encoder.relative_position_bias.relative_position_bias_table")
        shared = self.relative_position_bias(shared, win) if shared is not None else None
        ldb = (T + 3) // 4 * 4
        layers = (TfLayer * config.num_hidden_layers)()
        shared_ptr = None
        for l in range(config.num_hidden_layers):
            q = f"encoder.layer.{l}."
            a = q + "attention.attention."
            l1 = sd.get(q + "lambda_1", torch.ones(D))
            l2 = sd.get(q + "lambda_2", torch.ones(D))
            layers[l] = _tf_layer(
                hold, lo, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], None, sd[a + "value.weight"], sd[a + "value.bias"],
                sd[q + "attention.output.dense.weight"] * l1[:, None], sd[q + "attention.output.dense.bias"] * l1,
                (sd[q + "layernorm_before.weight"], sd[q + "layernorm_before.bias"]), sd[q + "intermediate.dense.weight"],
                sd[q + "intermediate.dense.bias"], sd[q + "output.dense.weight"] * l2[:, None], sd[q + "output.dense.bias"] * l2,
                (sd[q + "layernorm_after.weight"], sd[q + "layernorm_after.bias"]), mx=tmx)
            own = sd.get(a + "relative_position_bias.relative_position_bias_table")
            bias = self.relative_position_bias(own, win) if own is not None else None
            if shared is not None:
                bias = shared if bias is None else bias + shared
            if bias is not None:
                if own is None and shared_ptr is not None:
                    layers[l].attn_bias = shared_ptr        # shared table only: one device copy for all layers
                else:
                    padded = torch.zeros(Hn, T, ldb)
                    padded[:, :, :T] = bias
                    layers[l].attn_bias = hold.f32(padded)
                    if own is None:
                        shared_ptr = layers[l].attn_bias
        w.layers = C.cast(layers, C.POINTER(TfLayer))
        self._layers = layers
        _lib.check(_lib.lib().mer_vit_create(C.byref(cfg), C.byref(w), C.byref(self._handle)), "mer_vit_create")
        self._cfg = cfg
        self.tokens = T

    @staticmethod
    def relative_position_bias(table, w):
        """Data2VecVisionRelativePositionBias at the native window: [(2w-1)^2 + 3, H] -> [H, 1+w*w, 1+w*w]."""
        nrel = (2 * w - 1) * (2 * w - 1) + 3
        coords = torch.stack(torch.meshgrid(torch.arange(w), torch.arange(w), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += w - 1
        rel[:, :, 1] += w - 1
        rel[:, :, 0] *= 2 * w - 1
        idx = torch.zeros((w * w + 1,) * 2, dtype=rel.dtype)
        idx[1:, 1:] = rel.sum(-1)
        idx[0, 0:] = nrel - 3
        idx[0:, 0] = nrel - 2
        idx[0, 0] = nrel - 1
        return table[idx.view(-1)].view(w * w + 1, w * w + 1, -1).permute(2, 0, 1).contiguous()


HipBeitModel = HipData2VecVisionModel   # same module structure and state-dict keys (HF:beit/modeling_beit.py)


# =================================================================================================
def sinusoid_table(n_position, d_hid):
    """VideoMAE's fixed position table (HF:videomae/modeling_videomae.py:80-91), float64 numpy then float32."""
    import numpy as np
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.tensor(table, dtype=torch.float32)


class HipVideoMAEModel(_HipModule):
    """`model(inputs).last_hidden_state` of extract_vision_huggingface.py:155 (VideoMAE branch)."""
    _destroy = "mer_videomae_destroy"

    def __init__(self, state_dict, config, device="cuda:0", dtype="f16", precision="mean"):
        super().__init__()
        sd = _sd_of(state_dict)
        self.config = config
        self.device = torch.device(device)
        _, tf_passes, a32 = _prec(precision)
        lo, tmx = _planes(tf_passes)
        hold = self._hold = _Holder(device, dtype)
        D = config.hidden_size
        cfg = VideoMAEConfig()
        cfg.tf = _tf_config(D, config.num_attention_heads, config.intermediate_size, config.num_hidden_layers, True, MER_ACT_GELU,
                            config.layer_norm_eps, dtype, tf_passes, attn_f32=a32)
        cfg.image_size, cfg.patch_size, cfg.channels = config.image_size, config.patch_size, config.num_channels
        cfg.num_frames, cfg.tubelet_size = config.num_frames, config.tubelet_size
        cfg.final_ln = int("layernorm.weight" in sd)
        self.num_patches = (config.image_size // config.patch_size) ** 2 * (config.num_frames // config.tubelet_size)
        w = VideoMAEWeights()
        w.patch_w = hold.w16(sd["embeddings.patch_embeddings.projection.weight"].reshape(D, -1), lo, tmx, out="f32")
        w.patch_b = hold.f32(sd["embeddings.patch_embeddings.projection.bias"])
        w.pos = hold.f32(sinusoid_table(self.num_patches, D))
        if cfg.final_ln:
            w.final_ln_g, w.final_ln_b = hold.f32(sd["layernorm.weight"]), hold.f32(sd["layernorm.bias"])
        layers = (TfLayer * config.num_hidden_layers)()
        zeros = torch.zeros(D)
        for l in range(config.num_hidden_layers):
            q = f"encoder.layer.{l}."
            a = q + "attention.attention."
            if a + "query.bias" in sd:
                bq, bk, bv = sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]
            else:  # transformers <= 4.x: separate q_bias / v_bias parameters, no key bias
                bq, bk, bv = sd.get(a + "q_bias", zeros), zeros, sd.get(a + "v_bias", zeros)
            layers[l] = _tf_layer(
                hold, lo, sd[a + "query.weight"], bq, sd[a + "key.weight"], bk, sd[a + "value.weight"], bv,
                sd[q + "attention.output.dense.weight"], sd[q + "attention.output.dense.bias"],
                (sd[q + "layernorm_before.weight"], sd[q + "layernorm_before.bias"]),
                sd[q + "intermediate.dense.weight"], sd[q + "intermediate.dense.bias"], sd[q + "output.dense.weight"],
                sd[q + "output.dense.bias"], (sd[q + "layernorm_after.weight"], sd[q + "layernorm_after.bias"]), mx=tmx)
        w.layers = C.cast(layers, C.POINTER(TfLayer))
        self._layers = layers
        _lib.check(_lib.lib().mer_videomae_create(C.byref(cfg), C.byref(w), C.byref(self._handle)), "mer_videomae_create")
        self._cfg = cfg

    @classmethod
    def from_hf(cls, hf_model, **kw):
        """The drop-in call for a loaded HuggingFace module (AutoModel.from_pretrained(...) at the reference's call sites).  A real
        checkpoint always gets the load-time comparison against the `accurate` twin (self_check=True; the LayerNorm-ratio gate of
        "auto" is for hand-built state_dicts): `model.self_check_result` / `model.escalated` say what it found."""
        kw.setdefault("self_check", True)
        return cls(hf_model.state_dict(), hf_model.config, **kw)

    def _probe_features(self):
        """(video, token) features of the built-in calibration batch (load-time self-check): one video of blocky noise that drifts over time."""
        g = torch.Generator().manual_seed(20260926)
        c = self._cfg
        S, F, Cn = c.image_size, c.num_frames, c.channels
        base = torch.rand((1, 1, Cn, (S + 7) // 8, (S + 7) // 8), generator=g).repeat_interleave(8, 3).repeat_interleave(8, 4)[..., :S, :S]
        px = (base + 0.2 * torch.rand((1, F, Cn, S, S), generator=g) * torch.linspace(0.2, 1.0, F).view(1, F, 1, 1, 1) - 0.5) / 0.225
        out = self.forward_raw(px.contiguous().to(self.device))[0]
        return out.mean(1), out[0]

    def forward_raw(self, pixel_values, *, hidden=True, seg_start=None, seg_len=None):
        x = pixel_values
        if not x.is_cuda:
            x = x.to(self.device)
        x = x.to(torch.float32).contiguous()
        B = x.shape[0]
        c = self._cfg
        assert x.shape[1:] == (c.num_frames, c.channels, c.image_size, c.image_size), x.shape
        D = self.config.hidden_size
        out = torch.empty((B, self.num_patches, D), dtype=torch.float32, device=self.device) if hidden else None
        ss, sl, nseg = self._seg(seg_start, seg_len)
        pooled = torch.empty((nseg, D), dtype=torch.float32, device=self.device) if nseg else None
        wp, wn = self._workspace(_lib.lib().mer_videomae_workspace_bytes(self._handle, B))
        _lib.check(_lib.lib().mer_videomae_forward(
            self._handle, x.data_ptr(), B, wp, wn, out.data_ptr() if hidden else None, ss.data_ptr() if nseg else None,
            sl.data_ptr() if nseg else None, nseg, pooled.data_ptr() if nseg else None, stream()), "mer_videomae_forward")
        return out, pooled

    def __call__(self, pixel_values=None, **_):
        return EncoderOutput(last_hidden_state=self.forward_raw(pixel_values)[0])

    def extract_utterance(self, pixel_values):
        """UTTERANCE feature of the VideoMAE branch: the mean of the [F/ts, D] segment means (extract_vision_huggingface.py:156-158
        then :183-189) == the mean over all patches of the video (segments are equally sized) -> [B, D]."""
        B = pixel_values.shape[0]
        return self.forward_raw(pixel_values, hidden=False, seg_start=[b * self.num_patches for b in range(B)],
                                seg_len=[self.num_patches] * B)[1]

    def extract_segments(self, pixel_values):
        """Fused path of extract_vision_huggingface.py:156-158: view(F/ts, patches_per_frame, D).mean(1) per video
        -> [B * F/ts, D]."""
        B = pixel_values.shape[0]
        nseg = self._cfg.num_frames // self._cfg.tubelet_size
        per = self.num_patches // nseg
        starts = [b * self.num_patches + s * per for b in range(B) for s in range(nseg)]
        return self.forward_raw(pixel_values, hidden=False, seg_start=starts, seg_len=[per] * len(starts))[1]


# =================================================================================================
class HipBertModel(_HipModule):
    """BERT / RoBERTa family (post-LN encoder-only text models): BERT, RoBERTa, MacBERT / PERT / LERT / Chinese-RoBERTa-wwm
    (BERT keys), ELECTRA (+ `embeddings_project` for electra-small) and ALBERT (factorised embeddings, one block shared by
    all layers, gelu_new) — the encoder-only entries of extract_text_huggingface.py:20-58 with standard attention.
    Not covered: DeBERTa (disentangled attention), XLNet / T5 / MPNet (relative attention), the decoder-only LLMs."""
    _destroy = "mer_bert_destroy"

    def __init__(self, state_dict, config, device="cuda:0", dtype="f16", precision="mean"):
        super().__init__()
        sd = _sd_of(state_dict)
        self.config = config
        self.device = torch.device(device)
        _, tf_passes, a32 = _prec(precision)
        lo, tmx = _planes(tf_passes)
        hold = self._hold = _Holder(device, dtype)
        acts = {"gelu": MER_ACT_GELU, "gelu_new": MER_ACT_GELU_TANH}
        if config.hidden_act not in acts:
            raise _lib.MerError(f"hidden_act={config.hidden_act} unsupported")
        roberta = config.model_type in ("roberta", "xlm-roberta")
        albert = "encoder.embedding_hidden_mapping_in.weight" in sd        # ALBERT: factorised embeddings + ONE shared block
        cfg = BertConfig()
        cfg.tf = _tf_config(config.hidden_size, config.num_attention_heads, config.intermediate_size, config.num_hidden_layers,
                            False, acts[config.hidden_act], config.layer_norm_eps, dtype, tf_passes, attn_f32=a32)
        cfg.vocab, cfg.max_pos, cfg.type_vocab = config.vocab_size, config.max_position_embeddings, config.type_vocab_size
        cfg.pad_id = config.pad_token_id if config.pad_token_id is not None else 0
        cfg.pos_mode = 1 if roberta else 0
        cfg.emb_ln_eps = config.layer_norm_eps
        w = BertWeights()
        w.word = hold.f32(sd["embeddings.word_embeddings.weight"])
        w.pos = hold.f32(sd["embeddings.position_embeddings.weight"])
        w.type = hold.f32(sd["embeddings.token_type_embeddings.weight"])
        w.emb_ln_g, w.emb_ln_b = hold.f32(sd["embeddings.LayerNorm.weight"]), hold.f32(sd["embeddings.LayerNorm.bias"])
        # factorised embeddings: ELECTRA `embeddings_project` (electra-small), ALBERT `encoder.embedding_hidden_mapping_in`
        proj = "encoder.embedding_hidden_mapping_in." if albert else "embeddings_project."
        cfg.emb_dim = sd["embeddings.word_embeddings.weight"].shape[1]
        if proj + "weight" in sd:
            w.emb_proj_w = hold.w16(sd[proj + "weight"], lo, tmx)
            w.emb_proj_b = hold.f32(sd[proj + "bias"])
        elif cfg.emb_dim != config.hidden_size:
            raise _lib.MerError("embedding size differs from the hidden size but the checkpoint has no embedding projection")
        layers = (TfLayer * config.num_hidden_layers)()
        if albert:
            if getattr(config, "num_hidden_groups", 1) != 1 or getattr(config, "inner_group_num", 1) != 1:
                raise _lib.MerError("ALBERT with more than one layer group / inner group is not supported")
            q = "encoder.albert_layer_groups.0.albert_layers.0."
            a = q + "attention."
            shared = _tf_layer(
                hold, lo, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], sd[a + "key.bias"], sd[a + "value.weight"],
                sd[a + "value.bias"], sd[a + "dense.weight"], sd[a + "dense.bias"], (sd[a + "LayerNorm.weight"], sd[a + "LayerNorm.bias"]),
                sd[q + "ffn.weight"], sd[q + "ffn.bias"], sd[q + "ffn_output.weight"], sd[q + "ffn_output.bias"],
                (sd[q + "full_layer_layer_norm.weight"], sd[q + "full_layer_layer_norm.bias"]), mx=tmx)
            for l in range(config.num_hidden_layers):
                layers[l] = shared
        else:
            for l in range(config.num_hidden_layers):
                q = f"encoder.layer.{l}."
                a = q + "attention.self."
                layers[l] = _tf_layer(
                    hold, lo, sd[a + "query.weight"], sd[a + "query.bias"], sd[a + "key.weight"], sd[a + "key.bias"],
                    sd[a + "value.weight"], sd[a + "value.bias"], sd[q + "attention.output.dense.weight"],
                    sd[q + "attention.output.dense.bias"],
                    (sd[q + "attention.output.LayerNorm.weight"], sd[q + "attention.output.LayerNorm.bias"]),
                    sd[q + "intermediate.dense.weight"], sd[q + "intermediate.dense.bias"], sd[q + "output.dense.weight"],
                    sd[q + "output.dense.bias"], (sd[q + "output.LayerNorm.weight"], sd[q + "output.LayerNorm.bias"]), mx=tmx)
        w.layers = C.cast(layers, C.POINTER(TfLayer))
        self._layers = layers
        _lib.check(_lib.lib().mer_bert_create(C.byref(cfg), C.byref(w), C.byref(self._handle)), "mer_bert_create")
        self._cfg = cfg

    @classmethod
    def from_hf(cls, hf_model, **kw):
        """The drop-in call for a loaded HuggingFace module (AutoModel.from_pretrained(...) at the reference's call sites).  A real
        checkpoint always gets the load-time comparison against the `accurate` twin (self_check=True; the LayerNorm-ratio gate of
        "auto" is for hand-built state_dicts): `model.self_check_result` / `model.escalated` say what it found."""
        kw.setdefault("self_check", True)
        return cls(hf_model.state_dict(), hf_model.config, **kw)

    def _probe_features(self):
        """(utterance, token) features of the built-in calibration batch (load-time self-check): 4 sentences of 32 random ids."""
        g = torch.Generator().manual_seed(20260926)
        B, T = 4, 32
        hi = max(int(getattr(self.config, "vocab_size", 1000)) - 1, 8)
        ids = torch.randint(min(5, hi - 1), hi, (B, T), generator=g)
        _, fr, pooled = self.forward_raw(ids.to(self.device), lengths=[T] * B, frames=True, seg_start=[b * T + 1 for b in range(B)], seg_len=[T - 2] * B)
        return pooled, fr

    def forward_raw(self, input_ids, *, lengths=None, token_type_ids=None, hidden_states=False, frames=False, seg_start=None,
                    seg_len=None):
        ids = input_ids.to(self.device, torch.int64).contiguous()
        B, T = ids.shape
        D, nl = self.config.hidden_size, self.config.num_hidden_layers
        tt = token_type_ids.to(self.device, torch.int64).contiguous() if token_type_ids is not None else None
        ln = self._ints(lengths) if lengths is not None else None
        hs = torch.empty((nl + 1, B, T, D), dtype=torch.float32, device=self.device) if hidden_states else None
        fr = torch.empty((B * T, D), dtype=torch.float32, device=self.device) if frames else None
        ss, sl, nseg = self._seg(seg_start, seg_len)
        pooled = torch.empty((nseg, D), dtype=torch.float32, device=self.device) if nseg else None
        wp, wn = self._workspace(_lib.lib().mer_bert_workspace_bytes(self._handle, B, T, int(hidden_states)))
        _lib.check(_lib.lib().mer_bert_forward(
            self._handle, ids.data_ptr(), tt.data_ptr() if tt is not None else None, ln.data_ptr() if ln is not None else None,
            B, T, wp, wn, hs.data_ptr() if hs is not None else None, fr.data_ptr() if fr is not None else None,
            ss.data_ptr() if nseg else None, sl.data_ptr() if nseg else None, nseg, pooled.data_ptr() if nseg else None,
            stream()), "mer_bert_forward")
        return hs, fr, pooled

    def __call__(self, input_ids=None, attention_mask=None, token_type_ids=None, output_hidden_states=False, **_):
        lengths = None
        if attention_mask is not None:
            m = attention_mask.to("cpu")
            lengths = m.sum(dim=1).to(torch.int32)
            T = m.shape[1]
            if not bool((m == (torch.arange(T)[None] < lengths[:, None]).to(m.dtype)).all()):
                raise _lib.MerError("attention_mask must be a right-padded prefix mask")
        hs, _, _ = self.forward_raw(input_ids, lengths=lengths, token_type_ids=token_type_ids, hidden_states=True)
        return EncoderOutput(last_hidden_state=hs[-1], hidden_states=tuple(hs[i] for i in range(hs.shape[0])) if output_hidden_states else None)

    def extract_utterance(self, input_ids, lengths, start, end):
        """Fused path: last-4 sum, drop `start` leading / `-end` trailing special tokens
        (extract_text_huggingface.py:228-231), mean over the rest -> [B, D]."""
        B, T = input_ids.shape
        lengths = [int(x) for x in lengths]
        starts = [b * T + start for b in range(B)]
        lens = [max(0, (lengths[b] + (end if end is not None else 0)) - start) for b in range(B)]
        return self.forward_raw(input_ids, lengths=lengths, seg_start=starts, seg_len=lens)[2]
