"""Op-level parity (GPU): every HIP kernel against a plain torch fp32/fp64 CPU reference of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import assert_close

pytestmark = pytest.mark.gpu


def _ops():
    from mertools_amd import ops
    return ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _act_ref(x, act):
    if act == "gelu":
        return F.gelu(x)
    if act == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if act == "relu":
        return F.relu(x)
    if act in ("gelu_new", "gelu_tanh"):
        return F.gelu(x, approximate="tanh")
    return x


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("M,N,K,tile", [(128, 128, 64, 1), (300, 200, 136, 1), (257, 48, 72, 2), (1000, 768, 768, 0),
                                        (64, 2304, 768, 1), (130, 64, 3072, 2), (300, 200, 136, 3), (1000, 768, 768, 3),
                                        (513, 256, 64, 3), (2000, 520, 3072, 3)])
def test_gemm16_single_pass(dev, dtype, M, N, K, tile):
    ops = _ops()
    t16 = ops.torch16(dtype)
    a = _rand((M, K), 1).to(t16)
    w = (_rand((N, K), 2) * 0.05).to(t16)
    bias = _rand((N,), 3)
    res = _rand((M, N), 4)
    ref = a.double() @ w.double().T + bias.double()
    ref = _act_ref(ref, "gelu") + res.double()
    c32, c16, _ = ops.gemm16(a.to(dev), w.to(dev), bias=bias.to(dev), act="gelu", residual=res.to(dev), out32=True, out16=True,
                             dtype=dtype, tile=tile)
    torch.cuda.synchronize()
    assert_close(c32.cpu(), ref.float(), 2e-5, f"gemm16 {dtype} c32")
    assert_close(c16.float().cpu(), ref.float(), 1e-2 if dtype == "bf16" else 1.5e-3, f"gemm16 {dtype} c16")


@pytest.mark.parametrize("M,N,K,tile", [(300, 200, 136, 1), (1000, 768, 768, 1), (257, 48, 6144, 2), (1000, 768, 768, 3),
                                        (700, 264, 96, 3)])
def test_gemm16_three_pass_is_fp32_grade(dev, M, N, K, tile):
    ops = _ops()
    a = _rand((M, K), 5)
    w = _rand((N, K), 6) * 0.05
    ah, al = ops.split16(a.to(dev), "f16")
    wh, wl = ops.split16_host(w, "f16")
    ref = a.double() @ w.double().T
    c32, c16h, c16l = ops.gemm16(ah, wh.to(dev), a_lo=al, w_lo=wl.to(dev), out32=True, out16=True, out16_lo=True, passes=3,
                                 dtype="f16", tile=tile)
    torch.cuda.synchronize()
    assert_close(c32.cpu(), ref.float(), 2e-5, "gemm16 x3 c32")
    # hi+lo planes of the output reproduce the fp32 result to ~2^-21
    assert_close((c16h.float() + c16l.float()).cpu(), ref.float(), 1e-5, "gemm16 x3 hi+lo out")
    # and a single pass on the same data is visibly worse (guards against silently running 3 passes everywhere)
    c1, _, _ = ops.gemm16(ah, wh.to(dev), out32=True, passes=1, dtype="f16", tile=tile)
    e1 = (c1.cpu().double() - ref).abs().max() / ref.abs().max()
    assert e1 > 5e-5
    # 2-pass (weights split only): the weight-rounding error is gone, the activation rounding stays
    c2, _, _ = ops.gemm16(ah, wh.to(dev), w_lo=wl.to(dev), out32=True, passes=2, dtype="f16", tile=tile)
    ref2 = ah.float().cpu().double() @ w.double().T
    assert_close(c2.cpu(), ref2.float(), 2e-5, "gemm16 w2 vs exact-weights reference")


@pytest.mark.parametrize("M,N,K,tile", [(300, 200, 136, 1), (1000, 768, 768, 1), (257, 48, 6144, 2), (2048, 768, 768, 3), (700, 264, 96, 3)])
def test_gemm16_activation_split_passes6(dev, M, N, K, tile):
    """passes = 6 (round 5): a_hi*w_hi + a_lo*w_hi — the activation split alone.  Against A (fp32) x f16(W): fp32-grade, with a plain
    bias, with a per-sequence bias table (how the `mean_a2` preset applies the weight residual), and through the pre-blocked plane."""
    ops = _ops()
    a = _rand((M, K), 15)
    a[:, 3] *= 300.0          # a channel a single 16-bit plane cannot carry next to the others
    w = _rand((N, K), 16) * 0.05
    ah, al = ops.split16(a.to(dev), "f16")
    wh, _ = ops.split16_host(w, "f16")
    ref = a.double() @ wh.double().T
    c6, _, _ = ops.gemm16(ah, wh.to(dev), a_lo=al, out32=True, passes=6, dtype="f16", tile=tile)
    torch.cuda.synchronize()
    assert_close(c6.cpu(), ref.float(), 2e-5, "gemm16 passes=6 vs fp32 A x f16(W)")
    c1, _, _ = ops.gemm16(ah, wh.to(dev), out32=True, passes=1, dtype="f16", tile=tile)
    assert ((c1.cpu().double() - ref).abs().max() / ref.abs().max()) > 5e-5      # one plane of A is visibly worse on this data
    T = 50
    tab = _rand(((M + T - 1) // T, N), 17).to(dev)
    ct, c16, _ = ops.gemm16(ah, wh.to(dev), a_lo=al, bias=tab, bias_seg_rows=T, out32=True, out16=True, passes=6, dtype="f16", tile=tile)
    torch.cuda.synchronize()
    reft = ref + tab.cpu().double().repeat_interleave(T, 0)[:M]
    assert_close(ct.cpu(), reft.float(), 2e-5, "gemm16 passes=6 + bias table")
    assert_close(c16.float().cpu(), reft.float(), 1e-3, "gemm16 passes=6 16-bit output")
    if tile == 3 and K % 32 == 0 and N >= 192:
        cb, _, _ = ops.gemm16(ah, wh.to(dev), a_lo=al, out32=True, passes=6, dtype="f16", tile=tile, w_hi_blk=ops.w_block_pack(wh.to(dev)))
        torch.cuda.synchronize()
        assert torch.equal(cb, c6), "pre-blocked plane changes the bits of passes=6"


def test_gemm16_implicit_conv1d(dev):
    """Strided Conv1d over channels-last input as a row-mapped GEMM (HuBERT conv layers 1..6)."""
    ops = _ops()
    B, Tin, C, k, s, Cout = 3, 41, 64, 3, 2, 96
    x = _rand((B, Tin, C), 7).half()
    w = (_rand((Cout, C, k), 8) * 0.1).half()
    Tout = (Tin - k) // s + 1
    ref = F.conv1d(x.float().transpose(1, 2).double(), w.double(), stride=s).transpose(1, 2).reshape(B * Tout, Cout)
    w2 = w.permute(0, 2, 1).reshape(Cout, k * C).contiguous()
    c32, _, _ = ops.gemm16(x.to(dev).reshape(B * Tin, C), w2.to(dev), out32=True, dtype="f16", M=B * Tout, lda=s * C,
                           a_rows_per_batch=Tout, a_batch_stride=Tin * C)
    torch.cuda.synchronize()
    assert_close(c32.cpu(), ref.float(), 2e-5, "implicit conv1d")


def test_gemm16_batched_posconv(dev):
    """Grouped positional conv = pack + batched implicit GEMM with residual epilogue."""
    from mertools_amd._lib import GemmArgs
    ops = _ops()
    B, T, D, G, K = 2, 37, 96, 4, 16
    Dg = D // G
    x = _rand((B, T, D), 9)
    w = _rand((D, Dg, K), 10) * 0.05
    bias = _rand((D,), 11)
    pc = F.conv1d(x.transpose(1, 2).double(), w.double(), bias.double(), padding=K // 2, groups=G)[:, :, :-1]
    ref = (x.double() + F.gelu(pc).transpose(1, 2)).float()
    xd = x.to(dev)
    ph, pl = ops.posconv_pack(xd, G, K, dtype="f16", lo=True)
    # the pack itself
    xp = torch.zeros(B, G, T + K, Dg)
    xp[:, :, K // 2:K // 2 + T] = x.view(B, T, G, Dg).permute(0, 2, 1, 3)
    torch.cuda.synchronize()
    assert_close((ph.float() + pl.float()).cpu(), xp, 1e-6, "posconv_pack")
    wg = w.view(G, Dg, Dg, K).permute(0, 1, 3, 2).reshape(G * Dg, K * Dg).contiguous()
    wh, wl = ops.split16_host(wg, "f16")
    wh, wl = wh.to(dev), wl.to(dev)
    bd = bias.to(dev)
    out = torch.empty((B, T, D), device=dev)
    g = GemmArgs()
    g.M, g.N, g.K, g.dtype = T, Dg, K * Dg, 0
    g.a_hi, g.a_lo, g.lda = ph.data_ptr(), pl.data_ptr(), Dg
    g.w_hi, g.w_lo, g.ldw = wh.data_ptr(), wl.data_ptr(), K * Dg
    g.bias, g.act = bd.data_ptr(), 1
    g.residual, g.ldr = xd.data_ptr(), D
    g.c32, g.ldc32 = out.data_ptr(), D
    g.nbatch, g.nb_inner = B * G, G
    g.a_so, g.a_si = G * (T + K) * Dg, (T + K) * Dg
    g.w_si, g.bias_si = Dg * K * Dg, Dg
    g.c_so, g.c_si = T * D, Dg
    g.passes, g.tile = 3, 0
    ops.gemm16_raw(g)
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, 2e-5, "batched posconv gemm")


@pytest.mark.parametrize("D", [512, 768, 1024, 48 * 4])
def test_layernorm(dev, D):
    ops = _ops()
    M = 77
    x = _rand((M, D), 12, 3.0) + 0.5
    g, b = _rand((D,), 13) * 0.1 + 1.0, _rand((D,), 14) * 0.1
    ref = F.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-5)
    o32, oh, ol = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-5, out32=True, out16=True, out16_lo=True)
    torch.cuda.synchronize()
    assert_close(o32.cpu(), ref.float(), 2e-6, "layernorm fp32")
    assert_close((oh.float() + ol.float()).cpu(), ref.float(), 2e-6, "layernorm planes")
    o32g, _, _ = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-5, act="gelu")
    torch.cuda.synchronize()
    assert_close(o32g.cpu(), F.gelu(ref).float(), 2e-6, "layernorm+gelu")


@pytest.mark.parametrize("D", [512, 768, 1024])
def test_layernorm_many_rows(dev, D):
    """M large enough for the multi-row kernel (a wave walks several rows, gamma / beta in LDS) against the fp64 reference, and
    bit for bit against the one-row kernel the same rows take in a small launch (same arithmetic per row)."""
    ops = _ops()
    M = 33003                                                     # > 8 rows per resident wave slot, not a multiple of anything
    x = (_rand((M, D), 12, 3.0) + 0.5).to(dev)
    g, b = (_rand((D,), 13) * 0.1 + 1.0).to(dev), (_rand((D,), 14) * 0.1).to(dev)
    ref = F.layer_norm(x.cpu().double(), (D,), g.cpu().double(), b.cpu().double(), 1e-5).float()
    o32, oh, ol = ops.layernorm(x, g, b, 1e-5, out32=True, out16=True, out16_lo=True)
    og, _, _ = ops.layernorm(x, g, b, 1e-5, act="gelu")
    s32, _, _ = ops.layernorm(x[:1000].contiguous(), g, b, 1e-5, out32=True)     # small M: the one-row kernel
    torch.cuda.synchronize()
    assert_close(o32[:1000].cpu(), s32.cpu(), 1e-6, "multi-row vs one-row LayerNorm")
    assert_close(o32.cpu(), ref, 2e-6, "layernorm fp32 (multi-row)")
    assert_close((oh.float() + ol.float()).cpu(), ref, 2e-6, "layernorm planes (multi-row)")
    assert_close(og.cpu(), F.gelu(ref.double()).float(), 2e-6, "layernorm+gelu (multi-row)")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("B,T,H", [(2, 64, 2), (2, 50, 3), (1, 197, 12), (2, 249, 4), (1, 257, 2), (1, 499, 2), (1, 600, 2), (2, 1568, 1)])
def test_attention(dev, dtype, B, T, H):
    ops = _ops()
    t16 = ops.torch16(dtype)
    D = H * 64
    qkv = _rand((B * T, 3 * D), 15).to(t16)
    q, k, v = [t.float().view(B, T, H, 64).transpose(1, 2).double() for t in qkv.split(D, dim=1)]
    p = torch.softmax(q @ k.transpose(2, 3) / 8.0, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B * T, D).float()
    oh, ol = ops.attention(qkv.to(dev), B, T, H, 0.125, out_lo=True)
    torch.cuda.synchronize()
    # P is rounded to 16 bits before P@V: that bounds the error, not the fp32 softmax
    assert_close((oh.float() + ol.float()).cpu(), ref, 6e-3 if dtype == "bf16" else 8e-4, f"attention {dtype}")


@pytest.mark.parametrize("B,T,H,lens", [(3, 50, 2, None), (2, 197, 12, None), (2, 249, 3, [249, 100]), (1, 600, 2, None), (4, 16, 1, [16, 1, 7, 15]), (1, 1568, 1, None)])
def test_attention_f32(dev, B, T, H, lens):
    """mer_attention_f32 (the "accurate" preset's attention: fp32 q | k | v on the exact fp32 MFMA, hi + lo output planes) against fp64:
    1e-5, where the f16 kernel holds 8e-4 — any T, ragged key lengths, scores of large dynamic range."""
    ops = _ops()
    D = H * 64
    qkv = _rand((B * T, 3 * D), 16)
    qkv[:, :D] *= 3.0                       # logits up to +-60: the softmax is sharp on some rows, flat on others
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2).double() for t in qkv.split(D, dim=1)]
    s = q @ k.transpose(2, 3) / 8.0
    if lens is not None:
        mask = torch.arange(T)[None, :] >= torch.tensor(lens)[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * T, D)
    kv = torch.tensor(lens, dtype=torch.int32, device=dev) if lens is not None else None
    oh, ol = ops.attention_f32(qkv.to(dev), B, T, H, 0.125, kv_len=kv)
    oh1, _ = ops.attention_f32(qkv.to(dev), B, T, H, 0.125, kv_len=kv, out_lo=False)
    torch.cuda.synchronize()
    assert torch.equal(oh, oh1)
    got = (oh.float() + ol.float()).cpu().view(B, T, D)
    ref = ref.view(B, T, D)
    for b in range(B):                       # (rows >= the sequence's length are unspecified, as for mer_attention)
        n = lens[b] if lens is not None else T
        assert_close(got[b, :n], ref[b, :n].float(), 1e-5, f"attention_f32 B={B} T={T} sequence {b}")


@pytest.mark.parametrize("B,T,H,tile", [(3, 50, 2, 1), (2, 197, 4, 3), (1, 600, 2, 3)])
def test_qkv_headmajor_gemm_and_attention(dev, B, T, H, tile):
    """QKV projection written head-major by the GEMM epilogue + attention reading that layout == the row-major path."""
    ops = _ops()
    D = H * 64
    x = _rand((B * T, D), 60).half()
    w = (_rand((3 * D, D), 61) * 0.05).half()
    bias = _rand((3 * D,), 62)
    xd, wd, bd = x.to(dev), w.to(dev), bias.to(dev)
    _, qkv_rm, _ = ops.gemm16(xd, wd, bias=bd, out16=True, dtype="f16", tile=tile)
    _, qkv_hm, _ = ops.gemm16(xd, wd, bias=bd, out16=True, dtype="f16", tile=tile, headmajor=(T, H))
    torch.cuda.synchronize()
    exp = qkv_rm.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4).contiguous()
    assert torch.equal(qkv_hm.view(3, B, H, T, 64), exp), "head-major scatter must hold exactly the row-major values"
    o_rm, _ = ops.attention(qkv_rm, B, T, H, 0.125)
    o_hm, _ = ops.attention_hm(qkv_hm.view(3, B, H, T, 64), B, T, H, 0.125)
    torch.cuda.synchronize()
    assert torch.equal(o_rm, o_hm)


@pytest.mark.parametrize("T,lens", [(64, [64, 17, 40]),               # 4-wave single-pass kernel
                                    (197, [197, 30, 180, 100, 192, 193]),   # 8-wave kernel: masks in the last two tiles only / in every tile (short rows)
                                    (249, [249, 1, 225, 224, 16]),
                                    (600, [600, 64, 65, 300, 599]),      # streaming kernel: the mask lives in the last key block of each row
                                    (1568, [1568, 1000])])
def test_attention_kv_len_mask(dev, T, lens):
    ops = _ops()
    B, H = len(lens), 2
    D = H * 64
    lens = torch.tensor(lens, dtype=torch.int32)
    qkv = _rand((B * T, 3 * D), 16).half()
    q, k, v = [t.float().view(B, T, H, 64).transpose(1, 2).double() for t in qkv.split(D, dim=1)]
    s = q @ k.transpose(2, 3) / 8.0
    mask = torch.arange(T)[None, :] < lens[:, None]
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, D).float()
    oh, _ = ops.attention(qkv.to(dev), B, T, H, 0.125, kv_len=lens.to(dev))
    torch.cuda.synchronize()
    out = oh.float().cpu().view(B, T, D)
    for b in range(B):
        assert_close(out[b, :lens[b]], ref[b, :lens[b]], 1.5e-3, f"masked attention row {b}")


def test_hubert_conv0_groupnorm_gelu(dev):
    ops = _ops()
    B, L, C, k, s = 2, 4000, 512, 10, 5
    wav = _rand((B, L), 17)
    w = _rand((C, k), 18) * 0.3
    g, b = _rand((C,), 19) * 0.1 + 1.0, _rand((C,), 20) * 0.1
    y = F.conv1d(wav[:, None].double(), w[:, None].double(), stride=s)
    ref = F.gelu(F.group_norm(y, C, g.double(), b.double(), 1e-5)).transpose(1, 2).float()
    oh, ol = ops.hubert_conv0_gn(wav.to(dev), w.to(dev), g.to(dev), b.to(dev), 1e-5, stride=s, lo=True)
    torch.cuda.synchronize()
    assert_close((oh.float() + ol.float()).cpu(), ref, 5e-6, "conv0+GN+GELU")


def test_hubert_conv0_statistics_on_a_correlated_signal(dev):
    """The GroupNorm statistics come from the clip's 10x10 autocorrelation (w'Rw), not from the conv output.  The hard case for
    that identity: a DC offset plus a slow sine (R entries ~0.3, all nearly equal) under difference filters whose output energy
    is five orders of magnitude smaller — the moments must be exact enough to survive the cancellation.  Ragged rows: statistics
    over the row's own frames only, several moment chunks per clip."""
    ops = _ops()
    B, L, C, k, s = 3, 40000, 512, 10, 5
    t = torch.arange(L, dtype=torch.float64)
    wav = (0.3 + 0.5 * torch.sin(2 * math.pi * 50.0 * t / 16000.0))[None].repeat(B, 1)
    wav = (wav + 1e-3 * _rand((B, L), 31).double()).float()
    w = _rand((C, k), 32) * 0.3
    w[0] = 0
    w[0, 0], w[0, 1] = 1.0, -1.0                      # first difference
    w[1] = 0
    w[1, 3], w[1, 4], w[1, 5] = 1.0, -2.0, 1.0         # second difference
    w[2] = 0.1                                         # moving average (DC passes: mean >> std)
    w[3:64] -= w[3:64].mean(dim=1, keepdim=True)       # zero-DC random filters
    g, b = _rand((C,), 33) * 0.1 + 1.0, _rand((C,), 34) * 0.1
    T0 = (L - k) // s + 1
    valid = torch.tensor([T0, 4100, 257], dtype=torch.int32)
    y = F.conv1d(wav[:, None].double(), w[:, None].double(), stride=s)
    ref = torch.empty(B, T0, C)
    for i in range(B):
        v = int(valid[i])
        mu = y[i, :, :v].mean(dim=1, keepdim=True)
        var = y[i, :, :v].var(dim=1, unbiased=False, keepdim=True)
        z = (y[i] - mu) / torch.sqrt(var + 1e-5) * g.double()[:, None] + b.double()[:, None]
        ref[i] = F.gelu(z).t().float()
    oh, ol = ops.hubert_conv0_gn(wav.to(dev), w.to(dev), g.to(dev), b.to(dev), 1e-5, stride=s, lo=True, valid_frames=valid.to(dev))
    torch.cuda.synchronize()
    out = (oh.float() + ol.float()).cpu()
    for i in range(B):
        v = int(valid[i])
        assert_close(out[i, :v], ref[i, :v], 1e-4, f"conv0+GN+GELU, correlated signal, row {i} ({v} frames)")


def test_vit_patchify_and_assemble(dev):
    ops = _ops()
    N, P, S, D = 3, 16, 64, 128
    px = _rand((N, 3, S, S), 21)
    ph, pl = ops.vit_patchify(px.to(dev), P, lo=True)
    g = S // P
    ref = px.view(N, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(N * g * g, 3 * P * P)
    torch.cuda.synchronize()
    assert_close((ph.float() + pl.float()).cpu(), ref, 1e-6, "patchify")
    npatch = g * g
    patch = _rand((N * npatch, D), 22)
    cls, pos = _rand((D,), 23), _rand((npatch + 1, D), 24)
    gam, bet = _rand((D,), 25) * 0.1 + 1, _rand((D,), 26) * 0.1
    tok = torch.cat([cls.expand(N, 1, D), patch.view(N, npatch, D)], 1) + pos[None]
    out = ops.vit_assemble(patch.to(dev), cls.to(dev), pos.to(dev), gam.to(dev), bet.to(dev), 1e-5, N, npatch, D)
    torch.cuda.synchronize()
    assert_close(out.cpu(), F.layer_norm(tok.double(), (D,), gam.double(), bet.double(), 1e-5).float().view(-1, D), 2e-6, "assemble+LN")
    out2 = ops.vit_assemble(patch.to(dev), cls.to(dev), pos.to(dev), None, None, 1e-5, N, npatch, D)
    torch.cuda.synchronize()
    assert_close(out2.cpu(), tok.view(-1, D), 1e-7, "assemble")


@pytest.mark.parametrize("pos_mode", [0, 1])
def test_bert_embed(dev, pos_mode):
    ops = _ops()
    B, T, D, V = 3, 20, 256, 100
    g = torch.Generator().manual_seed(27)
    ids = torch.randint(3, V, (B, T), generator=g)
    pad = 1
    ids[1, 12:] = pad
    ids[2, 5:] = pad
    tt = torch.zeros_like(ids)
    word, pos, typ = _rand((V, D), 28), _rand((T + 4, D), 29), _rand((2, D), 30)
    gam, bet = _rand((D,), 31) * 0.1 + 1, _rand((D,), 32) * 0.1
    if pos_mode == 1:
        m = (ids != pad).long()
        pid = torch.cumsum(m, 1) * m + pad
    else:
        pid = torch.arange(T)[None].expand(B, T)
    ref = F.layer_norm((word[ids] + typ[tt] + pos[pid]).double(), (D,), gam.double(), bet.double(), 1e-12).float().view(-1, D)
    o32, oh = ops.bert_embed(ids.to(dev), None, word.to(dev), pos.to(dev), typ.to(dev), pos_mode, pad, gam.to(dev), bet.to(dev), 1e-12)
    torch.cuda.synchronize()
    assert_close(o32.cpu(), ref, 2e-6, "bert_embed")
    assert_close(oh.float().cpu(), ref, 1e-3, "bert_embed f16")


def test_sum_pool(dev):
    ops = _ops()
    M, D = 90, 768
    hs = [_rand((M, D), 40 + i) for i in range(4)]
    ref_fr = ((hs[0] + hs[1]) + hs[2]) + hs[3]
    starts = torch.tensor([0, 30, 31, 89], dtype=torch.int32)
    lens = torch.tensor([30, 1, 58, 1], dtype=torch.int32)
    fr, pooled = ops.sum_pool([h.to(dev) for h in hs], starts.to(dev), lens.to(dev), frames=True)
    torch.cuda.synchronize()
    assert torch.equal(fr.cpu(), ref_fr), "last-4 sum must be bit-exact (same add order as torch.stack(...).sum(0))"
    ref_pool = torch.stack([ref_fr[s:s + n].double().mean(0) for s, n in zip(starts.tolist(), lens.tolist())]).float()
    assert_close(pooled.cpu(), ref_pool, 1e-6, "segment mean")
    fr1, _ = ops.sum_pool([hs[0].to(dev)], frames=True)
    torch.cuda.synchronize()
    assert torch.equal(fr1.cpu(), hs[0])


@pytest.mark.parametrize("M,N,K", [(32, 128, 768), (33, 6, 130), (7, 3, 5)])
def test_gemm32_exact_fp32(dev, M, N, K):
    ops = _ops()
    a, w, b = _rand((M, K), 50), _rand((N, K), 51) * 0.1, _rand((N,), 52)
    ref = F.relu(a.double() @ w.double().T + b.double()).float()
    out = ops.gemm32(a.to(dev), w.to(dev), b.to(dev), "relu")
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, 2e-6, "gemm32")
    # transposed operand forms used by the backward pass
    out_t = ops.gemm32(a.T.contiguous().to(dev), w.T.contiguous().to(dev), None, None, trans_a=True, trans_w=True)
    torch.cuda.synchronize()
    assert_close(out_t.cpu(), (a.double() @ w.double().T).float(), 2e-6, "gemm32 transposed")
    acc = ops.gemm32(a.to(dev), w.to(dev), None, None, out=out_t.clone(), accumulate=True)
    torch.cuda.synchronize()
    assert_close(acc.cpu(), (2 * (a.double() @ w.double().T)).float(), 2e-6, "gemm32 accumulate")


def test_gemm16_random_shape_sweep(dev):
    """Ragged / tiny / odd shapes through every tile and pass count (row & column guards, K tails, scalar epilogue)."""
    ops = _ops()
    import random
    rnd = random.Random(1234)
    cases = [(1, 8, 8), (1, 1, 8), (5, 3, 16), (17, 24, 40), (129, 257, 72), (255, 100, 200), (256, 256, 32), (260, 8, 520)]
    cases += [(rnd.randint(1, 700), rnd.randint(1, 90) * 4 if rnd.random() < 0.7 else rnd.randint(1, 300), rnd.randint(1, 60) * 8) for _ in range(12)]
    for idx, (M, N, K) in enumerate(cases):
        a = _rand((M, K), 100 + idx)
        w = _rand((N, K), 200 + idx) * 0.1
        bias = _rand((N,), 300 + idx)
        res = _rand((M, N), 400 + idx)
        ah, al = ops.split16(a.to(dev), "f16")
        wh, wl = ops.split16_host(w, "f16")
        wh, wl = wh.to(dev), wl.to(dev)
        ref1 = F.relu(ah.float().cpu().double() @ wh.float().cpu().double().T + bias.double()) + res.double()
        ref3 = F.relu(a.double() @ w.double().T + bias.double()) + res.double()
        for tile in (1, 2, 3):
            for passes in (1, 2, 3):
                c32, c16, _ = ops.gemm16(ah, wh, a_lo=al if passes == 3 else None, w_lo=wl if passes >= 2 else None, bias=bias.to(dev),
                                         act="relu", residual=res.to(dev), out32=True, out16=True, passes=passes, dtype="f16", tile=tile)
                torch.cuda.synchronize()
                ref = ref1 if passes == 1 else ref3
                tol = 2e-5 if passes != 2 else 1e-3
                assert_close(c32.cpu(), ref.float(), tol, f"gemm16 M={M} N={N} K={K} tile={tile} passes={passes}")
                assert_close(c16.float().cpu(), ref.float(), 2e-3, f"gemm16 c16 M={M} N={N} K={K} tile={tile} passes={passes}")


@pytest.mark.parametrize("M,N,K,act", [(1024, 256, 128, None), (3000, 768, 768, "gelu"), (2048, 2304, 768, None),
                                       (1500, 512, 1536, "quick_gelu"), (1100, 300, 3072, None)])
def test_gemm16_mx_corrected(dev, M, N, K, act):
    """passes=4: a_hi*w_hi on the f16 MFMA + bf8(a_hi) * mxfp4(w - w_hi) on v_mfma_scale_f32_16x16x128_f8f6f4.
    (1) bit-level layout check: equals the fp64 emulation built from the packed plane by an independent decoder;
    (2) accuracy: as close to the true product as the f16 2-pass path (the residual only needs ~3 bits)."""
    from util import bf8_round, mx_decode
    ops = _ops()
    a = _rand((M, K), 31).half()
    w = _rand((N, K), 32) * 0.05
    wh, wl = ops.split16_host(w, "f16")
    packed = ops.mx_pack(w - wh.float())
    bias = _rand((N,), 33)
    res = _rand((M, N), 34)
    c32, c16, _ = ops.gemm16(a.to(dev), wh.to(dev), w_lo=wl.to(dev), w_mx=packed.to(dev), bias=bias.to(dev), act=act,
                             residual=res.to(dev), out32=True, out16=True, passes=4, tile=3)
    torch.cuda.synchronize()
    emu = a.double() @ wh.double().T + bf8_round(a) @ mx_decode(packed, N, K).T + bias.double()
    emu = _act_ref(emu, act) + res.double()
    assert_close(c32.cpu(), emu.float(), 3e-6, "gemm16 mx vs emulation")
    true = _act_ref(a.double() @ w.double().T + bias.double(), act) + res.double()
    e4 = assert_close(c32.cpu(), true.float(), 3e-5, "gemm16 mx vs exact")
    c1, _, _ = ops.gemm16(a.to(dev), wh.to(dev), bias=bias.to(dev), act=act, residual=res.to(dev), out32=True, passes=1, tile=3)
    e1 = (c1.cpu().double() - true).abs().max().item() / true.abs().max().item()
    assert e4 < 0.35 * e1, f"MX correction did not remove the weight-rounding error: {e4:.2e} vs 1-pass {e1:.2e}"
    assert_close(c16.float().cpu(), true.float(), 1.5e-3, "gemm16 mx c16")


def test_gemm16_mx_falls_back_to_two_pass(dev):
    """A K the MX kernel does not cover (K % 128 != 0) runs the f16 2-pass path with w_lo.  Few rows are NOT a reason to fall back
    (round 4: a clip must get the same bits alone and in a batch of 64): 100 rows run the same MX kernel as 4096."""
    ops = _ops()
    M, N, K = 600, 256, 136
    a = _rand((M, K), 41).half()
    w = _rand((N, K), 42) * 0.05
    wh, wl = ops.split16_host(w, "f16")
    packed = ops.mx_pack(w - wh.float())
    assert packed is None
    c4, _, _ = ops.gemm16(a.to(dev), wh.to(dev), w_lo=wl.to(dev), w_mx=None, out32=True, passes=4)
    c2, _, _ = ops.gemm16(a.to(dev), wh.to(dev), w_lo=wl.to(dev), out32=True, passes=2)
    torch.cuda.synchronize()
    assert torch.equal(c4, c2)
    with pytest.raises(Exception):
        ops.gemm16(a.to(dev), wh.to(dev), out32=True, passes=4)   # neither plane
    M, N, K = 4096, 768, 768
    a = _rand((M, K), 43).half()
    w = _rand((N, K), 44) * 0.05
    wh, wl = ops.split16_host(w, "f16")
    mx = ops.mx_pack(w - wh.float()).to(dev)
    big, _, _ = ops.gemm16(a.to(dev), wh.to(dev), w_lo=wl.to(dev), w_mx=mx, out32=True, passes=4)
    for rows in (1, 100, 1000):
        part, _, _ = ops.gemm16(a[:rows].to(dev), wh.to(dev), w_lo=wl.to(dev), w_mx=mx, out32=True, passes=4)
        torch.cuda.synchronize()
        assert torch.equal(part, big[:rows]), rows


@pytest.mark.parametrize("B,T,H,gated", [(2, 50, 2, True), (3, 197, 3, False), (2, 249, 4, True), (1, 499, 2, True)])
def test_attention_with_score_bias(dev, B, T, H, gated):
    """mer_attention_bias: scores += gate[b,h,q] * bias[h,q,k] (WavLM gated relative position bias / BEiT bias)."""
    ops = _ops()
    D = H * 64
    qkv = (_rand((B * T, 3 * D), 51) * 0.5).half()
    ldb = (T + 3) // 4 * 4
    bias = torch.zeros(H, T, ldb)
    bias[:, :, :T] = _rand((H, T, T), 52)
    gate = (1.0 + _rand((B, H, T), 53).abs()) if gated else None
    out = ops.attention_bias(qkv.to(dev), B, T, H, 0.125, bias.to(dev), gate=gate.to(dev) if gated else None)
    torch.cuda.synchronize()
    q, k, v = [t.double().view(B, T, H, 64).transpose(1, 2) for t in qkv.split(D, dim=1)]
    s = q @ k.transpose(2, 3) * 0.125 + (gate.double()[..., None] if gated else 1.0) * bias[None, :, :, :T].double()
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * T, D)
    assert_close(out.float().cpu(), ref.float(), 2e-3, "attention with score bias")


def test_wavlm_gate(dev):
    ops = _ops()
    B, T, H = 2, 37, 3
    x = _rand((B * T, H * 64), 61)
    w, b, cst = _rand((8, 64), 62) * 0.2, _rand((8,), 63) * 0.1, 0.5 + _rand((H,), 64).abs()
    g = ops.wavlm_gate(x.to(dev), w.to(dev), b.to(dev), cst.to(dev), B, T, H)
    torch.cuda.synchronize()
    from oracle.encoders_ref import wavlm_gate
    ref = wavlm_gate(x.view(B, T, H * 64), w, b, cst.view(1, H, 1, 1), H)[..., 0]
    assert_close(g.cpu(), ref, 1e-5, "wavlm gate")


def test_wave_normalize_matches_feature_extractor(dev):
    """GPU version of Wav2Vec2FeatureExtractor's normalisation, from int16 PCM and from fp32."""
    import numpy as np
    from mertools_amd.extract.audio import wav2vec2_normalize
    ops = _ops()
    rng = np.random.default_rng(0)
    pcm = (rng.standard_normal((3, 80000)) * 3000 + 200).clip(-32768, 32767).astype(np.int16)
    ref = torch.cat([wav2vec2_normalize(row.astype(np.float64) / 32768.0) for row in pcm], 0)      # what soundfile -> HF computes
    out = ops.wave_normalize(torch.from_numpy(pcm).to(dev))
    outf = ops.wave_normalize(torch.from_numpy(pcm.astype(np.float32) / 32768.0).to(dev))
    plain = ops.wave_normalize(torch.from_numpy(pcm).to(dev), do_normalize=False)
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, 2e-6, "wave_normalize int16")
    assert_close(outf.cpu(), ref, 2e-6, "wave_normalize fp32")
    assert torch.equal(plain.cpu(), torch.from_numpy(pcm.astype(np.float32) / 32768.0))


def test_image_normalize_u8_matches_clip_preprocess(dev):
    import numpy as np
    from mertools_amd.extract.visual import CLIP_MEAN, CLIP_STD, clip_preprocess
    ops = _ops()
    frames = np.random.default_rng(1).integers(0, 256, (5, 224, 224, 3), dtype=np.uint8)      # BGR, already model-sized
    ref = clip_preprocess(frames, 224)
    out = ops.image_normalize_u8(torch.from_numpy(frames).to(dev), CLIP_MEAN, CLIP_STD, bgr=True)
    torch.cuda.synchronize()
    assert_close(out.cpu(), ref, 2e-6, "image_normalize_u8")


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(2304, 1536, 256), (1000, 776, 96), (70000, 768, 64), (515, 72, 32)])
def test_gemm16_specialised_epilogues_equal_generic(dev, M, N, K, tile):
    """The two specialised epilogues — packed row pairs + streaming stores for 16-bit-only outputs, whole-line 4-columns-per-lane
    for fp32-only outputs with residual — against the generic fp32-staged epilogue (debug switch gemm_generic_epi), bit for bit,
    on every tile class, ragged M / N included; and the generic one against fp64."""
    from mertools_amd import _lib
    ops = _ops()
    lib = _lib.lib()
    a = _rand((M, K), 71)
    w = _rand((N, K), 72) * 0.05
    ah, _ = ops.split16(a.to(dev), "f16", lo=False)
    wh = ops.split16_host(w, "f16")[0].to(dev)
    bias, res = _rand((N,), 73).to(dev), _rand((M, N), 74).to(dev)
    z = a.double() @ w.double().T + bias.cpu().double()
    for act in (None, "gelu", "quick_gelu", "gelu_new", "relu"):
        kw16 = dict(bias=bias, act=act, out16=True, passes=1, tile=tile)
        kw32 = dict(bias=bias, act=act, residual=res, out32=True, passes=1, tile=tile)
        try:
            lib.mer_set_option(b"gemm_generic_epi", 1)
            _, ref16, _ = ops.gemm16(ah, wh, **kw16)
            ref32, _, _ = ops.gemm16(ah, wh, **kw32)
        finally:
            lib.mer_set_option(b"gemm_generic_epi", 0)
        _, c16, _ = ops.gemm16(ah, wh, **kw16)
        c32, _, _ = ops.gemm16(ah, wh, **kw32)
        inplace = res.clone()                                   # residual updated in place (the ABI allows residual == c32)
        g = ops.GemmArgs()
        g.M, g.N, g.K, g.dtype = M, N, K, ops.dt_code("f16")
        g.a_hi, g.lda, g.w_hi, g.ldw = ah.data_ptr(), K, wh.data_ptr(), K
        g.bias, g.act, g.residual, g.ldr, g.c32, g.ldc32 = bias.data_ptr(), ops.ACT[act], inplace.data_ptr(), N, inplace.data_ptr(), N
        g.nbatch, g.nb_inner, g.passes, g.tile = 1, 1, 1, tile
        ops.gemm16_raw(g)
        torch.cuda.synchronize()
        assert torch.equal(c16, ref16), f"packed-pair epilogue differs from the generic one (act={act})"
        assert torch.equal(c32, ref32) and torch.equal(inplace, ref32), f"fp32-only epilogue differs from the generic one (act={act})"
        true = _act_ref(z, act or "none")
        assert_close(ref16.float().cpu(), true.float(), 1.5e-3, f"generic 16-bit epilogue vs fp64 (act={act})")
        assert_close(ref32.cpu(), (true + res.cpu().double()).float(), 1e-3, f"generic fp32 epilogue vs fp64 (act={act})")


@pytest.mark.parametrize("passes", [1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(10240, 2048, 256), (4100, 768, 768), (9000, 1100, 96), (2048, 200, 32), (1500, 2304, 3072)])
def test_gemm16_preblocked_weights(dev, M, N, K, passes):
    """mer_w_block_pack: the 256x256 kernels read the weight planes from [n-tile][k-slab][16 KB LDS image] blocks (1 KiB
    contiguous DMA pieces) instead of the row-major planes; the LDS images are identical, so every output must equal the
    row-major launch bit for bit — ragged N (rows beyond N repeat row N-1), ragged M, K from one slab to 96 slabs."""
    ops = _ops()
    if passes == 4 and K % 128 != 0:
        pytest.skip("MX kernel needs K % 128 == 0")
    a = _rand((M, K), 81)
    w = _rand((N, K), 82) * 0.05
    ah, al = ops.split16(a.to(dev), "f16")
    wh, wl = ops.split16_host(w, "f16")
    wh, wl = wh.to(dev), wl.to(dev)
    mx = ops.mx_pack(w - wh.cpu().float()).to(dev) if passes == 4 else None
    bias, res = _rand((N,), 83).to(dev), _rand((M, N), 84).to(dev)
    kw = dict(a_lo=al if passes == 3 else None, w_lo=wl if passes in (2, 3) else None, w_mx=mx, bias=bias, act="gelu", residual=res,
              out32=True, out16=True, passes=passes, tile=3)
    ref32, ref16, _ = ops.gemm16(ah, wh, **kw)
    hb = ops.w_block_pack(wh)
    lb = ops.w_block_pack(wl) if passes in (2, 3) else None
    assert hb is not None and hb.numel() == (N + 255) // 256 * 256 * K * 2
    c32, c16, _ = ops.gemm16(ah, wh, w_hi_blk=hb, w_lo_blk=lb, **kw)
    torch.cuda.synchronize()
    assert torch.equal(c32, ref32) and torch.equal(c16, ref16)
    true = F.gelu(a.double() @ w.double().T + bias.cpu().double()) + res.cpu().double()
    assert_close(c32.cpu(), true.float(), {1: 1e-3, 2: 5e-4, 3: 2e-5, 4: 5e-4}[passes], "pre-blocked gemm16 vs fp64")


def test_w_block_pack_layout(dev):
    """Block (tn, kt) of the packed plane = rows tn*256 .. +255 (clamped to N-1), k-range [32 kt, 32 kt + 32), row r at byte
    64 r, 16-byte chunk pc holding logical chunk pc ^ ((-(r >> 2)) & 3)."""
    ops = _ops()
    N, K = 300, 96
    w = torch.arange(N * K, dtype=torch.int16).view(N, K).to(dev)
    out = ops.w_block_pack(w.view(torch.float16)).view(torch.int16).cpu().view(2, K // 32, 256, 4, 8)
    wc = w.cpu()
    for tn, kt, r, pc in [(0, 0, 0, 0), (0, 1, 5, 2), (0, 2, 255, 3), (1, 0, 17, 1), (1, 2, 43, 0), (1, 1, 200, 2)]:
        n = min(tn * 256 + r, N - 1)
        lc = pc ^ ((-(r >> 2)) & 3)
        assert torch.equal(out[tn, kt, r, pc], wc[n, kt * 32 + lc * 8: kt * 32 + lc * 8 + 8]), (tn, kt, r, pc)


def _unblock(plane, Mp, N):
    """Blocked [ceil(M/256)][N/32][256 rows][4 physical chunks][8] -> row-major [Mp, N] (physical chunk pc of row r holds
    logical chunk pc ^ ((-(r >> 2)) & 3))."""
    r = torch.arange(256)
    idx = (torch.arange(4)[None, :] ^ ((-(r >> 2)) & 3)[:, None]).to(plane.device)          # [256, 4]: logical -> physical
    b = plane.view(Mp // 256, N // 32, 256, 4, 8)
    logical = b[:, :, r.to(plane.device)[:, None], idx]                                     # [tm, kt, 256, 4, 8] in logical order
    return logical.permute(0, 2, 1, 3, 4).reshape(Mp, N)


@pytest.mark.parametrize("h,w,size", [(256, 320, 224), (300, 200, 224), (100, 100, 224), (480, 640, 224), (231, 517, 224), (224, 224, 224), (40, 52, 32)])
def test_image_resize_crop_u8_matches_pillow(dev, h, w, size):
    """mer_image_resize_crop_u8 (two integer passes on the GPU, cropped region only) == PIL Image.resize(BICUBIC) + centre crop,
    byte for byte — the reference's CLIPImageProcessor path on uint8 frames."""
    import numpy as np
    from PIL import Image
    from mertools_amd.extract.resize import resize_crop_u8, shortest_edge_geometry
    rng = np.random.RandomState(h * 1000 + w)
    frames = rng.randint(0, 256, (5, h, w, 3), dtype=np.uint8)
    frames[:, : h // 2] = np.linspace(0, 255, w)[None, None, :, None].astype(np.uint8)
    out = resize_crop_u8(torch.from_numpy(frames).to(dev), size).cpu().numpy()
    nw, nh, left, top, crop = shortest_edge_geometry(h, w, size)
    for n in range(frames.shape[0]):
        ref = np.asarray(Image.fromarray(frames[n]).resize((nw, nh), resample=Image.BICUBIC))[top:top + crop, left:left + crop]
        assert np.array_equal(out[n], ref), n


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("B,T,H,lens", [(5, 197, 12, None), (3, 50, 2, None), (2, 257, 4, None), (3, 64, 2, [64, 17, 1]), (1, 577, 2, None)])
def test_attention_cls(dev, dtype, B, T, H, lens):
    """mer_attention_cls: the CLS query of every sequence against all keys — softmax(q K^T * scale) V in fp64."""
    ops = _ops()
    t16 = ops.torch16(dtype)
    D = H * 64
    qkv = _rand((B * T, 3 * D), 93).to(t16)
    q = _rand((B, D), 94).to(t16)
    kv = torch.tensor(lens, dtype=torch.int32).to(dev) if lens else None
    out = ops.attention_cls(q.to(dev), qkv.to(dev), B, T, H, 0.125, kv_len=kv)
    torch.cuda.synchronize()
    k = qkv[:, D:2 * D].double().view(B, T, H, 64)
    v = qkv[:, 2 * D:].double().view(B, T, H, 64)
    sc = torch.einsum("bhd,bthd->bht", q.double().view(B, H, 64), k) * 0.125
    if lens:
        for b, n in enumerate(lens):
            sc[b, :, n:] = float("-inf")
    ref = torch.einsum("bht,bthd->bhd", torch.softmax(sc, -1), v).reshape(B, D)
    assert_close(out.float().cpu(), ref.float(), 6e-3 if dtype == "bf16" else 8e-4, f"attention_cls B={B} T={T} H={H}")


def test_w_block_pack_p_layout(dev):
    """mer_w_block_pack_p: block row 16 q + i of every 128-row group holds plane row 8 i + q (layout 0: a lane's eight accumulators
    are eight consecutive columns) or 64 (q / 4) + 4 i + q % 4 (layout 1: two runs of four); k-slab blocking and chunk swizzle as
    mer_w_block_pack."""
    ops = _ops()
    N, K = 512, 96
    w = torch.arange(N * K, dtype=torch.int32).to(torch.int16).view(N, K).to(dev)
    wc = w.cpu()
    for layout in (0, 1):
        out = ops.w_block_pack_p(w.view(torch.float16), layout).view(torch.int16).cpu().view(2, K // 32, 256, 4, 8)
        for tn in range(2):
            for kt in range(K // 32):
                for r in (0, 1, 15, 16, 17, 47, 63, 64, 100, 127, 128, 200, 255):
                    i, q = r & 15, (r >> 4) & 7
                    n = tn * 256 + (r & ~127) + (8 * i + q if layout == 0 else 64 * (q // 4) + 4 * i + q % 4)
                    for pc in range(4):
                        lc = pc ^ ((-(r >> 2)) & 3)
                        assert torch.equal(out[tn, kt, r, pc], wc[n, kt * 32 + lc * 8: kt * 32 + lc * 8 + 8]), (layout, tn, kt, r, pc)
    assert ops.w_block_pack_p(w.view(torch.float16)[:300], 0) is None      # N % 256 != 0: no persistent plane


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(10240, 2048, 256), (4100, 768, 768), (66000, 512, 1536), (1500, 2304, 3072), (25600, 2304, 768)])
def test_gemm16_persistent_equals_tile_kernel(dev, M, N, K, dtype):
    """The persistent 256x256 one-pass kernel (gemm16p_impl.h: register-direct epilogue over a row-permuted weight plane, next tile's
    ring issued before the stores, counted vmcnt across the tile switch) against gemm16_kernel on the same operands, bit for bit:
    every epilogue kind it takes (16-bit out with each activation; fp32 out; fp32 + residual, also updated in place), 51 .. 900
    tiles (0.2 .. 3.5 per workgroup: first-tile, steady-state and last-tile paths), ragged M (the predicated last row tile),
    8 .. 96 slabs.  Launched three times each (a stale LDS read or an early store shows up as run-to-run differences)."""
    from mertools_amd import _lib
    if dtype == "bf16" and (M, N, K) not in ((4100, 768, 768), (25600, 2304, 768)):
        pytest.skip("bf16 shares the code path: two shapes are enough")
    ops = _ops()
    lib = _lib.lib()
    a = _rand((M, K), 91)
    w = _rand((N, K), 92) * 0.05
    ah, _ = ops.split16(a.to(dev), dtype, lo=False)
    wh = ops.split16_host(w, dtype)[0].to(dev)
    hb, hp, hq = ops.w_block_pack(wh), ops.w_block_pack_p(wh, 0), ops.w_block_pack_p(wh, 1)
    assert hp is not None and hq is not None
    bias, res = _rand((N,), 93).to(dev), _rand((M, N), 94).to(dev)
    cases = [dict(out16=True, act=act, bias=bias) for act in (None, "gelu", "quick_gelu", "gelu_new")]
    cases += [dict(out16=True, act=None, bias=None)]
    cases += [dict(out32=True, act=act, bias=bias) for act in (None, "gelu")]
    cases += [dict(out32=True, act=act, bias=bias, residual=res) for act in (None, "gelu")]
    for kw in cases:
        kw = dict(kw, passes=1, tile=3, dtype=dtype)
        try:
            lib.mer_set_option(b"gemm_persist", 0)
            r32, r16, _ = ops.gemm16(ah, wh, w_hi_blk=hb, w_hi_blkp=hp, w_hi_blkq=hq, **kw)
        finally:
            lib.mer_set_option(b"gemm_persist", 1)
        for rep in range(6):
            try:     # both tile heights (256 rows, 192 rows: gemm16p_impl.h TM), three launches each
                lib.mer_set_option(b"gemm_tm", 4 if rep < 3 else 3)
                c32, c16, _ = ops.gemm16(ah, wh, w_hi_blk=hb, w_hi_blkp=hp, w_hi_blkq=hq, **kw)
                torch.cuda.synchronize()
            finally:
                lib.mer_set_option(b"gemm_tm", 0)
            what = f"persistent gemm16 {kw.get('act')} out16={kw.get('out16', False)} res={'residual' in kw} rep {rep}"
            if r16 is not None:
                assert torch.equal(c16.view(torch.int16), r16.view(torch.int16)), what
            if r32 is not None:
                assert torch.equal(c32, r32), what
    # in-place residual (the pre-LN residual stream: residual == c32)
    ref32, _, _ = ops.gemm16(ah, wh, w_hi_blk=hb, bias=bias, residual=res, out32=True, passes=1, tile=3, dtype=dtype)
    for tm in (4, 3):
        inplace = res.clone()
        g = ops.GemmArgs()
        g.M, g.N, g.K, g.dtype = M, N, K, ops.dt_code(dtype)
        g.a_hi, g.lda, g.w_hi, g.ldw = ah.data_ptr(), K, wh.data_ptr(), K
        g.w_hi_blk, g.w_hi_blkp, g.w_hi_blkq = hb.data_ptr(), hp.data_ptr(), hq.data_ptr()
        g.bias, g.act, g.residual, g.ldr, g.c32, g.ldc32 = bias.data_ptr(), ops.ACT[None], inplace.data_ptr(), N, inplace.data_ptr(), N
        g.nbatch, g.nb_inner, g.passes, g.tile = 1, 1, 1, 3
        try:
            lib.mer_set_option(b"gemm_tm", tm)
            ops.gemm16_raw(g)
            torch.cuda.synchronize()
        finally:
            lib.mer_set_option(b"gemm_tm", 0)
        assert torch.equal(inplace, ref32), f"persistent gemm16 ({64 * tm}-row tile), residual updated in place"
    true = a.double() @ w.double().T + bias.cpu().double() + res.cpu().double()
    assert_close(ref32.cpu(), true.float(), 1e-3 if dtype == "f16" else 8e-3, "persistent gemm16 vs fp64")


def test_gemm16_persistent_conv_rows(dev):
    """Implicit-im2col row mapping (HuBERT's strided Conv1d over channels-last planes: overlapping windows, a_rows_per_batch /
    a_batch_stride) through the persistent kernel == the tile kernel, bit for bit, with the GELU epilogue of the conv stack."""
    from mertools_amd import _lib
    ops = _ops()
    lib = _lib.lib()
    B, Tin, C, k, s = 24, 1599, 512, 3, 2
    Tout = (Tin - k) // s + 1
    x = _rand((B * Tin, C), 95)
    w = _rand((C, k * C), 96) * 0.03
    xh, _ = ops.split16(x.to(dev), "f16", lo=False)
    wh = ops.split16_host(w, "f16")[0].to(dev)
    hb, hp = ops.w_block_pack(wh), ops.w_block_pack_p(wh, 0)
    kw = dict(act="gelu", out16=True, passes=1, tile=3, M=B * Tout, lda=s * C, a_rows_per_batch=Tout, a_batch_stride=Tin * C)
    try:
        lib.mer_set_option(b"gemm_persist", 0)
        _, ref, _ = ops.gemm16(xh, wh, w_hi_blk=hb, w_hi_blkp=hp, **kw)
    finally:
        lib.mer_set_option(b"gemm_persist", 1)
    _, out, _ = ops.gemm16(xh, wh, w_hi_blk=hb, w_hi_blkp=hp, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    win = x.view(B, Tin, C).unfold(1, k, s).permute(0, 1, 3, 2).reshape(B * Tout, k * C)    # [B, Tout, C, k] -> (kk, ci) order
    assert_close(out.float().cpu(), F.gelu(win.double() @ w.double().T).float(), 2e-3, "persistent conv GEMM vs fp64")


@pytest.mark.parametrize("T,nseq,K,N", [(197, 40, 768, 2304), (64, 64, 768, 768), (249, 9, 3072, 768), (1568, 3, 1024, 512), (1, 70, 768, 3072), (7, 11, 96, 200)])
def test_seq_bias(dev, T, nseq, K, N):
    """mer_seq_bias: table[s, n] = bias[n] + mean_{rows h, h + st, ... < valid[s]}(A[s T + ., :]) . w_lo[n, :], st the largest power
    of two leaving >= 16 samples, h = st / 2; the 16-bit mean plane is the exact sample mean rounded once.  Against the same
    sample in fp64; a sequence's row must not depend on its batch mates (bit for bit: the same clips in another batch order)."""
    ops = _ops()
    M = T * nseq - (T // 3 if T > 3 else 0)          # last sequence partial
    a = (_rand((T * nseq, K), 101) * 0.7 + 0.3)[:M]
    wlo = _rand((N, K), 102) * 1e-4
    bias = _rand((N,), 103)
    g = torch.Generator().manual_seed(104)
    valid = torch.randint(max(T // 2, 1), T + 1, (nseq,), generator=g, dtype=torch.int32)
    ah = a.to(dev).half()
    wl = wlo.to(dev).half()
    st = 1
    while st * 2 * 16 <= T:
        st *= 2
    for use_valid in (False, True):
        tab = ops.seq_bias(ah, wl, T, bias=bias.to(dev), valid_rows=valid.to(dev) if use_valid else None)
        torch.cuda.synchronize()
        ref = torch.empty((nseq, N), dtype=torch.float64)
        for s in range(nseq):
            lim = min(int(valid[s]) if use_valid else T, M - s * T)
            rows = [s * T + t for t in range(st // 2, lim, st)]
            mean = ah[rows].cpu().double().mean(0).half().double() if rows else torch.zeros(K, dtype=torch.float64)
            ref[s] = mean @ wl.cpu().double().T + bias.double()
        assert_close(tab.cpu(), ref.float(), 2e-6, f"seq_bias T={T} valid={use_valid}")
    if N % 32 == 0:      # n_first: columns below it take the plain bias
        tab = ops.seq_bias(ah, wl, T, bias=bias.to(dev), n_first=N // 2)
        full = ops.seq_bias(ah, wl, T, bias=bias.to(dev))
        torch.cuda.synchronize()
        assert torch.equal(tab[:, N // 2:], full[:, N // 2:]) and torch.equal(tab[:, :N // 2], bias.to(dev)[None, :N // 2].expand(nseq, -1))
    # batch-mate independence: sequences 0 .. 3 alone == their rows of the full table
    if nseq >= 4 and M >= 4 * T:
        full = ops.seq_bias(ah, wl, T, bias=bias.to(dev))
        part = ops.seq_bias(ah[:4 * T].contiguous(), wl, T, bias=bias.to(dev))
        torch.cuda.synchronize()
        assert torch.equal(full[:4], part)


def test_gemm_with_seq_bias_matches_two_pass_on_the_mean(dev):
    """What the table is for: activations with a common mean (what a LayerNorm's beta / GELU leave behind) through a one-pass GEMM
    lose the (W - f16(W)) term; with mer_seq_bias's table as the bias the error drops to the part that does not act through the
    mean — less than half (in the encoders: 1e-3 -> 2-3e-4)."""
    ops = _ops()
    T, nseq, K, N = 197, 24, 768, 768
    a = _rand((T * nseq, K), 121) * 0.3 + 1.0
    w = _rand((N, K), 122) * 0.05
    ah, _ = ops.split16(a.to(dev), "f16", lo=False)
    wh, wl = ops.split16_host(w, "f16")
    tab = ops.seq_bias(ah, wl.to(dev), T)
    o1, _, _ = ops.gemm16(ah, wh.to(dev), bias=tab, bias_seg_rows=T, out32=True, passes=1)
    o0, _, _ = ops.gemm16(ah, wh.to(dev), out32=True, passes=1)
    torch.cuda.synchronize()
    true = ah.double().cpu() @ w.double().T
    e1 = (o1.double().cpu() - true).abs().max() / true.abs().max()
    e0 = (o0.double().cpu() - true).abs().max() / true.abs().max()
    assert e1 < 0.5 * e0, (e0.item(), e1.item())


@pytest.mark.parametrize("T,M", [(197, 9000), (64, 9000), (249, 9000), (40, 9000), (1568, 6000), (249, 70000), (64, 90000)])
def test_gemm16_bias_table(dev, T, M):
    """mer_gemm16 with a per-sequence bias table (bias_seg_rows): output row m takes row m / T of the table.  The persistent
    kernel (rows of the table staged in LDS, selected per output row), the 256x256 / 128x128 tile kernels (generic and fp32
    epilogues) and fp64 agree — bit for bit between the kernels — for 16-bit, fp32 and fp32 + residual outputs, ragged M."""
    from mertools_amd import _lib
    ops = _ops()
    lib = _lib.lib()
    K, N = 768, 768     # (M = 70000 / 90000: 822 / 1056 tiles, three or four per workgroup — the table rows of the NEXT tile are staged under this one)
    nseq = (M + T - 1) // T
    a = _rand((M, K), 111)
    w = _rand((N, K), 112) * 0.05
    tab = _rand((nseq, N), 113).to(dev)
    res = _rand((M, N), 114).to(dev)
    ah, _ = ops.split16(a.to(dev), "f16", lo=False)
    wh = ops.split16_host(w, "f16")[0].to(dev)
    hb, hp, hq = ops.w_block_pack(wh), ops.w_block_pack_p(wh, 0), ops.w_block_pack_p(wh, 1)
    rowseq = (torch.arange(M) // T)
    z = a.double() @ w.double().T + tab.cpu().double()[rowseq]
    outs = {}
    for name, persist, tile in (("persistent", 1, 3), ("persistent192", 1, 3), ("tile256", 0, 3)) + ((("tile128", 0, 1),) if M < 20000 else ()):
        try:
            lib.mer_set_option(b"gemm_persist", persist)
            lib.mer_set_option(b"gemm_tm", 3 if name == "persistent192" else 4)
            kw = dict(bias=tab, bias_seg_rows=T, passes=1, tile=tile, w_hi_blk=hb, w_hi_blkp=hp, w_hi_blkq=hq)
            _, c16, _ = ops.gemm16(ah, wh, act="gelu", out16=True, **kw)
            c32, _, _ = ops.gemm16(ah, wh, out32=True, **kw)
            r32, _, _ = ops.gemm16(ah, wh, out32=True, residual=res, **kw)
            torch.cuda.synchronize()
        finally:
            lib.mer_set_option(b"gemm_persist", 1)
            lib.mer_set_option(b"gemm_tm", 0)
        outs[name] = (c16, c32, r32)
    for name in [n for n in ("persistent192", "tile256", "tile128") if n in outs]:
        for i, what in enumerate(("16-bit gelu", "fp32", "fp32 + residual")):
            assert torch.equal(outs["persistent"][i].view(torch.int16 if i == 0 else torch.int32),
                               outs[name][i].view(torch.int16 if i == 0 else torch.int32)), f"bias table T={T}: persistent vs {name}, {what}"
    assert_close(outs["persistent"][0].float().cpu(), F.gelu(z).float(), 2e-3, f"bias table T={T}, 16-bit gelu vs fp64")
    assert_close(outs["persistent"][1].cpu(), z.float(), 1e-3, f"bias table T={T}, fp32 vs fp64")
    assert_close(outs["persistent"][2].cpu(), (z + res.cpu().double()).float(), 1e-3, f"bias table T={T}, fp32 + residual vs fp64")
