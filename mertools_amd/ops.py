"""Torch-tensor front ends for the op-level C ABI (include/mer_hip.h).

torch is used for device memory and the current stream only; every computation below is one
HIP kernel of libmer_hip.so.  These wrappers exist for the parity tests and for composing the
encoders; they validate devices/dtypes and otherwise pass raw pointers through.
"""
import torch

from . import _lib
from ._lib import GemmArgs, MER_DT_F16, MER_DT_BF16, MER_ACT_NONE, MER_ACT_GELU, MER_ACT_QUICK_GELU, MER_ACT_RELU, MER_ACT_GELU_TANH  # noqa: F401

ACT = {None: MER_ACT_NONE, "none": MER_ACT_NONE, "gelu": MER_ACT_GELU, "quick_gelu": MER_ACT_QUICK_GELU, "relu": MER_ACT_RELU,
       "gelu_new": MER_ACT_GELU_TANH, "gelu_tanh": MER_ACT_GELU_TANH}
_TORCH16 = {MER_DT_F16: torch.float16, MER_DT_BF16: torch.bfloat16}
_DT = {"f16": MER_DT_F16, "bf16": MER_DT_BF16, MER_DT_F16: MER_DT_F16, MER_DT_BF16: MER_DT_BF16,
       torch.float16: MER_DT_F16, torch.bfloat16: MER_DT_BF16}


def dt_code(dtype):
    return _DT[dtype]


def torch16(dtype):
    return _TORCH16[dt_code(dtype)]


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MerError("mertools_amd ops need CUDA/HIP device tensors (there is no CPU path)")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def split16(x, dtype="f16", lo=True):
    """fp32 tensor -> (hi, lo) 16-bit planes computed on the GPU."""
    x = x.contiguous()
    t16 = torch16(dtype)
    hi = torch.empty(x.shape, dtype=t16, device=x.device)
    lo_t = torch.empty(x.shape, dtype=t16, device=x.device) if lo else None
    _lib.check(_lib.lib().mer_split16(_p(x), _p(hi), _p(lo_t), x.numel(), dt_code(dtype), stream()), "mer_split16")
    return hi, lo_t


def split16_host(x, dtype="f16", lo=True):
    """Same split on a CPU tensor (weight preparation at load time)."""
    t16 = torch16(dtype)
    x = x.detach().to(torch.float32)
    hi = x.to(t16)
    lo_t = (x - hi.to(torch.float32)).to(t16) if lo else None
    return hi, lo_t


def gemm16(a_hi, w_hi, *, a_lo=None, w_lo=None, bias=None, act=None, residual=None, out32=False, out16=False,
           out16_lo=False, passes=1, dtype=None, tile=0, M=None, lda=None, a_rows_per_batch=0, a_batch_stride=0,
           headmajor=None, w_mx=None, w_hi_blk=None, w_lo_blk=None, w_hi_blkp=None, w_hi_blkq=None, bias_seg_rows=0):
    """C = epilogue(A @ W^T) — plain (non-batched) form used by the tests. A [M,K], W [N,K]."""
    dtype = dt_code(dtype if dtype is not None else a_hi.dtype)
    N, K = w_hi.shape
    if M is None:
        M = a_hi.shape[0]
    g = GemmArgs()
    g.M, g.N, g.K, g.dtype = M, N, K, dtype
    g.a_hi, g.a_lo = _p(a_hi), _p(a_lo)
    g.lda = lda if lda is not None else a_hi.stride(0)
    g.a_rows_per_batch, g.a_batch_stride = a_rows_per_batch, a_batch_stride
    g.w_hi, g.w_lo, g.ldw = _p(w_hi), _p(w_lo), w_hi.stride(0)
    g.w_mx = _p(w_mx)
    g.w_hi_blk, g.w_lo_blk, g.w_hi_blkp, g.w_hi_blkq = _p(w_hi_blk), _p(w_lo_blk), _p(w_hi_blkp), _p(w_hi_blkq)
    g.bias, g.act = _p(bias), ACT[act]
    if bias_seg_rows:   # `bias` is a per-sequence table [ceil(M / bias_seg_rows), N] (mer_seq_bias)
        g.bias_seg_rows, g.bias_ld = int(bias_seg_rows), bias.stride(0)
    g.residual, g.ldr = _p(residual), (residual.stride(0) if residual is not None else 0)
    dev = a_hi.device
    c32 = torch.empty((M, N), dtype=torch.float32, device=dev) if out32 else None
    c16h = torch.empty((M, N), dtype=torch16(dtype), device=dev) if out16 else None
    c16l = torch.empty((M, N), dtype=torch16(dtype), device=dev) if (out16 and out16_lo) else None
    g.c32, g.ldc32 = _p(c32), N
    g.c16_hi, g.c16_lo, g.ldc16 = _p(c16h), _p(c16l), N
    g.nbatch, g.nb_inner, g.passes, g.tile = 1, 1, passes, tile
    if headmajor is not None:
        g.headmajor_T, g.headmajor_H = headmajor
    _lib.check(_lib.lib().mer_gemm16(g, stream()), "mer_gemm16")
    return c32, c16h, c16l


def mx_pack(w_res):
    """Packs W - f16(W) ([N, K] fp32, host) into the MX-fp4 correction plane of gemm16(passes=4) (mer_mx_pack, host C++).
    Returns a uint8 CPU tensor, or None when the shape has no MX form (K % 128 != 0)."""
    w_res = w_res.detach().to("cpu", torch.float32).contiguous()
    N, K = w_res.shape
    nbytes = _lib.lib().mer_mx_packed_bytes(N, K)
    if nbytes <= 0:
        return None
    out = torch.empty(nbytes, dtype=torch.uint8)
    _lib.check(_lib.lib().mer_mx_pack(w_res.data_ptr(), w_res.stride(0), N, K, out.data_ptr()), "mer_mx_pack")
    return out


def w_block_pack(w16):
    """Device 16-bit plane [N, K] -> its pre-blocked copy for the 256-wide LDS-DMA GEMM kernels (mer_w_block_pack): per
    (256-row tile, 32-deep k-slab) the 16 KB LDS image.  Returns a device uint8 tensor, or None when K % 32 != 0."""
    assert w16.is_cuda and w16.dim() == 2 and w16.element_size() == 2 and w16.stride(1) == 1
    N, K = w16.shape
    nbytes = _lib.lib().mer_w_block_bytes(N, K)
    if nbytes <= 0 or w16.stride(0) % 8 != 0:
        return None
    out = torch.empty(nbytes, dtype=torch.uint8, device=w16.device)
    _lib.check(_lib.lib().mer_w_block_pack(w16.data_ptr(), w16.stride(0), N, K, out.data_ptr(), stream()), "mer_w_block_pack")
    return out


def w_block_pack_p(w16, layout):
    """Device 16-bit plane [N, K] -> a row-permuted pre-blocked copy for the PERSISTENT 256x256 one-pass kernel
    (mer_w_block_pack_p): layout 0 for GEMMs whose output is one 16-bit plane (a lane's eight accumulators of an output row become
    eight consecutive columns: one 16-byte store), layout 1 for fp32 outputs (two runs of four columns: two 16-byte stores / residual
    loads) — the epilogue stores whole lines from registers.  None when N % 256 or K % 32 != 0."""
    assert w16.is_cuda and w16.dim() == 2 and w16.element_size() == 2 and w16.stride(1) == 1
    N, K = w16.shape
    if N % 256 != 0 or K % 32 != 0 or w16.stride(0) % 8 != 0:
        return None
    out = torch.empty(N * K * 2, dtype=torch.uint8, device=w16.device)
    _lib.check(_lib.lib().mer_w_block_pack_p(w16.data_ptr(), w16.stride(0), N, K, int(layout), out.data_ptr(), stream()), "mer_w_block_pack_p")
    return out


def seq_bias(a16, w_lo, seg_rows, bias=None, valid_rows=None, M=None, n_first=0):
    """table[s, n] = bias[n] + mean_{sampled rows of sequence s}(a16)[k] * w_lo[n, k] (mer_seq_bias): the per-sequence weight-residual
    correction of a one-pass GEMM; pass the result as gemm16(bias=table, bias_seg_rows=seg_rows)."""
    assert a16.is_cuda and a16.dim() == 2 and a16.stride(1) == 1 and w_lo.dim() == 2 and w_lo.stride(1) == 1
    M = M if M is not None else a16.shape[0]
    K, N = a16.shape[1], w_lo.shape[0]
    nseq = (M + seg_rows - 1) // seg_rows
    scratch = torch.empty(_lib.lib().mer_seq_bias_scratch_bytes(nseq, K), dtype=torch.uint8, device=a16.device)
    out = torch.empty((nseq, N), dtype=torch.float32, device=a16.device)
    _lib.check(_lib.lib().mer_seq_bias(a16.data_ptr(), dt_code(a16.dtype), a16.stride(0), 0, 0, M, K, int(seg_rows), _p(valid_rows),
                                       w_lo.data_ptr(), w_lo.stride(0), _p(bias), N, int(n_first), scratch.data_ptr(), out.data_ptr(), N, stream()), "mer_seq_bias")
    return out


def gemm16_raw(args: GemmArgs):
    _lib.check(_lib.lib().mer_gemm16(args, stream()), "mer_gemm16")


def layernorm(x, gamma, beta, eps, *, act=None, out32=True, out16=False, out16_lo=False, dtype="f16", M=None, D=None, ldx=None):
    D = D if D is not None else x.shape[-1]
    M = M if M is not None else x.numel() // D
    ldx = ldx if ldx is not None else D
    dev = x.device
    o32 = torch.empty((M, D), dtype=torch.float32, device=dev) if out32 else None
    oh = torch.empty((M, D), dtype=torch16(dtype), device=dev) if out16 else None
    ol = torch.empty((M, D), dtype=torch16(dtype), device=dev) if (out16 and out16_lo) else None
    _lib.check(_lib.lib().mer_layernorm(_p(x), ldx, _p(gamma), _p(beta), eps, M, D, ACT[act], _p(o32), D, _p(oh), _p(ol), D,
                                        dt_code(dtype), stream()), "mer_layernorm")
    return o32, oh, ol


def attention_f32(qkv32, B, T, H, scale, *, kv_len=None, out_lo=True, dtype="f16"):
    """qkv32: fp32 [B*T, 3*H*64] (q | k | v column blocks) -> ctx hi (+ lo) 16-bit planes [B*T, H*64] on the exact fp32 MFMA
    (mer_attention_f32: the "accurate" preset's attention)."""
    D = H * 64
    assert qkv32.dtype == torch.float32 and qkv32.shape == (B * T, 3 * D) and qkv32.is_contiguous()
    oh = torch.empty((B * T, D), dtype=torch16(dtype), device=qkv32.device)
    ol = torch.empty_like(oh) if out_lo else None
    _lib.check(_lib.lib().mer_attention_f32(qkv32.data_ptr(), qkv32.data_ptr() + D * 4, qkv32.data_ptr() + 2 * D * 4, 3 * D,
                                            _p(oh), _p(ol), D, B, T, H, float(scale), _p(kv_len), dt_code(dtype), stream()), "mer_attention_f32")
    return oh, ol


def attention(qkv, B, T, H, scale, *, kv_len=None, out_lo=False):
    """qkv: 16-bit [B*T, 3*H*64] (q | k | v column blocks) -> ctx 16-bit [B*T, H*64]."""
    D = H * 64
    assert qkv.shape == (B * T, 3 * D) and qkv.is_contiguous()
    oh = torch.empty((B * T, D), dtype=qkv.dtype, device=qkv.device)
    ol = torch.empty_like(oh) if out_lo else None
    es = qkv.element_size()
    _lib.check(_lib.lib().mer_attention(qkv.data_ptr(), qkv.data_ptr() + D * es, qkv.data_ptr() + 2 * D * es, 3 * D,
                                        _p(oh), _p(ol), D, B, T, H, float(scale), _p(kv_len), dt_code(qkv.dtype), stream()),
               "mer_attention")
    return oh, ol


def attention_cls(q, qkv, B, T, H, scale, *, kv_len=None):
    """One query per sequence: q 16-bit [B, H*64], keys / values from qkv [B*T, 3*H*64] -> ctx 16-bit [B, H*64] (mer_attention_cls)."""
    D = H * 64
    assert qkv.shape == (B * T, 3 * D) and qkv.is_contiguous() and q.shape == (B, D) and q.is_contiguous()
    oh = torch.empty((B, D), dtype=qkv.dtype, device=qkv.device)
    es = qkv.element_size()
    _lib.check(_lib.lib().mer_attention_cls(q.data_ptr(), D, qkv.data_ptr() + D * es, qkv.data_ptr() + 2 * D * es, 3 * D, _p(oh), None, D,
                                            B, T, H, float(scale), _p(kv_len), dt_code(qkv.dtype), stream()), "mer_attention_cls")
    return oh


def wave_normalize(x, do_normalize=True):
    """Device int16 PCM or fp32 [B, L] -> fp32 [B, L], zero-mean / unit-variance per row (mer_wave_normalize)."""
    assert x.is_cuda and x.dim() == 2 and x.dtype in (torch.int16, torch.float32) and x.stride(1) == 1
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mer_wave_normalize(x.data_ptr(), int(x.dtype == torch.int16), x.stride(0), x.shape[0], x.shape[1],
                                             int(do_normalize), out.data_ptr(), out.stride(0), stream()), "mer_wave_normalize")
    return out


def image_normalize_u8(frames, mean, std, bgr=True):
    """Device uint8 [N, H, W, 3] -> fp32 [N, 3, H, W] RGB, (v / 255 - mean) / std (mer_image_normalize_u8)."""
    import ctypes
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
    N, H, W, _ = frames.shape
    out = torch.empty((N, 3, H, W), dtype=torch.float32, device=frames.device)
    m, s_ = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.check(_lib.lib().mer_image_normalize_u8(frames.data_ptr(), N, H, W, int(bgr), m, s_, out.data_ptr(), stream()), "mer_image_normalize_u8")
    return out


def small_attention(q, k, v, B, Tq, Tk, H, scale, causal=False):
    """fp32 attention for a handful of queries: q [B*Tq, H*64], k / v [B*Tk, H*64] (same row stride) -> [B*Tq, H*64]."""
    assert q.dtype == k.dtype == v.dtype == torch.float32 and k.stride(0) == v.stride(0)
    out = torch.empty((B * Tq, H * 64), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().mer_small_attention(_p(q), q.stride(0), _p(k), _p(v), k.stride(0), B, Tq, Tk, H, float(scale), int(causal),
                                              _p(out), out.stride(0), stream()), "mer_small_attention")
    return out


def add_pos(x, pos):
    """x[r, :] += pos[r % P, :] in place (x fp32 [rows, D], pos fp32 [P, D])."""
    _lib.check(_lib.lib().mer_add_pos(_p(x), _p(pos), x.shape[0], pos.shape[0], x.shape[1], stream()), "mer_add_pos")
    return x


def attention_bias(qkv, B, T, H, scale, bias, *, gate=None, kv_len=None):
    """attention() with scores += gate[b,h,q] * bias[h,q,k].  bias: fp32 [H, T, ldb] (ldb % 4 == 0), gate: fp32 [B,H,T] or None."""
    D = H * 64
    assert qkv.shape == (B * T, 3 * D) and qkv.is_contiguous() and bias.is_contiguous() and bias.shape[:2] == (H, T)
    oh = torch.empty((B * T, D), dtype=qkv.dtype, device=qkv.device)
    es = qkv.element_size()
    _lib.check(_lib.lib().mer_attention_bias(qkv.data_ptr(), qkv.data_ptr() + D * es, qkv.data_ptr() + 2 * D * es, 3 * D,
                                             _p(oh), None, D, B, T, H, float(scale), _p(kv_len), _p(bias), bias.shape[2], _p(gate),
                                             dt_code(qkv.dtype), stream()), "mer_attention_bias")
    return oh


def wavlm_gate(x, w, b, const, B, T, H):
    """gate [B,H,T] from the attention input x fp32 [B*T, H*64] (mer_wavlm_gate)."""
    gate = torch.empty((B, H, T), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mer_wavlm_gate(_p(x), x.stride(0), _p(w), _p(b), _p(const), B, T, H, _p(gate), stream()), "mer_wavlm_gate")
    return gate


def attention_hm(qkv_hm, B, T, H, scale, *, kv_len=None, out_lo=False):
    """qkv_hm: 16-bit [3, B, H, T, 64] (head-major, as written by gemm16(headmajor=(T, H))) -> ctx [B*T, H*64]."""
    D = H * 64
    assert qkv_hm.numel() == 3 * B * T * D and qkv_hm.is_contiguous()
    oh = torch.empty((B * T, D), dtype=qkv_hm.dtype, device=qkv_hm.device)
    ol = torch.empty_like(oh) if out_lo else None
    plane = B * T * D * qkv_hm.element_size()
    _lib.check(_lib.lib().mer_attention_hm(qkv_hm.data_ptr(), qkv_hm.data_ptr() + plane, qkv_hm.data_ptr() + 2 * plane, _p(oh), _p(ol),
                                           D, B, T, H, float(scale), _p(kv_len), dt_code(qkv_hm.dtype), stream()), "mer_attention_hm")
    return oh, ol


def hubert_conv0_gn(wav, w, gamma, beta, eps=1e-5, *, stride=5, dtype="f16", lo=False, valid_frames=None):
    """valid_frames: int32 [B] — GroupNorm statistics over each row's first valid_frames[b] frames only (ragged batch)."""
    B, L = wav.shape
    Cc, k = w.shape
    T0 = (L - k) // stride + 1
    stats = torch.empty((B, Cc, 2), dtype=torch.float64, device=wav.device)
    oh = torch.empty((B, T0, Cc), dtype=torch16(dtype), device=wav.device)
    ol = torch.empty_like(oh) if lo else None
    _lib.check(_lib.lib().mer_hubert_conv0_gn_ragged(_p(wav), B, L, _p(w), Cc, k, stride, _p(gamma), _p(beta), eps, _p(stats),
                                                     _p(oh), _p(ol), dt_code(dtype), _p(valid_frames), stream()),
               "mer_hubert_conv0_gn")
    return oh, ol


def posconv_pack(x, G, K, *, dtype="f16", lo=False):
    B, T, D = x.shape
    oh = torch.empty((B, G, T + K, D // G), dtype=torch16(dtype), device=x.device)
    ol = torch.empty_like(oh) if lo else None
    _lib.check(_lib.lib().mer_posconv_pack(_p(x), B, T, D, G, K, _p(oh), _p(ol), dt_code(dtype), stream()), "mer_posconv_pack")
    return oh, ol


def vit_patchify(px, P, *, dtype="f16", lo=False):
    N, Cc, H, W = px.shape
    oh = torch.empty((N * (H // P) * (W // P), (Cc * P * P + 7) // 8 * 8), dtype=torch16(dtype), device=px.device)
    ol = torch.empty_like(oh) if lo else None
    _lib.check(_lib.lib().mer_vit_patchify(_p(px), N, Cc, H, W, P, _p(oh), _p(ol), dt_code(dtype), stream()), "mer_vit_patchify")
    return oh, ol


def vit_assemble(patch, cls, pos, gamma, beta, eps, N, P, D):
    out = torch.empty((N * (P + 1), D), dtype=torch.float32, device=patch.device)
    _lib.check(_lib.lib().mer_vit_assemble(_p(patch), _p(cls), _p(pos), _p(gamma), _p(beta), eps, N, P, D, _p(out), None, None,
                                           MER_DT_F16, stream()), "mer_vit_assemble")
    return out


def bert_embed(ids, token_type, word, pos, type_emb, pos_mode, pad_id, gamma, beta, eps, *, dtype="f16"):
    B, T = ids.shape
    D = word.shape[1]
    o32 = torch.empty((B * T, D), dtype=torch.float32, device=ids.device)
    oh = torch.empty((B * T, D), dtype=torch16(dtype), device=ids.device)
    _lib.check(_lib.lib().mer_bert_embed(_p(ids), _p(token_type), B, T, D, _p(word), _p(pos), _p(type_emb), pos_mode, pad_id,
                                         _p(gamma), _p(beta), eps, _p(o32), _p(oh), None, dt_code(dtype), stream()),
               "mer_bert_embed")
    return o32, oh


def sum_pool(hs, seg_start=None, seg_len=None, *, frames=False):
    """hs: list of 1..4 fp32 [M,D] tensors. Returns (frames [M,D] or None, pooled [nseg,D] or None)."""
    M, D = hs[0].shape
    h = list(hs) + [None] * (4 - len(hs))
    fr = torch.empty((M, D), dtype=torch.float32, device=hs[0].device) if frames else None
    pooled = None
    nseg = 0
    if seg_start is not None:
        nseg = seg_start.numel()
        pooled = torch.empty((nseg, D), dtype=torch.float32, device=hs[0].device)
    _lib.check(_lib.lib().mer_sum_pool(_p(h[0]), _p(h[1]), _p(h[2]), _p(h[3]), M, D, _p(fr), _p(seg_start), _p(seg_len), nseg,
                                       _p(pooled), stream()), "mer_sum_pool")
    return fr, pooled


def gemm32(a, w, bias=None, act=None, *, trans_a=False, trans_w=False, out=None, accumulate=False):
    """Exact-fp32 C = act(A @ W^T + bias).  a: [M,K] (or [K,M] if trans_a); w: [N,K] (or [K,N] if trans_w)."""
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = w.shape[1] if trans_w else w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().mer_gemm32(_p(a), a.stride(0), int(trans_a), _p(w), w.stride(0), int(trans_w), _p(bias), ACT[act],
                                     _p(out), out.stride(0), int(accumulate), M, N, K, stream()), "mer_gemm32")
    return out
