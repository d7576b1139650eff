// encoders.cpp — host-side sequencing of the three encoder forwards on top of the op-level ABI.
// No device code here: every step is one of the mer_* ops (HIP kernels) launched on the caller's
// stream, with all scratch carved from the caller-provided workspace by a bump allocator.  The
// same planning code runs in "dry" mode to answer mer_*_workspace_bytes().
//
// Reference call sites this replaces:
//   audio : MERBench/feature_extraction/audio/extract_audio_huggingface.py:97-108
//   visual: MERBench/feature_extraction/visual/extract_vision_huggingface.py:118-122,183-189
//   text  : MERBench/feature_extraction/text/extract_text_huggingface.py:225-249
#include "common.h"
#include <vector>
#include <string.h>
#include <math.h>

namespace mer {

struct Arena {
  char* base;
  long long cap, off;
  bool dry;
  Arena(void* b, long long c, bool d) : base((char*)b), cap(c), off(0), dry(d) {}
  void* take(long long bytes) {
    off = (off + 255) & ~255LL;
    void* p = dry ? nullptr : (void*)(base + off);
    off += bytes;
    return p;
  }
  bool ok() const { return dry || off <= cap; }
};

struct P16 { void* hi; void* lo; };  // 16-bit activation planes

static P16 take16(Arena& a, long long elems, bool need_lo) {
  P16 p;
  p.hi = a.take(elems * 2);
  p.lo = need_lo ? a.take(elems * 2) : nullptr;
  return p;
}

// hidden-state addressing: either the caller's [layers+1, M, D] tensor or a small ring.
struct HsMap {
  float* base;
  long long stride;
  int ring;  // 0 = no ring (full tensor)
  float* at(int l) const { return base + (long long)(ring ? (l % ring) : l) * stride; }
};

// passes == 5: ONE f16 MFMA pass + the weight-rounding residual applied through each SEQUENCE's mean activation:
//   tab[s, n] = bias[n] + mean_{sampled rows of sequence s}(A)[k] * (W - f16(W))[n, k]     (mer_seq_bias: two small launches)
//   C[m, :] = epi(A[m, :] * f16(W)^T + tab[m / T, :])
// A clip's features depend on that clip alone (round 3's batch-mean bias made them depend on the batch mates).  `cw` carries the
// scratch and the sequence structure of the A plane; without it (or without a `lo` plane) 5 degrades to 4 / 2.
struct CorrArena {       // one forward's mer_seq_bias scratch, reused by every corrected GEMM (same stream: launches are ordered)
  char* mean16;          // [nseq, kmax] 16-bit mean plane; NULL = not available
  float* table;          // [nseq, nmax] fp32: the correction rows of the GEMM that is about to run
  int nseq, kmax;
  long long nmax;
};
struct CorrWs {
  CorrArena* arena;
  int seg_rows;          // rows per sequence of the A plane (1: every row is its own sequence; 0: no sequence structure -> no table) ...
  const int* valid;      // ... ragged batches: device int32 [nseq] valid rows of each (NULL: every row counts)
  int n_first;           // output columns below this take the plain bias (fused QKV: the Q | K columns)
};
static void corr_plan(Arena& ar, CorrArena& ca, bool on, int kmax, long long nmax, int nseq) {
  ca.nseq = nseq; ca.kmax = kmax; ca.nmax = (nmax + 3) / 4 * 4;
  ca.mean16 = on ? (char*)ar.take(mer_seq_bias_scratch_bytes(nseq, kmax)) : nullptr;
  ca.table = on ? (float*)ar.take((long long)nseq * ca.nmax * 4) : nullptr;
}

// mer_gemm16 with the passes == 5 preamble; `g` is complete except for the pass code handling
// passes == 6: the same table, but the GEMM runs TWO passes, a_hi*w_hi + a_lo*w_hi: every input is carried as hi + lo 16-bit planes
// (activations that do not fit 11 bits of mantissa: outlier channels riding a post-LN residual stream), the weight residual still
// goes through the sequence's mean token.  Without the table scratch it runs all three passes.
static int run_gemm(hipStream_t st, mer_gemm16_args g, const CorrWs* cw) {
  if (g.passes == 5 || g.passes == 6) {
    const bool a2 = g.passes == 6 && g.a_lo;
    CorrArena* ca = cw ? cw->arena : nullptr;
    const int nseq = (cw && cw->seg_rows > 0) ? (int)((g.M + cw->seg_rows - 1) / cw->seg_rows) : 0;
    const bool ok = ca && ca->mean16 && nseq > 0 && nseq <= ca->nseq && g.K <= ca->kmax && g.N <= ca->nmax && g.w_lo &&
                    g.N % 8 == 0 && g.K % 8 == 0 && g.nbatch <= 1;
    if (!ok) {
      if (a2) g.passes = g.w_lo ? 3 : 6;
      else g.passes = (g.w_mx || g.w_lo) ? 4 : 1;   // mer_gemm16 turns 4 into the 2-pass path where the MX kernel does not apply
    } else {
      int rc = mer_seq_bias(g.a_hi, g.dtype, g.lda, g.a_rows_per_batch, g.a_batch_stride, g.M, g.K, cw->seg_rows, cw->valid,
                            g.w_lo, g.ldw, g.bias, g.N, cw->n_first, ca->mean16, ca->table, g.N, (mer_stream_t)st);
      if (rc != MER_OK) return rc;
      g.passes = a2 ? 6 : 1;
      g.w_lo = nullptr; g.w_mx = nullptr; g.w_lo_blk = nullptr;
      g.bias = ca->table;
      g.bias_seg_rows = cw->seg_rows;
      g.bias_ld = g.N;
    }
  }
  return mer_gemm16(&g, (mer_stream_t)st);
}

static int gemm(hipStream_t st, int dtype, int passes, int M, int N, int K, P16 a, long long lda, const mer_w16& w,
                const float* bias, int act, const float* residual, long long ldr, float* c32, long long ldc32, P16 c16,
                long long ldc16, const CorrWs* cw = nullptr) {
  mer_gemm16_args g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K; g.dtype = dtype;
  g.a_hi = a.hi; g.a_lo = a.lo; g.lda = lda;
  g.w_hi = w.hi; g.w_lo = w.lo; g.w_mx = w.mx; g.ldw = K;
  g.w_hi_blk = w.hi_blk; g.w_lo_blk = w.lo_blk; g.w_hi_blkp = w.hi_blkp; g.w_hi_blkq = w.hi_blkq;
  g.bias = bias; g.act = act; g.residual = residual; g.ldr = ldr;
  g.c32 = c32; g.ldc32 = ldc32; g.c16_hi = c16.hi; g.c16_lo = c16.lo; g.ldc16 = ldc16;
  g.nbatch = 1; g.nb_inner = 1; g.passes = passes; g.tile = 0;
  return run_gemm(st, g, cw);
}

#define MER_TRY(expr)        \
  do {                       \
    int _rc = (expr);        \
    if (_rc != MER_OK) return _rc; \
  } while (0)

struct TfBufs {
  float* t32;
  float* h1_32;
  P16 cur16, qkv16, ctx16, h1_16, f16;
  float* qkv32;    // passes == 3: fp32 q | k | v for mer_attention_f32 (no operand of an "accurate" block is a single 16-bit plane)
  float* ffn32;    // SwiGLU: fp32 [M, 2F] output of weights_in awaiting the gate
  float* gate;     // WavLM: [B, H, T] gate of the current layer
  float* gin32;    // WavLM pre-LN: fp32 copy of the normalised attention input (the gate is computed from it)
  CorrArena corr;  // passes == 5: mer_seq_bias scratch (the encoder's other GEMMs share it)
};

// ViT whose caller only reads the [CLS] row of the last block (CLIP get_image_features): that block computes K and V for every
// token but Q, attention, output projection and the FFN for the CLS rows alone (nseq rows instead of nseq * T).
struct ClsBufs {
  P16 q16, ctx16, h16, f16;
  float* t32;      // CLS residual stream after the attention half
  float* y32;      // ... after the FFN: what post_layernorm reads
};
static void cls_plan(Arena& ar, const mer_tf_config& c, long long nseq, ClsBufs& b) {
  const bool lo = c.passes == 3 || c.passes == 6;
  const long long D = c.hidden, F = c.ffn;
  b.q16 = take16(ar, nseq * D, false);
  b.ctx16 = take16(ar, nseq * D, lo);
  b.h16 = take16(ar, nseq * D, lo);
  b.f16 = take16(ar, nseq * F, lo);
  b.t32 = (float*)ar.take(nseq * D * 4);
  b.y32 = (float*)ar.take(nseq * D * 4);
}

// extra_sites / extra_k / extra_n: corrected GEMMs outside the blocks that share the arena (conv stack, patch embedding)
static void tf_plan(Arena& ar, const mer_tf_config& c, long long M, int nseq, TfBufs& b, bool extra_on = false, int extra_k = 0, long long extra_n = 0) {
  const bool lo = c.passes == 3 || c.passes == 6;
  const long long D = c.hidden, F = c.ffn;
  b.t32 = (float*)ar.take(M * D * 4);
  b.h1_32 = c.pre_ln ? nullptr : (float*)ar.take(M * D * 4);
  b.cur16 = take16(ar, M * D, lo);
  b.qkv16 = take16(ar, M * 3 * D, false);
  b.qkv32 = (c.passes == 3 || (c.passes == 6 && c.attn_f32)) ? (float*)ar.take(M * 3 * D * 4) : nullptr;
  b.ctx16 = take16(ar, M * D, lo);
  b.h1_16 = take16(ar, M * D, lo);
  b.f16 = take16(ar, M * F, lo);
  b.ffn32 = c.ffn_swiglu ? (float*)ar.take(M * 2 * F * 4) : nullptr;
  b.gate = c.gated_rel_pos ? (float*)ar.take(M * c.heads * 4) : nullptr;
  b.gin32 = (c.gated_rel_pos && c.pre_ln) ? (float*)ar.take(M * D * 4) : nullptr;
  long long wide = 2 * F > 3 * D ? 2 * F : 3 * D;
  wide = extra_n > wide ? extra_n : wide;
  int kmax = (int)(F > D ? F : D);
  kmax = extra_k > kmax ? extra_k : kmax;
  corr_plan(ar, b.corr, c.passes == 5 || c.passes == 6 || extra_on, kmax, wide, nseq);
}

// Runs c.layers transformer blocks.  Post-LN: hs.at(0) and b.cur16 hold the (already normalised)
// input; pre-LN: hs.at(0) holds the raw residual stream.  Writes hs.at(l+1) for every layer.
static int tf_forward(hipStream_t st, const mer_tf_config& c, const mer_tf_layer* L, int Bseq, int T, const HsMap& hs,
                      TfBufs& b, const int* kv_len, const float* pos_bias = nullptr, long long ldb = 0, const ClsBufs* cls = nullptr) {
  const int M = Bseq * T, D = c.hidden, F = c.ffn, H = c.heads;
  const int dt = c.dtype, ps = c.passes;
  const bool sel = ps == 4 || ps == 5;                      // presets with a selective weight-residual correction (mx_skip)
  const int ps1 = (sel && (c.mx_skip & 2)) ? 1 : ps;       // fc1 without the correction
  const int ps2 = (sel && (c.mx_skip & 4)) ? 1 : ps;       // fc2 without the correction
  // passes == 5: the correction goes through the batch's mean token (run_gemm() above); a sequence = T rows, kv_len = its valid rows
  const CorrWs cwv = {&b.corr, T, kv_len, 0};
  const CorrWs cwqkv = {&b.corr, T, kv_len, (c.mx_skip & 1) && (2 * D) % 16 == 0 ? 2 * D : 0};   // Q | K uncorrected (mx_skip bit 0)
  const CorrWs* mc = ((ps == 5 || ps == 6) && b.corr.mean16) ? &cwv : nullptr;
  const CorrWs cw1 = {&b.corr, 1, nullptr, 0};                  // CLS-only block: one row per sequence (its "mean" is the row itself)
  const CorrWs* mc1 = mc ? &cw1 : nullptr;
  const CorrWs* mcq = mc ? &cwqkv : nullptr;
  const CorrWs cwkv = {&b.corr, T, kv_len, (c.mx_skip & 1) && D % 16 == 0 ? D : 0};          // CLS-only block's K | V GEMM: K uncorrected
  const CorrWs* mckv = mc ? &cwkv : nullptr;
  const float scale = 1.0f / sqrtf((float)(D / H));
  const P16 none = {nullptr, nullptr};
  for (int l = 0; l < c.layers; ++l) {
    const mer_tf_layer& w = L[l];
    float* x = hs.at(l);
    float* y = hs.at(l + 1);
    if (c.pre_ln)
      MER_TRY(mer_layernorm(x, D, w.ln1_g, w.ln1_b, c.ln_eps, M, D, MER_ACT_NONE, b.gin32, D, b.cur16.hi, b.cur16.lo, D, dt, st));
    if (cls && l == c.layers - 1) {
      // last block, CLS rows only (the caller checked: pre-LN, plain FFN, no score bias).  K | V for every token ...
      const long long woff = (long long)D * D * 2;   // bytes into the 16-bit planes: row D of the fused [3D, D] weight
      const bool tiles = D % 256 == 0;               // the pre-blocked / MX planes are stored per 256-row tile
      const mer_w16 wkv = {(const char*)w.wqkv.hi + woff, w.wqkv.lo ? (const char*)w.wqkv.lo + woff : nullptr,
                           (w.wqkv.mx && tiles) ? (const char*)w.wqkv.mx + (long long)(D / 256) * (D / 32) * 5120 : nullptr,
                           (w.wqkv.hi_blk && tiles) ? (const char*)w.wqkv.hi_blk + woff : nullptr,
                           (w.wqkv.lo_blk && tiles) ? (const char*)w.wqkv.lo_blk + woff : nullptr,
                           (w.wqkv.hi_blkp && tiles) ? (const char*)w.wqkv.hi_blkp + woff : nullptr, nullptr};
      const P16 ckv = {(char*)b.qkv16.hi + (long long)D * 2, nullptr};
      MER_TRY(gemm(st, dt, ps, M, 2 * D, D, b.cur16, D, wkv, w.bqkv + D, MER_ACT_NONE, nullptr, 0, nullptr, 0, ckv, 3 * D, mckv));
      // ... Q for the CLS rows (row n of the A operand = token 0 of sequence n: lda = T * D)
      const mer_w16 wq = {w.wqkv.hi, w.wqkv.lo, nullptr, nullptr, nullptr, nullptr, nullptr};
      MER_TRY(gemm(st, dt, ps, Bseq, D, D, b.cur16, (long long)T * D, wq, w.bqkv, MER_ACT_NONE, nullptr, 0, nullptr, 0, cls->q16, D, mc1));
      MER_TRY(mer_attention_cls(cls->q16.hi, D, (char*)b.qkv16.hi + (long long)D * 2, (char*)b.qkv16.hi + (long long)2 * D * 2, 3 * D,
                                cls->ctx16.hi, cls->ctx16.lo, D, Bseq, T, H, scale, kv_len, dt, (mer_stream_t)st));
      MER_TRY(gemm(st, dt, ps, Bseq, D, D, cls->ctx16, D, w.wo, w.bo, MER_ACT_NONE, x, (long long)T * D, cls->t32, D, none, 0, mc1));
      MER_TRY(mer_layernorm(cls->t32, D, w.ln2_g, w.ln2_b, c.ln_eps, Bseq, D, MER_ACT_NONE, nullptr, 0, cls->h16.hi, cls->h16.lo, D, dt, st));
      MER_TRY(gemm(st, dt, ps1, Bseq, F, D, cls->h16, D, w.w1, w.b1, c.act, nullptr, 0, nullptr, 0, cls->f16, F, mc1));
      MER_TRY(gemm(st, dt, ps2, Bseq, D, F, cls->f16, F, w.w2, w.b2, MER_ACT_NONE, cls->t32, D, cls->y32, D, none, 0, mc1));
      continue;
    }
    const float* ab = w.attn_bias ? w.attn_bias : pos_bias;
    // three passes: q | k | v stay fp32 and attention runs on the exact fp32 MFMA (not with a score bias: WavLM / BEiT keep the f16 kernel)
    const bool f32attn = (ps == 3 || (ps == 6 && c.attn_f32)) && b.qkv32 != nullptr && !ab;
    // (a head-major QKV layout — mer_gemm16's headmajor_* output + mer_attention_hm — was measured: attention gains
    //  nothing from the contiguous K/V streams while the scatter epilogue costs the QKV GEMM ~4 %, so row-major stays)
    if (ps == 4 && (c.mx_skip & 1) && (2 * D) % 256 == 0 && w.wqkv.mx != nullptr) {
      // Q | K columns: one f16 pass (weight rounding there only perturbs softmax logits: no measurable effect on the features);
      // V columns: MX-corrected.  The MX plane is stored per 256-column tile, so the V block starts at tile 2D/256.
      const mer_w16 wqk = {w.wqkv.hi, nullptr, nullptr, w.wqkv.hi_blk, nullptr, w.wqkv.hi_blkp, nullptr};
      MER_TRY(gemm(st, dt, 1, M, 2 * D, D, b.cur16, D, wqk, w.bqkv, MER_ACT_NONE, nullptr, 0, nullptr, 0, b.qkv16, 3 * D));
      const long long woff = (long long)2 * D * D * 2;   // bytes into the 16-bit planes
      const mer_w16 wv = {(const char*)w.wqkv.hi + woff, w.wqkv.lo ? (const char*)w.wqkv.lo + woff : nullptr,
                          w.wqkv.mx ? (const char*)w.wqkv.mx + (long long)(2 * D / 256) * (D / 32) * 5120 : nullptr,
                          // the pre-blocked planes are stored per 256-row tile as well: tile 2D/256 starts woff bytes in
                          w.wqkv.hi_blk ? (const char*)w.wqkv.hi_blk + woff : nullptr,
                          w.wqkv.lo_blk ? (const char*)w.wqkv.lo_blk + woff : nullptr,
                          w.wqkv.hi_blkp ? (const char*)w.wqkv.hi_blkp + woff : nullptr, nullptr};
      const P16 cv = {(char*)b.qkv16.hi + (long long)2 * D * 2, nullptr};
      MER_TRY(gemm(st, dt, ps, M, D, D, b.cur16, D, wv, w.bqkv + 2 * D, MER_ACT_NONE, nullptr, 0, nullptr, 0, cv, 3 * D, mc));
    } else if (f32attn)
    MER_TRY(gemm(st, dt, ps, M, 3 * D, D, b.cur16, D, w.wqkv, w.bqkv, MER_ACT_NONE, nullptr, 0, b.qkv32, 3 * D, none, 0, mcq));
    else
    MER_TRY(gemm(st, dt, ps, M, 3 * D, D, b.cur16, D, w.wqkv, w.bqkv, MER_ACT_NONE, nullptr, 0, nullptr, 0, b.qkv16, 3 * D, mcq));
    if (f32attn) {
      MER_TRY(mer_attention_f32(b.qkv32, b.qkv32 + D, b.qkv32 + 2 * D, 3 * D, b.ctx16.hi, b.ctx16.lo, D, Bseq, T, H, scale, kv_len, dt, (mer_stream_t)st));
    } else if (ab) {   // additive score bias (BEiT) with WavLM's per-layer gate computed from the attention input
      const float* gate = nullptr;
      if (w.gru_w) {
        MER_TRY(mer_wavlm_gate(c.pre_ln ? b.gin32 : x, D, w.gru_w, w.gru_b, w.gru_const, Bseq, T, H, b.gate, (mer_stream_t)st));
        gate = b.gate;
      }
      MER_TRY(mer_attention_bias(b.qkv16.hi, (char*)b.qkv16.hi + (long long)D * 2, (char*)b.qkv16.hi + (long long)2 * D * 2, 3 * D,
                                 b.ctx16.hi, b.ctx16.lo, D, Bseq, T, H, scale, kv_len, ab, w.attn_bias ? (T + 3) / 4 * 4 : ldb, gate, dt, st));
    } else
    MER_TRY(mer_attention(b.qkv16.hi, (char*)b.qkv16.hi + (long long)D * 2, (char*)b.qkv16.hi + (long long)2 * D * 2, 3 * D,
                          b.ctx16.hi, b.ctx16.lo, D, Bseq, T, H, scale, kv_len, dt, st));
    MER_TRY(gemm(st, dt, ps, M, D, D, b.ctx16, D, w.wo, w.bo, MER_ACT_NONE, x, D, b.t32, D, none, 0, mc));
    if (c.pre_ln) {
      MER_TRY(mer_layernorm(b.t32, D, w.ln2_g, w.ln2_b, c.ln_eps, M, D, MER_ACT_NONE, nullptr, 0, b.h1_16.hi, b.h1_16.lo, D, dt, st));
      if (c.ffn_swiglu) {  // weights_in -> fp32 [M, 2F]; silu(first half) * second half -> 16-bit planes [M, F]
        MER_TRY(gemm(st, dt, ps1, M, 2 * F, D, b.h1_16, D, w.w1, w.b1, MER_ACT_NONE, nullptr, 0, b.ffn32, 2 * F, none, 0, mc));
        MER_TRY(mer_swiglu(b.ffn32, 2 * F, M, F, b.f16.hi, b.f16.lo, dt, (mer_stream_t)st));
      } else
      MER_TRY(gemm(st, dt, ps1, M, F, D, b.h1_16, D, w.w1, w.b1, c.act, nullptr, 0, nullptr, 0, b.f16, F, mc));
      MER_TRY(gemm(st, dt, ps2, M, D, F, b.f16, F, w.w2, w.b2, MER_ACT_NONE, b.t32, D, y, D, none, 0, mc));
    } else {
      MER_TRY(mer_layernorm(b.t32, D, w.ln1_g, w.ln1_b, c.ln_eps, M, D, MER_ACT_NONE, b.h1_32, D, b.h1_16.hi, b.h1_16.lo, D, dt, st));
      MER_TRY(gemm(st, dt, ps1, M, F, D, b.h1_16, D, w.w1, w.b1, c.act, nullptr, 0, nullptr, 0, b.f16, F, mc));
      MER_TRY(gemm(st, dt, ps2, M, D, F, b.f16, F, w.w2, w.b2, MER_ACT_NONE, b.h1_32, D, b.t32, D, none, 0, mc));
      MER_TRY(mer_layernorm(b.t32, D, w.ln2_g, w.ln2_b, c.ln_eps, M, D, MER_ACT_NONE, y, D, b.cur16.hi, b.cur16.lo, D, dt, st));
    }
  }
  return MER_OK;
}

static int check_tf(const mer_tf_config& c, const char* who) {
  MER_REQUIRE(c.hidden > 0 && c.heads > 0 && c.hidden % c.heads == 0, MER_EINVAL, "%s: bad hidden/heads", who);
  MER_REQUIRE(c.hidden / c.heads == 64, MER_EUNSUPPORTED, "%s: head_dim %d != 64 unsupported", who, c.hidden / c.heads);
  MER_REQUIRE(c.hidden % 8 == 0 && c.ffn % 8 == 0, MER_ESHAPE, "%s: hidden/ffn must be multiples of 8", who);
  MER_REQUIRE(c.passes >= 1 && c.passes <= 6, MER_EINVAL, "%s: passes must be 1, 2, 3, 4 (MX-corrected), 5 (mean-corrected) or 6 (5 with hi + lo activation planes)", who);
  MER_REQUIRE(c.layers >= 1, MER_EINVAL, "%s: layers < 1", who);
  MER_REQUIRE(c.gated_rel_pos == 0 || c.gated_rel_pos == 1, MER_EINVAL, "%s: gated_rel_pos must be 0 or 1", who);
  MER_REQUIRE(c.mx_skip >= 0 && c.mx_skip <= 7, MER_EINVAL, "%s: mx_skip must be a 3-bit mask", who);
  MER_REQUIRE(!c.ffn_swiglu || c.pre_ln, MER_EUNSUPPORTED, "%s: the SwiGLU feed-forward is only built for pre-LN blocks", who);
  MER_REQUIRE(c.dtype == MER_DT_F16 || c.dtype == MER_DT_BF16, MER_EINVAL, "%s: bad dtype", who);
  return MER_OK;
}

static int last4_pool(hipStream_t st, const HsMap& hs, int layers, long long M, int D, float* frames, const int* seg_start,
                      const int* seg_len, int nseg, float* pooled) {
  if (!frames && !pooled) return MER_OK;
  // hidden_states[-4..-1] (extract_audio_huggingface.py:93,98); fewer than 4 states -> as many as exist
  const float* h[4] = {nullptr, nullptr, nullptr, nullptr};
  int n = 0;
  for (int l = layers - 3; l <= layers; ++l)
    if (l >= 0) h[n++] = hs.at(l);
  return mer_sum_pool(h[0], h[1], h[2], h[3], M, D, frames, seg_start, seg_len, nseg, pooled, st);
}

}  // namespace mer

using namespace mer;

// =============================================================================================
// HuBERT / wav2vec2
// =============================================================================================
struct mer_hubert {
  mer_hubert_config cfg;
  mer_hubert_weights w;
  std::vector<mer_tf_layer> layers;
};

extern "C" int mer_hubert_create(const mer_hubert_config* cfg, const mer_hubert_weights* w, mer_hubert** out) {
  MER_REQUIRE(cfg && w && out, MER_EINVAL, "mer_hubert_create: null argument");
  MER_TRY(check_tf(cfg->tf, "mer_hubert_create"));
  MER_REQUIRE(cfg->n_conv >= 2 && cfg->n_conv <= MER_MAX_CONV, MER_EINVAL, "mer_hubert_create: n_conv=%d", cfg->n_conv);
  MER_REQUIRE(cfg->conv_dim % 8 == 0, MER_ESHAPE, "mer_hubert_create: conv_dim %% 8 != 0");
  MER_REQUIRE(cfg->conv_passes >= 1 && cfg->conv_passes <= 5, MER_EINVAL, "mer_hubert_create: conv_passes must be 1 .. 5");
  MER_REQUIRE(cfg->tf.hidden % cfg->pos_groups == 0 && (cfg->tf.hidden / cfg->pos_groups) % 8 == 0, MER_ESHAPE,
              "mer_hubert_create: hidden/pos_groups must be a multiple of 8");
  MER_REQUIRE(cfg->pos_layers >= 0 && cfg->pos_layers <= MER_MAX_POS, MER_EINVAL, "mer_hubert_create: pos_layers=%d", cfg->pos_layers);
  MER_REQUIRE(w->layers && w->conv0_w && w->fp_w.hi && (cfg->pos_layers > 0 || w->pos_w.hi), MER_EINVAL, "mer_hubert_create: missing weights");
  for (int i = 0; i < cfg->pos_layers; ++i)
    MER_REQUIRE(w->pos_ws[i].hi && w->pos_bs[i], MER_EINVAL, "mer_hubert_create: missing positional conv layer %d", i);
  if (cfg->tf.gated_rel_pos)
    for (int l = 0; l < cfg->tf.layers; ++l)
      MER_REQUIRE(w->layers[l].gru_w && w->layers[l].gru_b && w->layers[l].gru_const, MER_EINVAL,
                  "mer_hubert_create: WavLM layer %d lacks the gru_rel_pos parameters", l);
  mer_hubert* h = new mer_hubert();
  h->cfg = *cfg;
  h->w = *w;
  h->layers.assign(w->layers, w->layers + cfg->tf.layers);
  h->w.layers = h->layers.data();
  *out = h;
  return MER_OK;
}
extern "C" void mer_hubert_destroy(mer_hubert* h) { delete h; }

static void hubert_lengths(const mer_hubert_config& c, int L, int* T) {
  int t = L;
  for (int i = 0; i < c.n_conv; ++i) {
    t = (t - c.conv_kernel[i]) / c.conv_stride[i] + 1;
    T[i] = t;
  }
}
extern "C" int mer_hubert_out_frames(const mer_hubert* h, int L) {
  if (!h) return MER_EINVAL;
  int T[MER_MAX_CONV];
  hubert_lengths(h->cfg, L, T);
  return T[h->cfg.n_conv - 1];
}

struct HubertPlan {
  double* stats;
  P16 convA, convB;
  float* conv32;       // feat_extract_norm == "layer": fp32 conv output awaiting its LayerNorm
  float* conv_last32;
  P16 fp16;
  float* hproj;
  P16 pospack;
  float* posbuf;       // data2vec-audio: output of a positional conv layer (input of the next)
  float* ring;
  int* vlen;           // ragged batches: valid frames of every clip after each conv layer, [n_conv][B] (device int32)
  TfBufs tf;
  int T[MER_MAX_CONV];
};

static long long hubert_plan(const mer_hubert* h, Arena& ar, int B, int L, bool want_hs, HubertPlan& p) {
  const mer_hubert_config& c = h->cfg;
  hubert_lengths(c, L, p.T);
  const long long C = c.conv_dim, D = c.tf.hidden;
  const int Tn = p.T[c.n_conv - 1];
  const long long M = (long long)B * Tn;
  const bool clo = c.conv_passes == 3;
  p.stats = (double*)ar.take((long long)B * C * 2 * 8);
  p.vlen = (int*)ar.take((long long)c.n_conv * B * 4);
  p.convA = take16(ar, (long long)B * p.T[0] * C, clo);
  p.convB = take16(ar, (long long)B * p.T[1] * C, clo);
  p.conv32 = c.feat_norm_group ? nullptr : (float*)ar.take((long long)B * p.T[0] * C * 4);
  p.conv_last32 = (float*)ar.take(M * C * 4);
  p.fp16 = take16(ar, M * C, clo);
  p.hproj = (float*)ar.take(M * D * 4);
  p.pospack = take16(ar, (long long)B * (Tn + c.pos_k) * D, clo);
  p.posbuf = c.pos_layers > 0 ? (float*)ar.take(M * D * 4) : nullptr;
  p.ring = want_hs ? nullptr : (float*)ar.take(5 * M * D * 4);
  int kmax = (int)C;
  for (int i = 1; i < c.n_conv; ++i) kmax = c.conv_kernel[i] * (int)C > kmax ? c.conv_kernel[i] * (int)C : kmax;
  tf_plan(ar, c.tf, M, B, p.tf, c.conv_passes == 5, kmax, C > D ? C : D);
  return ar.off;
}

extern "C" long long mer_hubert_workspace_bytes(const mer_hubert* h, int B, int L, int want_hidden_states) {
  if (!h || B <= 0 || L <= 0) return MER_EINVAL;
  Arena ar(nullptr, 0, true);
  HubertPlan p;
  return hubert_plan(h, ar, B, L, want_hidden_states != 0, p) + 256;
}

extern "C" int mer_hubert_forward(const mer_hubert* h, const float* wav, int B, int L, void* workspace, long long workspace_bytes,
                                  float* hidden_states, float* frames, const int* seg_start, const int* seg_len, int nseg,
                                  float* pooled, mer_stream_t stream) {
  return mer_hubert_forward_bias(h, wav, B, L, workspace, workspace_bytes, hidden_states, frames, seg_start, seg_len, nseg, pooled,
                                 nullptr, 0, stream);
}

extern "C" int mer_hubert_forward_bias(const mer_hubert* h, const float* wav, int B, int L, void* workspace,
                                  long long workspace_bytes, float* hidden_states, float* frames, const int* seg_start,
                                  const int* seg_len, int nseg, float* pooled, const float* pos_bias, long long ldb,
                                  mer_stream_t stream) {
  return mer_hubert_forward_ragged(h, wav, B, L, nullptr, workspace, workspace_bytes, hidden_states, frames, seg_start, seg_len, nseg,
                                   pooled, pos_bias, ldb, stream);
}

extern "C" int mer_hubert_forward_ragged(const mer_hubert* h, const float* wav, int B, int L, const int* valid_samples, void* workspace,
                                  long long workspace_bytes, float* hidden_states, float* frames, const int* seg_start,
                                  const int* seg_len, int nseg, float* pooled, const float* pos_bias, long long ldb,
                                  mer_stream_t stream) {
  MER_REQUIRE(h && wav && workspace, MER_EINVAL, "mer_hubert_forward: null argument");
  MER_REQUIRE(!h->cfg.tf.gated_rel_pos || pos_bias, MER_EINVAL,
              "mer_hubert_forward: this is a WavLM handle — call mer_hubert_forward_bias with the relative position bias table");
  MER_REQUIRE(B > 0 && L >= 400, MER_ESHAPE, "mer_hubert_forward: need B>0 and L>=400 samples (B=%d L=%d)", B, L);
  MER_REQUIRE(((uintptr_t)workspace & 255) == 0, MER_EINVAL, "mer_hubert_forward: workspace must be 256-byte aligned");
  const mer_hubert_config& c = h->cfg;
  const mer_hubert_weights& w = h->w;
  hipStream_t st = (hipStream_t)stream;
  Arena ar(workspace, workspace_bytes, false);
  HubertPlan p;
  hubert_plan(h, ar, B, L, hidden_states != nullptr, p);
  MER_REQUIRE(ar.ok(), MER_ENOMEM, "mer_hubert_forward: workspace too small (%lld < %lld bytes)", workspace_bytes, ar.off);
  const int C = c.conv_dim, D = c.tf.hidden, dt = c.tf.dtype, cps = c.conv_passes;
  const int Tn = p.T[c.n_conv - 1];
  MER_REQUIRE(Tn >= 1, MER_ESHAPE, "mer_hubert_forward: input too short");
  const int M = B * Tn;
  const P16 none = {nullptr, nullptr};
  // conv_passes == 5 (the "mean_all" STUDY preset — not what "mean" ships): batch-mean correction of the conv GEMMs / the projection.
  // Unsound for real audio: the conv layers read un-normalised GELU outputs, a quiet passage's rows are 20-50x smaller than the batch
  // mean, the bias is an absolute offset they cannot absorb and the projection's LayerNorm magnifies it (1e-2 on loud / quiet clips,
  // DESIGN.md §4, test_hubert_loud_and_quiet_passages).  "mean" runs the conv stack with conv_passes == 4 (per-row MX correction) and
  // keeps the batch-mean bias behind LayerNorms.  In a ragged batch only the output frames that come from a clip's own samples count.
  const bool corr_on = cps == 5 && p.tf.corr.mean16;

  // ragged batch: per-row valid frame counts after conv 0 (GroupNorm statistics) and after the stack (positional conv zeros,
  // attention key mask), derived on the device from the rows' sample counts
  const int* t0_len = nullptr;
  const int* tn_len = nullptr;
  if (valid_samples) {
    MER_TRY(mer_hubert_valid_frames_all(valid_samples, B, L, c.n_conv, c.conv_kernel, c.conv_stride, p.vlen, stream));
    t0_len = p.vlen;
    tn_len = p.vlen + (long long)(c.n_conv - 1) * B;
  }
  P16 src = p.convA, dst = p.convB;
  if (c.feat_norm_group) {
    // conv0 + GroupNorm + GELU -> channels-last planes
    MER_TRY(mer_hubert_conv0_gn_ragged(wav, B, L, w.conv0_w, C, c.conv_kernel[0], c.conv_stride[0], w.conv_norm_g[0], w.conv_norm_b[0],
                                       1e-5f, p.stats, p.convA.hi, p.convA.lo, dt, t0_len, st));
  } else {
    // "layer" front end: every conv is followed by LayerNorm(C) over channels, then GELU (HF:hubert/modeling_hubert.py:127-151)
    MER_TRY(mer_hubert_conv0_plain(wav, B, L, w.conv0_w, c.conv_bias ? w.conv_b[0] : nullptr, C, c.conv_kernel[0], c.conv_stride[0],
                                   p.conv32, st));
    MER_TRY(mer_layernorm(p.conv32, C, w.conv_norm_g[0], w.conv_norm_b[0], 1e-5f, B * p.T[0], C, MER_ACT_GELU, nullptr, 0,
                          p.convA.hi, p.convA.lo, C, dt, st));
  }
  // conv1.. as implicit-im2col GEMMs (row m=(b,t) starts at b*T_in*C + t*stride*C, K = k*C contiguous)
  for (int i = 1; i < c.n_conv; ++i) {
    const bool last = i == c.n_conv - 1;
    mer_gemm16_args g;
    memset(&g, 0, sizeof(g));
    g.M = B * p.T[i]; g.N = C; g.K = c.conv_kernel[i] * C; g.dtype = dt;
    g.a_hi = src.hi; g.a_lo = src.lo; g.lda = (long long)c.conv_stride[i] * C;
    g.a_rows_per_batch = p.T[i]; g.a_batch_stride = (long long)p.T[i - 1] * C;
    g.w_hi = w.conv_w[i].hi; g.w_lo = w.conv_w[i].lo; g.w_mx = w.conv_w[i].mx; g.ldw = g.K;
    g.w_hi_blk = w.conv_w[i].hi_blk; g.w_lo_blk = w.conv_w[i].lo_blk; g.w_hi_blkp = w.conv_w[i].hi_blkp; g.w_hi_blkq = w.conv_w[i].hi_blkq;
    g.bias = c.conv_bias ? w.conv_b[i] : nullptr;
    g.nbatch = 1; g.nb_inner = 1; g.passes = cps;
    const CorrWs ccw = {&p.tf.corr, p.T[i], valid_samples ? p.vlen + (long long)i * B : nullptr, 0};
    const CorrWs* cw = corr_on ? &ccw : nullptr;
    if (c.feat_norm_group) {
      g.act = MER_ACT_GELU;
      if (last) { g.c32 = p.conv_last32; g.ldc32 = C; }
      if (!last || !c.feat_proj_layer_norm) { g.c16_hi = (last ? p.fp16.hi : dst.hi); g.c16_lo = (last ? p.fp16.lo : dst.lo); g.ldc16 = C; }
      MER_TRY(run_gemm(st, g, cw));
    } else {
      g.act = MER_ACT_NONE;
      g.c32 = p.conv32; g.ldc32 = C;
      MER_TRY(run_gemm(st, g, cw));
      const bool want32 = last && c.feat_proj_layer_norm;
      P16 o = last ? p.fp16 : dst;
      MER_TRY(mer_layernorm(p.conv32, C, w.conv_norm_g[i], w.conv_norm_b[i], 1e-5f, B * p.T[i], C, MER_ACT_GELU,
                            want32 ? p.conv_last32 : nullptr, C, want32 ? nullptr : o.hi, want32 ? nullptr : o.lo, C, dt, st));
    }
    P16 t = src; src = dst; dst = t;
  }
  // feature projection: LayerNorm(C) -> Linear(C -> D)   (HF:hubert/modeling_hubert.py:216-231)
  if (c.feat_proj_layer_norm)
    MER_TRY(mer_layernorm(p.conv_last32, C, w.fp_ln_g, w.fp_ln_b, c.tf.ln_eps, M, C, MER_ACT_NONE, nullptr, 0, p.fp16.hi, p.fp16.lo, C, dt, st));
  const CorrWs pcw = {&p.tf.corr, Tn, tn_len, 0};
  // (behind a LayerNorm: rows of one size, so the per-sequence table applies under "mean" as it does in the blocks; the conv stack above
  //  reads un-normalised GELU outputs and keeps its per-row MX correction — DESIGN.md §4)
  const bool fp_tab = (cps == 5 || ((c.tf.passes == 5 || c.tf.passes == 6) && c.feat_proj_layer_norm)) && p.tf.corr.mean16;
  // (where the preset carries the projection's input as hi + lo planes — conv stack three passes: mean_conv3 / a2_conv3, or blocks on
  //  passes == 6 — the table rides on the two-pass activation split, a_hi*w_hi + a_lo*w_hi, not on the hi plane alone: ADVICE r5)
  const int fp_passes = fp_tab ? ((cps == 3 || c.tf.passes == 6) ? 6 : 5) : cps;
  MER_TRY(gemm(st, dt, fp_passes, M, D, C, p.fp16, C, w.fp_w, w.fp_b, MER_ACT_NONE, nullptr, 0, p.hproj, D, none, 0, fp_tab ? &pcw : nullptr));

  // positional conv: x + GELU(Conv1d(D, D, k, pad k/2, groups G)(x)[..., :-1])   (HF:...:45-103)
  HsMap hs;
  hs.stride = (long long)M * D;
  if (hidden_states) { hs.base = hidden_states; hs.ring = 0; } else { hs.base = p.ring; hs.ring = 5; }
  {
    const int G = c.pos_groups, Dg = D / G, K = c.pos_k;
    const int nl = c.pos_layers > 0 ? c.pos_layers : 1;
    const float* src = p.hproj;
    for (int i = 0; i < nl; ++i) {
      const bool d2v = c.pos_layers > 0;
      MER_TRY(mer_posconv_pack_ragged(src, B, Tn, D, G, K, p.pospack.hi, p.pospack.lo, dt, tn_len, st));
      mer_gemm16_args g;
      memset(&g, 0, sizeof(g));
      g.M = Tn; g.N = Dg; g.K = K * Dg; g.dtype = dt;
      g.a_hi = p.pospack.hi; g.a_lo = p.pospack.lo; g.lda = Dg;
      const mer_w16& pw = d2v ? w.pos_ws[i] : w.pos_w;
      g.w_hi = pw.hi; g.w_lo = pw.lo; g.ldw = g.K;
      g.bias = d2v ? w.pos_bs[i] : w.pos_b;
      g.nbatch = B * G; g.nb_inner = G;
      g.a_so = (long long)G * (Tn + K) * Dg; g.a_si = (long long)(Tn + K) * Dg;
      g.w_si = (long long)Dg * K * Dg; g.bias_si = Dg;
      g.c_so = (long long)Tn * D; g.c_si = Dg;
      g.passes = (cps == 4 || cps == 5) ? 2 : cps;   // batched narrow GEMM: neither the MX kernel nor the mean correction covers it, say so up front
      if (!d2v) {  // HuBERT / wav2vec2: GELU and the residual ride in the epilogue
        g.act = MER_ACT_GELU;
        g.residual = p.hproj; g.ldr = D;
        g.c32 = c.stable_layer_norm ? hs.at(0) : p.tf.t32; g.ldc32 = D;
        MER_TRY(mer_gemm16(&g, stream));
      } else {     // data2vec-audio: conv -> LayerNorm(no affine, eps 1e-5) -> GELU, five times; the residual once at the end
        g.act = MER_ACT_NONE;
        g.c32 = p.tf.t32; g.ldc32 = D;
        MER_TRY(mer_gemm16(&g, stream));
        MER_TRY(mer_layernorm(p.tf.t32, D, nullptr, nullptr, 1e-5f, M, D, MER_ACT_GELU, p.posbuf, D, nullptr, nullptr, 0, dt, st));
        src = p.posbuf;
      }
    }
    if (c.pos_layers > 0)   // hidden = hidden + pos_conv_embed(hidden)
      MER_TRY(mer_sum_pool(p.hproj, p.posbuf, nullptr, nullptr, M, D, c.stable_layer_norm ? hs.at(0) : p.tf.t32, nullptr, nullptr, 0, nullptr, st));
  }
  if (!c.stable_layer_norm)
    MER_TRY(mer_layernorm(p.tf.t32, D, w.enc_ln_g, w.enc_ln_b, c.tf.ln_eps, M, D, MER_ACT_NONE, hs.at(0), D, p.tf.cur16.hi, p.tf.cur16.lo, D, dt, st));

  MER_TRY(tf_forward(st, c.tf, h->layers.data(), B, Tn, hs, p.tf, tn_len, c.tf.gated_rel_pos ? pos_bias : nullptr, ldb));

  if (c.stable_layer_norm)  // final LayerNorm only on the last state (HF:hubert/modeling_hubert.py:612)
    MER_TRY(mer_layernorm(hs.at(c.tf.layers), D, w.enc_ln_g, w.enc_ln_b, c.tf.ln_eps, M, D, MER_ACT_NONE, hs.at(c.tf.layers), D, nullptr, nullptr, 0, dt, st));

  return last4_pool(st, hs, c.tf.layers, M, D, frames, seg_start, seg_len, nseg, pooled);
}

// =============================================================================================
// CLIP vision tower
// =============================================================================================
struct mer_vit {
  mer_vit_config cfg;
  mer_vit_weights w;
  std::vector<mer_tf_layer> layers;
};

extern "C" int mer_vit_create(const mer_vit_config* cfg, const mer_vit_weights* w, mer_vit** out) {
  MER_REQUIRE(cfg && w && out, MER_EINVAL, "mer_vit_create: null argument");
  MER_TRY(check_tf(cfg->tf, "mer_vit_create"));
  MER_REQUIRE(cfg->tf.pre_ln == 1, MER_EUNSUPPORTED, "mer_vit_create: ViT blocks are pre-LN");
  MER_REQUIRE(cfg->image_size % cfg->patch_size == 0, MER_ESHAPE, "mer_vit_create: image/patch size");
  MER_REQUIRE(w->layers && w->patch_w.hi && w->cls && w->pos, MER_EINVAL, "mer_vit_create: missing weights");
  MER_REQUIRE(cfg->variant == 0 || cfg->variant == 1, MER_EINVAL, "mer_vit_create: variant must be 0 (CLIP) or 1 (DINOv2)");
  if (cfg->variant == 0) {
    MER_REQUIRE(w->pre_ln_g && w->post_ln_g && w->proj_w.hi, MER_EINVAL, "mer_vit_create: CLIP needs pre/post LayerNorm and the projection");
  } else {
    MER_REQUIRE(cfg->proj_dim == cfg->tf.hidden, MER_ESHAPE, "mer_vit_create: DINOv2 features are hidden-sized (proj_dim == hidden)");
  }
  mer_vit* h = new mer_vit();
  h->cfg = *cfg;
  h->w = *w;
  h->layers.assign(w->layers, w->layers + cfg->tf.layers);
  h->w.layers = h->layers.data();
  *out = h;
  return MER_OK;
}
extern "C" void mer_vit_destroy(mer_vit* h) { delete h; }

struct VitPlan {
  P16 patches;
  float* patch32;
  float* x;
  P16 cls16;
  float* feats;
  TfBufs tf;
  ClsBufs cls;
};

// CLIP features come from the CLS row alone: the last block runs for those rows only unless the caller wants every token
// (mer_attention_cls keeps a sequence's scores in registers: T <= 584 tokens; the CLS branch of tf_forward assumes pre-LN blocks)
static bool vit_cls_only(const mer_vit_config& c) {
  const long long g = c.image_size / (c.patch_size > 0 ? c.patch_size : 1);
  // (not under the three-pass preset: its blocks run mer_attention_f32 on fp32 q | k | v, the CLS kernel reads 16-bit K / V planes)
  return c.variant == 0 && c.tf.pre_ln && !c.tf.ffn_swiglu && !c.tf.gated_rel_pos && c.tf.layers >= 1 && g * g + 1 <= 584 && c.tf.passes != 3 && !(c.tf.passes == 6 && c.tf.attn_f32);
}

static long long vit_plan(const mer_vit* h, Arena& ar, int N, VitPlan& p) {
  const mer_vit_config& c = h->cfg;
  const long long g = c.image_size / c.patch_size, P = g * g, D = c.tf.hidden;
  const long long cols = ((long long)c.channels * c.patch_size * c.patch_size + 7) / 8 * 8;
  const bool lo = c.tf.passes == 3 || c.tf.passes == 6;
  p.patches = take16(ar, N * P * cols, lo);
  p.patch32 = (float*)ar.take(N * P * D * 4);
  p.x = (float*)ar.take(N * (P + 1) * D * 4);
  p.cls16 = take16(ar, (long long)N * D, lo);
  p.feats = (float*)ar.take((long long)N * c.proj_dim * 4);
  tf_plan(ar, c.tf, N * (P + 1), N, p.tf, false, (int)cols);
  if (vit_cls_only(c)) cls_plan(ar, c.tf, N, p.cls);
  return ar.off;
}

extern "C" long long mer_vit_workspace_bytes(const mer_vit* h, int N) {
  if (!h || N <= 0) return MER_EINVAL;
  Arena ar(nullptr, 0, true);
  VitPlan p;
  return vit_plan(h, ar, N, p) + 256;
}

extern "C" int mer_vit_forward(const mer_vit* h, const float* pixels, int N, void* workspace, long long workspace_bytes,
                               float* image_features, const int* seg_start, const int* seg_len, int nseg, float* pooled,
                               mer_stream_t stream) {
  return mer_vit_forward_tokens(h, pixels, N, workspace, workspace_bytes, image_features, seg_start, seg_len, nseg, pooled, nullptr, stream);
}

extern "C" int mer_vit_forward_tokens(const mer_vit* h, const float* pixels, int N, void* workspace, long long workspace_bytes,
                                      float* image_features, const int* seg_start, const int* seg_len, int nseg, float* pooled,
                                      float* tokens_out, mer_stream_t stream) {
  MER_REQUIRE(h && pixels && workspace && N > 0, MER_EINVAL, "mer_vit_forward: bad argument");
  MER_REQUIRE(((uintptr_t)workspace & 255) == 0, MER_EINVAL, "mer_vit_forward: workspace must be 256-byte aligned");
  const mer_vit_config& c = h->cfg;
  const mer_vit_weights& w = h->w;
  hipStream_t st = (hipStream_t)stream;
  Arena ar(workspace, workspace_bytes, false);
  VitPlan p;
  vit_plan(h, ar, N, p);
  MER_REQUIRE(ar.ok(), MER_ENOMEM, "mer_vit_forward: workspace too small (%lld < %lld bytes)", workspace_bytes, ar.off);
  const int g = c.image_size / c.patch_size, P = g * g, D = c.tf.hidden, dt = c.tf.dtype, ps = c.tf.passes;
  const int cols = (c.channels * c.patch_size * c.patch_size + 7) / 8 * 8;
  const P16 none = {nullptr, nullptr};
  // patch embedding: Conv2d(stride == kernel, no bias) == GEMM over patch rows   (HF:clip/modeling_clip.py:138-217)
  MER_TRY(mer_vit_patchify(pixels, N, c.channels, c.image_size, c.image_size, c.patch_size, p.patches.hi, p.patches.lo, dt, st));
  const CorrWs pcw = {&p.tf.corr, P, nullptr, 0};
  MER_TRY(gemm(st, dt, ps, N * P, D, cols, p.patches, cols, w.patch_w, c.variant == 1 ? w.patch_b : nullptr, MER_ACT_NONE, nullptr, 0,
               p.patch32, D, none, 0, &pcw));
  // [CLS] + position embeddings (+ pre_layrnorm for CLIP; DINOv2 has no embedding LayerNorm: gamma == NULL stores the plain sum)
  MER_TRY(mer_vit_assemble(p.patch32, w.cls, w.pos, c.variant == 0 ? w.pre_ln_g : nullptr, c.variant == 0 ? w.pre_ln_b : nullptr,
                           c.tf.ln_eps, N, P, D, p.x, nullptr, nullptr, dt, st));
  HsMap hs;
  hs.base = p.x; hs.stride = 0; hs.ring = 1;  // pre-LN blocks update the residual stream in place
  bool cls_only = vit_cls_only(c) && !tokens_out;
  for (int l = 0; l < c.tf.layers; ++l) cls_only = cls_only && !h->layers[l].attn_bias;
  MER_TRY(tf_forward(st, c.tf, h->layers.data(), N, P + 1, hs, p.tf, nullptr, nullptr, 0, cls_only ? &p.cls : nullptr));
  if (tokens_out)
    MER_REQUIRE(hipMemcpyAsync(tokens_out, p.x, (size_t)N * (P + 1) * D * 4, hipMemcpyDeviceToDevice, st) == hipSuccess, MER_ELAUNCH,
                "mer_vit_forward: copy of the token states failed");
  if (c.variant == 1) {   // DINOv2: token sum of the last residual stream (extract_vision_huggingface.py:142), then the per-clip mean
    float* f = image_features ? image_features : p.feats;
    MER_TRY(mer_token_reduce(p.x, N, P + 1, D, 1.0f, f, stream));
    if (pooled) MER_TRY(mer_sum_pool(f, nullptr, nullptr, nullptr, N, D, nullptr, seg_start, seg_len, nseg, pooled, st));
    return MER_OK;
  }
  // pooled = post_layernorm(x[:, 0]); features = visual_projection(pooled)   (HF:clip/modeling_clip.py:719-748)
  MER_TRY(mer_layernorm(cls_only ? p.cls.y32 : p.x, cls_only ? (long long)D : (long long)(P + 1) * D, w.post_ln_g, w.post_ln_b, c.tf.ln_eps, N, D,
                        MER_ACT_NONE, nullptr, 0, p.cls16.hi, p.cls16.lo, D, dt, st));
  float* feats = image_features ? image_features : p.feats;
  MER_TRY(gemm(st, dt, ps, N, c.proj_dim, D, p.cls16, D, w.proj_w, nullptr, MER_ACT_NONE, nullptr, 0, feats, c.proj_dim, none, 0));
  if (pooled) MER_TRY(mer_sum_pool(feats, nullptr, nullptr, nullptr, N, c.proj_dim, nullptr, seg_start, seg_len, nseg, pooled, st));
  return MER_OK;
}

// =============================================================================================
// VideoMAE
// =============================================================================================
struct mer_videomae {
  mer_videomae_config cfg;
  mer_videomae_weights w;
  std::vector<mer_tf_layer> layers;
};

extern "C" int mer_videomae_create(const mer_videomae_config* cfg, const mer_videomae_weights* w, mer_videomae** out) {
  MER_REQUIRE(cfg && w && out, MER_EINVAL, "mer_videomae_create: null argument");
  MER_TRY(check_tf(cfg->tf, "mer_videomae_create"));
  MER_REQUIRE(cfg->tf.pre_ln == 1, MER_EUNSUPPORTED, "mer_videomae_create: VideoMAE blocks are pre-LN");
  MER_REQUIRE(cfg->image_size % cfg->patch_size == 0 && cfg->patch_size % 4 == 0 && cfg->num_frames % cfg->tubelet_size == 0,
              MER_ESHAPE, "mer_videomae_create: image/patch/tubelet size");
  MER_REQUIRE(w->layers && w->patch_w.hi && w->pos, MER_EINVAL, "mer_videomae_create: missing weights");
  mer_videomae* h = new mer_videomae();
  h->cfg = *cfg;
  h->w = *w;
  h->layers.assign(w->layers, w->layers + cfg->tf.layers);
  h->w.layers = h->layers.data();
  *out = h;
  return MER_OK;
}
extern "C" void mer_videomae_destroy(mer_videomae* h) { delete h; }

struct VmaePlan {
  P16 patches;
  float* x;
  TfBufs tf;
};
static long long vmae_plan(const mer_videomae* h, Arena& ar, int B, VmaePlan& p) {
  const mer_videomae_config& c = h->cfg;
  const long long g = c.image_size / c.patch_size, NP = g * g * (c.num_frames / c.tubelet_size), D = c.tf.hidden;
  const long long cols = (long long)c.channels * c.tubelet_size * c.patch_size * c.patch_size;
  p.patches = take16(ar, B * NP * cols, c.tf.passes == 3 || c.tf.passes == 6);
  p.x = (float*)ar.take(B * NP * D * 4);
  tf_plan(ar, c.tf, B * NP, B, p.tf, false, (int)cols);
  return ar.off;
}
extern "C" long long mer_videomae_workspace_bytes(const mer_videomae* h, int B) {
  if (!h || B <= 0) return MER_EINVAL;
  Arena ar(nullptr, 0, true);
  VmaePlan p;
  return vmae_plan(h, ar, B, p) + 256;
}

extern "C" int mer_videomae_forward(const mer_videomae* h, const float* pixels, int B, void* workspace, long long workspace_bytes,
                                    float* last_hidden_state, const int* seg_start, const int* seg_len, int nseg, float* pooled,
                                    mer_stream_t stream) {
  MER_REQUIRE(h && pixels && workspace && B > 0, MER_EINVAL, "mer_videomae_forward: bad argument");
  MER_REQUIRE(((uintptr_t)workspace & 255) == 0, MER_EINVAL, "mer_videomae_forward: workspace must be 256-byte aligned");
  const mer_videomae_config& c = h->cfg;
  const mer_videomae_weights& w = h->w;
  hipStream_t st = (hipStream_t)stream;
  Arena ar(workspace, workspace_bytes, false);
  VmaePlan p;
  vmae_plan(h, ar, B, p);
  MER_REQUIRE(ar.ok(), MER_ENOMEM, "mer_videomae_forward: workspace too small (%lld < %lld bytes)", workspace_bytes, ar.off);
  const int g = c.image_size / c.patch_size, NP = g * g * (c.num_frames / c.tubelet_size), D = c.tf.hidden;
  const int cols = c.channels * c.tubelet_size * c.patch_size * c.patch_size, dt = c.tf.dtype, ps = c.tf.passes;
  const P16 none = {nullptr, nullptr};
  float* x = last_hidden_state ? last_hidden_state : p.x;
  // tubelet embedding: Conv3d(stride == kernel, bias) == GEMM over tubelet rows; + fixed sin-cos positions
  MER_TRY(mer_video_patchify(pixels, B, c.num_frames, c.channels, c.image_size, c.image_size, c.patch_size, c.tubelet_size,
                             p.patches.hi, p.patches.lo, dt, st));
  const CorrWs pcw = {&p.tf.corr, NP, nullptr, 0};
  MER_TRY(gemm(st, dt, ps, B * NP, D, cols, p.patches, cols, w.patch_w, w.patch_b, MER_ACT_NONE, nullptr, 0, x, D, none, 0, &pcw));
  MER_TRY(mer_add_pos(x, w.pos, (long long)B * NP, NP, D, st));
  HsMap hs;
  hs.base = x; hs.stride = 0; hs.ring = 1;
  MER_TRY(tf_forward(st, c.tf, h->layers.data(), B, NP, hs, p.tf, nullptr));
  if (c.final_ln)
    MER_TRY(mer_layernorm(x, D, w.final_ln_g, w.final_ln_b, c.tf.ln_eps, B * NP, D, MER_ACT_NONE, x, D, nullptr, nullptr, 0, dt, st));
  if (pooled) MER_TRY(mer_sum_pool(x, nullptr, nullptr, nullptr, (long long)B * NP, D, nullptr, seg_start, seg_len, nseg, pooled, st));
  return MER_OK;
}

// =============================================================================================
// BERT / RoBERTa
// =============================================================================================
struct mer_bert {
  mer_bert_config cfg;
  mer_bert_weights w;
  std::vector<mer_tf_layer> layers;
};

extern "C" int mer_bert_create(const mer_bert_config* cfg, const mer_bert_weights* w, mer_bert** out) {
  MER_REQUIRE(cfg && w && out, MER_EINVAL, "mer_bert_create: null argument");
  MER_TRY(check_tf(cfg->tf, "mer_bert_create"));
  MER_REQUIRE(cfg->tf.pre_ln == 0, MER_EUNSUPPORTED, "mer_bert_create: BERT blocks are post-LN");
  MER_REQUIRE(w->layers && w->word && w->pos, MER_EINVAL, "mer_bert_create: missing weights");
  if (cfg->emb_dim > 0 && cfg->emb_dim != cfg->tf.hidden)
    MER_REQUIRE(cfg->emb_dim % 8 == 0 && w->emb_proj_w.hi && w->emb_proj_b, MER_EINVAL,
                "mer_bert_create: emb_dim=%d needs emb_proj_w / emb_proj_b (and emb_dim %% 8 == 0)", cfg->emb_dim);
  mer_bert* h = new mer_bert();
  h->cfg = *cfg;
  h->w = *w;
  h->layers.assign(w->layers, w->layers + cfg->tf.layers);
  h->w.layers = h->layers.data();
  *out = h;
  return MER_OK;
}
extern "C" void mer_bert_destroy(mer_bert* h) { delete h; }

struct BertPlan {
  float* ring;
  P16 emb16;   // emb_dim != hidden: the LayerNorm-ed embeddings awaiting their projection
  TfBufs tf;
};
static long long bert_plan(const mer_bert* h, Arena& ar, int B, int T, bool want_hs, BertPlan& p) {
  const long long M = (long long)B * T, D = h->cfg.tf.hidden;
  const int E = h->cfg.emb_dim;
  p.ring = want_hs ? nullptr : (float*)ar.take(5 * M * D * 4);
  p.emb16 = (E > 0 && E != D) ? take16(ar, M * E, h->cfg.tf.passes == 3 || h->cfg.tf.passes == 6) : P16{nullptr, nullptr};
  tf_plan(ar, h->cfg.tf, M, B, p.tf);
  return ar.off;
}
extern "C" long long mer_bert_workspace_bytes(const mer_bert* h, int B, int T, int want_hidden_states) {
  if (!h || B <= 0 || T <= 0) return MER_EINVAL;
  Arena ar(nullptr, 0, true);
  BertPlan p;
  return bert_plan(h, ar, B, T, want_hidden_states != 0, p) + 256;
}

extern "C" int mer_bert_forward(const mer_bert* h, const int64_t* ids, const int64_t* token_type, const int* lengths, int B,
                                int T, void* workspace, long long workspace_bytes, float* hidden_states, float* frames,
                                const int* seg_start, const int* seg_len, int nseg, float* pooled, mer_stream_t stream) {
  MER_REQUIRE(h && ids && workspace && B > 0 && T > 0, MER_EINVAL, "mer_bert_forward: bad argument");
  MER_REQUIRE(T <= 512, MER_EUNSUPPORTED, "mer_bert_forward: T=%d > 512", T);
  MER_REQUIRE(((uintptr_t)workspace & 255) == 0, MER_EINVAL, "mer_bert_forward: workspace must be 256-byte aligned");
  const mer_bert_config& c = h->cfg;
  const mer_bert_weights& w = h->w;
  MER_REQUIRE(c.pos_mode == 0 ? T <= c.max_pos : T + c.pad_id + 1 <= c.max_pos, MER_ESHAPE,
              "mer_bert_forward: T=%d exceeds the position table (%d)", T, c.max_pos);
  hipStream_t st = (hipStream_t)stream;
  Arena ar(workspace, workspace_bytes, false);
  BertPlan p;
  bert_plan(h, ar, B, T, hidden_states != nullptr, p);
  MER_REQUIRE(ar.ok(), MER_ENOMEM, "mer_bert_forward: workspace too small (%lld < %lld bytes)", workspace_bytes, ar.off);
  const int D = c.tf.hidden, dt = c.tf.dtype;
  const long long M = (long long)B * T;
  HsMap hs;
  hs.stride = M * D;
  if (hidden_states) { hs.base = hidden_states; hs.ring = 0; } else { hs.base = p.ring; hs.ring = 5; }
  if (c.emb_dim > 0 && c.emb_dim != D) {   // factorised embeddings: E-wide tables + LayerNorm, then Linear(E -> D) = hidden_states[0]
    MER_TRY(mer_bert_embed(ids, token_type, B, T, c.emb_dim, w.word, w.pos, w.type, c.pos_mode, c.pad_id, w.emb_ln_g, w.emb_ln_b,
                           c.emb_ln_eps, nullptr, p.emb16.hi, p.emb16.lo, dt, st));
    MER_TRY(gemm(st, dt, c.tf.passes, (int)M, D, c.emb_dim, p.emb16, c.emb_dim, w.emb_proj_w, w.emb_proj_b, MER_ACT_NONE, nullptr, 0,
                 hs.at(0), D, p.tf.cur16, D));
  } else
  MER_TRY(mer_bert_embed(ids, token_type, B, T, D, w.word, w.pos, w.type, c.pos_mode, c.pad_id, w.emb_ln_g, w.emb_ln_b,
                         c.emb_ln_eps, hs.at(0), p.tf.cur16.hi, p.tf.cur16.lo, dt, st));
  MER_TRY(tf_forward(st, c.tf, h->layers.data(), B, T, hs, p.tf, lengths));
  return last4_pool(st, hs, c.tf.layers, M, D, frames, seg_start, seg_len, nseg, pooled);
}

// ABI self-description: lets a binding check its struct layouts against the library it loaded.
extern "C" int mer_abi_sizeof(const char* name) {
  if (!name) return MER_EINVAL;
#define MER_SZ(T) if (strcmp(name, #T) == 0) return (int)sizeof(T)
  MER_SZ(mer_gemm16_args);
  MER_SZ(mer_w16);
  MER_SZ(mer_tf_layer);
  MER_SZ(mer_tf_config);
  MER_SZ(mer_hubert_config);
  MER_SZ(mer_hubert_weights);
  MER_SZ(mer_vit_config);
  MER_SZ(mer_vit_weights);
  MER_SZ(mer_bert_config);
  MER_SZ(mer_bert_weights);
  MER_SZ(mer_videomae_config);
  MER_SZ(mer_videomae_weights);
#undef MER_SZ
  return MER_EINVAL;
}
