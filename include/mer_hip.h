/*
 * mer_hip.h — C ABI of libmer_hip.so, the MI355X (gfx950 / CDNA4) implementation of the MERTools
 * feature-extraction + fusion hot path.
 *
 * The reference (zeroQiaoba/MERTools) is pure Python and has no FFI of its own: its "operator
 * API" for this path is the call signature of a HuggingFace model object at four call sites
 * (SURVEY.md §8b).  Each encoder-level entry point below names the reference call it replaces;
 * the op-level entry points are the building blocks those are made of and exist so every kernel
 * can be parity-tested on its own through the same ABI.
 *
 * Conventions
 *   - every pointer named d_* or documented "device" is a HIP device pointer; the library never
 *     allocates, frees or copies caller memory.  Scratch comes from a caller-provided workspace.
 *   - all shapes are explicit ints, row-major, strides in ELEMENTS;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream;
 *   - return value: 0 = MER_OK, <0 = error (mer_last_error() gives a thread-local message);
 *     no C++ exceptions cross this boundary;
 *   - handles are immutable after create and may be used concurrently from several host threads
 *     provided each call uses its own workspace and stream.
 *   - "16-bit planes": an fp32 tensor x is carried as hi = rn16(x) and optionally
 *     lo = rn16(x - hi) (MER_DT_F16 or MER_DT_BF16).  A GEMM with both operands in two planes
 *     runs three MFMA passes (hi*hi + hi*lo + lo*hi) and is fp32-grade accurate; with one plane
 *     it is a plain fp16/bf16 MFMA GEMM with fp32 accumulation.
 */
#ifndef MER_HIP_H
#define MER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mer_stream_t; /* hipStream_t */

enum { MER_OK = 0, MER_EINVAL = -1, MER_ESHAPE = -2, MER_ELAUNCH = -3, MER_ENOMEM = -4, MER_EUNSUPPORTED = -5 };
enum { MER_DT_F16 = 0, MER_DT_BF16 = 1 };
enum { MER_ACT_NONE = 0, MER_ACT_GELU = 1, MER_ACT_QUICK_GELU = 2, MER_ACT_RELU = 3,
       MER_ACT_GELU_TANH = 4 /* HF "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) — ALBERT */ };

const char* mer_version(void);
const char* mer_last_error(void);
/* Hardware the library was built for ("gfx950"). */
const char* mer_target_arch(void);

/* Per-launch timing for the roofline report: while enabled, instrumented launches are bracketed
 * by hipEvents on their own stream.  mer_prof_report synchronises the device, writes a JSON array
 * [{"name","calls","ms","flops","bytes"}] aggregated per kernel and clears the records. */
/* sizeof() of a struct of this header by name ("mer_gemm16_args", ...) so bindings can verify layouts. */
int mer_abi_sizeof(const char* name);
/* Debug switches — process-global, NOT thread-safe (set them while no forward is in flight); nothing on the product path needs them:
 * "gemm_glds" 1 = global->LDS DMA loader (default), 0 = register-staged; "gemm_generic_epi" 1 = every GEMM takes the generic
 * epilogue (bit-equality tests of the specialised ones); "gemm_dbg_skip" 1 / 2 = skip the epilogue's stores / the whole epilogue
 * (timing decomposition), + 8 = no free stagger of the persistent fp32 + residual launches (A/B); "gemm_stamp" 1 = s_memtime-instrumented kernels writing into mer_set_debug_buffer's buffer. */
int mer_set_option(const char* name, int value);
/* Current value of a mer_set_option switch (so that a caller that flips one can put back what it found, e.g. a user's
 * MER_OPTIONS=gemm_persist=0 kill-switch); MER_EINVAL for an unknown name.  "gemm_persist": 0 = tile kernels only, 1 = the persistent
 * kernel where it applies (default); "gemm_tm": 3 / 4 = forced tile rows / 64 of the persistent kernel. */
int mer_get_option(const char* name, int* value);
/* Kernel-phase timing for tuning: when non-NULL, every mer_gemm16 workgroup writes 4 s_memtime stamps (start, first
 * slab ready, K loop done, end) at buffer[4*workgroup ..], per-phase counters at [4*W + 16*workgroup ..] (gemm_stamp builds)
 * and (XCC id << 32 | HW_ID) of the CU it ran on at [20*W + workgroup], W = workgroups of the launch: the caller
 * provides 21*W 8-byte words. NULL disables. */
int mer_set_debug_buffer(void* device_u64_buffer);
int mer_prof_enable(int on);
int mer_prof_report(char* buf, int buflen);

/* ------------------------------------------------------------------------------------------ */
/* Op level                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* C = epilogue(A * W^T).  A: [M,K] 16-bit planes, W: [N,K] 16-bit planes (the torch Linear
 * layout), fp32 accumulate.  epilogue: v = acc + bias[n]; v = act(v); v += residual[m,n];
 * stores fp32 (c32) and/or 16-bit planes (c16_hi / c16_lo).
 * Implicit-im2col row mapping for strided Conv1d over a channels-last tensor
 * (HF:hubert/modeling_hubert.py:106-124 conv layers, :45-92 positional conv):
 *   address(A row m) = a + (m / a_rows_per_batch) * a_batch_stride + (m % a_rows_per_batch) * lda
 * (a_rows_per_batch <= 0 -> plain row-major).
 * Batched form: z in [0,nbatch); zo = z / nb_inner, zi = z % nb_inner;
 *   A += zo*a_so + zi*a_si; W += zi*w_si; bias += zi*bias_si;
 *   {c32,c16,residual} += zo*c_so + zi*c_si.
 * Requirements: K % 8 == 0, lda % 8 == 0, ldw % 8 == 0, all strides % 8 == 0 (16-byte loads).
 */
typedef struct {
  int M, N, K;
  int dtype;
  const void* a_hi; const void* a_lo; long long lda;
  int a_rows_per_batch; long long a_batch_stride;
  const void* w_hi; const void* w_lo; long long ldw;
  const float* bias;
  int act;
  const float* residual; long long ldr;
  float* c32; long long ldc32;
  void* c16_hi; void* c16_lo; long long ldc16;
  int nbatch, nb_inner;
  long long a_so, a_si, w_si, bias_si, c_so, c_si;
  int passes;   /* 1 = a_hi*w_hi; 2 = + a_hi*w_lo (needs w_lo); 3 = + a_lo*w_hi (needs a_lo too);
                 * 4 = a_hi*w_hi + bf8(a_hi)*mxfp4(w - w_hi): the weight residual as an MX-scaled fp4 plane (w_mx, from
                 * mer_mx_pack) through v_mfma_scale_f32_16x16x128_f8f6f4, 1/4 of an f16 pass; f16, 256x256 tile,
                 * K % 128 == 0, no batching — other shapes run passes=2 with w_lo;
                 * 6 = a_hi*w_hi + a_lo*w_hi (needs a_lo, no w_lo): the ACTIVATION split alone — two MFMA passes; the weight residual is
                 * then applied through a bias table (bias_seg_rows, mer_seq_bias) by the caller */
  int tile;     /* 0 = auto; 1 = 128x128; 2 = 128x64 (narrow N); 3 = 256x256 (8 waves, 1 workgroup/CU) */
  /* headmajor_T > 0: the 16-bit output is written head-major for attention instead of row-major:
   * element (row = b*T + t, col = which*64*H + h*64 + d) goes to c16[which][b][h][t][d] (contiguous [T,64] per head);
   * needs M % T == 0, N % (64*H) == 0, c16_hi, no batching. */
  int headmajor_T, headmajor_H;
  const void* w_mx;   /* passes == 4: packed MX-fp4 residual plane of W (mer_mx_pack), 16-byte aligned */
  /* optional pre-blocked copies of w_hi / w_lo (mer_w_block_pack): used instead of the row-major planes by the
   * 256x256 LDS-DMA kernels (a DMA piece becomes 1 KiB contiguous); NULL = not available.  w_lo_blk is only
   * needed for passes 2 / 3. */
  const void* w_hi_blk; const void* w_lo_blk;
  /* optional pre-blocked, ROW-PERMUTED copies of w_hi (mer_w_block_pack_p) for the persistent 256x256 one-pass kernel (register-direct
   * epilogue, csrc/gemm16p_impl.h): w_hi_blkp = layout 0, read when the output is ONE 16-bit plane; w_hi_blkq = layout 1, read when
   * the output is fp32 (+ residual).  Used when passes == 1, nbatch <= 1, N % 256 == 0, K % 32 == 0, K >= 256; NULL = not available. */
  const void* w_hi_blkp; const void* w_hi_blkq;
  /* bias_seg_rows > 0: `bias` is a TABLE [ceil(M / bias_seg_rows), bias_ld] (fp32) and output row m takes row m / bias_seg_rows of it
   * (mer_seq_bias: one correction row per sequence); 0: `bias` is a vector [N].  Not with batching. */
  int bias_seg_rows; long long bias_ld;
} mer_gemm16_args;
int mer_gemm16(const mer_gemm16_args* args, mer_stream_t stream);

/* Per-SEQUENCE weight-residual correction (precision "mean"): table[s, n] = bias[n] + mean_{sampled rows of sequence s}(A)[k] * w_lo[n, k],
 * w_lo = the 16-bit plane of W - f16(W) ([N, K], row stride ldw).  The rounding error of a weight matrix is the same perturbation for
 * every token, so what it does to the features goes almost entirely through the mean activation of the sequence
 * (tests/studies/mean_correction.py) — i.e. it is a bias, one row per sequence.  Computed
 * for the ceil(M / seg_rows) sequences of seg_rows consecutive rows of the A plane (rows addressed like mer_gemm16's A operand).  The
 * sample of a sequence is rows h, h + s, h + 2 s, ... below its valid length (valid_rows[s], or seg_rows), s the largest power of two
 * that leaves at least 16 samples, h = s / 2; sums are exact 64-bit fixed-point integers, one owner per element (no atomics): the row
 * of a sequence depends on that sequence alone, bit for bit — whatever else is in the batch.  The one-pass GEMM that follows takes the
 * table through mer_gemm16_args.bias / bias_seg_rows / bias_ld.  Columns below n_first (a multiple of 16) get the plain bias (the
 * Q | K columns of a fused QKV weight need no correction).  scratch: device, mer_seq_bias_scratch_bytes(nseq, K) bytes, 16-byte
 * aligned (the 16-bit mean plane); table: device fp32 [nseq, ldt].  Two small launches on `stream`. */
long long mer_seq_bias_scratch_bytes(int nseq, int K);
int mer_seq_bias(const void* a, int dtype, long long lda, int a_rows_per_batch, long long a_batch_stride, int M, int K,
                 int seg_rows, const int* valid_rows, const void* w_lo, long long ldw, const float* bias, int N, int n_first,
                 void* scratch, float* table, long long ldt, mer_stream_t stream);

/* Pre-blocked weight plane: a DEVICE 16-bit plane w [N, K] (row stride ldw, K % 32 == 0) is re-laid as
 * [ceil(N/256)][K/32] blocks of 16 KB, each the exact LDS image (256 rows x 64 B, 16-byte chunks XOR-swizzled) of
 * that (column tile, k-slab) — rows beyond N repeat row N-1.  out: DEVICE buffer of mer_w_block_bytes(N, K) bytes.
 * Row-range views stay addressable: the block of column tile t starts at byte t * 256 * K * 2. */
long long mer_w_block_bytes(int N, int K);
int mer_w_block_pack(const void* w, long long ldw, int N, int K, void* out, mer_stream_t stream);
/* The same blocks with the rows of every 128-row group permuted so that the eight accumulators a lane of the 16x16 MFMA holds for
 * one output row (eight column tiles of the wave's 128 columns) are CONSECUTIVE columns and the persistent kernel's epilogue stores
 * whole 128-byte lines straight from registers (no LDS transposition): block row 16 q + i (i < 16, q < 8) holds plane row
 *   layout 0 (16-bit outputs: one 16-byte store per lane and row):  8 i + q
 *   layout 1 (fp32 outputs: two 16-byte stores per lane and row):   64 (q / 4) + 4 i + q % 4.
 * N % 256 == 0.  Same size as mer_w_block_bytes(N, K). */
int mer_w_block_pack_p(const void* w, long long ldw, int N, int K, int layout, void* out, mer_stream_t stream);

/* Host-side packer of the MX correction plane.  w_res: HOST fp32 [N, K] (row stride ldw) = W - f16(W);
 * out: HOST buffer of mer_mx_packed_bytes(N, K) bytes (then copied to the device once).  Per (256-column tile,
 * 32-deep k-slab) one 5 KB block: four 16-column tiles of e2m1 codes in the lane order of the MFMA B operand
 * + the E8M0 scales (one per column and 32 k-slots) of the slab's 128-k group; K % 128 == 0.  Returns 0 bytes / MER_ESHAPE for unsupported shapes. */
long long mer_mx_packed_bytes(int N, int K);
int mer_mx_pack(const float* w_res, long long ldw, int N, int K, void* out);

/* Exact-fp32 GEMM on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain): C = act(A*W^T + bias).
 * A [M,K] fp32 (lda), W [N,K] fp32 (ldw), C [M,N] fp32 (ldc).  Used by the fusion classifier
 * (reference: MERBench/toolkit/models/modules/encoder.py:30-41) and as a tight-parity check.
 * trans_a: A is given as [K,M] (A^T), trans_w: W is given as [K,N]; both needed by the backward
 * pass.  accumulate != 0: C += result. */
int mer_gemm32(const float* a, long long lda, int trans_a, const float* w, long long ldw, int trans_w,
               const float* bias, int act, float* c, long long ldc, int accumulate,
               int M, int N, int K, mer_stream_t stream);

/* y = act(LayerNorm(x) * gamma + beta) over the last dim D (biased variance, two-pass fp32) —
 * torch.nn.LayerNorm semantics.  x: fp32 rows with stride ldx.  Optional outputs: fp32 (out32,
 * ld32) and 16-bit planes (out16_hi, out16_lo, ld16).  D % 4 == 0. */
int mer_layernorm(const float* x, long long ldx, const float* gamma, const float* beta, float eps,
                  int M, int D, int act, float* out32, long long ld32,
                  void* out16_hi, void* out16_lo, long long ld16, int dtype, mer_stream_t stream);

/* Multi-head self-attention, softmax(q k^T * scale) v, no causal mask; optional per-sequence key
 * length (keys >= kv_len[b] are masked out; rows >= kv_len[b] produce unspecified output).
 * q,k,v: 16-bit [B*T, ld] with head h at columns [h*64, h*64+64) of each pointer; head_dim = 64.
 * out: 16-bit planes [B*T, ldo].  T <= 512 uses the single-pass kernel (K and V^T of one head
 * resident in LDS); larger T uses the streaming (online-softmax) kernel.
 * (HF:hubert/modeling_hubert.py:236-259 eager_attention_forward; same math in CLIP/RoBERTa.) */
int mer_attention(const void* q, const void* k, const void* v, long long ld,
                  void* out_hi, void* out_lo, long long ldo,
                  int B, int T, int H, float scale, const int* kv_len, int dtype, mer_stream_t stream);

/* The same attention with fp32 q | k | v (the fp32 output of the QKV GEMM) on the exact fp32 MFMA: what the three-pass ("accurate")
 * preset runs, so that no operand of a block is a single 16-bit plane (HF:hubert/modeling_hubert.py:236-259 in fp32).  q, k, v:
 * device fp32, row stride ld floats (ld % 4 == 0, 16-byte aligned), head h at column 64 h; out_hi / out_lo: 16-bit hi / lo planes of
 * the context rows (out_lo may be NULL).  Any T; keys >= kv_len[b] are masked. */
int mer_attention_f32(const float* q, const float* k, const float* v, long long ld, void* out_hi, void* out_lo,
                      long long ldo, int B, int T, int H, float scale, const int* kv_len, int dtype, mer_stream_t stream);

/* Same attention on head-major operands: q, k, v are [B][H][T][64] planes (as written by mer_gemm16's head-major
 * output), so every head's K/V tile is one contiguous 128*T-byte stream.  out stays row-major [B*T, ldo]. */
int mer_attention_hm(const void* q, const void* k, const void* v, void* out_hi, void* out_lo, long long ldo,
                     int B, int T, int H, float scale, const int* kv_len, int dtype, mer_stream_t stream);

/* Attention for ONE query per sequence (the [CLS] row of a ViT's last block, which is all get_image_features reads):
 * q: 16-bit [B, ldq] (head h at columns [64h, 64h+64)), k / v: 16-bit [B*T, ld] as in mer_attention, out: 16-bit planes [B, ldo];
 * fp32 softmax and accumulation; T <= 584. */
int mer_attention_cls(const void* q, long long ldq, const void* k, const void* v, long long ld, void* out_hi, void* out_lo,
                      long long ldo, int B, int T, int H, float scale, const int* kv_len, int dtype, mer_stream_t stream);

/* fp32 -> 16-bit planes (lo may be NULL). n elements. */
int mer_split16(const float* x, void* hi, void* lo, long long n, int dtype, mer_stream_t stream);

/* HuBERT / wav2vec2 layer-0 feature extractor for feat_extract_norm == "group":
 * Conv1d(1->C, k, stride, no bias) -> GroupNorm(C groups == per-channel over time, eps 1e-5,
 * affine) -> GELU, written channels-last as 16-bit planes [B, T0, C].
 * (HF:hubert/modeling_hubert.py:154-175.)  stats: device scratch of 2*B*C doubles (no initialisation needed).
 * For k <= 10, C % 8 == 0, C >= 65 the statistics come from the clip's input autocorrelation (sum y^2 = w'Rw, fp64) and the
 * conv is computed once; other shapes take a two-pass kernel. */
int mer_hubert_conv0_gn(const float* wav, int B, int L, const float* w /*[C,k]*/, int C, int k, int stride,
                        const float* gamma, const float* beta, float eps, double* stats,
                        void* out_hi, void* out_lo, int dtype, mer_stream_t stream);

/* Ragged batches (rows zero-padded to a common L): valid_frames[b] (device int32, NULL = all T0) = the conv0 output frames
 * of row b that come from its own samples; the GroupNorm statistics run over those only, which is what the reference's
 * batch-of-one forward (extract_audio_huggingface.py:93-100, no padding, no mask) computes for that clip. */
int mer_hubert_conv0_gn_ragged(const float* wav, int B, int L, const float* w /*[C,k]*/, int C, int k, int stride,
                               const float* gamma, const float* beta, float eps, double* stats,
                               void* out_hi, void* out_lo, int dtype, const int* valid_frames, mer_stream_t stream);
/* valid_samples[b] (device int32) -> frames after conv 0 (t0_len) and after the whole valid-conv stack (tn_len), device
 * int32 [B] each; kernels / strides: HOST int [n_conv]. */
int mer_hubert_valid_frames(const int* valid_samples, int B, int L, int n_conv, const int* kernels, const int* strides,
                            int* t0_len, int* tn_len, mer_stream_t stream);
/* The same after EVERY conv layer: lens = device int32 [n_conv, B] (row 0 == t0_len, row n_conv - 1 == tn_len). */
int mer_hubert_valid_frames_all(const int* valid_samples, int B, int L, int n_conv, const int* kernels, const int* strides,
                                int* lens, mer_stream_t stream);

/* Layer-0 conv for feat_extract_norm == "layer" (HuBERT-large / wav2vec2-large, HF:hubert/modeling_hubert.py:127-151):
 * out[b,t,c] = bias[c] + conv, fp32 channels-last [B,T0,C]; the per-frame LayerNorm + GELU is mer_layernorm(act=GELU). */
int mer_hubert_conv0_plain(const float* wav, int B, int L, const float* w, const float* bias, int C, int k, int stride,
                           float* out, mer_stream_t stream);

/* Channels-last hidden [B,T,D] fp32 -> zero-padded, group-major 16-bit planes [B, G, T+K, D/G]
 * with x[b,t,g*Dg+c] at row t + K/2, so that the grouped positional Conv1d becomes one batched
 * implicit-im2col GEMM (HF:hubert/modeling_hubert.py:45-92). */
int mer_posconv_pack(const float* x, int B, int T, int D, int G, int K, void* out_hi, void* out_lo,
                     int dtype, mer_stream_t stream);
/* Ragged batches: frames t >= valid_frames[b] are packed as zeros, so the conv's padding region of a short row is exactly
 * the zero padding of its batch-of-one forward. */
int mer_posconv_pack_ragged(const float* x, int B, int T, int D, int G, int K, void* out_hi, void* out_lo,
                            int dtype, const int* valid_frames, mer_stream_t stream);

/* ViT patchify: pixel_values fp32 [N,3,H,W] -> 16-bit planes [N*(H/P)*(W/P), ceil8(3*P*P)] with the
 * (c, i, j) ordering of a flattened Conv2d weight (HF:clip/modeling_clip.py:138-217); rows are zero-padded to a
 * multiple of 8 columns (CLIP-L/14: 588 -> 592) and the weight must be padded the same way. */
int mer_vit_patchify(const float* pixels, int N, int C, int H, int W, int P, void* out_hi, void* out_lo,
                     int dtype, mer_stream_t stream);

/* VideoMAE tubelet patchify: pixel_values fp32 [B,F,C,H,W] -> 16-bit planes [B*(F/ts)*(H/P)*(W/P), C*ts*P*P] in the
 * (c, dt, i, j) order of the flattened Conv3d weight (HF:videomae/modeling_videomae.py:119-176). */
int mer_video_patchify(const float* pixels, int B, int F, int C, int H, int W, int P, int ts, void* out_hi, void* out_lo,
                       int dtype, mer_stream_t stream);
/* x[r,:] += pos[r % P,:] in place (fixed sin-cos position table). */
int mer_add_pos(float* x, const float* pos, long long rows, int P, int D, mer_stream_t stream);

/* ViT token assembly + optional LayerNorm: tok[n,0,:] = cls + pos[0]; tok[n,1+p,:] = patch[n,p,:]
 * + pos[1+p]; y = LN(tok) if gamma != NULL else tok.  out32 [N*(1+P), D]; optional 16-bit. */
int mer_vit_assemble(const float* patch, const float* cls, const float* pos, const float* gamma,
                     const float* beta, float eps, int N, int P, int D, float* out32,
                     void* out16_hi, void* out16_lo, int dtype, mer_stream_t stream);

/* BERT/RoBERTa embeddings + LayerNorm: x = word[ids] + pos[position] + type[tt]; y = LN(x).
 * ids: int64 [B,T]; token_type may be NULL (-> 0).  pos_mode 0: position = t (BERT);
 * pos_mode 1: RoBERTa, position = pad_id + cumsum(ids != pad_id) for non-pad tokens, pad_id
 * otherwise (HF:roberta/modeling_roberta.py:56-155). */
int mer_bert_embed(const int64_t* ids, const int64_t* token_type, int B, int T, int D,
                   const float* word, const float* pos, const float* type, int pos_mode, int pad_id,
                   const float* gamma, const float* beta, float eps, float* out32,
                   void* out16_hi, void* out16_lo, int dtype, mer_stream_t stream);

/* out_frames[r,:] = h0[r,:] (+ h1 + h2 + h3)   (sequential fp32 adds, the order of
 * torch.stack(hs)[[-4,-3,-2,-1]].sum(0): extract_audio_huggingface.py:98);
 * out_pool[i,:]  = mean over rows [seg_start[i], seg_start[i] + seg_len[i]) of that sum
 *                  (extract_audio_huggingface.py:104-108 / extract_text_huggingface.py:243-249).
 * h1..h3, out_frames, out_pool may be NULL.  seg_* are device int32 arrays of nseg entries. */
int mer_sum_pool(const float* h0, const float* h1, const float* h2, const float* h3, long long M, int D,
                 float* out_frames, const int* seg_start, const int* seg_len, int nseg, float* out_pool,
                 mer_stream_t stream);

/* ---- fusion classifier pieces (MERBench/toolkit/models/attention.py:36-57, modules/encoder.py:30-41,
 *      utils/loss.py:5-28, main-release.py:50-66).  Linear layers run on mer_gemm32; these are the
 *      element-wise / tiny-reduction kernels around them, forward and backward. ------------------- */
int mer_relu_bwd(const float* dy, const float* y, float* dz, long long n, mer_stream_t stream);          /* dz = y>0 ? dy : 0 */
int mer_colsum(const float* x, int M, int N, long long ldx, float* out, int accumulate, mer_stream_t stream); /* bias grad */
int mer_dropout(const float* x, const uint8_t* keep, float scale, float* out, long long n, mer_stream_t stream);
/* fused[b,j] = sum_e h[b,e*H+j]*att[b,e]  (h = cat of the E modality encoders' outputs, att NOT softmaxed) */
int mer_fuse_fwd(const float* h, const float* att, float* out, int B, int H, int E, mer_stream_t stream);
int mer_fuse_bwd(const float* dout, const float* h, const float* att, float* dh, float* datt, int B, int H, int E,
                 mer_stream_t stream);
/* CELoss = NLL(log_softmax(logits,1), target, 'sum')/B; probs [B,C] and row_scratch [B] are device scratch
 * kept for the backward; loss is a device scalar. */
int mer_ce_loss(const float* logits, const int64_t* target, int B, int C, float* probs, float* row_scratch, float* loss,
                mer_stream_t stream);
int mer_ce_loss_bwd(const float* probs, const int64_t* target, const float* gout, float* dlogits, int B, int C,
                    mer_stream_t stream);
/* MSELoss = sum((pred-target)^2)/B on [B] vectors */
int mer_mse_loss(const float* pred, const float* target, int B, float* row_scratch, float* loss, mer_stream_t stream);
int mer_mse_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred, int B, mer_stream_t stream);
/* torch.optim.Adam step on one flat tensor (L2 weight decay added to the gradient; clip_value > 0 applies
 * clip_grad_value_ first); step counts from 1. */
int mer_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float clip_value, mer_stream_t stream);

/* hipGraph-replayable Adam: reads the 0-based step count from device memory (uses step+1); call mer_inc_i32 on the
 * counter once per optimiser step after all tensors were updated. */
int mer_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int* step_dev, float clip_value, mer_stream_t stream);
int mer_inc_i32(int* x, mer_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Encoder level                                                                               */
/* ------------------------------------------------------------------------------------------ */

typedef struct { const void* hi; const void* lo; const void* mx;   /* 16-bit weight planes [N,K] (+ MX residual plane for passes == 4, may be null) */
                 const void* hi_blk; const void* lo_blk;   /* optional pre-blocked copies of hi / lo (mer_w_block_pack), may be null */
                 const void* hi_blkp; const void* hi_blkq; } mer_w16;   /* optional row-permuted pre-blocked copies of hi (mer_w_block_pack_p layout 0 / 1), may be null */

/* One transformer block (HuBERT / wav2vec2 / CLIP-ViT / VideoMAE / BERT / RoBERTa).
 * wqkv = cat(q,k,v) rows [3D, D]; bqkv fp32 [3D] (zeros where the model has no bias). */
typedef struct {
  mer_w16 wqkv; const float* bqkv;
  mer_w16 wo; const float* bo;
  const float* ln1_g; const float* ln1_b;
  mer_w16 w1; const float* b1;
  mer_w16 w2; const float* b2;
  const float* ln2_g; const float* ln2_b;
  /* optional additive attention-score bias (all NULL for the models above):
   *   attn_bias  fp32 [H, T, ceil4(T)] baked for a fixed T (BEiT / data2vec-vision relative position bias); when NULL the
   *              table passed to the forward call is used (WavLM: one table per T shared by every layer);
   *   gru_w [8,64], gru_b [8], gru_const [H]: WavLM's per-layer gate on that table (mer_wavlm_gate). */
  const float* attn_bias;
  const float* gru_w; const float* gru_b; const float* gru_const;
} mer_tf_layer;

typedef struct {
  int hidden, heads, ffn, layers;
  int pre_ln;     /* 0: post-LN blocks (HuBERT-base, BERT, RoBERTa); 1: pre-LN (CLIP, VideoMAE, HuBERT-large) */
  int act;        /* MER_ACT_GELU | MER_ACT_QUICK_GELU */
  float ln_eps;
  int dtype;      /* MER_DT_F16 | MER_DT_BF16 */
  int passes;     /* GEMM passes inside the blocks: 1, 2 (weights split), 3 (both split), 4 (MX-corrected, see mer_gemm16) or
                   * 5 (one pass + the per-sequence weight-residual correction table, see mer_seq_bias; needs the `lo` planes) or
                   * 6 (5 with every GEMM input carried as hi + lo activation planes: a_hi*w_hi + a_lo*w_hi + the table — two MFMA
                   * passes; the preset for checkpoints whose activations do not fit one 16-bit plane) */
  int gated_rel_pos;  /* 1: WavLM — every layer carries gru_* and the forward call must be given the position-bias table */
  int ffn_swiglu;     /* 1: DINOv2-giant SwiGLU feed-forward (HF:dinov2/modeling_dinov2.py Dinov2SwiGLUFFN): w1 = weights_in [2*ffn, D],
                       * h = silu(y[:, :ffn]) * y[:, ffn:], w2 = weights_out [D, ffn]; `ffn` is the post-gate width, `act` is ignored */
  int mx_skip;        /* passes == 4 / 5: GEMMs that run WITHOUT the weight-residual correction (plain one-pass f16) because
                       * their rounding error does not reach the saved features (tests/studies/mx_selective.py):
                       * bit 0 = the Q and K projections (their error only perturbs softmax logits), bit 1 = FFN fc1, bit 2 = FFN fc2 */
  int attn_f32;       /* passes == 6: 1 = attention on fp32 q | k | v (mer_attention_f32: the QKV GEMM writes fp32) as passes == 3 always does;
                       * 0 = the 16-bit attention kernels */
} mer_tf_config;

/* ---- HuBERT / wav2vec2 audio encoder --------------------------------------------------------
 * Replaces `model(input_values, output_hidden_states=True).hidden_states` at
 * MERBench/feature_extraction/audio/extract_audio_huggingface.py:97 and the last-4 sum / mean
 * of :98-108. */
#define MER_MAX_CONV 8
#define MER_MAX_POS 8
typedef struct {
  mer_tf_config tf;
  int n_conv; int conv_dim; int conv_kernel[MER_MAX_CONV]; int conv_stride[MER_MAX_CONV];
  int feat_norm_group;      /* 1: GroupNorm on conv0 only ("group"); 0: LayerNorm after every conv ("layer") */
  int conv_bias;
  int feat_proj_layer_norm;
  int pos_k, pos_groups;
  int stable_layer_norm;    /* 1: HubertEncoderStableLayerNorm (large) */
  int conv_passes;          /* GEMM passes in the conv stack + projection + positional conv: 1 .. 5 (as mer_tf_config.passes) */
  int pos_layers;           /* 0: HuBERT / wav2vec2 — one weight-normed conv, x + GELU(conv(x));
                             * n >= 1: data2vec-audio (HF:data2vec/modeling_data2vec_audio.py Data2VecAudioPositionalConvEmbedding) —
                             * n x [grouped conv (kernel pos_k, padding pos_k/2) -> LayerNorm without affine (eps 1e-5) -> GELU],
                             * x + that stack; weights in pos_ws / pos_bs */
} mer_hubert_config;

typedef struct {
  const float* conv0_w;                 /* [C, k0] fp32 */
  const float* conv_norm_g[MER_MAX_CONV]; const float* conv_norm_b[MER_MAX_CONV];
  mer_w16 conv_w[MER_MAX_CONV];         /* i>=1: [C, k_i*C] with column index kk*C + ci */
  const float* conv_b[MER_MAX_CONV];
  const float* fp_ln_g; const float* fp_ln_b;
  mer_w16 fp_w; const float* fp_b;      /* [D, C] */
  mer_w16 pos_w; const float* pos_b;    /* [G, D/G, pos_k * D/G] (weight-norm folded), column kk*Dg + ci */
  const float* enc_ln_g; const float* enc_ln_b;
  const mer_tf_layer* layers;           /* host array of tf.layers entries */
  mer_w16 pos_ws[MER_MAX_POS]; const float* pos_bs[MER_MAX_POS];   /* pos_layers >= 1: per layer, same layout as pos_w / pos_b */
} mer_hubert_weights;

typedef struct mer_hubert mer_hubert;
int mer_hubert_create(const mer_hubert_config* cfg, const mer_hubert_weights* w, mer_hubert** out);
void mer_hubert_destroy(mer_hubert* h);
/* frames after the conv stack for L input samples (HF _get_feat_extract_output_lengths) */
int mer_hubert_out_frames(const mer_hubert* h, int L);
long long mer_hubert_workspace_bytes(const mer_hubert* h, int B, int L, int want_hidden_states);
/* wav: device fp32 [B,L].  Outputs (each may be NULL):
 *   hidden_states fp32 [layers+1, B, T, D]    (the HF tuple, stacked)
 *   frames        fp32 [B*T, D]               sum of the last four hidden states
 *   pooled        fp32 [nseg, D]              mean of `frames` rows over each segment; seg_start /
 *                 seg_len are device int32 [nseg] in units of rows of the flattened [B*T] axis
 *                 (one segment per clip; a >10 s clip split into chunks is one segment spanning
 *                 its chunks, as in the reference). */
int mer_hubert_forward(const mer_hubert* h, const float* wav, int B, int L,
                       void* workspace, long long workspace_bytes,
                       float* hidden_states, float* frames,
                       const int* seg_start, const int* seg_len, int nseg, float* pooled,
                       mer_stream_t stream);
/* WavLM (tf.gated_rel_pos = 1): same, plus the relative position bias table fp32 [H, T, ldb] for T = mer_hubert_out_frames(L)
 * (WavLMAttention.compute_bias of layer 0's rel_attn_embed; rows padded to ldb % 4 == 0).  The table depends only on T and
 * the checkpoint, so the caller builds it once per input length. */
int mer_hubert_forward_bias(const mer_hubert* h, const float* wav, int B, int L,
                            void* workspace, long long workspace_bytes,
                            float* hidden_states, float* frames,
                            const int* seg_start, const int* seg_len, int nseg, float* pooled,
                            const float* pos_bias, long long ldb, mer_stream_t stream);
/* Ragged batch: rows are clips of different lengths, zero-padded to L; valid_samples (device int32 [B], NULL = all L) holds
 * each row's own sample count.  Every frame whose receptive field lies inside the clip then equals the batch-of-one result
 * (GroupNorm statistics over valid frames, zeros past the clip for the positional conv, keys past the clip masked);
 * frames past a row's length are unspecified — select the valid ones with seg_start / seg_len.  The reference handles any
 * length only because it runs batch 1 (extract_audio_huggingface.py:93-100). */
int mer_hubert_forward_ragged(const mer_hubert* h, const float* wav, int B, int L, const int* valid_samples,
                              void* workspace, long long workspace_bytes,
                              float* hidden_states, float* frames,
                              const int* seg_start, const int* seg_len, int nseg, float* pooled,
                              const float* pos_bias, long long ldb, mer_stream_t stream);

/* ---- CLIP vision tower ----------------------------------------------------------------------
 * Replaces `model.get_image_features(pixel_values)` at
 * MERBench/feature_extraction/visual/extract_vision_huggingface.py:118-122 and the frame mean of
 * :183-189. */
typedef struct {
  mer_tf_config tf;
  int image_size, patch_size, channels, proj_dim;
  int variant;   /* 0 = CLIP vision tower (no patch bias, pre_layrnorm, post_layernorm(CLS) -> projection);
                  * 1 = DINOv2 (HF:dinov2/modeling_dinov2.py): patch bias, no embedding LN, layer scale folded into wo/w2 by
                  *     the host, image_features = SUM over the 1+P tokens of the last residual stream (the reference's
                  *     `torch.stack(hidden_states)[-1].sum(dim=1)`, extract_vision_huggingface.py:142); proj_dim == hidden */
} mer_vit_config;
typedef struct {
  mer_w16 patch_w;                      /* [D, ceil8(C*P*P)] (no bias; zero-padded columns) */
  const float* cls; const float* pos;   /* [D], [1+P, D] */
  const float* pre_ln_g; const float* pre_ln_b;
  const float* post_ln_g; const float* post_ln_b;
  mer_w16 proj_w;                       /* [proj_dim, D] (no bias) */
  const mer_tf_layer* layers;
  const float* patch_b;                 /* variant 1: [D] patch-embedding bias (NULL for CLIP) */
} mer_vit_weights;
typedef struct mer_vit mer_vit;
int mer_vit_create(const mer_vit_config* cfg, const mer_vit_weights* w, mer_vit** out);
void mer_vit_destroy(mer_vit* h);
long long mer_vit_workspace_bytes(const mer_vit* h, int N);
/* pixels: device fp32 [N,3,S,S].  image_features fp32 [N, proj_dim] (may be NULL);
 * pooled fp32 [nseg, proj_dim] = mean of image_features rows per segment (may be NULL). */
int mer_vit_forward(const mer_vit* h, const float* pixels, int N, void* workspace, long long workspace_bytes,
                    float* image_features, const int* seg_start, const int* seg_len, int nseg, float* pooled,
                    mer_stream_t stream);
/* Same, and additionally copies the last residual stream [N, 1+P, D] (the last entry of HF's `hidden_states`) to
 * `tokens_out` when it is not NULL — what the drop-in `model(..., output_hidden_states=True).hidden_states[-1]` returns. */
int mer_vit_forward_tokens(const mer_vit* h, const float* pixels, int N, void* workspace, long long workspace_bytes,
                           float* image_features, const int* seg_start, const int* seg_len, int nseg, float* pooled,
                           float* tokens_out, mer_stream_t stream);

/* mer_attention with an additive score bias: scores = q k^T * scale + gate[b,h,q] * bias[h,q,k].
 * bias: fp32 [H, T, ldb] (ldb >= T, ldb % 4 == 0, 16-byte aligned; one table for the whole batch); gate: fp32 [B, H, T] or NULL
 * (= 1).  WavLM's gated relative position bias (HF:wavlm/modeling_wavlm.py WavLMAttention) and, ungated, BEiT /
 * data2vec-vision's relative position bias.  T <= 512. */
int mer_attention_bias(const void* q, const void* k, const void* v, long long ld, void* out_hi, void* out_lo,
                       long long ldo, int B, int T, int H, float scale, const int* kv_len, const float* bias,
                       long long ldb, const float* gate, int dtype, mer_stream_t stream);

/* WavLM gate: gate[b,h,t] = ga * (gb * cst[h] - 1) + 2 with (ga, gb) = sigmoid of the two 4-sums of W[8,64] x[m, 64h:64h+64] + b[8],
 * x fp32 [B*T, ldx] = the attention input of the layer (head_dim 64). */
int mer_wavlm_gate(const float* x, long long ldx, const float* w, const float* b, const float* cst, int B, int T, int H,
                   float* gate, mer_stream_t stream);

/* ---- LSTM recurrence for the frame-level fusion encoders (MERBench/toolkit/models/modules/encoder.py:45-72: nn.LSTM, one
 * layer, unidirectional, batch_first; only the final hidden state is used).  All fp32, gate order i,f,g,o as in torch.
 * mer_lstm_fwd: gx [B,T,4H] = X W_ih^T + b_ih + b_hh (one mer_gemm32), w_hh_t [H,4H] = W_hh^T  ->  gates [B,T,4H] (post-
 *   activation), cs [B,T,H], hs [B,T,H] (h_T = hs[:, T-1]).
 * mer_lstm_bwd: dh_last [B,H] = dL/dh_T, w_hh [4H,H]  ->  dA [B,T,4H] = dL/d(pre-activations); then dW_ih = dA^T X,
 *   dW_hh = dA^T [0, hs[:, :-1]], db_ih = db_hh = column sums of dA.   H <= 256, H % 16 == 0. */
int mer_lstm_fwd(const float* gx, const float* w_hh_t, int B, int T, int H, float* gates, float* cs, float* hs,
                 mer_stream_t stream);
int mer_lstm_bwd(const float* dh_last, const float* gates, const float* cs, const float* w_hh, int B, int T, int H,
                 float* dA, mer_stream_t stream);

/* Small fp32 attention (head_dim 64, Tk <= 2048): out[b,tq,h,:] = softmax_k(q[b,tq,h,:]·k[b,k,h,:] * scale [masked k > tq if causal]) v.
 * q [B*Tq, ldq], k / v [B*Tk, ldkv], out [B*Tq, ldo], head h at column 64h.  Decoder-side work of Whisper (two tokens per clip:
 * causal self-attention and cross-attention to the 1500 encoder states; HF:whisper/modeling_whisper.py WhisperAttention). */
int mer_small_attention(const float* q, long long ldq, const float* k, const float* v, long long ldkv, int B, int Tq, int Tk,
                        int H, float scale, int causal, float* out, long long ldo, mer_stream_t stream);

/* ---- host pre-processing on the GPU (what the reference does on the CPU before the H2D copy) ----
 * mer_wave_normalize: Wav2Vec2FeatureExtractor's per-utterance (x - mean) / sqrt(var + 1e-7) (do_normalize != 0) or a plain
 * conversion, from device int16 PCM (x = pcm / 32768, is_int16 != 0) or device fp32; one row per utterance / chunk.
 * mer_image_normalize_u8: device uint8 frames [N,H,W,3] (bgr != 0: OpenCV channel order) -> fp32 [N,3,H,W] RGB,
 * (v / 255 - mean[c]) / std[c]; mean3 / std3 are HOST arrays of three floats (RGB order).  Frames must already have the
 * model's resolution (the resize of CLIPImageProcessor stays on the host: PIL's fixed-point bicubic is not reproduced). */
int mer_wave_normalize(const void* x, int is_int16, long long ldx, int B, int L, int do_normalize, float* out,
                       long long ldo, mer_stream_t stream);
int mer_image_normalize_u8(const unsigned char* frames, int N, int H, int W, int bgr, const float* mean3,
                           const float* std3, float* out, mer_stream_t stream);

/* Pillow-exact bicubic resize + centre crop of 8-bit frames on the GPU (reference a6: processor(images=...) ->
 * CLIPImageProcessor.resize -> PIL Image.resize(BICUBIC) on uint8, then center_crop;
 * MERBench/feature_extraction/visual/extract_vision_huggingface.py:116).  frames: DEVICE u8 [N,H,W,3]; out: DEVICE u8
 * [N,crop_h,crop_w,3] = the window (left, top, crop_w, crop_h) of the resized image.  xb / yb: DEVICE int32 [new, 2]
 * (first input index, tap count) per output column / row; xk / yk: DEVICE int32 [new, ksize] fixed-point coefficients
 * (22 fractional bits) — Pillow's precompute_coeffs + normalize_coeffs_8bpc, built on the host per (input, output) size
 * (mertools_amd/extract/resize.py:pil_coeffs).  y0..y1-1: the input rows the cropped output rows touch; tmp: DEVICE u8
 * [N, y1-y0, crop_w, 3] (the 8-bit image between the two passes, as in Pillow).  Byte-identical to Pillow. */
int mer_image_resize_crop_u8(const unsigned char* frames, int N, int H, int W, int left, int top, int crop_w, int crop_h,
                             const int* xb, const int* xk, int xksize, const int* yb, const int* yk, int yksize, int y0, int y1,
                             unsigned char* tmp, unsigned char* out, mer_stream_t stream);

/* SwiGLU gate: out[m, j] = silu(y[m, j]) * y[m, F + j] for y fp32 [M, 2F] (row stride ldy) -> 16-bit planes [M, F]. */
int mer_swiglu(const float* y, long long ldy, int M, int F, void* out_hi, void* out_lo, int dtype, mer_stream_t stream);

/* out[n, :] = scale * sum over t of x[n, t, :]   (x fp32 [N, T, D]; token sum / mean of a hidden state). */
int mer_token_reduce(const float* x, int N, int T, int D, float scale, float* out, mer_stream_t stream);

/* ---- VideoMAE video encoder ------------------------------------------------------------------
 * Replaces `model(inputs).last_hidden_state` at
 * MERBench/feature_extraction/visual/extract_vision_huggingface.py:155 and the per-segment patch mean of :156-158. */
typedef struct {
  mer_tf_config tf;          /* pre_ln = 1 */
  int image_size, patch_size, channels, num_frames, tubelet_size;
  int final_ln;              /* VideoMAEModel.layernorm present (config.use_mean_pooling == False) */
} mer_videomae_config;
typedef struct {
  mer_w16 patch_w; const float* patch_b;   /* [D, C*ts*P*P], [D] */
  const float* pos;                          /* [num_patches, D] fixed sin-cos table */
  const float* final_ln_g; const float* final_ln_b;
  const mer_tf_layer* layers;
} mer_videomae_weights;
typedef struct mer_videomae mer_videomae;
int mer_videomae_create(const mer_videomae_config* cfg, const mer_videomae_weights* w, mer_videomae** out);
void mer_videomae_destroy(mer_videomae* h);
long long mer_videomae_workspace_bytes(const mer_videomae* h, int B);
/* pixels: device fp32 [B,F,C,S,S].  last_hidden_state fp32 [B*num_patches, D] (may be NULL);
 * pooled fp32 [nseg, D] = mean of last_hidden_state rows per segment (may be NULL). */
int mer_videomae_forward(const mer_videomae* h, const float* pixels, int B, void* workspace, long long workspace_bytes,
                         float* last_hidden_state, const int* seg_start, const int* seg_len, int nseg, float* pooled,
                         mer_stream_t stream);

/* ---- BERT / RoBERTa text encoder ------------------------------------------------------------
 * Replaces `model(**inputs, output_hidden_states=True).hidden_states` at
 * MERBench/feature_extraction/text/extract_text_huggingface.py:225 and the sum/slice/mean of
 * :226-249. */
typedef struct {
  mer_tf_config tf;
  int vocab, max_pos, type_vocab, pad_id, pos_mode;
  float emb_ln_eps;
  int emb_dim;   /* 0 or == tf.hidden: embeddings are hidden-sized (BERT, RoBERTa).  Otherwise the embedding tables and their
                  * LayerNorm are emb_dim wide and emb_proj maps them to tf.hidden before the first block: ELECTRA-small's
                  * `embeddings_project`, ALBERT's `embedding_hidden_mapping_in`; hidden_states[0] is the projected tensor. */
} mer_bert_config;
typedef struct {
  const float* word; const float* pos; const float* type;
  const float* emb_ln_g; const float* emb_ln_b;
  const mer_tf_layer* layers;              /* ALBERT: every entry points at the one shared block */
  mer_w16 emb_proj_w; const float* emb_proj_b;   /* [hidden, emb_dim], [hidden] when emb_dim != hidden */
} mer_bert_weights;
typedef struct mer_bert mer_bert;
int mer_bert_create(const mer_bert_config* cfg, const mer_bert_weights* w, mer_bert** out);
void mer_bert_destroy(mer_bert* h);
long long mer_bert_workspace_bytes(const mer_bert* h, int B, int T, int want_hidden_states);
/* ids: device int64 [B,T] (right-padded with pad_id); token_type may be NULL; lengths: device
 * int32 [B] = number of real tokens per row (keys beyond it are masked; NULL = all T).
 * Outputs as for mer_hubert_forward; seg_start/seg_len select e.g. rows [b*T+start, b*T+len+end)
 * to drop the special tokens (extract_text_huggingface.py:228-231). */
int mer_bert_forward(const mer_bert* h, const int64_t* ids, const int64_t* token_type, const int* lengths,
                     int B, int T, void* workspace, long long workspace_bytes,
                     float* hidden_states, float* frames,
                     const int* seg_start, const int* seg_len, int nseg, float* pooled,
                     mer_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MER_HIP_H */
