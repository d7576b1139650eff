// gemm16.hip — C ABI of the 16-bit MFMA GEMM (mer_gemm16), its options and the weight packers; the kernel template lives in
// gemm16_impl.h and is instantiated per (tile class, dtype) in gemm16_t3_*.hip / gemm16_small_*.hip.
#include "gemm16p_impl.h"

namespace mer {
// debug / tuning switches (mer_set_option; process-global and NOT thread-safe: set them before any forward is in flight)
int g_gemm_skip = 0;          // "gemm_dbg_skip": 1 = skip the epilogue's global stores, 2 = skip the whole epilogue (timing decomposition)
int g_gemm_stamp = 0;         // "gemm_stamp": the instrumented (s_memtime) build of the 8-wave kernels, with mer_set_debug_buffer
int g_gemm_glds = 1;          // "gemm_glds": 0 = register-staged loader instead of LDS-DMA (A/B and the K % 32 != 0 fallback's twin)
int g_gemm_tm = 0;            // "gemm_tm": rows per persistent tile / 64: 0 = chosen per shape (p_pick_tm), 3 or 4 = forced
int g_gemm_persist = 1;       // "gemm_persist": 0 = never take the persistent kernel (A/B against gemm16_kernel on the same planes)
int g_gemm_generic_epi = 0;   // "gemm_generic_epi": 1 = every launch takes the generic epilogue (the specialised ones must equal it)
unsigned long long* g_gemm_dbg = nullptr;  // mer_set_debug_buffer(); also stamped by attn_sp_kernel

template <typename T>
static int dispatch(const Gemm16Params& p, int nbatch, int passes, int tile, hipStream_t st) {
  if (passes == 4 || tile == 3) return dispatch_t3<T>(p, nbatch, passes, st);   // MX-corrected: eligibility was checked by mer_gemm16
  return dispatch_small<T>(p, nbatch, passes, tile, st);
}

}  // namespace mer

extern "C" int mer_set_debug_buffer(void* device_u64_buffer) {
  mer::g_gemm_dbg = (unsigned long long*)device_u64_buffer;
  return MER_OK;
}

extern "C" int mer_set_option(const char* name, int value) {
  if (name && strcmp(name, "gemm_glds") == 0) { mer::g_gemm_glds = value; return MER_OK; }
  if (name && strcmp(name, "gemm_dbg_skip") == 0) { mer::g_gemm_skip = value; return MER_OK; }
  if (name && strcmp(name, "gemm_stamp") == 0) { mer::g_gemm_stamp = value; return MER_OK; }
  if (name && strcmp(name, "gemm_persist") == 0) { mer::g_gemm_persist = value; return MER_OK; }
  if (name && strcmp(name, "gemm_tm") == 0) { mer::g_gemm_tm = value; return MER_OK; }
  if (name && strcmp(name, "gemm_generic_epi") == 0) { mer::g_gemm_generic_epi = value; return MER_OK; }
  mer::set_error("mer_set_option: unknown option '%s'", name ? name : "(null)");
  return MER_EINVAL;
}

extern "C" int mer_get_option(const char* name, int* value) {
  const struct { const char* n; const int* v; } opts[] = {
      {"gemm_glds", &mer::g_gemm_glds}, {"gemm_dbg_skip", &mer::g_gemm_skip}, {"gemm_stamp", &mer::g_gemm_stamp}, {"gemm_persist", &mer::g_gemm_persist},
      {"gemm_tm", &mer::g_gemm_tm}, {"gemm_generic_epi", &mer::g_gemm_generic_epi}};
  if (name && value)
    for (const auto& o : opts)
      if (strcmp(name, o.n) == 0) { *value = *o.v; return MER_OK; }
  mer::set_error("mer_get_option: unknown option '%s'", name ? name : "(null)");
  return MER_EINVAL;
}

// ---- device-side packer of pre-blocked weight planes (weights are prepared once, offline) ----
namespace mer {
// one thread per 16-byte chunk: out[(tn, kt)][row r][physical chunk pc] = w[min(tn*256 + r, N-1)][kt*32 + 8*(pc ^ swz(r)) ..]
__global__ void w_block_pack_kernel(const u32x4* w, long long ldw8, int N, int nk, long long total, u32x4* out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int pc = (int)(idx & 3), r = (int)((idx >> 2) & 255);
  const long long blk = idx >> 10;
  const int kt = (int)(blk % nk);
  const long long tn = blk / nk;
  long long n = tn * 256 + r;
  n = n < N ? n : N - 1;
  out[idx] = w[n * ldw8 + kt * 4 + (pc ^ swz_of<4>(r))];
}

// row-permuted variants for the persistent kernel (gemm16p_impl.h: p_perm_row): block row r of every 128-row group holds the plane
// row that makes a lane's eight accumulators of an output row consecutive columns (layout 0: 8 li + nt; layout 1: two runs of four)
__global__ void w_block_pack_p_kernel(const u32x4* w, long long ldw8, int N, int nk, long long total, int layout, u32x4* out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int pc = (int)(idx & 3), r = (int)((idx >> 2) & 255);
  const long long blk = idx >> 10;
  const int kt = (int)(blk % nk);
  const long long tn = blk / nk;
  long long n = tn * 256 + p_perm_row(r, layout);
  n = n < N ? n : N - 1;
  out[idx] = w[n * ldw8 + kt * 4 + (pc ^ swz_of<4>(r))];
}
}  // namespace mer

extern "C" int mer_w_block_pack_p(const void* w, long long ldw, int N, int K, int layout, void* out, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(w && out, MER_EINVAL, "mer_w_block_pack_p: null pointer");
  MER_REQUIRE(layout == 0 || layout == 1, MER_EINVAL, "mer_w_block_pack_p: layout must be 0 (16-bit outputs) or 1 (fp32 outputs)");
  MER_REQUIRE(N > 0 && N % 256 == 0 && K > 0 && K % 32 == 0 && ldw % 8 == 0, MER_ESHAPE,
              "mer_w_block_pack_p: N must be a multiple of 256, K of 32 and ldw of 8 (N=%d K=%d ldw=%lld)", N, K, ldw);
  MER_REQUIRE((((uintptr_t)w | (uintptr_t)out) & 15) == 0, MER_EINVAL, "mer_w_block_pack_p: planes must be 16-byte aligned");
  const long long total = (long long)(N / 256) * (K / 32) * 1024;
  hipLaunchKernelGGL(w_block_pack_p_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)w, ldw / 8, N, K / 32, total, layout, (u32x4*)out);
  return check_launch("w_block_pack_p");
}

extern "C" long long mer_w_block_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
  return (long long)((N + 255) / 256) * 256 * K * 2;
}

extern "C" int mer_w_block_pack(const void* w, long long ldw, int N, int K, void* out, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(w && out, MER_EINVAL, "mer_w_block_pack: null pointer");
  MER_REQUIRE(N > 0 && K > 0 && K % 32 == 0 && ldw % 8 == 0, MER_ESHAPE, "mer_w_block_pack: K must be a multiple of 32 and ldw of 8 (N=%d K=%d ldw=%lld)", N, K, ldw);
  MER_REQUIRE((((uintptr_t)w | (uintptr_t)out) & 15) == 0, MER_EINVAL, "mer_w_block_pack: planes must be 16-byte aligned");
  const long long total = (long long)((N + 255) / 256) * (K / 32) * 1024;
  hipLaunchKernelGGL(w_block_pack_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)w, ldw / 8, N, K / 32, total, (u32x4*)out);
  return check_launch("w_block_pack");
}

// ---- host-side packer of the MX correction plane (weights are prepared once, offline) ----
extern "C" long long mer_mx_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 128 != 0) return 0;
  return (long long)((N + 255) / 256) * (K / 32) * mer::MX_BLOCK;
}

extern "C" int mer_mx_pack(const float* w_res, long long ldw, int N, int K, void* out) {
  using namespace mer;
  MER_REQUIRE(w_res && out, MER_EINVAL, "mer_mx_pack: null pointer");
  MER_REQUIRE(N > 0 && K > 0 && K % 128 == 0, MER_ESHAPE, "mer_mx_pack: K must be a positive multiple of 128 (N=%d K=%d)", N, K);
  static const float GRID[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};   // e2m1 magnitudes
  unsigned char* o = (unsigned char*)out;
  const int nslab = K / 32, ntile = (N + 255) / 256;
  memset(o, 0, (size_t)mer_mx_packed_bytes(N, K));
  auto quant = [&](float x, float inv_scale) -> int {   // nearest e2m1 code (ties to the even code), saturating
    const float m = fabsf(x) * inv_scale;
    int best = 0; float bd = m;
    for (int c = 1; c < 8; ++c) {
      const float d = fabsf(m - GRID[c]);
      if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = c; }
    }
    return best | (x < 0.f && best ? 8 : 0);
  };
  for (int tn = 0; tn < ntile; ++tn)
    for (int kg = 0; kg < K / 128; ++kg)
      for (int ctg = 0; ctg < 16; ++ctg)          // 16-column tile of the 256-column block
        for (int n16 = 0; n16 < 16; ++n16) {
          const int n = tn * 256 + ctg * 16 + n16;
          float v[128];
          for (int h = 0; h < 128; ++h) {   // k-slot h of the instruction <-> k of the group (see gemm16_kernel)
            const int hh = h & 63, g = hh >> 4, r = hh & 15, s = (h >> 6) * 2 + (r >> 3), e = r & 7;
            v[h] = n < N ? w_res[(long long)n * ldw + kg * 128 + 32 * s + 8 * g + e] : 0.f;
          }
          // column tile ctg travels with slab (ctg / 4) of the group, as entry (ctg % 4) of that slab's block
          unsigned char* frag = o + ((size_t)tn * nslab + kg * 4 + ctg / 4) * MX_BLOCK + (ctg % 4) * 1024;
          for (int b = 0; b < 4; ++b) {
            float amax = 0.f;
            for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[32 * b + j]));
            int sbyte = 0;
            if (amax > 0.f) {
              int ex; frexpf(amax, &ex);                 // amax = f * 2^ex, f in [0.5, 1)  ->  floor(log2) = ex - 1
              const int e0 = ex - 1 - 2;                 // OCP MX: 2^(floor(log2 amax) - emax(e2m1))
              double best_err = -1.0;
              for (int cand = e0; cand <= e0 + 1; ++cand) {   // amax/2^e0 is in [4, 8): the coarser scale may fit better
                const int sb = cand + 127 < 0 ? 0 : (cand + 127 > 254 ? 254 : cand + 127);
                const float sc = ldexpf(1.f, sb - 127), inv = 1.f / sc;
                double err = 0.0;
                for (int j = 0; j < 32; ++j) {
                  const float x = v[32 * b + j];
                  const int c = quant(x, inv);
                  const double d = (double)x - (double)((c & 8) ? -GRID[c & 7] : GRID[c & 7]) * sc;
                  err += d * d;
                }
                if (best_err < 0.0 || err < best_err) { best_err = err; sbyte = sb; }
              }
            }
            const float inv = 1.f / ldexpf(1.f, sbyte - 127);
            const int lane = n16 + 16 * b;
            for (int j = 0; j < 32; j += 2)
              frag[lane * 16 + j / 2] = (unsigned char)(quant(v[32 * b + j], inv) | (quant(v[32 * b + j + 1], inv) << 4));
            for (int q = 0; q < 4; ++q)   // every slab of the group carries the group's scales: dword [ctg / 4][lane], byte ctg % 4
              o[((size_t)tn * nslab + kg * 4 + q) * MX_BLOCK + 4096 + (ctg / 4) * 256 + lane * 4 + (ctg % 4)] = (unsigned char)sbyte;
          }
        }
  return MER_OK;
}

static inline int nbatch_of(const mer_gemm16_args* a) { return a->nbatch > 0 ? a->nbatch : 1; }

extern "C" int mer_gemm16(const mer_gemm16_args* a, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(a != nullptr, MER_EINVAL, "mer_gemm16: null args");
  MER_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, MER_ESHAPE, "mer_gemm16: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  MER_REQUIRE(a->a_hi && a->w_hi, MER_EINVAL, "mer_gemm16: a_hi / w_hi must be non-null");
  MER_REQUIRE((a->passes >= 1 && a->passes <= 4) || a->passes == 6, MER_EINVAL, "mer_gemm16: passes must be 1, 2, 3, 4 or 6 (got %d)", a->passes);
  MER_REQUIRE(a->passes == 4 || a->passes == 6 || a->passes < 2 || a->w_lo, MER_EINVAL, "mer_gemm16: passes 2/3 need w_lo");
  MER_REQUIRE((a->passes != 3 && a->passes != 6) || a->a_lo, MER_EINVAL, "mer_gemm16: passes=3 / 6 need a_lo");
  MER_REQUIRE(a->passes != 4 || a->w_mx || a->w_lo, MER_EINVAL, "mer_gemm16: passes=4 needs w_mx (or w_lo for the 2-pass fallback)");
  MER_REQUIRE(a->K % 8 == 0 && a->lda % 8 == 0 && a->ldw % 8 == 0, MER_ESHAPE,
              "mer_gemm16: K, lda, ldw must be multiples of 8 (K=%d lda=%lld ldw=%lld)", a->K, a->lda, a->ldw);
  MER_REQUIRE(a->a_so % 8 == 0 && a->a_si % 8 == 0 && a->w_si % 8 == 0 && a->a_batch_stride % 8 == 0, MER_ESHAPE,
              "mer_gemm16: batch strides of A / W must be multiples of 8 elements");
  MER_REQUIRE(a->dtype == MER_DT_F16 || a->dtype == MER_DT_BF16, MER_EINVAL, "mer_gemm16: bad dtype %d", a->dtype);
  MER_REQUIRE(a->c32 || a->c16_hi, MER_EINVAL, "mer_gemm16: no output given");
  MER_REQUIRE(a->headmajor_T == 0 || (a->headmajor_T > 0 && a->headmajor_H > 0 && a->M % a->headmajor_T == 0 && a->N % (64 * a->headmajor_H) == 0 &&
                                      a->c16_hi && a->N % 8 == 0 && (a->nbatch <= 1)),
              MER_ESHAPE, "mer_gemm16: head-major output needs M %% T == 0, N %% (64*H) == 0, a 16-bit output and no batching");
  MER_REQUIRE(!a->c16_lo || a->c16_hi, MER_EINVAL, "mer_gemm16: c16_lo without c16_hi");
  MER_REQUIRE(a->bias_seg_rows >= 0 && (a->bias_seg_rows == 0 || !a->bias || (a->nbatch <= 1 && a->bias_ld >= a->N && a->bias_ld % 4 == 0 &&
                                                                             (((uintptr_t)a->bias) & 15) == 0)),
              MER_ESHAPE, "mer_gemm16: a bias table (bias_seg_rows > 0) needs nbatch <= 1, bias_ld >= N, bias_ld %% 4 == 0 and 16-byte alignment");
  const int nbatch = a->nbatch > 0 ? a->nbatch : 1;
  Gemm16Params p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.a_hi = a->a_hi; p.a_lo = a->a_lo; p.lda = a->lda; p.a_rpb = a->a_rows_per_batch; p.a_bstride = a->a_batch_stride;
  p.w_hi = a->w_hi; p.w_lo = a->w_lo; p.ldw = a->ldw; p.w_mx = a->w_mx; p.w_blk = 0;
  p.bias = a->bias; p.act = a->act;
  p.bias_T = a->bias ? a->bias_seg_rows : 0; p.bias_ld = a->bias_ld;
  p.residual = a->residual; p.ldr = a->ldr;
  p.c32 = a->c32; p.ldc32 = a->ldc32;
  p.c16_hi = a->c16_hi; p.c16_lo = a->c16_lo; p.ldc16 = a->ldc16;
  p.nb_inner = a->nb_inner > 0 ? a->nb_inner : 1;
  p.a_so = a->a_so; p.a_si = a->a_si; p.w_si = a->w_si; p.bias_si = a->bias_si; p.c_so = a->c_so; p.c_si = a->c_si;
  p.tiles_m = p.tiles_n = 0;
  p.dbg = g_gemm_dbg;
  p.dbg_skip = g_gemm_skip;
  p.hm_T = a->headmajor_T; p.hm_H = a->headmajor_H;
  // the vector epilogue moves 8 columns per lane with 16-byte accesses
  bool vec = (a->N % 8 == 0) && (a->c_so % 8 == 0) && (a->c_si % 8 == 0);
  if (a->residual) vec = vec && (a->ldr % 4 == 0) && (((uintptr_t)a->residual & 15) == 0);
  if (a->c32) vec = vec && (a->ldc32 % 4 == 0) && (((uintptr_t)a->c32 & 15) == 0);
  if (a->c16_hi) vec = vec && (a->ldc16 % 8 == 0) && (((uintptr_t)a->c16_hi & 15) == 0);
  if (a->c16_lo) vec = vec && (((uintptr_t)a->c16_lo & 15) == 0);
  p.vec_ok = vec ? 1 : 0;
  // fp32-only epilogue: whole-line residual loads / stores (4 columns per lane); N % 8 == 0 and 16-byte alignment are `vec`
  p.epi32 = (!g_gemm_generic_epi && vec && a->c32 && !a->c16_hi && !a->c16_lo && a->headmajor_T == 0 && (a->act == MER_ACT_NONE || a->act == MER_ACT_GELU) &&
             (!a->bias || ((((uintptr_t)a->bias) & 15) == 0 && a->bias_si % 4 == 0))) ? 1 : 0;   // its static bias is one 16-byte load per lane
  // packed-pair epilogue: 16-bit output only (no fp32 copy, residual or lo plane), row-major, every column group of 8 in range
  p.pk_epi = (!g_gemm_generic_epi && vec && a->c16_hi && !a->c16_lo && !a->c32 && !a->residual && a->headmajor_T == 0 &&
              a->act != MER_ACT_RELU && p.bias_T == 0) ? 1 : 0;   // (a bias table is per row: the packed form adds the bias per column)
  MER_REQUIRE((((uintptr_t)a->a_hi | (uintptr_t)a->w_hi | (uintptr_t)a->a_lo | (uintptr_t)a->w_lo | (uintptr_t)a->w_mx) & 15) == 0, MER_EINVAL,
              "mer_gemm16: operand planes must be 16-byte aligned");
  int tile = a->tile;
  // 256x256 (8 waves, staggered schedule) wins whenever there are enough tiles to occupy the chip; narrow / short problems keep
  // 4-wave tiles: N <= 64 -> 128x64; fewer than 1024 rows, or a narrow output (N <= 1024) that makes at most 96 tiles of 256x256
  // (RoBERTa's attention-output / fc2 GEMMs at 64 clips, HuBERT's at 32: 48 / 96 tiles on 256 CUs) -> 128x128, four times the
  // workgroups (measured, profiles/r03_gemm16_bench_tile1_short_m.txt: 27.9 -> 17.9 us, 67.4 -> 46.8 us, 30.7 -> 24.0 us,
  // 72.2 -> 60.5 us; the wide QKV / fc1 launches and anything with 144+ tiles are faster on 256x256).  The MX preset keeps its kernel.
  if (tile == 0) {
    const long long t256 = cdiv(a->M, 256) * cdiv(a->N, 256);
    const bool few_narrow = t256 <= 96 && a->N <= 1024 && a->passes != 4;
    tile = (a->N <= 64) ? 2 : ((a->M >= 1024 && a->N >= 192 && !few_narrow) ? 3 : 1);
    // an MX-corrected GEMM keeps the MX kernel however few rows it has: its 2-pass stand-in is different arithmetic, and a clip's
    // features would depend on whether its batch reached 1024 rows (HuBERT's last conv layers at batch 1 .. 4: 249 .. 996 rows)
    if (a->passes == 4 && a->w_mx && a->N >= 192 && a->dtype == MER_DT_F16 && a->K % 128 == 0 && nbatch == 1) tile = 3;
  }
  hipStream_t st = (hipStream_t)stream;
  int passes = a->passes;
  if (passes == 4) {
    // the MX correction lives in the 256x256 f16 kernel; anything it does not cover runs the f16 2-pass path
    // (its DMA uses 32-bit byte offsets from the plane bases, so both planes must span < 4 GB)
    const long long a_last = a->a_rows_per_batch > 0
        ? (long long)((a->M - 1) / a->a_rows_per_batch) * a->a_batch_stride + (long long)((a->M - 1) % a->a_rows_per_batch) * a->lda
        : (long long)(a->M - 1) * a->lda;
    const bool small = (a_last + a->K) * 2 < (1ll << 32) && ((long long)a->N * a->ldw) * 2 < (1ll << 32);
    const bool mx_ok = a->w_mx && tile == 3 && a->dtype == MER_DT_F16 && a->K % 128 == 0 && nbatch == 1 && g_gemm_glds == 1 && small;
    if (!mx_ok) {
      MER_REQUIRE(a->w_lo, MER_EINVAL, "mer_gemm16: passes=4 on a shape the MX kernel does not cover (tile %d, K=%d, nbatch=%d, dtype %d) needs w_lo",
                  tile, a->K, nbatch, a->dtype);
      passes = 2;
    }
  }
  // pre-blocked weight planes feed the 256-wide LDS-DMA kernels only (their block is one 256 x 32 stage plane)
  if (a->w_hi_blk && tile == 3 && g_gemm_glds == 1 && a->K % 32 == 0 && nbatch == 1 && (passes == 1 || passes == 4 || passes == 6 || a->w_lo_blk)) {
    MER_REQUIRE((((uintptr_t)a->w_hi_blk | (uintptr_t)a->w_lo_blk) & 15) == 0, MER_EINVAL, "mer_gemm16: pre-blocked planes must be 16-byte aligned");
    p.w_hi = a->w_hi_blk;
    p.w_lo = passes == 6 ? nullptr : a->w_lo_blk;
    p.w_blk = 1;
  }
  // the persistent 256x256 kernel (gemm16p_impl.h): one pass, its own row-permuted pre-blocked plane (layout A for a 16-bit output,
  // layout B for an fp32 one), whole 256-column tiles, a K loop long enough for its counted waits (8 slabs), one 16-bit plane OR fp32
  // (+ residual, then without activation) out, every plane within 32-bit byte offsets
  const void* wp = a->c16_hi ? a->w_hi_blkp : a->w_hi_blkq;
  if (wp && tile == 3 && passes == 1 && g_gemm_persist && g_gemm_glds == 1 && !g_gemm_generic_epi && nbatch == 1 &&
      a->N % 256 == 0 && a->K % 32 == 0 && a->K >= 256 && a->headmajor_T == 0 && !a->c16_lo && (!a->c16_hi != !a->c32) &&
      (a->c16_hi ? !a->residual : true) && (p.bias_T == 0 || p.bias_T >= 40)) {
    const long long a_last = a->a_rows_per_batch > 0
        ? (long long)((a->M - 1) / a->a_rows_per_batch) * a->a_batch_stride + (long long)((a->M - 1) % a->a_rows_per_batch) * a->lda
        : (long long)(a->M - 1) * a->lda;
    const bool act_ok = a->c16_hi ? (a->act != MER_ACT_RELU) : (a->act == MER_ACT_NONE || (a->act == MER_ACT_GELU && !a->residual));
    bool al = (((uintptr_t)wp | (uintptr_t)a->bias) & 15) == 0;
    if (a->c16_hi) al = al && a->ldc16 % 8 == 0 && (((uintptr_t)a->c16_hi) & 15) == 0 && 256ll * a->ldc16 * 2 < (1ll << 31);
    if (a->c32) al = al && a->ldc32 % 4 == 0 && (((uintptr_t)a->c32) & 15) == 0 && 256ll * a->ldc32 * 4 < (1ll << 31);
    if (a->residual) al = al && a->ldr % 4 == 0 && (((uintptr_t)a->residual) & 15) == 0 && 256ll * a->ldr * 4 < (1ll << 31);
    if (act_ok && al && (a_last + a->K) * 2 < (1ll << 32)) {
      p.w_hi = wp;
      p.w_lo = nullptr;
      p.w_blk = 2;
      if (a->dtype == MER_DT_F16) return dispatch_p<f16>(p, st);
      return dispatch_p<bf16>(p, st);
    }
  }
  if (a->dtype == MER_DT_F16) return dispatch<f16>(p, nbatch, passes, tile, st);
  return dispatch<bf16>(p, nbatch, passes, tile, st);
}
