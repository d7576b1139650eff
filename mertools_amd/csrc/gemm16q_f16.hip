// lagged-group persistent GEMM (gemm16q_impl.h), f16: the entry point (tile rows per shape as gemm16p: p_pick_tm).
#include "gemm16q_impl.h"

namespace mer {
template <> int dispatch_q<f16>(const Gemm16Params& p, hipStream_t st) {
  const int tm = g_gemm_tm == 3 || g_gemm_tm == 4 ? g_gemm_tm : p_pick_tm(p.M, p.N, device_cu_count());
  return tm == 3 ? dispatch_q_tm<f16, 3>(p, st) : dispatch_q_tm<f16, 4>(p, st);
}
}  // namespace mer
