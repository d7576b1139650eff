// lagged-group persistent GEMM (gemm16q_impl.h), f16: 256-row tiles, prefetch distance 2, lag 1 slab(s).
#include "gemm16q_impl.h"

namespace mer {
template <> int dispatch_q_cfg<f16, 4, 2, 1>(const Gemm16Params& p, hipStream_t st) { return dispatch_q_impl<f16, 4, 2, 1>(p, st); }
}  // namespace mer
