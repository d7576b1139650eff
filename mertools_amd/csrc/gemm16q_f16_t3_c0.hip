// lagged-group persistent GEMM (gemm16q_impl.h), f16: 192-row tiles, prefetch distance 3, lag 1 slab(s).
#include "gemm16q_impl.h"

namespace mer {
template <> int dispatch_q_cfg<f16, 3, 3, 1>(const Gemm16Params& p, hipStream_t st) { return dispatch_q_impl<f16, 3, 3, 1>(p, st); }
}  // namespace mer
