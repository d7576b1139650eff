// persistent one-pass GEMM, bf16: the 256-row tile's instantiations and the entry point (see gemm16p_impl.h).
#include "gemm16p_impl.h"

namespace mer {
template <> int dispatch_p_tm<bf16, 4>(const Gemm16Params& p, hipStream_t st) { return dispatch_p_impl<bf16, 4>(p, st); }
template <> int dispatch_p_tm<bf16, 3>(const Gemm16Params& p, hipStream_t st);   // gemm16p_bf16_r192.hip
template <> int dispatch_p<bf16>(const Gemm16Params& p, hipStream_t st) { return dispatch_p_pick<bf16>(p, st); }
}  // namespace mer
