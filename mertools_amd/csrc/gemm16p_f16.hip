// persistent one-pass GEMM, f16: the 256-row tile's instantiations and the entry point (see gemm16p_impl.h).
#include "gemm16p_impl.h"

namespace mer {
template <> int dispatch_p_tm<f16, 4>(const Gemm16Params& p, hipStream_t st) { return dispatch_p_impl<f16, 4>(p, st); }
template <> int dispatch_p_tm<f16, 3>(const Gemm16Params& p, hipStream_t st);   // gemm16p_f16_r192.hip
template <> int dispatch_p<f16>(const Gemm16Params& p, hipStream_t st) { return dispatch_p_pick<f16>(p, st); }
}  // namespace mer
