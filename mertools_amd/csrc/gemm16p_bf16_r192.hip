// persistent one-pass GEMM, bf16 instantiations of the 192-row tile (see gemm16p_impl.h).
#include "gemm16p_impl.h"

namespace mer {
template <> int dispatch_p_tm<bf16, 3>(const Gemm16Params& p, hipStream_t st) { return dispatch_p_impl<bf16, 3>(p, st); }
}  // namespace mer
