// persistent 256x256 one-pass GEMM, bf16 instantiations (see gemm16p_impl.h).
#include "gemm16p_impl.h"

namespace mer {
template <> int dispatch_p<bf16>(const Gemm16Params& p, hipStream_t st) { return dispatch_p_impl<bf16>(p, st); }
}  // namespace mer
