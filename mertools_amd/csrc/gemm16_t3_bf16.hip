// 256x256 8-wave instantiations of gemm16_kernel for bf16 (see gemm16_impl.h).
#include "gemm16_impl.h"

namespace mer {

template <>
int dispatch_t3<bf16>(const Gemm16Params& p, int nbatch, int passes, hipStream_t st) {
  if (passes == 4) return MER_EINVAL;   // the MX correction exists for f16 only
  // 256x256, 8 waves (2x4), one workgroup per CU: twice the FLOP per byte pulled into the CU
  if (passes == 3) return launch<bf16, 256, 256, 32, 2, 4, 2, 2, 2>(p, nbatch, st);
  if (passes == 2) return launch<bf16, 256, 256, 32, 2, 4, 1, 2, 3>(p, nbatch, st);
  if (passes == 6) return launch<bf16, 256, 256, 32, 2, 4, 2, 1, 3>(p, nbatch, st);   // activation planes hi + lo against w_hi
  return launch<bf16, 256, 256, 32, 2, 4, 1, 1, 4>(p, nbatch, st);
}

}  // namespace mer
