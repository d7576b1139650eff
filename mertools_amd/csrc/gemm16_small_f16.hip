// 4-wave (128x128, 128x64) instantiations of gemm16_kernel for f16 (see gemm16_impl.h).
#include "gemm16_impl.h"

namespace mer {

template <>
int dispatch_small<f16>(const Gemm16Params& p, int nbatch, int passes, int tile, hipStream_t st) {
  if (tile == 2) {
    if (passes == 3) return launch<f16, 128, 64, 32, 2, 2, 2, 2, 3>(p, nbatch, st);
    if (passes == 2) return launch<f16, 128, 64, 32, 2, 2, 1, 2, 3>(p, nbatch, st);
    if (passes == 6) return launch<f16, 128, 64, 32, 2, 2, 2, 1, 3>(p, nbatch, st);
    return launch<f16, 128, 64, 32, 2, 2, 1, 1, 4>(p, nbatch, st);
  }
  // stage counts keep the LDS footprint at <= 80 KB so two workgroups share a CU (the C staging tile of the
  // epilogue needs 69.6 KB anyway): 1-pass 4 x 16 KB, 2-pass 3 x 24 KB, 3-pass 2 x 32 KB.
  if (passes == 3) return launch<f16, 128, 128, 32, 2, 2, 2, 2, 2>(p, nbatch, st);
  if (passes == 2) return launch<f16, 128, 128, 32, 2, 2, 1, 2, 3>(p, nbatch, st);
  if (passes == 6) return launch<f16, 128, 128, 32, 2, 2, 2, 1, 3>(p, nbatch, st);
  return launch<f16, 128, 128, 32, 2, 2, 1, 1, 4>(p, nbatch, st);
}

}  // namespace mer
