#!/bin/bash
# Builds the standalone probe binaries for gfx950 (cross-compiles without a GPU); they travel to the GPU box with gpurun.
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ceiling_probe.bin ceiling_probe.hip
ls -la ceiling_probe.bin
# C-ABI self-test of libmer_hip.so (needs the library built: python -m mertools_amd.build); finds it next to the binary's repo copy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -x hip abi_selftest.cpp -I../../include -L../../mertools_amd -lmer_hip \
  -Wl,-rpath,'$ORIGIN/../../mertools_amd' -o abi_selftest.bin
ls -la abi_selftest.bin
# real-kernel GEMM timing over the C ABI (no Python): the cheap A/B tool
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -x hip gemm16_bench.cpp -I../../include -L../../mertools_amd -lmer_hip \
  -Wl,-rpath,'$ORIGIN/../../mertools_amd' -o gemm16_bench.bin
ls -la gemm16_bench.bin
# what bounds the GEMM epilogue's global stores (per-CU rate vs chip rate, store flavours)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store_probe.bin store_probe.hip
ls -la store_probe.bin
