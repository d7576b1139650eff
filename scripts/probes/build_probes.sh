#!/bin/bash
# Builds the standalone probe binaries for gfx950 (cross-compiles without a GPU); they travel to the GPU box with gpurun.
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ceiling_probe.bin ceiling_probe.hip
ls -la ceiling_probe.bin
