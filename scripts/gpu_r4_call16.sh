#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4c16
timeout 600 python tests/studies/outlier_block_bisect_gpu.py 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c16/outlier_block_bisect_b8.txt | cut -c1-330
