#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4c18
timeout 600 python -m pytest tests/test_ops_gpu.py -k "attention" -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-250
timeout 600 python tests/studies/outlier_block_chain_gpu.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c18/outlier_block_chain.txt | cut -c1-300
