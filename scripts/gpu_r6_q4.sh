#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6q4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm16q_gpu.py tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "gemm16" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
