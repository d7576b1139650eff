#!/bin/bash
# PMC passes of the shipped tree (FETCH_SIZE / WRITE_SIZE / SQ counters: each its own rocprofv3 run), the fixed driver test, smoke
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash scripts/pmc_traffic.sh 2>&1 | tail -25
bash scripts/pmc_mfma.sh 2>&1 | tail -25
mkdir -p gpurun_out/r04b
timeout 600 python -m pytest tests/test_extract_gpu.py -k base_size -m gpu -q -s --no-header -p no:cacheprovider > gpurun_out/r04b/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|driver," gpurun_out/r04b/tests.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
