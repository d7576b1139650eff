#!/bin/bash
# the round's last GPU action: PMC passes of the shipped tree (FETCH_SIZE / WRITE_SIZE / SQ counters: each its own rocprofv3 run), smoke
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash scripts/pmc_traffic.sh 2>&1 | grep -E "rc=|gemm16p|layernorm_rows|seqbias|seqmean" | head
bash scripts/pmc_mfma.sh 2>&1 | grep -E "rc=|gemm16p|attn_sp" | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
