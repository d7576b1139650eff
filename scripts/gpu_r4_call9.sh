#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c9; mkdir -p "$O"
timeout 600 python tests/studies/batch_rows_ops_gpu.py > "$O/batch_rows_ops.txt" 2>&1; cat "$O/batch_rows_ops.txt" | tail -14
