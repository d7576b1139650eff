#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c8; mkdir -p "$O"
export TMPDIR=/tmp
timeout 600 python tests/studies/hubert_batch_split_gpu.py fast balanced > "$O/hubert_batch_split.txt" 2>&1; grep "layer  [0-3]" "$O/hubert_batch_split.txt"
