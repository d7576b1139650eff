#!/bin/bash
# Per-CU timeline of the CLIP GEMMs + check of the workgroup -> XCD assumption behind the kernel's tile map.
set -u
out=gpurun_out/r2_call23
mkdir -p $out
WARM=10 timeout 200 python scripts/gemm_timeline.py > $out/timeline.txt 2> $out/timeline.err; echo "rc=$?"; cut -c1-400 $out/timeline.txt; tail -3 $out/timeline.err
