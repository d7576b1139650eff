#!/bin/bash
# Standalone GEMM timing with hot (one A / output plane re-used, Infinity-Cache resident) vs cold (4 rotating planes) operands.
set -u
out=gpurun_out/r2_call22
mkdir -p $out
timeout 150 scripts/probes/gemm16_bench.bin 20 40 all > $out/gemm16_bench.jsonl 2> $out/gemm16_bench.err; echo "gemm16_bench rc=$?"
python - <<'PY' | tee $out/summary.txt
import json
rows = {}
for l in open('gpurun_out/r2_call22/gemm16_bench.jsonl'):
    d = json.loads(l)
    if 'shape' in d:
        rows.setdefault(d['shape'], {})[d['variant']] = (d['us'], d['TFLOPs'])
for s, v in rows.items():
    hot, cold = v.get('pre-blocked W'), v.get('pre-blocked W, 4 rotating A / output planes (cold operands)')
    print(f"{s:64s} hot {hot[0]:7.1f} us {hot[1]:5.0f} TF   cold {cold[0]:7.1f} us {cold[1]:5.0f} TF   x{cold[0] / hot[0]:.3f}")
PY
