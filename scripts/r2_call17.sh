#!/bin/bash
# Does a working set that fits the Infinity Cache (256 MB) speed up the HBM-bound kernels?  visual-only at 16 / 32 / 64 clips per step.
set -u
out=gpurun_out/r2_call17
mkdir -p $out
for b in 16 32 64; do
  timeout 200 python bench.py --modalities v --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $out/bench_v_b$b.json 2> $out/bench_v_b$b.err
  echo "v b=$b rc=$? $(python -c "import json;d=json.load(open('$out/bench_v_b$b.json'));r=d['roofline'];print(d['value'], r['kernel'], r['achieved'], r['avg_launch_us'], r['whole_step_tflops'], {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
done
