#!/bin/bash
# Headline bench at larger per-step batches, and with the visual batch split over two streams.
set -u
out=gpurun_out/r2_call18
mkdir -p $out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $out/smoke.txt; exit 1; }
run() {  # tag, flags...
  tag=$1; shift
  timeout 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity "$@" > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "$tag rc=$? $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['whole_step_tflops'], {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
}
run b64 --batch 64
run b96 --batch 96
run b128 --batch 128
run b64_vsplit2 --batch 64 --split 2 --split-mods v
