#!/bin/bash
# HuBERT conv stack in one f16 pass (transformer blocks stay MX-corrected): parity and audio-only throughput against the mx preset.
# (needs the one-line experiment preset  _PREC["mxc1"] = (1, 4)  in mertools_amd/encoders.py and "mxc1" in bench.py's --precision choices; not kept)
set -u
out=gpurun_out/r2_call21
mkdir -p $out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $out/smoke.txt; exit 1; }
timeout 300 python scripts/conv_precision_ab.py mx mxc1 > $out/parity.txt 2> $out/parity.err; echo "parity rc=$?"; cat $out/parity.txt | tee -a $out/summary.txt
for prec in mx mxc1; do
  timeout 200 python bench.py --modalities a --precision $prec --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_a_$prec.json 2> $out/bench_a_$prec.err
  echo "a $prec rc=$? $(python -c "import json;d=json.load(open('$out/bench_a_$prec.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], d['parity'], {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16','gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
done
