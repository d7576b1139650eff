#!/bin/bash
set -u
out=gpurun_out/r2_call12
mkdir -p $out
export TMPDIR=/tmp
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 400 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider -k "gemm16 or seg_mean" > $out/ops.log 2>&1; rc=$?; echo "ops rc=$rc $(tail -1 $out/ops.log)" | tee -a $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: ops tests failed"; grep -E "FAILED|Error|assert" $out/ops.log | head -10; exit 1; }
timeout 400 python -m pytest tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "bench_tiles or clip_base16 or ragged" > $out/enc.log 2>&1; rc=$?; echo "enc rc=$rc $(tail -1 $out/enc.log)" | tee -a $out/summary.txt
grep -E "^\.?(hubert|roberta|clip)" $out/enc.log | tee -a $out/summary.txt
grep -E "^(FAILED|E  )" $out/enc.log | head -10 | tee -a $out/summary.txt
for prec in mean mx; do
  timeout 300 python bench.py --precision $prec --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$prec.json 2> $out/bench_$prec.err
  echo "bench[$prec] rc=$? $(python -c "import json;d=json.load(open('$out/bench_$prec.json'));r=d['roofline'];print(d['value'], r['kernel'], r['achieved'], r['share_of_gpu_time'], d.get('parity'), {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items()})" 2>/dev/null)" | tee -a $out/summary.txt
  tail -2 $out/bench_$prec.err | grep -v amdgpu.ids | tee -a $out/summary.txt
done
