#!/bin/bash
set -u
out=gpurun_out/r2_call10
mkdir -p $out
export TMPDIR=/tmp
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider -k "attention" > $out/ops.log 2>&1; rc=$?; echo "ops(attention) rc=$rc $(tail -1 $out/ops.log)" | tee -a $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: attention tests failed"; grep -E "FAILED|Error|assert" $out/ops.log | head -10; exit 1; }
timeout 300 python -m pytest tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "videomae" > $out/enc.log 2>&1; echo "enc(videomae) rc=$? $(tail -1 $out/enc.log)" | tee -a $out/summary.txt
grep -E "^\.?videomae" $out/enc.log | tee -a $out/summary.txt
for opt in "" "attn_stream_qs=1"; do
  tag=${opt:-default}; tag=${tag//=/_}
  MER_OPTIONS="$opt" timeout 300 python bench.py --config large --steps 5 --warmup 2 --no-cpu-baseline $( [[ -n "$opt" ]] && echo --no-parity ) > $out/bench_large_$tag.json 2> $out/bench_large_$tag.err
  echo "bench[large,$tag] rc=$? $(python -c "import json;d=json.load(open('$out/bench_large_$tag.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['whole_step_tflops'], d.get('parity'), {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
done
timeout 400 python bench.py --config large --dtype bf16 --precision accurate --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_large_bf16.json 2> $out/bench_large_bf16.err
echo "bench[large,bf16 accurate] rc=$? $(python -c "import json;d=json.load(open('$out/bench_large_bf16.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['whole_step_tflops'], d.get('parity'))" 2>/dev/null)" | tee -a $out/summary.txt
tail -2 $out/bench_large_bf16.err | grep -v amdgpu.ids | tee -a $out/summary.txt
