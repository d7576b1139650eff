#!/bin/bash
set -u
out=gpurun_out/r2_call8
mkdir -p $out
export TMPDIR=/tmp
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider -k "gemm16 or mx" > $out/ops.log 2>&1; rc=$?; echo "ops rc=$rc $(tail -1 $out/ops.log)" | tee -a $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: ops tests failed"; grep -E "FAILED|Error|assert" $out/ops.log | head -10; exit 1; }
timeout 120 scripts/probes/gemm16_bench.bin 30 30 all > $out/gemm16_bench.jsonl 2>&1; echo "gemm16_bench rc=$?" | tee -a $out/summary.txt
python - <<'PY' | tee -a $out/summary.txt
import json
for l in open('gpurun_out/r2_call8/gemm16_bench.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if 'shape' in d and d['variant'] == 'pre-blocked W': print(d['shape'][:60].ljust(60), f"{d['us']:.0f}us/{d['TFLOPs']:.0f}TF")
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench.json'));r=d['roofline'];print(d['value'], r['achieved'], r['avg_launch_us'], d.get('parity'), {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
timeout 200 python bench.py --modalities a --batch 32 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_audio_b32.json 2> $out/bench_a.err
echo "bench[a,b32] rc=$? $(python -c "import json;d=json.load(open('$out/bench_audio_b32.json'));r=d['roofline'];print(d['value'], r['kernel'], r['achieved'], r['whole_step_tflops'], d.get('parity'))" 2>/dev/null)" | tee -a $out/summary.txt
timeout 200 python bench.py --modalities v --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_visual_b64.json 2> $out/bench_v.err
echo "bench[v,b64] rc=$? $(python -c "import json;d=json.load(open('$out/bench_visual_b64.json'));r=d['roofline'];print(d['value'], r['kernel'], r['achieved'], r['whole_step_tflops'], d.get('parity'))" 2>/dev/null)" | tee -a $out/summary.txt
