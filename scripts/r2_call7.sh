#!/bin/bash
# Guarded: a hung / failing first step aborts the whole call (round-2 call 6 burned 30 GPU-minutes on timeouts behind a hung GPU).
set -u
out=gpurun_out/r2_call7
mkdir -p $out
export TMPDIR=/tmp
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 60 scripts/probes/gemm16_bench.bin 5 5 clip > $out/gemm16_quick.jsonl 2>&1; rc=$?; echo "gemm16 quick rc=$rc" | tee -a $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: gemm16_bench failed"; tail -3 $out/gemm16_quick.jsonl; exit 1; }
timeout 400 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider > $out/ops.log 2>&1; rc=$?; echo "ops rc=$rc $(tail -1 $out/ops.log)" | tee -a $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: ops tests failed"; grep -E "FAILED|Error|assert" $out/ops.log | head -10; exit 1; }
for set in "" "gemm_epi32=0" "gemm_store32=2" "gemm_res_nt=1" "gemm_store32=2,gemm_res_nt=1"; do
  tag=${set:-default}; tag=${tag//=/_}; tag=${tag//,/_}
  MER_SET="$set" timeout 90 scripts/probes/gemm16_bench.bin 30 30 clip > $out/gemm16_bench_$tag.jsonl 2>&1; echo "gemm16_bench[$tag] rc=$?" | tee -a $out/summary.txt
done
python - <<'PY' | tee -a $out/summary.txt
import json, glob
rows = {}
for f in sorted(glob.glob('gpurun_out/r2_call7/gemm16_bench_*.jsonl')):
    tag = f.split('gemm16_bench_')[1][:-6]
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if 'shape' in d and d['variant'] == 'pre-blocked W':
            rows.setdefault(d['shape'], {})[tag] = (d['us'], d['TFLOPs'])
for s, r in rows.items():
    print(s[:40].ljust(40), '  '.join(f"{t}: {v[0]:.0f}us/{v[1]:.0f}TF" for t, v in r.items()))
PY
for opt in "" "ln_nt=1" "attn_nt=1" "gemm_store32=2,gemm_res_nt=1" "gemm_epi32=0,gemm_store=0"; do
  tag=${opt:-default}; tag=${tag//=/_}; tag=${tag//,/_}
  MER_OPTIONS="$opt" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $( [[ -n "$opt" ]] && echo --no-parity ) > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "bench[$tag] rc=$? $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));r=d['roofline'];print(d['value'], r['achieved'], r['avg_launch_us'], d.get('parity'), {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
done
