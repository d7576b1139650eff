#!/bin/bash
set -u
out=gpurun_out/r2_call15
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider -k "gemm16" > $out/ops.log 2>&1; rc=$?; echo "ops rc=$rc $(tail -1 $out/ops.log)" | tee -a $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: ops tests failed"; grep -E "FAILED|Error|assert" $out/ops.log | head -10; exit 1; }
timeout 120 scripts/probes/gemm16_bench.bin 30 30 all > $out/gemm16_bench.jsonl 2>&1; echo "gemm16_bench rc=$?" | tee -a $out/summary.txt
python - <<'PY' | tee -a $out/summary.txt
import json
for l in open('gpurun_out/r2_call15/gemm16_bench.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if 'shape' in d and d['variant'] == 'pre-blocked W': print(d['shape'][:60].ljust(60), f"{d['us']:.0f}us/{d['TFLOPs']:.0f}TF")
PY
timeout 300 python -m pytest tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "bench_tiles or clip_base16_8frames" > $out/enc.log 2>&1; echo "enc rc=$? $(tail -1 $out/enc.log)" | tee -a $out/summary.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], d.get('parity'), d['cpu_baseline']['value'])" 2>/dev/null)" | tee -a $out/summary.txt
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc/$c
  rm -rf "$d"
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- \
     python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > "$R/gpurun_out/pmc/$c.log" 2>&1; echo "$c rc=$?" | tee -a "$R/$out/summary.txt")
done
python scripts/pmc_summarize.py gpurun_out/pmc > $out/pmc_summary.txt 2>&1; head -4 $out/pmc_summary.txt | cut -c1-160
cp gpurun_out/pmc/summary.json $out/pmc_summary.json
find gpurun_out/pmc -name "*.csv" -size +5M -delete
