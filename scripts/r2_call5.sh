#!/bin/bash
set -u
out=gpurun_out/r2_call5
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider -k "attention or gemm16" > $out/ops.log 2>&1; echo "ops rc=$? $(tail -1 $out/ops.log)" | tee $out/summary.txt
for set in "" "gemm_store=1" "gemm_store=2" "gemm_store=3"; do
  tag=${set:-default}; tag=${tag//=/_}
  MER_SET="$set" timeout 120 scripts/probes/gemm16_bench.bin 30 30 clip > $out/gemm16_bench_$tag.jsonl 2>&1; echo "gemm16_bench[$tag] rc=$?" | tee -a $out/summary.txt
done
python - <<'PY' | tee -a $out/summary.txt
import json, glob
rows = {}
for f in sorted(glob.glob('gpurun_out/r2_call5/gemm16_bench_*.jsonl')):
    tag = f.split('gemm16_bench_')[1][:-6]
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if 'shape' in d and d['variant'] == 'pre-blocked W':
            rows.setdefault(d['shape'], {})[tag] = (d['us'], d['TFLOPs'])
for s, r in rows.items():
    print(s.ljust(48), '  '.join(f"{t}: {v[0]:.0f}us/{v[1]:.0f}TF" for t, v in r.items()))
PY
WARM=30 timeout 300 python scripts/gemm_timeline.py > $out/timeline_default.txt 2>&1; cat $out/timeline_default.txt | tee -a $out/summary.txt
MER_OPTIONS="gemm_store=1" WARM=30 timeout 300 python scripts/gemm_timeline.py > $out/timeline_sc1.txt 2>&1; cat $out/timeline_sc1.txt | tee -a $out/summary.txt
for opt in "" "attn_waves=4" "gemm_store=1"; do
  tag=${opt:-default}; tag=${tag//=/_}
  MER_OPTIONS="$opt" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $( [[ -n "$opt" ]] && echo --no-parity ) > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "bench[$tag] rc=$? $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));r=d['roofline'];print(d['value'], r['achieved'], r['avg_launch_us'], d.get('parity'), {k:(v['ms_share'],v['tflops']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
done
