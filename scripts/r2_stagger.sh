#!/bin/bash
# (HISTORICAL: the gemm_stagger option was removed after this A/B — negative result, profiles/r02_gemm_stagger_ab.txt.)
# Phase-spreading A/B: real-kernel timings and the bench with gemm_stagger = 0 / 4 / 2, plus the new fusion golden test.
set -u
out=gpurun_out/r2_stagger
mkdir -p $out
export TMPDIR=/tmp
for set in "" "gemm_stagger=4" "gemm_stagger=2"; do
  tag=${set:-default}; tag=${tag//=/_}
  MER_SET="$set" timeout 120 scripts/probes/gemm16_bench.bin 30 30 all > $out/gemm16_bench_$tag.jsonl 2>&1; echo "gemm16_bench[$tag] rc=$?" | tee -a $out/summary.txt
done
python - <<'PY' | tee -a $out/summary.txt
import json, glob
rows = {}
for f in sorted(glob.glob('gpurun_out/r2_stagger/gemm16_bench_*.jsonl')):
    tag = f.split('gemm16_bench_')[1][:-6]
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if 'shape' in d and d['variant'] == 'pre-blocked W':
            rows.setdefault(d['shape'], {})[tag] = (d['us'], d['TFLOPs'])
for s, r in rows.items():
    print(s.ljust(64), '  '.join(f"{t}: {v[0]:.0f}us/{v[1]:.0f}TF" for t, v in r.items()))
PY
for opt in "" "gemm_stagger=4" "gemm_stagger=2"; do
  tag=${opt:-default}; tag=${tag//=/_}
  MER_OPTIONS="$opt" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "bench[$tag] rc=$? $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));r=d['roofline'];print(d['value'], r['achieved'], r['avg_launch_us'], {k:(v['ms_share'],v['tflops']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
done
timeout 300 python -m pytest tests/test_fusion_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "frame_level or graph_loop" > $out/fusion.log 2>&1; echo "fusion rc=$? $(tail -1 $out/fusion.log)" | tee -a $out/summary.txt
