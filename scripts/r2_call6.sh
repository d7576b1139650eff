#!/bin/bash
set -u
out=gpurun_out/r2_call6
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider > $out/ops.log 2>&1; echo "ops rc=$? $(tail -1 $out/ops.log)" | tee $out/summary.txt
for set in "" "gemm_epi32=0" "gemm_store32=2" "gemm_store=0"; do
  tag=${set:-default}; tag=${tag//=/_}
  MER_SET="$set" timeout 120 scripts/probes/gemm16_bench.bin 30 30 all > $out/gemm16_bench_$tag.jsonl 2>&1; echo "gemm16_bench[$tag] rc=$?" | tee -a $out/summary.txt
done
python - <<'PY' | tee -a $out/summary.txt
import json, glob
rows = {}
for f in sorted(glob.glob('gpurun_out/r2_call6/gemm16_bench_*.jsonl')):
    tag = f.split('gemm16_bench_')[1][:-6]
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if 'shape' in d and d['variant'] == 'pre-blocked W':
            rows.setdefault(d['shape'], {})[tag] = (d['us'], d['TFLOPs'])
for s, r in rows.items():
    print(s[:44].ljust(44), '  '.join(f"{t}: {v[0]:.0f}us/{v[1]:.0f}TF" for t, v in r.items()))
PY
for opt in "" "gemm_store32=2" "gemm_store=0,gemm_epi32=0"; do
  tag=${opt:-default}; tag=${tag//=/_}; tag=${tag//,/_}
  MER_OPTIONS="$opt" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "bench[$tag] rc=$? $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));r=d['roofline'];print(d['value'], r['achieved'], r['avg_launch_us'], d.get('parity'), {k:(v['ms_share'],v['tflops']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
  tail -2 $out/bench_$tag.err | grep -v amdgpu.ids | tee -a $out/summary.txt
done
timeout 900 python -m pytest tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -s > $out/enc.log 2>&1; echo "enc rc=$? $(tail -1 $out/enc.log)" | tee -a $out/summary.txt
grep -E "^\.?(hubert|roberta|clip|large|videomae)" $out/enc.log > $out/parity_lines.txt
