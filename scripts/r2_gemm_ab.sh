#!/bin/bash
# Round-2 GEMM A/B: ops tests (epilogue correctness), real-kernel timings with the packed-pair epilogue on/off and the
# epilogue decomposition (dbg_skip 1 = no global stores, 2 = no epilogue at all), then bench.py, then the fusion loop tests.
set -u
out=gpurun_out/r2_gemm_ab
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider > $out/ops.log 2>&1; echo "ops rc=$? $(tail -1 $out/ops.log)" | tee $out/summary.txt
for set in "" "gemm_pkepi=0" "gemm_dbg_skip=1" "gemm_dbg_skip=2"; do
  tag=${set:-default}; tag=${tag//=/_}
  MER_SET="$set" timeout 120 scripts/probes/gemm16_bench.bin 30 30 all > $out/gemm16_bench_$tag.jsonl 2>&1; echo "gemm16_bench[$tag] rc=$?" | tee -a $out/summary.txt
done
python - <<'PY' | tee -a $out/summary.txt
import json, glob
rows = {}
for f in sorted(glob.glob('gpurun_out/r2_gemm_ab/gemm16_bench_*.jsonl')):
    tag = f.split('gemm16_bench_')[1][:-6]
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if 'shape' in d and d['variant'] == 'pre-blocked W':
            rows.setdefault(d['shape'], {})[tag] = (d['us'], d['TFLOPs'])
for s, r in rows.items():
    print(s.ljust(64), '  '.join(f"{t}: {v[0]:.0f}us/{v[1]:.0f}TF" for t, v in r.items()))
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench.json'));print(d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d.get('parity'), {k:(v['ms_share'],v['tflops']) for k,v in d['roofline']['other_kernels'].items()})" 2>/dev/null)" | tee -a $out/summary.txt
MER_OPTIONS="gemm_pkepi=0" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $out/bench_nopk.json 2> $out/bench_nopk.err
echo "bench[pkepi=0] rc=$? $(python -c "import json;d=json.load(open('$out/bench_nopk.json'));print(d['value'], d['roofline']['achieved'])" 2>/dev/null)" | tee -a $out/summary.txt
timeout 600 python -m pytest tests/test_fusion_gpu.py tests/test_extract_gpu.py -m gpu -q --no-header -p no:cacheprovider > $out/fusion_extract.log 2>&1; echo "fusion+extract rc=$? $(tail -1 $out/fusion_extract.log)" | tee -a $out/summary.txt
timeout 300 python tests/studies/bench_fusion.py > $out/bench_fusion.json 2> $out/bench_fusion.err; echo "bench_fusion rc=$? $(cat $out/bench_fusion.json)" | tee -a $out/summary.txt
timeout 300 python -m pytest tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "bench_tiles or hubert_base_5s or clip_base16_8frames" > $out/enc.log 2>&1; echo "enc rc=$? $(tail -1 $out/enc.log)" | tee -a $out/summary.txt
grep -E "^\.?(hubert|roberta|clip)" $out/enc.log | tee -a $out/summary.txt
