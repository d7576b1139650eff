#!/bin/bash
set -u
out=gpurun_out/r2_call4
mkdir -p $out
export TMPDIR=/tmp
timeout 120 scripts/probes/store_probe.bin > $out/store_probe.jsonl 2>&1; echo "store_probe rc=$?" | tee $out/summary.txt
cat $out/store_probe.jsonl | tee -a $out/summary.txt
timeout 600 python bench.py --config large --steps 5 --warmup 2 > $out/bench_large.json 2> $out/bench_large.err
echo "bench[large] rc=$? $(python -c "import json;d=json.load(open('$out/bench_large.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['whole_step_tflops'], d.get('parity'), d.get('cpu_baseline',{}).get('value'))" 2>/dev/null)" | tee -a $out/summary.txt
tail -3 $out/bench_large.err | tee -a $out/summary.txt
timeout 600 python scripts/run_config4.py --steps 10 --warmup 2 > $out/config4.json 2> $out/config4.err; echo "config4 rc=$? $(cat $out/config4.json)" | tee -a $out/summary.txt
tail -3 $out/config4.err | tee -a $out/summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $out/bench_base.json 2> $out/bench_base.err
echo "bench[base] rc=$? $(python -c "import json;d=json.load(open('$out/bench_base.json'));r=d['roofline'];print(d['value'], r['achieved'], d.get('parity'), json.dumps(d.get('cpu_baseline'))[:600])" 2>/dev/null)" | tee -a $out/summary.txt
tail -3 $out/bench_base.err | tee -a $out/summary.txt
