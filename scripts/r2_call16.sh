#!/bin/bash
set -u
out=gpurun_out/r2_call16
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['traffic'], d.get('parity'), d['cpu_baseline']['value'])" 2>/dev/null)" | tee -a $out/summary.txt
rm -rf $out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/prof" -o trace -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --streams 0 > "$R/$out/prof.log" 2>&1; echo "prof rc=$?" | tee -a "$R/$out/summary.txt")
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" $out/kernel_stats.csv && head -8 $out/kernel_stats.csv | cut -c1-160
find $out/prof -name "*kernel_trace.csv" -delete
