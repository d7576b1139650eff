#!/bin/bash
# End-of-round validation after the attention change: smoke -> whole GPU suite -> headline bench (with cpu_baseline) -> rocprofv3 kernel stats.
# (no PMC pass: csrc/gemm16_impl.h is unchanged since profiles/r02_pmc_hbm_traffic.json was collected, its _source_sha still matches)
set -u
out=gpurun_out/r2_final2
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s --durations=10 > $out/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc $(grep -E 'passed|failed' $out/pytest.log | tail -1)" | tee -a $out/summary.txt
grep -E "^\.?(hubert|roberta|clip|large|videomae|wavlm|data2vec|whisper)" $out/pytest.log > $out/parity_lines.txt
grep -E "^(FAILED|ERROR)" $out/pytest.log | head -10 | tee -a $out/summary.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], d.get('parity'), d['cpu_baseline']['value'])" 2>/dev/null)" | tee -a $out/summary.txt
rm -rf $out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/prof" -o trace -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --streams 0 > "$R/$out/prof.log" 2>&1; echo "prof rc=$?" | tee -a "$R/$out/summary.txt")
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" $out/kernel_stats.csv
find $out/prof -name "*kernel_trace.csv" -delete
