#!/bin/bash
# rocprofv3 kernel stats of the bench step + PMC HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), guarded by a smoke run.
set -u
out=gpurun_out/r2_profile
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
rm -rf $out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/prof" -o trace -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --streams 0 > "$R/$out/prof.log" 2>&1; echo "prof rc=$?" | tee -a "$R/$out/summary.txt")
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" $out/kernel_stats.csv && head -14 $out/kernel_stats.csv | cut -c1-200
find $out/prof -name "*kernel_trace.csv" -delete
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc/$c
  rm -rf "$d"
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- \
     python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > "$R/gpurun_out/pmc/$c.log" 2>&1; echo "$c rc=$?" | tee -a "$R/$out/summary.txt")
done
python scripts/pmc_summarize.py gpurun_out/pmc > $out/pmc_summary.txt 2>&1; head -12 $out/pmc_summary.txt | cut -c1-160
cp gpurun_out/pmc/summary.json $out/pmc_summary.json
find gpurun_out/pmc -name "*.csv" -size +5M -delete
