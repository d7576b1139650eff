#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (MI355X_MICROARCH.md §HBM / rocprofv3 PMC slots):
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots: 3 + 2 do not fit), --pmc alone with --kernel-trace.
# Output: gpurun_out/pmc/{fetch,write}/…counter_collection.csv  -> scripts/pmc_summarize.py
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc/$c
  rm -rf "$d"
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- \
     python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-large --no-sustained --no-parity --e2e 0 > "$R/gpurun_out/pmc/$c.log" 2>&1; echo "$c rc=$?")
  ls "$d" | head
done
python scripts/pmc_summarize.py gpurun_out/pmc | tee gpurun_out/pmc/summary.txt
find gpurun_out/pmc -name "*.csv" -size +30M -delete
