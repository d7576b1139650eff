#!/bin/bash
# MFMA-utilisation counters of the bench's kernels (north_star: "rocprof HBM GB/s and MFMA utilisation"): SQ counters in their own
# rocprofv3 pass, --pmc with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots: SQ has 8).
# Output: gpurun_out/pmc_mfma/…counter_collection.csv -> scripts/pmc_mfma_summarize.py -> gpurun_out/pmc_mfma/summary.json
set -u
export TMPDIR=/tmp
R=$PWD
d=$R/gpurun_out/pmc_mfma
rm -rf "$d"; mkdir -p "$d"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d "$d" -o pmc -- \
   python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-large --no-sustained --no-parity --e2e 0 --streams 0 > "$d/run.log" 2>&1; echo "pmc_mfma rc=$?")
python scripts/pmc_mfma_summarize.py "$d" | tee "$d/summary.txt"
find "$d" -name "*.csv" -size +30M -delete
