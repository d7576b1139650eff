#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, bench, rocprof — each step under its own timeout,
# logs into gpurun_out/ (merged back by gpurun).  Usage: scripts/gpu_suite.sh [tests|bench|prof|all]
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name ===" | tee -a gpurun_out/suite.log
  timeout "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/suite.log
  tail -n 25 "gpurun_out/$name.log"
}
if [[ $what == tests || $what == all ]]; then
  [[ -x scripts/probes/abi_selftest.bin ]] && run abi_selftest 60 scripts/probes/abi_selftest.bin   # C ABI without Python (seconds)
  run ops 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider
  run enc 900 python -m pytest tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -s
  run fusion 900 python -m pytest tests/test_fusion_gpu.py tests/test_extract_gpu.py tests/test_dinov2.py tests/test_affectgpt.py tests/test_whisper.py -m gpu -q --no-header -p no:cacheprovider
  run smoke 300 python __graft_entry__.py smoke
fi
if [[ $what == bench || $what == all ]]; then
  run bench 900 python bench.py --steps 5 --warmup 2
fi
if [[ $what == prof || $what == all ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o trace -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-large --no-sustained --no-parity --e2e 0 --streams 0 > "$OLDPWD/gpurun_out/prof.log" 2>&1; echo "prof rc=$?" | tee -a "$OLDPWD/gpurun_out/suite.log")
  find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && head -30 "$f"
  # keep only the small summaries
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
cat gpurun_out/suite.log
