#!/bin/bash
# First GPU call of the next round: runs what round 1 wrote after its GPU budget ran out, then A/Bs the opt-in switches.
#   1. the gated tests (MER_EXPERIMENTAL=1): blocked fc1 -> fc2 activation plane (bit-exact vs row-major), tf_ablk option,
#      the tri-modal pipeline against direct encoder calls, the Pillow-exact GPU resize (bytes vs PIL, driver vs host path)
#   2. bench.py default vs MER_OPTIONS=tf_ablk=1 (probe estimate: -15 % on CLIP's fc2 = +1.5-2 % clips/s)
# Everything lands in gpurun_out/ab_round2/.  Budget: ~2 GPU-minutes.
set -u
out=gpurun_out/ab_round2
mkdir -p $out
export TMPDIR=/tmp
# 0. no-Python checks first (seconds): C-ABI self-test, then the real GEMM kernels on the bench shapes (row-major W / pre-blocked W /
#    blocked activation planes) — scripts/probes/build_probes.sh must have been run before gpurun ships the tree
timeout 60 scripts/probes/abi_selftest.bin > $out/abi_selftest.jsonl 2>&1; echo "abi_selftest rc=$?" | tee $out/summary.txt
timeout 120 scripts/probes/gemm16_bench.bin 30 30 all > $out/gemm16_bench.jsonl 2>&1; echo "gemm16_bench rc=$?" | tee -a $out/summary.txt
MER_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_extract_gpu.py -m gpu -q --no-header -p no:cacheprovider \
  -k "blocked_activation or tf_ablk or trimodal or resize" > $out/gated_tests.log 2>&1
echo "gated tests rc=$?" | tee -a $out/summary.txt
tail -5 $out/gated_tests.log
for opt in "" "tf_ablk=1"; do
  tag=${opt:-default}; tag=${tag//=/_}
  MER_OPTIONS="$opt" timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "bench[$tag] rc=$? $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));print(d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" 2>/dev/null)" | tee -a $out/summary.txt
done
