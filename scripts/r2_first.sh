#!/bin/bash
# Round-2 first GPU call: whole GPU suite (un-gated tests, bench-tile parity, ragged audio), then bench default vs tf_ablk.
set -u
out=gpurun_out/r2_first
mkdir -p $out
export TMPDIR=/tmp
timeout 60 scripts/probes/abi_selftest.bin > $out/abi_selftest.jsonl 2>&1; echo "abi_selftest rc=$?" | tee $out/summary.txt
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s --durations=15 > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/summary.txt
grep -E "passed|failed|error" $out/pytest.log | tail -3 | tee -a $out/summary.txt
grep -E "^(hubert|roberta|clip|large|videomae|wavlm|data2vec)" $out/pytest.log > $out/parity_lines.txt
for opt in "" "tf_ablk=1"; do
  tag=${opt:-default}; tag=${tag//=/_}
  MER_OPTIONS="$opt" timeout 300 python bench.py --steps 10 --warmup 3 $( [[ -n "$opt" ]] && echo --no-cpu-baseline ) > $out/bench_$tag.json 2> $out/bench_$tag.err
  echo "bench[$tag] rc=$? $(python -c "import json;d=json.load(open('$out/bench_$tag.json'));print(d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d.get('parity'))" 2>/dev/null)" | tee -a $out/summary.txt
done
grep -E "FAILED|Error" $out/pytest.log | head -20
