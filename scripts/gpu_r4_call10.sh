#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c10; mkdir -p "$O"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s --no-header -p no:cacheprovider > "$O/suite.log" 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" "$O/suite.log" | tail -8
grep -E "(utt|frames?)=" "$O/suite.log" | grep -v "print(" > "$O/parity_lines.txt"; grep -E "outliers|self-check" "$O/parity_lines.txt" | cut -c1-330
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - "$O/bench.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["parity"], r["kernel"], r["achieved"], r["frac"], r["whole_step_tflops"]); print(r["other_kernels"])
except Exception as e:
    print("bench parse failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2500:])
P
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o step -- python "$OLDPWD/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-roofline --no-large --no-sustained --e2e 0 --streams 0 > /dev/null 2>&1; echo "prof rc=$?")
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" "$O/kernel_stats.csv"; rm -rf "$O/prof"; head -16 "$O/kernel_stats.csv" | cut -c1-170
