#!/bin/bash
# One GPU-box call that produces a round's evidence: C-ABI self-test, the whole GPU suite, the default bench line (with the large-trio
# sub-object, the sustained line, the CPU baseline and the e2e drivers), the per-config lines of BASELINE configs[1] / [2], the kernel
# stats of the step (rocprofv3 --kernel-trace --stats) and the standalone GEMM table.  Usage (via gpurun): bash scripts/gpu_round.sh <tag>
tag=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/$tag; mkdir -p "$O"
export TMPDIR=/tmp
timeout 60 scripts/probes/abi_selftest.bin > "$O/abi_selftest.jsonl" 2>&1; echo "abi rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -s --no-header -p no:cacheprovider > "$O/suite.log" 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" "$O/suite.log" | tail -6
grep -E "(utt|frames?)( diff)?=|self-check|escalated" "$O/suite.log" | grep -v "print(" > "$O/parity_lines.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities a --batch 32 --no-cpu-baseline --no-sustained --e2e 0 > "$O/bench_audio_b32.json" 2>> "$O/bench.err"; echo "audio rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities a --no-cpu-baseline --no-sustained --e2e 0 > "$O/bench_audio_b64.json" 2>> "$O/bench.err"; echo "audio64 rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities v --no-cpu-baseline --no-sustained --e2e 0 > "$O/bench_visual_b64.json" 2>> "$O/bench.err"; echo "visual rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities t --no-cpu-baseline --no-sustained --e2e 0 > "$O/bench_text_b64.json" 2>> "$O/bench.err"; echo "text rc=$?"
for prec in mx balanced accurate; do   # the other presets on the same box (A/B of the default)
  timeout 300 python bench.py --steps 20 --warmup 5 --precision $prec --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/bench_$prec.json" 2>> "$O/bench.err"; echo "$prec rc=$?"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o step -- python "$OLDPWD/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-roofline --no-large --no-sustained --e2e 0 --streams 0 > /dev/null 2>&1; echo "prof rc=$?")
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" "$O/kernel_stats.csv"; rm -rf "$O/prof"
timeout 200 scripts/probes/gemm16_bench.bin 20 20 all > "$O/gemm16_bench.jsonl" 2>&1; echo "gemm16_bench rc=$?"
python - "$O" <<'P'
import json, sys
O = sys.argv[1]
d = json.load(open(f"{O}/bench.json"))
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], d["parity"], "| dominant", r["kernel"], r["achieved"], r["frac"], "whole step", r["whole_step_tflops"], r["whole_step_frac"], "mfma_busy", (r.get("mfma_busy") or {}).get("mfma_busy"), "traffic", r["traffic"])
print("sustained", d.get("sustained")); L = d.get("large") or {}; print("large", L.get("value"), L.get("whole_step_frac"), L.get("parity"))
c = d.get("cpu_baseline", {}); print("cpu", c.get("value"), c.get("cores"), c.get("threads_tried")); e = d.get("e2e", {}); print("e2e warm", e.get("clips_per_s"), e.get("frac_of_kernel_only"), "cold", e.get("cold"), e.get("three_threads_at_once"))
for n in ("audio_b32", "audio_b64", "visual_b64", "text_b64", "mx", "balanced", "accurate"):
    x = json.load(open(f"{O}/bench_{n}.json")); print(n, x["value"], x["roofline"]["whole_step_tflops"], x["roofline"]["whole_step_frac"], x["parity"])
P
