#!/bin/bash
# Round 6's evidence in ONE GPU-box call, on the final tree: C-ABI self-test, the whole GPU suite, the default bench line (large trio, sustained,
# CPU baseline, e2e warm + cold), the per-config lines of BASELINE configs[1] / [2] + text, the other presets on the same box (mx, balanced,
# mean_a2, accurate), rocprofv3 kernel stats of the headline / audio-b32 / large steps, the HBM-traffic and MFMA-busy PMC passes (each --pmc alone
# with --kernel-trace) and the standalone GEMM table.  Usage (via gpurun): bash scripts/gpu_round6.sh <tag>
tag=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/$tag; mkdir -p "$O"
export TMPDIR=/tmp
R=$PWD
timeout 60 scripts/probes/abi_selftest.bin > "$O/abi_selftest.jsonl" 2>&1; echo "abi rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -s --no-header -p no:cacheprovider > "$O/suite.log" 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" "$O/suite.log" | tail -8
grep -E "(utt|frames?)( diff)?=|self-check|escalated|by name" "$O/suite.log" | grep -v "print(" > "$O/parity_lines.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities a --batch 32 --no-cpu-baseline --no-sustained --no-ladder --e2e 0 > "$O/bench_audio_b32.json" 2>> "$O/bench.err"; echo "audio rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities a --no-cpu-baseline --no-sustained --no-ladder --e2e 0 > "$O/bench_audio_b64.json" 2>> "$O/bench.err"; echo "audio64 rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities v --no-cpu-baseline --no-sustained --no-ladder --e2e 0 > "$O/bench_visual_b64.json" 2>> "$O/bench.err"; echo "visual rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities t --no-cpu-baseline --no-sustained --no-ladder --e2e 0 > "$O/bench_text_b64.json" 2>> "$O/bench.err"; echo "text rc=$?"
for prec in mx balanced; do   # the other one-plane presets on the same box (the ladder rungs ride in the default line: `ladder`)
  timeout 300 python bench.py --steps 20 --warmup 5 --precision $prec --no-cpu-baseline --no-sustained --no-large --no-ladder --e2e 0 > "$O/bench_$prec.json" 2>> "$O/bench.err"; echo "$prec rc=$?"
done
prof() {  # name, bench flags...
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_$name" -o step -- python "$R/bench.py" "$@" --no-cpu-baseline --no-parity --no-roofline --no-sustained --no-ladder --e2e 0 --streams 0 > /dev/null 2>&1; echo "prof $name rc=$?")
  f=$(find "$O/prof_$name" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" "$O/${name}_kernel_stats.csv"; rm -rf "$O/prof_$name"
}
prof headline --steps 4 --warmup 1 --no-large
prof audio_b32 --modalities a --batch 32 --steps 8 --warmup 2 --no-large
prof large --config large --steps 2 --warmup 1
# PMC passes: counters alone with --kernel-trace (never with other trace domains)
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc/$c; rm -rf "$d"; mkdir -p "$R/gpurun_out/pmc"
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-large --no-sustained --no-ladder --no-parity --e2e 0 > "$R/gpurun_out/pmc/$c.log" 2>&1; echo "$c rc=$?")
done
python scripts/pmc_summarize.py gpurun_out/pmc > "$O/pmc_hbm_traffic.txt" 2>&1; tail -12 "$O/pmc_hbm_traffic.txt"
cp gpurun_out/pmc/summary.json "$O/pmc_hbm_traffic.json" 2>/dev/null
d=$R/gpurun_out/pmc_mfma; rm -rf "$d"; mkdir -p "$d"
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$d" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-large --no-sustained --no-ladder --no-parity --e2e 0 --streams 0 > "$d/run.log" 2>&1; echo "pmc_mfma rc=$?")
python scripts/pmc_mfma_summarize.py "$d" > "$O/pmc_mfma.txt" 2>&1; tail -8 "$O/pmc_mfma.txt"
cp "$d/summary.json" "$O/pmc_mfma.json" 2>/dev/null
find gpurun_out/pmc gpurun_out/pmc_mfma -name "*.csv" -size +20M -delete
MER_CHECK=1 MER_DECOMP=1 MER_STAGGER_AB=1 timeout 500 scripts/probes/gemm16_bench.bin 20 20 all > "$O/gemm16_bench.jsonl" 2>&1; echo "gemm16_bench rc=$?"
timeout 300 python scripts/load_time_ladder.py > "$O/load_time_ladder.json" 2>/dev/null; echo "load_time rc=$?"
# the headline line once more, now that the PMC collections of THIS tree exist next to it (roofline.traffic / mfma_busy filled in)
mkdir -p profiles_tmp && cp "$O/pmc_hbm_traffic.json" profiles/r06_pmc_hbm_traffic.json 2>/dev/null; cp "$O/pmc_mfma.json" profiles/r06_pmc_mfma.json 2>/dev/null; rmdir profiles_tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-ladder --e2e 0 > "$O/bench_with_pmc.json" 2>> "$O/bench.err"; echo "bench+pmc rc=$?"
python - "$O" <<'P'
import json, sys
O = sys.argv[1]
d = json.load(open(f"{O}/bench.json"))
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], d["parity"], "| dominant", r["kernel"], r["achieved"], r["frac"], "whole step", r["whole_step_tflops"], r["whole_step_frac"])
print("sustained", d.get("sustained")); print("ladder", d.get("ladder")); L = d.get("large") or {}; print("large", L.get("value"), L.get("whole_step_frac"), L.get("parity"))
c = d.get("cpu_baseline", {}); print("cpu", c.get("value"), c.get("cores"), c.get("threads_tried")); e = d.get("e2e", {}); print("e2e warm", e.get("clips_per_s"), e.get("frac_of_kernel_only"), "cold", e.get("cold"))
for n in ("audio_b32", "audio_b64", "visual_b64", "text_b64", "mx", "balanced"):
    try:
        x = json.load(open(f"{O}/bench_{n}.json")); print(n, x["value"], x["roofline"]["whole_step_tflops"], x["roofline"]["whole_step_frac"], x["parity"])
    except Exception as ex: print(n, "failed", ex)
try:
    x = json.load(open(f"{O}/bench_with_pmc.json"))["roofline"]; print("with pmc: traffic", x["traffic"], x["traffic_detail"], "mfma_busy", (x.get("mfma_busy") or {}).get("mfma_busy"))
except Exception as ex: print("with pmc failed", ex)
P
