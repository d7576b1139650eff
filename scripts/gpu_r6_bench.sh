#!/bin/bash
# round 6: the headline step and the per-modality lines (kernel-only), + rocprofv3 kernel stats of the headline / audio-b32 steps.  Usage: bash scripts/gpu_r6_bench.sh <tag> [noprof]
tag=${1:-r6b}; cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/$tag; mkdir -p "$O"; export TMPDIR=/tmp; R=$PWD
Q="--no-cpu-baseline --no-sustained --no-large --no-ladder --e2e 0"
timeout 300 python bench.py --steps 20 --warmup 5 $Q > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities v $Q > $O/bench_visual_b64.json 2>> $O/bench.err; echo "visual rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities a --batch 32 $Q > $O/bench_audio_b32.json 2>> $O/bench.err; echo "audio32 rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities a $Q > $O/bench_audio_b64.json 2>> $O/bench.err; echo "audio64 rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --modalities t $Q > $O/bench_text_b64.json 2>> $O/bench.err; echo "text rc=$?"
python - "$O" <<'P'
import json, sys
O = sys.argv[1]
for n in ("bench", "bench_visual_b64", "bench_audio_b32", "bench_audio_b64", "bench_text_b64"):
    try:
        x = json.load(open(f"{O}/{n}.json")); r = x["roofline"]
        print(n, x["value"], x["ms_per_step"], "dominant", r["kernel"], r["achieved"], r["frac"], "whole", r["whole_step_tflops"], r["whole_step_frac"], x["parity"])
    except Exception as ex: print(n, "failed", ex)
P
[[ "$2" == noprof ]] && exit 0
prof() {
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_$name" -o step -- python "$R/bench.py" "$@" --no-cpu-baseline --no-parity --no-roofline --no-sustained --no-ladder --e2e 0 --streams 0 > /dev/null 2>&1; echo "prof $name rc=$?")
  f=$(find "$O/prof_$name" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" "$O/${name}_kernel_stats.csv"; rm -rf "$O/prof_$name"
}
prof headline --steps 4 --warmup 1 --no-large
prof audio_b32 --modalities a --batch 32 --steps 8 --warmup 2 --no-large
