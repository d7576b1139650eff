#!/bin/bash
# One library, several settings of its mer_set_option switches (MER_OPTIONS, read by mertools_amd/_lib.py at load), alternating on ONE GPU
# box: what a switch is worth in the step.  Usage: bash scripts/gpu_ab_options.sh <tag> "<modalities ...>" "<options A>" "<options B>" ...
# ("-" = the defaults; modalities: any subset string of avt, a32 = audio at batch 32, large / vlarge = the large trio / its visual tower), e.g.  bash scripts/gpu_ab_options.sh ab1 "avt v a32 t" - seq_bias_fused=0 attn_handout=0
tag=$1; mods=$2; shift 2; cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/$tag; mkdir -p "$O"
Q="--no-cpu-baseline --no-sustained --no-large --no-ladder --no-parity --no-roofline --e2e 0 --steps 20 --warmup 5"
for round in 1 2 3; do
  for opt in "$@"; do
    o=$opt; [[ $opt == - ]] && o=""
    for m in $mods; do
      extra="--modalities $m"; [[ $m == a32 ]] && extra="--modalities a --batch 32"; [[ $m == large ]] && extra="--config large --steps 8 --warmup 2"; [[ $m == vlarge ]] && extra="--config large --modalities v --steps 8 --warmup 2"
      MER_OPTIONS=$o timeout 200 python bench.py $Q $extra > "$O/${opt//[=,]/_}_${m}_$round.json" 2>> $O/err.log
      python - "$O/${opt//[=,]/_}_${m}_$round.json" "$opt" $m $round <<'P'
import json, sys
try:
    x = json.load(open(sys.argv[1])); print(sys.argv[2].ljust(34), sys.argv[3].ljust(4), "round", sys.argv[4], x["value"], x["ms_per_step"], flush=True)
except Exception as ex: print(sys.argv[2], sys.argv[3], "failed", ex, flush=True)
P
    done
  done
done
