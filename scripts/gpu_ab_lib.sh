#!/bin/bash
# Two builds of libmer_hip.so on ONE GPU box, alternating (box-to-box spread is +-2 %, a round's gain can be less): the library given as
# $1 (e.g. scratch/ab/libmer_hip_r5.so, built from an older tree) against the tree's own.  Usage: bash scripts/gpu_ab_lib.sh <other.so> <tag>
other=$1; tag=${2:-ab}; cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/$tag; mkdir -p "$O"
Q="--no-cpu-baseline --no-sustained --no-large --no-ladder --no-parity --no-roofline --e2e 0 --steps 20 --warmup 5"
for round in 1 2 3; do
  for which in other tree; do
    lib=""; [[ $which == other ]] && lib="$PWD/$other"
    for m in avt v a; do
      MER_LIB_PATH=$lib timeout 200 python bench.py $Q --modalities $m > $O/${which}_${m}_$round.json 2>> $O/err.log
      python - "$O/${which}_${m}_$round.json" $which $m $round <<'P'
import json, sys
x = json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], "round", sys.argv[4], x["value"], x["ms_per_step"])
P
    done
  done
done
