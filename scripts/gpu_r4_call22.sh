#!/bin/bash
# small batches: the reference's own mode (one clip per forward) and 8 clips per step
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c22; mkdir -p "$O"
for b in 1 8 16; do
  timeout 200 python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-sustained --no-large --no-roofline --e2e 0 > "$O/bench_b$b.json" 2>> "$O/bench.err"; echo "b$b rc=$?"
done
python - "$O" <<'P'
import json, sys
for b in (1, 8, 16):
    d = json.load(open(f"{sys.argv[1]}/bench_b{b}.json")); print(b, d["value"], d["ms_per_step"], d["parity"])
P
