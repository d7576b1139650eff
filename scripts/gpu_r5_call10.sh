#!/bin/bash
# round 5, call 10: the headline step at 1 / 8 / 16 clips per step (re-take of round 4's small-batch line on the frozen kernels)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c10; mkdir -p "$O"
for B in 1 8 16; do
  timeout 200 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-sustained --e2e 0 > "$O/bench_b$B.json" 2>> "$O/bench.err"; echo "b$B rc=$?"
done
python - <<'P'
import json
out={}
for B in (1,8,16):
    d=json.loads([l for l in open(f"gpurun_out/r5c10/bench_b{B}.json") if l.startswith("{")][-1])
    out[f"batch_{B}"]={"clips_per_s":d["value"],"ms_per_step":d["ms_per_step"],"parity":d["parity"],"whole_step_frac":d["roofline"]["whole_step_frac"]}
    print(B, out[f"batch_{B}"])
json.dump(out, open("gpurun_out/r5c10/small_batches.json","w"), indent=1)
P
