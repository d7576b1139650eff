#!/bin/bash
# round 5, call 9: the multi-GPU code path with a one-rank RCCL group on this 1-GPU box (bench.py --force-dist), and the whole GPU suite
# on the final Python tree (the kernels are the frozen ones: sha 3414b1fa25665e00)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c9; mkdir -p "$O"
export TMPDIR=/tmp
timeout 300 python bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-large --no-sustained --e2e 0 > "$O/bench_force_dist.json" 2> "$O/bench_force_dist.err"; echo "force-dist rc=$?"
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r5c9/bench_force_dist.json") if l.startswith("{")][-1]); print("force-dist", d["value"], d["ms_per_step"], d.get("rccl_ranks"), d.get("allgather"), d["config"].get("parallelism"))
except Exception as ex: print("no line", ex)
P
tail -2 "$O/bench_force_dist.err"
timeout 300 python scripts/run_config4.py --steps 10 > "$O/config4.json" 2> "$O/config4.err"; echo "config4 rc=$?"; tail -2 "$O/config4.json" | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > "$O/suite.log" 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" "$O/suite.log" | tail -8
