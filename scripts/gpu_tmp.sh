cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6e1; mkdir -p $O
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sustained --no-large --no-ladder --e2e 256 > $O/bench_e2e.json 2> $O/err.log; echo rc=$?
python - $O/bench_e2e.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); e=d["e2e"]
print("value", d["value"]); print("e2e warm", e.get("clips_per_s"), e.get("frac_of_kernel_only"), "3 threads", e.get("three_threads_at_once"))
print("tri", e.get("trimodal_pipeline")); print("cold", (e.get("cold") or {}).get("clips_per_s"))
P
tail -3 $O/err.log
