#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c4; mkdir -p "$O"
export TMPDIR=/tmp
for persist in 1 0; do
  MER_OPTIONS="gemm_persist=$persist" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/bench_persist$persist.json" 2> "$O/bench_persist$persist.err"; echo "bench persist=$persist rc=$?"
  python - "$O/bench_persist$persist.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["parity"], r["kernel"], r["achieved"], r["frac"], r["whole_step_tflops"])
    print({k: (round(v.get("frac_time", 0), 3), round(v.get("tflops") or 0)) for k, v in (r.get("kernels") or {}).items()} if isinstance(r.get("kernels"), dict) else r.keys())
except Exception as e:
    print("bench parse failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
P
done
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider --deselect tests/test_parity_hardening_gpu.py > "$O/suite.log" 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" "$O/suite.log" | tail -6
