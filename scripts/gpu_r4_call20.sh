#!/bin/bash
# the default bench line with the committed PMC collections in place (traffic / mfma_busy filled)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c20; mkdir -p "$O"
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - "$O/bench.json" <<'P'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(d["value"], d["ms_per_step"], d["parity"], r["achieved"], r["frac"], r["traffic"], (r.get("mfma_busy") or {}).get("mfma_busy"), r["whole_step_frac"], r["whole_step_frac_executed"])
print("large", d["large"]["value"], "e2e", d["e2e"]["clips_per_s"], "cold", d["e2e"]["cold"]["clips_per_s"], "sustained", d["sustained"]["clips_per_s"])
P
