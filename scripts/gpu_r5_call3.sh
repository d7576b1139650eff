#!/bin/bash
# round 5, call 3: full live-HF file, outlier sweep with the a2 / fp32-attention study presets, text driver tests, quick headline + e2e bench
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c3; mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_from_hf_gpu.py -m gpu -q --no-header -p no:cacheprovider -s > "$O/from_hf.log" 2>&1; echo "from_hf rc=$?"
grep -E "from_hf\[|by name|vs the live|passed|failed|Error|error" "$O/from_hf.log" | grep -v Warning | tail -40
timeout 900 python -m pytest tests/test_encoders_gpu.py tests/test_extract_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "activation_outliers_post_ln and hubert or text_extract" > "$O/enc.log" 2>&1; echo "enc rc=$?"
grep -E "outliers|text driver|passed|failed|Error|assert" "$O/enc.log" | grep -v Warning | tail -40
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-large --no-sustained --e2e 1024 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r5c3/bench.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity"))
    e=d.get("e2e",{})
    print("e2e", e.get("clips_per_s"), e.get("frac_of_kernel_only"), "cold", (e.get("cold") or {}).get("clips_per_s"), (e.get("cold") or {}).get("per_modality_seconds"))
    for m,v in e.get("per_modality",{}).items(): print(m, v)
    r=d.get("roofline",{}); print({k:r.get(k) for k in ("achieved","frac","traffic","whole_step_frac","whole_step_frac_executed")})
except Exception as ex: print("no bench line", ex)
P
tail -5 "$O/bench.err"
