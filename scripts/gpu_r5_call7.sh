#!/bin/bash
# round 5, call 7: smoke() with the persistent-GEMM self-test, the audio driver reading PCM16 into pinned memory (driver tests + e2e)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c7; mkdir -p "$O"
timeout 300 python __graft_entry__.py smoke > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; grep smoke "$O/smoke.log"
timeout 900 python -m pytest tests/test_extract_gpu.py tests/test_from_hf_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "audio or device_preprocess or trimodal or async_save" > "$O/t.log" 2>&1; echo "tests rc=$?"; tail -4 "$O/t.log"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-large --no-sustained --e2e 1024 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r5c7/bench.json") if l.startswith("{")][-1])
    e=d.get("e2e",{}); c=e.get("cold") or {}
    print("value", d["value"], "e2e", e.get("clips_per_s"), e.get("frac_of_kernel_only"), "cold", c.get("clips_per_s"), c.get("per_modality_seconds"), "same bytes", e.get("byte_identical_to_sync_path"))
    for m,v in e.get("per_modality",{}).items(): print(m, v)
except Exception as ex: print("no bench line", ex)
P
tail -3 "$O/bench.err"
