#!/bin/bash
# round 5, call 4: the self-check ladder on the HF-initialised data2vec-audio module, attention ops after the rescale skip, the persistent
# GEMM on the guide's calibration shapes, the large trio and the cold e2e
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c4; mkdir -p "$O"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_from_hf_gpu.py tests/test_ops_gpu.py tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "other_architectures and (data2vec or wav2vec2 or wavlm) or attention or large_trio or activation_outliers_post_ln" > "$O/t.log" 2>&1; echo "tests rc=$?"
grep -E "from_hf\[|self-check|outliers \[default|large|passed|failed|Error|assert" "$O/t.log" | grep -v Warning | cut -c1-330 | tail -40
timeout 120 scripts/probes/gemm16_bench.bin 20 20 square > "$O/gemm16_square.jsonl" 2>&1; echo "square rc=$?"; grep -v "tile kernel\|rotating" "$O/gemm16_square.jsonl" | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sustained --e2e 1024 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r5c4/bench.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity"))
    L=d.get("large") or {}; print("large", L.get("value"), L.get("ms_per_step"), L.get("parity"), (L.get("roofline") or {}).get("other_kernels",{}).get("attention"))
    e=d.get("e2e",{}); c=e.get("cold") or {}
    print("e2e", e.get("clips_per_s"), e.get("frac_of_kernel_only"), "cold", c.get("clips_per_s"), c.get("per_modality_seconds"), "kernel-only", e.get("kernel_only_clips_per_s_same_schedule"))
    r=d.get("roofline",{}); print({k:r.get(k) for k in ("kernel","achieved","frac","launches","avg_launch_us","share_of_gpu_time","whole_step_frac")})
    print({k:v for k,v in r.get("other_kernels",{}).items()})
except Exception as ex: print("no bench line", ex)
P
tail -3 "$O/bench.err"
