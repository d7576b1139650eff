#!/bin/bash
# LDS-staged fp32 attention: operator tests, accurate-preset encoder tests, accurate bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c21; mkdir -p "$O"
timeout 600 python -m pytest tests/test_ops_gpu.py -k "attention" -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-220
timeout 900 python -m pytest tests/test_encoders_gpu.py -k "outliers_post_ln or base_5s or base16_8frames or roberta_base_64tok or roberta_large_bf16" -m gpu -q -s --no-header -p no:cacheprovider > "$O/enc.log" 2>&1; echo "enc rc=$?"; grep -E "passed|failed|^FAILED|^E  " "$O/enc.log" | tail -5; grep -E "\[accurate" "$O/enc.log" | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --precision accurate --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/bench_accurate.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - "$O/bench_accurate.json" <<'P'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(d["value"], d["ms_per_step"], d["parity"], r["kernel"], r["achieved"], {k: (v["ms_share"], v["tflops"]) for k, v in r["other_kernels"].items()})
P
