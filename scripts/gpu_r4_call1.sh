#!/bin/bash
# round 4, GPU call 1: the persistent GEMM (bit-equality tests, A/B timing, decomposition), the heterogeneous-batch parity tests, headline A/B
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c1; mkdir -p "$O"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider -k "persistent or block_pack" > "$O/gemm_tests.log" 2>&1; echo "gemm tests rc=$?"; tail -5 "$O/gemm_tests.log"
MER_DECOMP=1 timeout 300 scripts/probes/gemm16_bench.bin 20 20 all > "$O/gemm16_bench.jsonl" 2>&1; echo "gemm16_bench rc=$?"
python - "$O" <<'P'
import json, sys
for l in open(sys.argv[1] + "/gemm16_bench.jsonl"):
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    if "shape" in d: print(f'{d["shape"][:44]:44s} {d["variant"][:58]:58s} {d["us"]:8.1f} us {d["TFLOPs"]:6.0f} TF')
P
timeout 900 python -m pytest tests/test_parity_hardening_gpu.py -m gpu -q -s --no-header -p no:cacheprovider > "$O/parity_hardening.log" 2>&1; echo "parity hardening rc=$?"
grep -E "utt|frame|passed|failed|Error" "$O/parity_hardening.log" | grep -v "^tests\|def \|assert \|print(" | tail -40
for persist in 1 0; do
  MER_OPTIONS="gemm_persist=$persist" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/bench_persist$persist.json" 2> "$O/bench_persist$persist.err"; echo "bench persist=$persist rc=$?"
  python - "$O/bench_persist$persist.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["parity"], r["kernel"], r["achieved"], r["frac"], r["whole_step_tflops"], {k: (round(v["frac_time"], 3), round(v["tflops"] or 0)) for k, v in (r.get("kernels") or {}).items()} if isinstance(r.get("kernels"), dict) else "")
except Exception as e:
    print("bench parse failed", e)
P
done
