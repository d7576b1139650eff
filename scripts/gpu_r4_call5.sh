#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c5; mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "persistent or block_pack or seq_bias or bias_table or bias_corr" > "$O/gemm_tests.log" 2>&1; echo "gemm tests rc=$?"; grep -E "passed|failed|^FAILED|AssertionError" "$O/gemm_tests.log" | cut -c1-400 | tail -14
MER_STAMP="$O" MER_DECOMP=1 timeout 300 scripts/probes/gemm16_bench.bin 20 20 clip > "$O/gemm16_bench_clip.jsonl" 2>&1; echo "gemm16_bench rc=$?"
timeout 300 scripts/probes/gemm16_bench.bin 20 20 hubert > "$O/gemm16_bench_hubert.jsonl" 2>&1
python - "$O" <<'P'
import json, sys
for f in ("clip", "hubert"):
    for l in open(sys.argv[1] + f"/gemm16_bench_{f}.jsonl"):
        try: d = json.loads(l)
        except Exception: print(l.strip()); continue
        if "us" in d: print(f'{d["shape"][:40]:40s} {d["variant"][:50]:50s} {d["us"]:8.1f} us {d["TFLOPs"]:6.0f} TF')
P
python scripts/gemm16p_timeline.py "$O"/stamps_*.bin > "$O/timeline.txt" 2>&1; cat "$O/timeline.txt"
timeout 900 python -m pytest tests/test_parity_hardening_gpu.py -m gpu -q -s --no-header -p no:cacheprovider > "$O/parity_hardening.log" 2>&1; echo "parity hardening rc=$?"
grep -E "utt|frame|passed|failed" "$O/parity_hardening.log" | grep -v "^tests\|def \|assert \|print(\|    " | tail -40
timeout 1200 python -m pytest tests/test_encoders_gpu.py -m gpu -q -s -x --no-header -p no:cacheprovider -k "bench_tiles or ragged or loud or base16 or base_5s or 64tok or large_trio or outliers" > "$O/enc.log" 2>&1; echo "enc rc=$?"
grep -E "utt=|passed|failed|^FAILED|Error" "$O/enc.log" | grep -v "print(" | tail -60
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - "$O/bench.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["parity"], r["kernel"], r["achieved"], r["frac"], r["whole_step_tflops"]); print(r["other_kernels"])
except Exception as e:
    print("bench parse failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2500:])
P
