#!/bin/bash
# seq_bias latency kernels: targeted tests, headline, per-config lines
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c11; mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_parity_hardening_gpu.py "tests/test_encoders_gpu.py::test_bert_tiny_hidden_states" -m gpu -q -s --no-header -p no:cacheprovider > "$O/tests.log" 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" "$O/tests.log" | tail -8
grep -E "(utt|frames?)[ =]" "$O/tests.log" | grep -v "print(" | cut -c1-250 | tail -20
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
for cfg in "a 32" "a 64" "v 64" "t 64"; do set -- $cfg
  timeout 200 python bench.py --modalities $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-parity --e2e 0 > "$O/bench_$1_b$2.json" 2>> "$O/bench.err"; echo "bench $1 b$2 rc=$?"
done
python - "$O" <<'P'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("parity"), r["kernel"], r["achieved"], r["frac"], r["whole_step_tflops"], {k: (v["ms_share"], v["tflops"], v["gbps"]) for k, v in r["other_kernels"].items()})
    except Exception as e:
        print(f, "parse failed", e)
P
tail -5 "$O/bench.err"
