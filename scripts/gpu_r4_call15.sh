#!/bin/bash
# phase stagger of the persistent GEMM (the `gemm_stagger` switch and the probe's MER_STAGGER_AB leg were removed with the experiment: this script is the record of how profiles/r04_gemm16_stagger_ab.txt was taken and no longer runs as is)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c15; mkdir -p "$O"
export TMPDIR=/tmp
for set in clip hubert; do
  MER_STAGGER_AB=1 timeout 200 scripts/probes/gemm16_bench.bin 20 20 $set > "$O/ab_$set.jsonl" 2>&1; echo "ab $set rc=$?"
done
python - "$O" <<'P'
import json, sys, glob, collections
for f in sorted(glob.glob(sys.argv[1] + "/ab_*.jsonl")):
    rows = collections.OrderedDict()
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "variant" in d: rows.setdefault(d["shape"], collections.OrderedDict()).setdefault(d["variant"][:28], []).append(d["TFLOPs"])
    for sh, v in rows.items():
        print(f"{sh[:44]:44s} " + "  ".join(f"{k[:26]}: {'/'.join(str(int(x)) for x in xs)}" for k, xs in v.items() if not k.startswith("4 rot")))
P
timeout 600 python -m pytest tests/test_ops_gpu.py -k "persistent" -m gpu -q --no-header -p no:cacheprovider > "$O/tests.log" 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error" "$O/tests.log" | tail -5
for st in 0 -1; do
  MER_OPTIONS="gemm_stagger=$st" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-parity --e2e 0 > "$O/bench_st$st.json" 2>> "$O/bench.err"; echo "bench stagger=$st rc=$?"
  MER_OPTIONS="gemm_stagger=$st" timeout 300 python bench.py --modalities v --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --no-parity --e2e 0 > "$O/bench_v_st$st.json" 2>> "$O/bench.err"; echo "bench v stagger=$st rc=$?"
done
python - "$O" <<'P'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], r["whole_step_tflops"])
    except Exception as e:
        print(f, "parse failed", e)
P
tail -3 "$O/bench.err"
