#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c3; mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "persistent or block_pack" > "$O/gemm_tests.log" 2>&1; echo "gemm tests rc=$?"; grep -E "passed|failed|^FAILED|AssertionError: persistent" "$O/gemm_tests.log" | cut -c1-420 | tail -14
MER_STAMP="$O" MER_DECOMP=1 timeout 300 scripts/probes/gemm16_bench.bin 20 20 clip > "$O/gemm16_bench_clip.jsonl" 2>&1; echo "gemm16_bench rc=$?"
timeout 300 scripts/probes/gemm16_bench.bin 20 20 hubert > "$O/gemm16_bench_hubert.jsonl" 2>&1
timeout 300 scripts/probes/gemm16_bench.bin 20 20 roberta > "$O/gemm16_bench_roberta.jsonl" 2>&1
python - "$O" <<'P'
import json, sys
for f in ("clip", "hubert", "roberta"):
    for l in open(sys.argv[1] + f"/gemm16_bench_{f}.jsonl"):
        try: d = json.loads(l)
        except Exception: print(l.strip()); continue
        if "us" in d: print(f'{d["shape"][:40]:40s} {d["variant"][:50]:50s} {d["us"]:8.1f} us {d["TFLOPs"]:6.0f} TF')
P
python scripts/gemm16p_timeline.py "$O"/stamps_*.bin > "$O/timeline.txt" 2>&1; cat "$O/timeline.txt"
