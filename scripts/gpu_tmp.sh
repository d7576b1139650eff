cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6dec; mkdir -p $O
MER_DECOMP=1 timeout 300 scripts/probes/gemm16_bench.bin 20 20 clip > $O/decomp_clip.jsonl 2>&1; echo rc=$?
grep variant $O/decomp_clip.jsonl | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(f\"{d['shape'][:30]:30s} {d['variant'][:48]:48s} {d['us']:8.1f}\")"
