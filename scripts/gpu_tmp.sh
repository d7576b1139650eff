cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6s2; mkdir -p $O
MER_BOUNDARY_AB=1 timeout 300 scripts/probes/gemm16_bench.bin 20 20 clip > $O/bench_clip.jsonl 2>&1; echo "rc=$?"
grep -h variant $O/bench_clip.jsonl | python3 -c "
import sys,json,collections
d=collections.OrderedDict()
for l in sys.stdin:
    x=json.loads(l); d.setdefault((x['shape'][:34],x['variant'][:44]),[]).append(x['us'])
for k,v in d.items(): print(f'{k[0]:34s} {k[1]:44s}', ' '.join(f'{u:7.1f}' for u in v))"
