#!/bin/bash
# round 5, call 5: A/B of the persistent GEMM with the DMA pieces issued behind the fragment reads (gemm_dbg_skip bit 2)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c5; mkdir -p "$O"
for rep in 1 2; do
  for v in 0 4; do
    MER_SET="gemm_dbg_skip=$v" timeout 100 scripts/probes/gemm16_bench.bin 20 20 clip 2>&1 | grep "persistent kernel" | sed "s/^/[skip=$v] /" >> "$O/ab_clip.txt"
  done
done
for v in 0 4; do MER_SET="gemm_dbg_skip=$v" timeout 100 scripts/probes/gemm16_bench.bin 20 20 square 2>&1 | grep "persistent kernel" | sed "s/^/[skip=$v] /" >> "$O/ab_square.txt"; done
for v in 0 4; do MER_SET="gemm_dbg_skip=$v" timeout 100 scripts/probes/gemm16_bench.bin 20 20 hubert 2>&1 | grep "persistent kernel" | sed "s/^/[skip=$v] /" >> "$O/ab_hubert.txt"; done
python - <<'P'
import json,collections,re
for f in ("ab_clip","ab_square","ab_hubert"):
    acc=collections.defaultdict(list)
    for l in open(f"gpurun_out/r5c5/{f}.txt"):
        m=re.match(r"\[skip=(\d)\] (\{.*\})",l)
        if not m: continue
        d=json.loads(m.group(2)); acc[(d["shape"][:40],m.group(1))].append(d["TFLOPs"])
    for k in sorted(acc): print(f, k, acc[k])
P
