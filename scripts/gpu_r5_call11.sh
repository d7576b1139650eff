#!/bin/bash
# round 5, call 11: the K-loop probe — 32 MFMAs per barrier phase (the product's loop) against 16 per phase (the guide's 8-phase grain), random operands
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c11; mkdir -p "$O"
timeout 150 scripts/probes/ceiling_probe.bin 10 20 phase8 > "$O/phase8.jsonl" 2>&1; echo "probe rc=$?"
python - <<'P'
import json
for l in open("gpurun_out/r5c11/phase8.jsonl"):
    if not l.startswith("{"): print(l.strip()); continue
    d=json.loads(l)
    if "probe" in d: print(f'{d["name"][:70]:70s} M={d["M"]:6d} N={d["N"]:4d} K={d["K"]:4d} us={d["us"]:7.1f} cyc/slab={d["cycles_per_slab"]:6.0f} TF={d["mfma_TFLOPs"]:5.0f}')
P
