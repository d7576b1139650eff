#!/bin/bash
# What the single-pass attention kernel spends its SIMD cycles on: SQ counters over scripts/attn_phases.py's three shapes
# (CLIP 512 x 197, HuBERT 64 x 249, RoBERTa 64 x 64; 12 heads), each --pmc set in its own rocprofv3 pass with --kernel-trace only.
# Output: gpurun_out/pmc_attn/summary.txt (per kernel: counters per launch and, against SQ_BUSY_CYCLES, the share of busy SIMD cycles
# an instruction class was issuing in).
set -u
export TMPDIR=/tmp
R=$PWD
d=$R/gpurun_out/pmc_attn
rm -rf "$d"; mkdir -p "$d"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
           "SQ_BUSY_CYCLES SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i + 1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$d/p$i" -o pmc -- python "$R/scripts/attn_phases.py" > "$d/run$i.log" 2>&1; echo "pmc_attn pass $i rc=$?")
done
python - "$d" <<'P' | tee "$d/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "attn_sp_kernel" not in n: continue
        k = "attn_sp<NKT=%s,NW=%s>" % (n.split("Li")[1].split("E")[0], n.split("Li")[2].split("E")[0])
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    busy = m.get("SQ_BUSY_CYCLES", 0)
    print(k, "launches", len(v.get("SQ_BUSY_CYCLES", [])))
    for c in sorted(m):
        print(f"   {c:34s} {m[c]:16.0f}" + (f"   {m[c] / busy:8.3f} x SQ_BUSY_CYCLES" if busy and c != "SQ_BUSY_CYCLES" else ""))
P
find "$d" -name "*.csv" -size +30M -delete
