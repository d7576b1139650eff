#!/usr/bin/env python3
"""Aggregates a rocprofv3 SQ counter pass (scripts/pmc_mfma.sh) per kernel: mean per launch of every counter, and
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
i.e. the fraction of the launch's SIMD-cycles in which the matrix pipe was busy.  Calibration on this chip (profiles/r03_pmc_mfma.txt):
SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA exactly for the 16x16x32 f16 kernels (16 cycles per MFMA per SIMD, summed over the chip);
GRBM_GUI_ACTIVE is summed over the 8 XCDs (divided by 8 and by the launch's duration from the kernel trace it gives a 2.1 GHz shader
clock under the profiler).  mfma_busy x (that clock / 2.4 GHz) is the achieved fraction of the 2.5 PFLOP/s peak."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
SIMDS = 256 * 4
XCDS = 8
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if "gemm16p_kernel" in name:   # the persistent one-pass family: every <dtype, EPI, ACT> instantiation pooled
                short = "gemm16p"
            elif "gemm16_kernel" in name:
                targs = name.split("gemm16_kernelI")[1].split("EEvNS")[0]
                short = "gemm16<" + ",".join([("f16" if targs.startswith("DF16_") else "bf16")] + re.findall(r"L[ib](\d+)E", targs)) + ">"
            else:
                short = re.sub(r"^_ZN3mer\d+", "", name.split("(")[0])[:60]
            agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
print(f"{'kernel':64s} {'launches':>8s} {'mfma_busy':>9s} {'MFMA insts/launch':>18s} {'GUI_ACTIVE':>12s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    mean = {c: sum(x) / len(x) for c, x in v.items()}
    n = len(next(iter(v.values())))
    act = mean.get("GRBM_GUI_ACTIVE", 0.0)
    busy = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out[k] = dict(launches=n, counters_per_launch={c: round(x, 1) for c, x in mean.items()},
                  mfma_busy=round(busy / (act / XCDS * SIMDS), 4) if act else None)
    print(f"{k[:64]:64s} {n:8d} {out[k]['mfma_busy'] if out[k]['mfma_busy'] is not None else float('nan'):9.4f} {mean.get('SQ_INSTS_MFMA', 0):18.0f} {act:12.0f}")
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in sorted(glob.glob(os.path.join(_root, "mertools_amd", "csrc", "*"))):   # every source of the library: the stamp bench.py checks
    if _f.endswith((".h", ".hip", ".cpp")):
        _h.update(os.path.basename(_f).encode())
        _h.update(open(_f, "rb").read())
out["_source_sha"] = _h.hexdigest()[:16]
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
