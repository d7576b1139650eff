#!/usr/bin/env python3
"""Aggregates rocprofv3 counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes) per kernel: mean per launch.
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B (bytes = value * 1024); on gfx950 FETCH_SIZE reports half
the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md §HBM), so fetch is also shown doubled."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                if "gemm16p_kernel" in name:   # the persistent one-pass family: every <dtype, EPI, ACT> instantiation pooled
                    short = "gemm16p"
                elif "gemm16_kernel" in name:   # keep every template argument: <dtype,BM,BN,BK,WM,WN,AP,WP,GLDS,NS,MX[,STAMP]>
                    import re
                    targs = name.split("gemm16_kernelI")[1].split("EEvNS")[0]
                    short = "gemm16<" + ",".join([("f16" if targs.startswith("DF16_") else "bf16")] + re.findall(r"L[ib](\d+)E", targs)) + ">"
                else:
                    short = name.split("(")[0][-60:]
                agg[short][counter].append(float(row["Counter_Value"]))
out = {}
print(f"{'kernel':70s} {'launches':>8s} {'fetch MB/launch':>16s} {'x2 (gfx950)':>12s} {'write MB/launch':>16s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    fe, wr = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    fmb = sum(fe) / max(len(fe), 1) * 1024 / 1e6
    wmb = sum(wr) / max(len(wr), 1) * 1024 / 1e6
    out[k] = dict(launches=len(fe) or len(wr), fetch_mb=fmb, fetch_mb_x2=2 * fmb, write_mb=wmb)
    print(f"{k[:70]:70s} {len(fe) or len(wr):8d} {fmb:16.2f} {2 * fmb:12.2f} {wmb:16.2f}")
# stamp the library sources the counters belong to: bench.py only quotes a traffic figure whose stamp matches the tree it runs
import glob as _g
import hashlib
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in sorted(_g.glob(os.path.join(_root, "mertools_amd", "csrc", "*"))):
    if _f.endswith((".h", ".hip", ".cpp")):
        _h.update(os.path.basename(_f).encode())
        _h.update(open(_f, "rb").read())
out["_source_sha"] = _h.hexdigest()[:16]
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
