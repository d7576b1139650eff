#!/usr/bin/env python3
"""Stamps a rocprofv3 --kernel-trace --stats summary with the sha of the library sources it was taken on:
    python scripts/stamp_kernel_stats.py <kernel_stats.csv> <profiles/rNN_name> [command]   ->  <..>_kernel_stats.csv + <..>_kernel_stats.json
tests/test_bench_cpu.py refuses a committed summary whose stamp is not the tree's (the round-3 CSV described a tree that no longer existed)."""
import glob
import hashlib
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
command = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --streams 0 (single stream, no roofline / parity legs)"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "mertools_amd", "csrc", "*"))):
    if f.endswith((".h", ".hip", ".cpp")):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
shutil.copy(src, dst + "_kernel_stats.csv")
json.dump({"_source_sha": h.hexdigest()[:16], "csv": os.path.basename(dst) + "_kernel_stats.csv",
           "command": command},
          open(dst + "_kernel_stats.json", "w"), indent=1)
print("stamped", dst + "_kernel_stats.csv", h.hexdigest()[:16])
