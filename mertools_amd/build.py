"""Builds libmer_hip.so (HIP kernels + C-ABI, gfx950 only) in-tree with hipcc.

`python -m mertools_amd.build` or `mertools_amd.build.build()`; hipcc cross-compiles for gfx950
without a GPU present.  Objects are cached under mertools_amd/csrc/_obj keyed on source mtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmer_hip.so")
# the gemm16 kernel family is instantiated in four translation units (tile class x dtype) so that it compiles in parallel;
# the long ones go first
SOURCES = ["gemm16p_f16.hip", "gemm16p_bf16.hip", "gemm16p_f16_r192.hip", "gemm16p_bf16_r192.hip", "gemm16_t3_f16.hip", "gemm16_t3_bf16.hip", "gemm16_small_f16.hip", "gemm16_small_bf16.hip", "attention.hip", "attention_f32.hip",
           "common.cpp", "gemm16.hip", "gemm32.hip", "norm.hip", "frontend.hip", "fusion.hip", "encoders.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-x", "hip", "-Rpass-analysis=kernel-resource-usage"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "mer_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, hdr_mtime, verbose):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src + ".o")
    if os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_mtime):
        return o
    cmd = [_hipcc()] + FLAGS + ["-c", s, "-o", o]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    # per-kernel VGPR / scratch / occupancy remarks: tests/test_abi.py checks that no hot kernel touches scratch memory
    remarks = [l for l in r.stderr.splitlines() if "remark:" in l]
    with open(o[:-2] + ".resources.txt", "w") as fh:
        fh.write("\n".join(l.split("remark:", 1)[1].split("[-Rpass")[0].rstrip() for l in remarks) + "\n")
    other = [l for l in r.stderr.splitlines() if "remark:" not in l and l.strip()]
    if verbose and other:
        print("\n".join(other), file=sys.stderr)
    return o


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdr_mtime = _deps_mtime() if not force else float("inf")
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr_mtime, verbose), srcs))
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) > max(os.path.getmtime(o) for o in objs)):
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
