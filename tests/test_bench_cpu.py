"""bench.py's roofline.traffic lookup (CPU): a PMC collection is quoted only when it was taken on the kernel source in the tree."""
import importlib.util
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_pmc_collection_matches_the_kernel_source():
    """Editing ANY source under mertools_amd/csrc/ without re-collecting the PMC passes (scripts/pmc_traffic.sh, scripts/pmc_mfma.sh) makes this
    fail — on purpose: the bench would otherwise print traffic = null / mfma_busy = null on the GPU box."""
    b = _bench()
    traffic, detail = b.pmc_traffic("gemm16p", 5e8)   # (the default preset launches the one-pass kernel only)
    assert traffic is not None and traffic > 1e8, detail
    assert detail["kernel_source_sha"] == b.kernel_source_sha() and detail["source"].startswith("profiles/")
    busy = b.pmc_mfma_busy("gemm16p")
    assert busy is not None and 0.05 < busy["mfma_busy"] < 1.0 and busy["kernel_source_sha"] == b.kernel_source_sha(), busy
    # the counter calibration the definition rests on: 16 busy cycles per 16x16x32 MFMA
    assert abs(busy["SQ_VALU_MFMA_BUSY_CYCLES"] / busy["SQ_INSTS_MFMA"] - 16.0) < 0.01


def test_committed_kernel_stats_match_the_kernel_source():
    """The rocprofv3 --kernel-trace --stats summary under profiles/ (the per-kernel times DESIGN.md and the judge's recomputation rest on)
    carries the sha of the library sources it was taken on (scripts/stamp_kernel_stats.py): a summary of another tree fails here."""
    import glob
    b = _bench()
    stamps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*kernel_stats.json")))   # (the latest round's sorts last)
    assert stamps, "no stamped kernel-stats summary under profiles/"
    d = json.load(open(stamps[-1]))
    assert d["_source_sha"] == b.kernel_source_sha(), (stamps[-1], d["_source_sha"], b.kernel_source_sha())
    csv = open(os.path.join(ROOT, "profiles", d["csv"])).read()
    assert "gemm16p_kernel" in csv and "gemm16_kernel" in csv      # the persistent family and the MX conv-stack kernel both have rows


def test_stale_pmc_collection_is_refused(tmp_path):
    import glob
    b = _bench()
    root = tmp_path / "tree"
    (root / "mertools_amd" / "csrc").mkdir(parents=True)
    (root / "profiles").mkdir()
    for f in glob.glob(os.path.join(ROOT, "mertools_amd", "csrc", "*")):
        if f.endswith((".h", ".hip", ".cpp")):
            shutil.copy(f, root / "mertools_amd" / "csrc" / os.path.basename(f))
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*pmc_hbm_traffic*.json")))[-1]
    shutil.copy(pmc, root / "profiles" / os.path.basename(pmc))
    assert b.pmc_traffic("gemm16p", 5e8, root=str(root))[0] is not None
    with open(root / "mertools_amd" / "csrc" / "norm.hip", "a") as f:      # ANY source of the library, not only the GEMM template
        f.write("// a kernel edit\n")
    traffic, detail = b.pmc_traffic("gemm16p", 5e8, root=str(root))
    assert traffic is None and os.path.basename(pmc) in detail["note"] and b.kernel_source_sha(str(root)) in detail["note"]
    # a collection without a stamp (round 1's) is never quoted either
    d = json.load(open(root / "profiles" / os.path.basename(pmc)))
    d.pop("_source_sha")
    json.dump(d, open(root / "profiles" / os.path.basename(pmc), "w"))
    assert b.pmc_traffic("gemm16p", 5e8, root=str(root))[0] is None
