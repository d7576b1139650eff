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
    """Editing csrc/gemm16_impl.h without re-collecting the PMC passes (scripts/r2_final.sh) makes this fail — on purpose: the
    bench would otherwise print traffic = null on the GPU box."""
    b = _bench()
    for kernel in ("gemm16", "gemm16_mx"):
        traffic, detail = b.pmc_traffic(kernel, 5e8)
        assert traffic is not None and traffic > 1e8, (kernel, detail)
        assert detail["kernel_source_sha"] == b.kernel_source_sha() and detail["source"].startswith("profiles/")


def test_stale_pmc_collection_is_refused(tmp_path):
    b = _bench()
    root = tmp_path / "tree"
    (root / "mertools_amd" / "csrc").mkdir(parents=True)
    (root / "profiles").mkdir()
    shutil.copy(os.path.join(ROOT, "mertools_amd", "csrc", "gemm16_impl.h"), root / "mertools_amd" / "csrc" / "gemm16_impl.h")
    shutil.copy(os.path.join(ROOT, "profiles", "r02_pmc_hbm_traffic.json"), root / "profiles" / "r02_pmc_hbm_traffic.json")
    assert b.pmc_traffic("gemm16", 5e8, root=str(root))[0] is not None
    with open(root / "mertools_amd" / "csrc" / "gemm16_impl.h", "a") as f:
        f.write("// a kernel edit\n")
    traffic, detail = b.pmc_traffic("gemm16", 5e8, root=str(root))
    assert traffic is None and "r02_pmc_hbm_traffic.json" in detail["note"] and b.kernel_source_sha(str(root)) in detail["note"]
    # a collection without a stamp (round 1's) is never quoted either
    d = json.load(open(root / "profiles" / "r02_pmc_hbm_traffic.json"))
    d.pop("_source_sha")
    json.dump(d, open(root / "profiles" / "r03_pmc_hbm_traffic.json", "w"))
    assert b.pmc_traffic("gemm16", 5e8, root=str(root))[0] is None
