import torch


def rel_err(out, ref):
    """max|out-ref| / max|ref| and relative L2 — the two figures every parity test reports."""
    out = out.detach().double().cpu()
    ref = ref.detach().double().cpu()
    d = (out - ref).abs().max().item()
    return d / max(ref.abs().max().item(), 1e-30), ((out - ref).norm() / max(ref.norm().item(), 1e-30)).item()


def assert_close(out, ref, tol, what=""):
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite values in output"
    e, l2 = rel_err(out, ref)
    assert e <= tol, f"{what}: max-norm rel err {e:.3e} (rel-L2 {l2:.3e}) > tol {tol:.1e}"
    return e
