import torch


def rel_err(out, ref):
    """max|out-ref| / max|ref| and relative L2 — the two figures every parity test reports."""
    out = out.detach().double().cpu()
    ref = ref.detach().double().cpu()
    d = (out - ref).abs().max().item()
    return d / max(ref.abs().max().item(), 1e-30), ((out - ref).norm() / max(ref.norm().item(), 1e-30)).item()


def dim_rel(out, ref):
    """Norm-free companion of rel_err for a [rows, D] feature matrix: the WORST feature dimension's RMS error over the rows relative to
    that dimension's own RMS — max-norm figures are blind to the small-magnitude dimensions of a feature vector (VERDICT r4 weak #1 iii)."""
    out = out.detach().double().cpu().reshape(-1, out.shape[-1])
    ref = ref.detach().double().cpu().reshape(-1, ref.shape[-1])
    err = (out - ref).pow(2).mean(0).sqrt()
    mag = ref.pow(2).mean(0).sqrt()
    return float((err / mag.clamp_min(1e-30)).max())


def assert_close(out, ref, tol, what=""):
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite values in output"
    e, l2 = rel_err(out, ref)
    assert e <= tol, f"{what}: max-norm rel err {e:.3e} (rel-L2 {l2:.3e}) > tol {tol:.1e}"
    return e


FP4_GRID = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], dtype=torch.float64)


def mx_decode(packed, N, K):
    """Independent reader of the MX correction plane (include/mer_hip.h: mer_mx_pack): returns the fp64 [N, K] matrix the
    v_mfma_scale_f32_16x16x128_f8f6f4 B operand represents.  Layout restated from the header / DESIGN.md, not from the packer:
    per (256-column tile tn, slab t) a 5120-byte block; bytes [0,4096) = column tiles 4*(t%4) .. +3 of the 128-k group
    t//4, each 64 lanes x 16 B (lane = n16 + 16*b holds k-slots 32b .. 32b+31, low nibble first); bytes [4096,5120) =
    scales[set 0..3][lane] dwords, byte c of set s = E8M0 scale of column tile 4s + c.  k-slot h of a group maps to
    k = 32*s + 8*g + e with (h < 64: g = h//16, r = h%16, s = r//8) / (h >= 64: g = (h-64)//16, r = (h-64)%16, s = 2 + r//8), e = r%8."""
    buf = packed.cpu().numpy() if hasattr(packed, "cpu") else packed
    import numpy as np
    nslab = K // 32
    out = np.zeros((N, K), dtype=np.float64)
    grid = FP4_GRID.numpy()
    slot_k = np.zeros(128, dtype=np.int64)
    for h in range(128):
        hh = h % 64
        g, r = hh // 16, hh % 16
        s, e = (h // 64) * 2 + r // 8, r % 8
        slot_k[h] = 32 * s + 8 * g + e
    for tn in range((N + 255) // 256):
        for kg in range(K // 128):
            for ctg in range(16):
                blk = (tn * nslab + kg * 4 + ctg // 4) * 5120
                frag = buf[blk + (ctg % 4) * 1024: blk + (ctg % 4) * 1024 + 1024].reshape(64, 16)
                sc = buf[blk + 4096 + (ctg // 4) * 256: blk + 4096 + (ctg // 4) * 256 + 256].reshape(64, 4)[:, ctg % 4]
                lo, hi = frag & 15, frag >> 4
                codes = np.stack([lo, hi], axis=2).reshape(64, 32)            # element j of lane
                vals = grid[codes & 7] * np.where(codes & 8, -1.0, 1.0) * np.exp2(sc.astype(np.float64) - 127.0)[:, None]
                for n16 in range(16):
                    n = tn * 256 + ctg * 16 + n16
                    if n >= N:
                        continue
                    for b in range(4):
                        out[n, kg * 128 + slot_k[32 * b: 32 * b + 32]] = vals[n16 + 16 * b]
    return torch.from_numpy(out)


def bf8_round(x):
    """f16 -> bf8 (e5m2, round to nearest even) -> fp64: what v_cvt_scalef32_pk_bf8_f16 with scale 1 produces."""
    return x.to(torch.float16).to(torch.float8_e5m2).double()
