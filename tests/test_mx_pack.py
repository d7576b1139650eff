"""Host-side packer of the MX-fp4 correction plane (mer_mx_pack) against an independent decoder (tests/util.mx_decode):
layout round trip on exactly representable data, quantisation error bound on real rounding residuals, edge shapes."""
import pytest
import torch

from util import FP4_GRID, mx_decode


@pytest.fixture(scope="module")
def ops():
    from mertools_amd import ops
    return ops


def test_unsupported_shapes_have_no_plane(ops):
    assert ops.mx_pack(torch.zeros(16, 96)) is None        # K % 128 != 0
    assert ops.mx_pack(torch.zeros(16, 592)) is None


def test_representable_values_round_trip_exactly(ops):
    g = torch.Generator().manual_seed(0)
    N, K = 300, 256                                           # ragged N: second column tile is partial
    codes = torch.randint(0, 8, (N, K), generator=g)
    sign = torch.where(torch.rand(N, K, generator=g) < 0.5, -1.0, 1.0)
    assert ops.mx_pack(torch.zeros(N, K)).numel() == ((N + 255) // 256) * (K // 32) * 5120
    # blocks of the packer are 32 k-SLOTS (a permutation of k inside each 128-k group), not 32 consecutive k: give every
    # slot block of a row the same exponent by making the exponent constant per (row, 128-k group)
    expo_g = torch.randint(-20, -8, (N, K // 128), generator=g).repeat_interleave(128, dim=1).double()
    w = FP4_GRID[codes] * sign * torch.exp2(expo_g)
    w.view(N, K // 8, 8)[:, :, 0] = 6.0 * torch.exp2(expo_g.view(N, K // 8, 8)[:, :, 0])   # a 6.0 in every 8-run => in every slot block
    back = mx_decode(ops.mx_pack(w.float()), N, K)
    assert torch.equal(back, w), (back - w).abs().max()


def test_rounding_residual_error_bound(ops):
    g = torch.Generator().manual_seed(1)
    N, K = 768, 768
    w = torch.randn(N, K, generator=g) / K ** 0.5
    res = w - w.half().float()
    back = mx_decode(ops.mx_pack(res), N, K)
    rel = ((back - res.double()).norm() / res.double().norm()).item()
    assert rel < 0.2, rel                                     # e2m1 + per-32 scale on a ~uniform residual: ~0.1
    assert (back - res.double()).abs().max() <= res.abs().max().item()   # never worse than dropping the residual
    # zero input -> zero plane
    assert mx_decode(ops.mx_pack(torch.zeros(32, 128)), 32, 128).abs().max() == 0


def test_column_tile_blocks_are_independent(ops):
    """The engine splits the fused QKV GEMM into a one-pass [Q|K] launch and an MX-corrected V launch by offsetting into the
    packed plane at tile 2D/256 (csrc/encoders.cpp tf_forward): the plane of W[3D, D] from byte (2D/256)*(D/32)*5120 on must be
    byte-identical to the plane of the V rows packed alone."""
    g = torch.Generator().manual_seed(3)
    for D in (128, 768):
        w = torch.randn(3 * D, D, generator=g) * 1e-5
        full = ops.mx_pack(w)
        v_only = ops.mx_pack(w[2 * D:].contiguous())
        off = (2 * D // 256) * (D // 32) * 5120
        assert torch.equal(full[off:], v_only)
