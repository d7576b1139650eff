"""The lagged-group persistent GEMM (csrc/gemm16q_impl.h, round 6) against gemm16_kernel (the tile kernel) on the same operands, bit for
bit: group 1 runs S slabs + a phase behind group 0 on a shared W ring, the epilogue is cut into barrier-separated pieces.  A stale LDS
stage, a half of a W slab read before its DMA landed or a store counted wrongly shows up here as differing bits or as run-to-run
differences (every case is launched several times)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CFGS = [0, 1, 2]     # gemm_q_cfg: (D 3, S 1), (D 2, S 1), (D 2, S 2)


def _ops():
    from mertools_amd import ops
    return ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


class _Mode:
    """gemm_persist / gemm_q_cfg / gemm_tm for the duration of a block, restored afterwards."""

    def __init__(self, persist, cfg=0, tm=0):
        self.v = (persist, cfg, tm)

    def __enter__(self):
        from mertools_amd import _lib
        self.lib = _lib.lib()
        self.was = _lib.get_option("gemm_persist")
        self.lib.mer_set_option(b"gemm_persist", self.v[0])
        self.lib.mer_set_option(b"gemm_q_cfg", self.v[1])
        self.lib.mer_set_option(b"gemm_tm", self.v[2])

    def __exit__(self, *a):
        self.lib.mer_set_option(b"gemm_persist", self.was)
        self.lib.mer_set_option(b"gemm_q_cfg", 0)
        self.lib.mer_set_option(b"gemm_tm", 0)


def _same(x, y):
    return torch.equal(x.view(torch.int16 if x.element_size() == 2 else torch.int32), y.view(torch.int16 if y.element_size() == 2 else torch.int32))


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("M,N,K", [(10240, 2048, 256), (4100, 768, 768), (66000, 512, 1536), (1500, 2304, 3072), (25600, 2304, 768), (140000, 768, 768)])
def test_gemm16q_equals_tile_kernel(dev, M, N, K, cfg):
    """Every epilogue kind (16-bit out with each activation, without bias; fp32; fp32 + residual), 0.2 .. 6.4 tiles per workgroup
    (first-tile, steady-state and last-tile paths of both groups), ragged M, 8 .. 96 slabs, 256- and 192-row tiles."""
    ops = _ops()
    a = _rand((M, K), 91)
    w = _rand((N, K), 92) * 0.05
    ah, _ = ops.split16(a.to(dev), "f16", lo=False)
    wh = ops.split16_host(w, "f16")[0].to(dev)
    hb, hp, hq = ops.w_block_pack(wh), ops.w_block_pack_p(wh, 0), ops.w_block_pack_p(wh, 1)
    bias, res = _rand((N,), 93).to(dev), _rand((M, N), 94).to(dev)
    cases = [dict(out16=True, act=act, bias=bias) for act in (None, "gelu", "quick_gelu", "gelu_new")]
    cases += [dict(out16=True, act=None, bias=None)]
    cases += [dict(out32=True, act=act, bias=bias) for act in (None, "gelu")]
    cases += [dict(out32=True, act=None, bias=bias, residual=res)]
    for kw in cases:
        kw = dict(kw, passes=1, tile=3, dtype="f16", w_hi_blk=hb, w_hi_blkp=hp, w_hi_blkq=hq)
        with _Mode(0):
            r32, r16, _ = ops.gemm16(ah, wh, **kw)
        for tm in (4, 3):
            for rep in range(3):
                with _Mode(2, cfg, tm):
                    c32, c16, _ = ops.gemm16(ah, wh, **kw)
                    torch.cuda.synchronize()
                what = f"gemm16q cfg {cfg} tm {tm} act={kw.get('act')} out16={kw.get('out16', False)} res={'residual' in kw} rep {rep}"
                if r16 is not None:
                    assert _same(c16, r16), what
                if r32 is not None:
                    assert _same(c32, r32), what
    # the residual stream updated in place (pre-LN blocks): residual == c32
    with _Mode(0):
        ref32, _, _ = ops.gemm16(ah, wh, w_hi_blk=hb, bias=bias, residual=res, out32=True, passes=1, tile=3, dtype="f16")
    for tm in (4, 3):
        inplace = res.clone()
        g = ops.GemmArgs()
        g.M, g.N, g.K, g.dtype = M, N, K, ops.dt_code("f16")
        g.a_hi, g.lda, g.w_hi, g.ldw = ah.data_ptr(), K, wh.data_ptr(), K
        g.w_hi_blk, g.w_hi_blkp, g.w_hi_blkq = hb.data_ptr(), hp.data_ptr(), hq.data_ptr()
        g.bias, g.act, g.residual, g.ldr, g.c32, g.ldc32 = bias.data_ptr(), ops.ACT[None], inplace.data_ptr(), N, inplace.data_ptr(), N
        g.nbatch, g.nb_inner, g.passes, g.tile = 1, 1, 1, 3
        with _Mode(2, cfg, tm):
            ops.gemm16_raw(g)
            torch.cuda.synchronize()
        assert torch.equal(inplace, ref32), f"gemm16q cfg {cfg} ({64 * tm}-row tile), residual updated in place"


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("T,M", [(197, 9000), (64, 9000), (249, 9000), (40, 9000), (1568, 6000), (249, 70000), (64, 90000)])
def test_gemm16q_bias_table(dev, T, M, cfg):
    """Per-sequence bias tables (mer_seq_bias): the table rows of a tile are staged by group 0 one tile ahead and read by both groups'
    epilogue pieces."""
    ops = _ops()
    K, N = 768, 768
    nseq = (M + T - 1) // T
    a = _rand((M, K), 111)
    w = _rand((N, K), 112) * 0.05
    tab = _rand((nseq, N), 113).to(dev)
    res = _rand((M, N), 114).to(dev)
    ah, _ = ops.split16(a.to(dev), "f16", lo=False)
    wh = ops.split16_host(w, "f16")[0].to(dev)
    hb, hp, hq = ops.w_block_pack(wh), ops.w_block_pack_p(wh, 0), ops.w_block_pack_p(wh, 1)
    kw = dict(bias=tab, bias_seg_rows=T, passes=1, tile=3, w_hi_blk=hb, w_hi_blkp=hp, w_hi_blkq=hq)

    def run():
        _, c16, _ = ops.gemm16(ah, wh, act="gelu", out16=True, **kw)
        c32, _, _ = ops.gemm16(ah, wh, out32=True, **kw)
        r32, _, _ = ops.gemm16(ah, wh, out32=True, residual=res, **kw)
        torch.cuda.synchronize()
        return c16, c32, r32

    with _Mode(0):
        ref = run()
    for tm in (4, 3):
        for rep in range(2):
            with _Mode(2, cfg, tm):
                out = run()
            for i, what in enumerate(("16-bit gelu", "fp32", "fp32 + residual")):
                assert _same(out[i], ref[i]), f"gemm16q cfg {cfg} tm {tm} bias table T={T}: {what}, rep {rep}"


@pytest.mark.parametrize("cfg", CFGS)
def test_gemm16q_conv_rows(dev, cfg):
    """Implicit-im2col row mapping (overlapping windows of a channels-last plane) through the per-group A rings."""
    ops = _ops()
    B, Tin, C, k, s = 24, 1599, 512, 3, 2
    Tout = (Tin - k) // s + 1
    x = _rand((B * Tin, C), 95)
    w = _rand((C, k * C), 96) * 0.03
    xh, _ = ops.split16(x.to(dev), "f16", lo=False)
    wh = ops.split16_host(w, "f16")[0].to(dev)
    hb, hp = ops.w_block_pack(wh), ops.w_block_pack_p(wh, 0)
    kw = dict(act="gelu", out16=True, passes=1, tile=3, M=B * Tout, lda=s * C, a_rows_per_batch=Tout, a_batch_stride=Tin * C)
    with _Mode(0):
        _, ref, _ = ops.gemm16(xh, wh, w_hi_blk=hb, w_hi_blkp=hp, **kw)
    for tm in (4, 3):
        with _Mode(2, cfg, tm):
            _, out, _ = ops.gemm16(xh, wh, w_hi_blk=hb, w_hi_blkp=hp, **kw)
            torch.cuda.synchronize()
        assert _same(out, ref), f"gemm16q cfg {cfg} tm {tm}: conv rows"
