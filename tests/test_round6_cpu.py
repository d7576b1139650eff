"""Round-6 host logic (CPU): constant-chunk detection of chunked clips, the public precision surface."""
import numpy as np
import pytest


def test_constant_chunks_of_a_long_clip():
    """A chunked (> 10 s) clip's silent chunks are found on the host's raw samples: a full chunk iff its raw samples hold
    one value; the zero-padded last chunk only when its samples normalise to exactly 0."""
    from mertools_amd.extract.audio import MAXLEN, constant_chunks, split_into_batch, wav2vec2_normalize, constant_rows
    rng = np.random.RandomState(0)
    n = 2 * MAXLEN + 5000
    x = (rng.randn(n) * 3000).astype(np.int16)
    x[MAXLEN:2 * MAXLEN] = 7                       # a DC stretch exactly covering chunk 1
    assert constant_chunks(x) == [1]
    # ... which is what the reference's host path finds on the normalised, chunked rows
    assert constant_rows(split_into_batch(wav2vec2_normalize(x.astype(np.float64)))) == [1]
    x[2 * MAXLEN:] = 7                              # the partial last chunk constant too: after zero padding it is NOT one value
    assert constant_chunks(x) == [1]
    assert constant_rows(split_into_batch(wav2vec2_normalize(x.astype(np.float64)))) == [1]
    z = np.zeros(n, np.int16)                       # an entirely silent long file: every chunk normalises to zero
    assert constant_chunks(z) == [0, 1, 2]
    assert constant_rows(split_into_batch(wav2vec2_normalize(z.astype(np.float64)))) == [0, 1, 2]
    c = np.full(n, 11, np.int16)                    # a DC file: (x - mean) is exactly 0 everywhere
    assert constant_chunks(c) == [0, 1, 2]
    assert constant_chunks(c, do_normalize=False) == [0, 1]        # un-normalised: the last chunk is 11 ... 11 0 ... 0
    assert constant_chunks(z, do_normalize=False) == [0, 1, 2]
    short = np.zeros(5000, np.int16)
    assert constant_chunks(short) == [0] and constant_chunks(x[:5000]) == []


def test_study_presets_are_not_on_the_public_surface(monkeypatch):
    """the eight deployment presets resolve anywhere; the numerics-study pass combinations only under MER_STUDY_PRESETS=1."""
    from mertools_amd import encoders, _lib
    assert sorted(encoders._PREC) == ["a2_conv3", "accurate", "balanced", "fast", "mean", "mean_a2", "mean_conv3", "mx"]
    monkeypatch.setenv("MER_STUDY_PRESETS", "0")
    for name in encoders._PREC:
        assert len(encoders._prec(name)) == 3
    for name in ("mean_all", "mixed", "a2f_conv3", "x3"):
        with pytest.raises(_lib.MerError, match="study preset"):
            encoders._prec(name)
    with pytest.raises(_lib.MerError, match="unknown precision"):
        encoders._prec("no_such_preset")
    monkeypatch.setenv("MER_STUDY_PRESETS", "1")
    assert encoders._prec("mean_all") == (5, 5, 0)


def test_builds_of_one_load_share_their_weight_planes():
    """VERDICT r5 #6c: inside one load (encoders._plane_cache) a second build takes the planes the first made — the same device tensor for the
    same weight content and plane kind — and adds only what it alone needs; outside a load nothing is shared; a weight that differs in one
    element is a different key."""
    import torch
    from mertools_amd import encoders as E
    g = torch.Generator().manual_seed(0)
    w = torch.randn(256, 64, generator=g)
    w2 = w.clone()
    w2[17, 3] += 1e-3
    b = torch.randn(256, generator=g)
    a, c = E._Holder("cpu", "f16"), E._Holder("cpu", "f16")
    assert E._PLANES is None
    x0, y0 = a.w16(w, lo=False), c.w16(w, lo=False)
    assert x0.hi != y0.hi                                  # no load in progress: every build packs its own planes
    with E._plane_cache():
        x = a.w16(w, lo=False)                             # the one-plane object ...
        y = c.w16(w, lo=True)                              # ... and its twin: same hi plane, its own lo plane
        z = c.w16(w2, lo=False)
        assert x.hi == y.hi and y.lo and not x.lo and z.hi != x.hi
        assert a.f32(b) == c.f32(b) and a.f32(b + 1) != c.f32(b)
        with E._plane_cache():                             # nested (a rung built inside the self-check): the same cache
            assert E._Holder("cpu", "f16").w16(w, lo=False).hi == x.hi
        assert E._PLANES is not None
    assert E._PLANES is None
    # the shared plane holds what a private build holds
    hi_shared = next(t for t in a.keep if t.data_ptr() == x.hi)
    hi_private = next(t for t in a.keep if t.data_ptr() == x0.hi)
    assert torch.equal(hi_shared, hi_private)
