"""TriModalExtractor (mertools_amd/extract/trimodal.py): pipeline bookkeeping on a CPU device with stand-in encoders —
batch order, slot / buffer reuse across ragged batch sizes, the .npy layout.  The real encoders only run on a GPU."""
import os

import numpy as np
import torch

from mertools_amd.extract.trimodal import TriModalExtractor


class _Audio:
    def extract_utterance(self, x):
        return torch.stack([x.mean(1), x.abs().max(1).values, x[:, 0]], 1)


class _Visual:
    def extract_utterance(self, frames, frames_per_clip):
        flat = frames.reshape(frames.shape[0], -1).float()
        out, r = [], 0
        for n in frames_per_clip:
            out.append(flat[r:r + n].mean(0)[:5])
            r += n
        return torch.stack(out)


class _Text:
    def extract_utterance(self, ids, lengths, start, end):
        return torch.stack([ids[b, start:lengths[b] + end].float().mean().reshape(1) for b in range(ids.shape[0])])


def _batches(sizes, seed=0):
    g = torch.Generator().manual_seed(seed)
    out, c = [], 0
    for B in sizes:
        fpc = [1 + (c + i) % 3 for i in range(B)]
        out.append({"names": [f"clip{c + i:03d}" for i in range(B)],
                    "audio": torch.randn(B, 50, generator=g),
                    "frames": torch.randn(sum(fpc), 3, 4, 4, generator=g), "frames_per_clip": fpc,
                    "input_ids": torch.randint(3, 100, (B, 9), generator=g), "lengths": [9 - (i % 3) for i in range(B)]})
        c += B
    return out


def test_pipeline_matches_direct_calls():
    sizes = [3, 5, 1, 4, 4, 2]
    batches = _batches(sizes)
    eng = TriModalExtractor(_Audio(), _Visual(), _Text(), device="cpu")
    got = list(eng.run(batches))
    assert [len(n) for n, _ in got] == sizes
    for (names, feats), b in zip(got, batches):
        assert names == b["names"]
        assert np.array_equal(feats["audio"], _Audio().extract_utterance(b["audio"]).numpy())
        assert np.array_equal(feats["visual"], _Visual().extract_utterance(b["frames"], b["frames_per_clip"]).numpy())
        assert np.array_equal(feats["text"], _Text().extract_utterance(b["input_ids"], b["lengths"], 1, -1).numpy())


def test_results_survive_buffer_reuse():
    """The arrays handed out for batch k must not alias the pinned buffers that batch k+2 overwrites."""
    batches = _batches([2, 2, 2, 2, 2], seed=3)
    eng = TriModalExtractor(_Audio(), None, None, device="cpu")
    got = list(eng.run(batches))
    for (_, feats), b in zip(got, batches):
        assert set(feats) == {"audio"}
        assert np.array_equal(feats["audio"], _Audio().extract_utterance(b["audio"]).numpy())


def test_extract_to_dirs_layout(tmp_path):
    batches = _batches([3, 2], seed=5)
    eng = TriModalExtractor(_Audio(), _Visual(), _Text(), device="cpu")
    dirs = {"audio": str(tmp_path / "hubert-UTT"), "visual": str(tmp_path / "clip-UTT"), "text": str(tmp_path / "roberta-UTT")}
    assert eng.extract_to_dirs(batches, dirs) == 5
    for m, d in dirs.items():
        files = sorted(os.listdir(d))
        assert files == [f"clip{i:03d}.npy" for i in range(5)]
        x = np.load(os.path.join(d, files[0]))
        assert x.ndim == 1 and x.dtype == np.float32          # the reference's UTT layout: float32 [D]


def test_per_clip_lists_equal_stacked_batches(tmp_path):
    """`audio` / `frames` as lists of per-clip arrays (gathered straight into the pinned block, on copy threads) == the stacked
    tensors, ragged frame counts and compact dtypes included; the saves run on worker threads and write the same files."""
    batches = _batches([3, 5, 1, 4], seed=7)
    listed = []
    for b in batches:
        rows, r = [], 0
        for n in b["frames_per_clip"]:
            rows.append(b["frames"][r:r + n].numpy())
            r += n
        listed.append(dict(b, audio=[x.numpy() for x in b["audio"]], frames=rows))
    a = TriModalExtractor(_Audio(), _Visual(), _Text(), device="cpu")
    c = TriModalExtractor(_Audio(), _Visual(), _Text(), device="cpu", copy_workers=3)
    for (n1, f1), (n2, f2) in zip(a.run(batches), c.run(listed)):
        assert n1 == n2
        for m in f1:
            assert np.array_equal(f1[m], f2[m]), m
    d1 = {m: str(tmp_path / f"one_{m}") for m in ("audio", "visual", "text")}
    d2 = {m: str(tmp_path / f"two_{m}") for m in ("audio", "visual", "text")}
    assert a.extract_to_dirs(batches, d1) == c.extract_to_dirs(listed, d2) == 13
    for m in d1:
        assert sorted(os.listdir(d1[m])) == sorted(os.listdir(d2[m])) and len(os.listdir(d1[m])) == 13
        for f in os.listdir(d1[m]):
            assert np.array_equal(np.load(os.path.join(d1[m], f)), np.load(os.path.join(d2[m], f)))
    # compact forms keep their dtype through the gather (int16 PCM, uint8 frames)
    pcm = [np.arange(40, dtype=np.int16) + i for i in range(3)]
    u8 = [np.full((2, 4, 4, 3), i, dtype=np.uint8) for i in range(3)]
    e = TriModalExtractor(None, None, None, device="cpu")
    assert e._gather(e.slots[0], "audio", pcm).dtype == torch.int16 and tuple(e._gather(e.slots[0], "audio", pcm).shape) == (3, 40)
    g = e._gather(e.slots[0], "frames", u8)
    assert g.dtype == torch.uint8 and tuple(g.shape) == (6, 4, 4, 3) and int(g[4, 0, 0, 0]) == 2
    import pytest
    with pytest.raises(ValueError):
        e._gather(e.slots[0], "audio", [pcm[0], pcm[1][:10]])
