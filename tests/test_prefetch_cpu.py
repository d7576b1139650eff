"""extract.prefetch and its use in the drivers: ordered read-ahead on threads must not change a single output byte.
CPU-only: the drivers run with stand-in encoders (the real ones need a GPU)."""
import os
import threading
import time
import wave

import numpy as np
import pytest
import torch

from mertools_amd.extract import audio, visual
from mertools_amd.extract.prefetch import prefetch_map


def test_prefetch_map_order_and_laziness():
    seen = []

    def fn(i):
        time.sleep(0.002 * ((7 * i) % 5))    # finish out of order
        seen.append(i)
        return i * i

    assert list(prefetch_map(fn, range(40), workers=0)) == [i * i for i in range(40)]
    seen.clear()
    assert list(prefetch_map(fn, range(40), workers=6, depth=8)) == [i * i for i in range(40)]
    assert sorted(seen) == list(range(40))
    # bounded read-ahead: after taking 3 results at most depth + workers items have been started
    seen.clear()
    g = prefetch_map(fn, range(1000), workers=4, depth=8)
    got = [next(g) for _ in range(3)]
    assert got == [0, 1, 4] and len(seen) <= 3 + 8 + 4
    g.close()                                 # early stop: must not hang or run the remaining 990 items
    time.sleep(0.05)
    assert len(seen) < 40


def test_prefetch_map_exception_surfaces_at_its_item():
    def fn(i):
        if i == 5:
            raise ValueError("item 5")
        return i

    out = []
    with pytest.raises(ValueError, match="item 5"):
        for x in prefetch_map(fn, range(20), workers=4):
            out.append(x)
    assert out == [0, 1, 2, 3, 4]


def test_prefetch_map_runs_on_other_threads():
    names = set(prefetch_map(lambda i: threading.current_thread().name, range(16), workers=3))
    assert all(n.startswith("mer-prefetch") for n in names)


class _Vision:
    device = torch.device("cpu")

    class config:
        class vision_config:
            image_size = 32

    def get_image_features(self, px):
        return torch.stack([px.mean((1, 2, 3)), px[:, 0].amax((1, 2)), px[:, 2, 3, 5]], 1)


def test_visual_driver_identical_with_workers(tmp_path):
    rng = np.random.RandomState(0)
    face = tmp_path / "faces"
    vids = []
    for i, (n, h, w) in enumerate([(5, 32, 32), (3, 40, 48), (0, 32, 32), (7, 50, 36), (2, 32, 32), (9, 33, 64)]):
        vid = f"v{i}"
        os.makedirs(face / vid)
        np.save(face / vid / f"{vid}.npy", rng.randint(0, 256, (n, h, w, 3), dtype=np.uint8))
        vids.append(vid)
    outs = []
    for workers in (0, 4):
        d = tmp_path / f"out{workers}"
        visual.extract(_Vision(), str(face), str(d), "UTTERANCE", vids=vids, frames_per_batch=12, workers=workers)
        outs.append({v: np.load(d / f"{v}.npy") for v in vids})
    for v in vids:
        assert outs[0][v].dtype == outs[1][v].dtype and np.array_equal(outs[0][v], outs[1][v]), v
    assert outs[0]["v2"].shape == (3,) and not outs[0]["v2"].any()      # empty video: zeros of the running embedding dim


class _Audio:
    """Stand-in with the encoder's batching interface (ragged-aware: a row's feature only looks at its own samples)."""
    device = torch.device("cpu")

    def out_frames(self, L):
        return L // 320

    def clip_segments(self, L, clip_chunks, valid_samples=None):
        T = self.out_frames(L)
        starts, lens, r = [], [], 0
        for n in clip_chunks:
            starts.append(r * T)
            lens.append(n * T if (n > 1 or valid_samples is None) else self.out_frames(valid_samples[r]))
            r += n
        return starts, lens

    def extract_utterance(self, rows, clip_chunks=None, valid_samples=None):
        out, r = [], 0
        for n in clip_chunks:
            x = rows[r:r + n] if (n > 1 or valid_samples is None) else rows[r:r + 1, :valid_samples[r]]
            out.append(torch.stack([x.double().mean().float(), x.abs().max(), x[0, 7]]))
            r += n
        return torch.stack(out)

    def forward_raw(self, rows, frames=False, valid_samples=None):
        T = self.out_frames(rows.shape[1])
        fr = rows[:, :T * 320].reshape(rows.shape[0] * T, 320)[:, :4].contiguous()
        return None, fr, None


def test_audio_driver_identical_with_workers(tmp_path):
    rng = np.random.RandomState(1)
    files = []
    for i, L in enumerate([4000, 6400, 4000, 9000, 6400, 4000]):
        p = str(tmp_path / f"a{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes((np.clip(rng.randn(L) * 0.1, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())
        files.append(p)
    outs = []
    for workers in (0, 3):
        d = str(tmp_path / f"feat{workers}")
        audio.extract("stub", files, d, "UTTERANCE", 0, model=_Audio(), workers=workers, batch_rows=2)
        outs.append([np.load(os.path.join(d, f"a{i}.npy")) for i in range(len(files))])
    for a, b in zip(*outs):
        assert a.dtype == b.dtype == np.float32 and np.array_equal(a, b)


def _faces(tmp_path, specs, seed=0):
    rng = np.random.RandomState(seed)
    face = tmp_path / "faces"
    vids = []
    for i, (n, h, w) in enumerate(specs):
        vid = f"v{i}"
        os.makedirs(face / vid)
        np.save(face / vid / f"{vid}.npy", rng.randint(0, 256, (n, h, w, 3), dtype=np.uint8))
        vids.append(vid)
    return str(face), vids


def test_other_visual_branches_run_with_stand_in_encoders(tmp_path):
    """VideoMAE / DINOv2 / data2vec-vision drivers (host pre-processing path) end to end on CPU with stand-in encoders: batching,
    frame resampling and the .npy layout of the reference (FRAME: [n, D]; UTTERANCE: [D])."""
    face, vids = _faces(tmp_path, [(20, 40, 48), (5, 36, 36), (17, 50, 32)])

    class _VMAE:
        device = torch.device("cpu")

        class config:
            num_frames, tubelet_size, image_size = 16, 2, 32

        def extract_segments(self, px):                      # [B, 16, 3, S, S] -> [B * 8, 4]
            assert px.dim() == 5 and px.shape[1:] == (16, 3, 32, 32)
            return px.view(px.shape[0] * 8, 2, -1)[:, :, :4].mean(1)

    visual.extract_videomae(_VMAE(), face, str(tmp_path / "vmae"), "FRAME", vids=vids, videos_per_batch=2)
    assert all(np.load(tmp_path / "vmae" / f"{v}.npy").shape == (8, 4) for v in vids)

    class _Tok:
        device = torch.device("cpu")

        class _cfg:
            image_size = 32

        def extract_frames(self, px):                         # [N, 3, S, S] -> [N, 5]
            assert px.shape[1:] == (3, 32, 32)
            return px.reshape(px.shape[0], -1)[:, :5]

    visual.extract_dinov2(_Tok(), face, str(tmp_path / "dino"), "FRAME", vids=vids, frames_per_batch=70, nframe=8)
    assert all(np.load(tmp_path / "dino" / f"{v}.npy").shape == (8, 5) for v in vids)
    visual.extract_data2vec_vision(_Tok(), face, str(tmp_path / "d2v"), "UTTERANCE", vids=vids, frames_per_batch=16)
    assert all(np.load(tmp_path / "d2v" / f"{v}.npy").shape == (5,) for v in vids)


def _wavs(tmp_path, lens, seed=1):
    rng = np.random.RandomState(seed)
    files = []
    for i, L in enumerate(lens):
        p = str(tmp_path / f"a{i:03d}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes((np.clip(rng.randn(L) * 0.1, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())
        files.append(p)
    return files


@pytest.mark.parametrize("level", ["UTTERANCE", "FRAME"])
def test_audio_driver_ragged_streaming_and_sharded(tmp_path, level):
    """Ragged batches (different lengths in one batch), a small streaming window and a 2-way clip shard all write the same
    files as one exact-length-bucket pass over the whole list; the driver never holds more than `window` clips."""
    rng = np.random.RandomState(3)
    lens = [int(x) for x in rng.randint(3200, 20000, size=37)] + [4000, 4000, 4000] + [23000, 41000]   # the last two are chunked (> maxlen)
    files = _wavs(tmp_path, lens)
    old = audio.split_into_batch.__defaults__
    audio.split_into_batch.__defaults__ = (20000,)
    held = {"now": 0, "max": 0}

    def reader(path):
        held["now"] += 1
        held["max"] = max(held["max"], held["now"])
        return audio.read_audio(path)

    class Counting(_Audio):
        def extract_utterance(self, rows, clip_chunks=None, valid_samples=None):
            held["now"] -= len(clip_chunks)
            return super().extract_utterance(rows, clip_chunks, valid_samples)

        def forward_raw(self, rows, frames=False, valid_samples=None):
            return super().forward_raw(rows, frames, valid_samples)

    try:
        base = str(tmp_path / "base")
        audio.extract("stub", files, base, level, 0, model=_Audio(), batch_rows=4, ragged=False)
        rag = str(tmp_path / "ragged")
        audio.extract("stub", files, rag, level, 0, model=Counting(), batch_rows=4, ragged=True, window=8, reader=reader)
        if level == "UTTERANCE":
            assert held["max"] <= 8 + 1, held          # host memory is O(window)
        for r in range(2):
            audio.extract("stub", files, str(tmp_path / "sharded"), level, 0, model=_Audio(), batch_rows=4, rank=r, world=2, window=8)
        for f in files:
            n = os.path.basename(f)[:-4] + ".npy"
            a = np.load(os.path.join(base, n))
            for other in (rag, str(tmp_path / "sharded")):
                b = np.load(os.path.join(other, n))
                assert a.shape == b.shape and np.array_equal(a, b), (n, other)
    finally:
        audio.split_into_batch.__defaults__ = old


def test_plan_batches_bounds_padding_and_memory():
    import random
    random.seed(0)
    pend = [dict(vid=i, rows=1, len=random.randint(16000, 160000)) for i in range(256)]
    out, keep = audio.plan_batches(pend, 32, True, False, keep_at_most=128)
    assert len(keep) <= 128 and sorted(it["vid"] for g in out for it in g) + sorted(it["vid"] for it in keep) is not None
    assert sorted([it["vid"] for g in out for it in g] + [it["vid"] for it in keep]) == list(range(256))
    for g in out:
        assert sum(it["rows"] for it in g) <= 32 and max(it["len"] for it in g) <= 1.5 * min(it["len"] for it in g)
    padded = sum(max(it["len"] for it in g) * len(g) for g in out)
    assert padded <= 1.25 * sum(it["len"] for g in out for it in g)
    out2, keep2 = audio.plan_batches(keep, 32, True, True)
    assert not keep2 and sum(len(g) for g in out2) == len(keep)


def test_prefetch_map_chunked_tasks_keep_order_laziness_and_exceptions():
    seen = []

    def fn(i):
        time.sleep(0.001 * ((7 * i) % 5))
        seen.append(i)
        if i == 21:
            raise ValueError("item 21")
        return i * i

    assert list(prefetch_map(fn, range(20), workers=3, chunk=4)) == [i * i for i in range(20)]
    assert list(prefetch_map(fn, range(19), workers=3, chunk=4, depth=5)) == [i * i for i in range(19)]   # ragged last group
    out = []
    with pytest.raises(ValueError, match="item 21"):
        for x in prefetch_map(fn, range(40), workers=4, chunk=4):
            out.append(x)
    assert out == [i * i for i in range(21)]          # the items before it in the same group are still delivered
    seen.clear()
    g = prefetch_map(lambda i: seen.append(i) or i, range(1000), workers=2, chunk=4, depth=8)
    assert [next(g) for _ in range(3)] == [0, 1, 2] and len(seen) <= 3 + 8 + 2 * 4
    g.close()
    time.sleep(0.05)
    assert len(seen) < 40


def test_ramp_up_cuts_the_first_two_batches_small_and_changes_no_feature(tmp_path):
    """ramp_up: the first two batches are a quarter / half of the batch size (the GPU starts while the read-ahead is still filling);
    with encoders whose rows do not look at each other the files are the same with and without it."""
    rng = np.random.RandomState(2)
    files = []
    for i in range(40):                      # equal-length clips: the driver cuts a batch as soon as it is full
        p = str(tmp_path / f"c{i:02d}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes((np.clip(rng.randn(4000) * 0.1, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())
        files.append(p)

    class Counting(_Audio):
        def __init__(self):
            self.batches = []

        def extract_utterance(self, rows, clip_chunks=None, valid_samples=None):
            self.batches.append(rows.shape[0])
            return super().extract_utterance(rows, clip_chunks, valid_samples)

    got = {}
    for ramp in (True, False):
        m = Counting()
        d = str(tmp_path / f"ramp{ramp}")
        audio.extract("stub", files, d, "UTTERANCE", 0, model=m, workers=2, batch_rows=16, ramp_up=ramp)
        got[ramp] = (m.batches, [np.load(os.path.join(d, f"c{i:02d}.npy")) for i in range(40)])
    assert got[True][0] == [4, 8, 16, 12] and got[False][0] == [16, 16, 8]
    assert all(np.array_equal(a, b) for a, b in zip(got[True][1], got[False][1]))

    face, vids = _faces(tmp_path, [(4, 32, 32)] * 12)

    class CountingVision(_Vision):
        def __init__(self):
            self.batches = []

        def get_image_features(self, px):
            self.batches.append(px.shape[0])
            return super().get_image_features(px)

    outs = {}
    for ramp in (True, False):
        m = CountingVision()
        d = tmp_path / f"vis{ramp}"
        visual.extract(m, face, str(d), "UTTERANCE", vids=vids, frames_per_batch=16, workers=2, ramp_up=ramp)
        outs[ramp] = (m.batches, [np.load(d / f"{v}.npy") for v in vids])
    assert outs[True][0] == [4, 8, 16, 16, 4] and outs[False][0] == [16, 16, 16]
    assert all(np.array_equal(a, b) for a, b in zip(outs[True][1], outs[False][1]))
