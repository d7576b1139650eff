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
    device = torch.device("cpu")

    def out_frames(self, L):
        return L // 320

    def extract_utterance(self, rows, clip_chunks=None):
        out, r = [], 0
        for n in clip_chunks:
            out.append(torch.stack([rows[r:r + n].mean(), rows[r:r + n].abs().max(), rows[r, 7]]))
            r += n
        return torch.stack(out)


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
