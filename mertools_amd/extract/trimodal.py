"""Tri-modal extraction engine: the three encoders of one clip batch on three HIP streams, with the next batch's
host -> device copies running under the current batch's kernels.

The reference extracts one modality per script and one clip per forward
(MERBench/feature_extraction/{audio,visual,text}/extract_*_huggingface.py); `bench.py` measures the three encoders of a
64-clip batch together with the inputs already in HBM.  This module is the deployable form of that step:

  stage(k)    the batch's tensors are copied into pinned host buffers and sent up on a dedicated copy stream
              (one event per batch slot) — this overlaps compute(k-1)
  compute(k)  each modality's stream waits for the copy event, runs `extract_utterance` (the fused HIP forward) and
              queues the [B, D] result into a pinned output buffer (another event)
  finish(k-1) the host waits for batch k-1's output events only, then hands out numpy arrays

Two batch slots are alive at any time, so device input buffers, pinned buffers and the encoders' per-stream workspaces
are reused without a device-wide synchronisation.  Inputs may be the reference's fp32 tensors or the compact forms
(int16 PCM, uint8 BGR frames) that the pre-processing kernels expand on the GPU (a quarter of the H2D bytes).

On a CPU device (unit tests with stub encoders) the same code runs without streams.  There is no CPU fallback for the
real encoders: they raise on CPU tensors.
"""
import os

import numpy as np
import torch

MODALITIES = ("visual", "audio", "text")   # launch order: longest first


class _Slot:
    """Buffers of one in-flight batch."""

    def __init__(self):
        self.pinned_in = {}
        self.dev_in = {}
        self.pinned_out = {}
        self.copied = None      # event: H2D of this slot finished
        self.done = {}          # modality -> event: result is in pinned_out
        self.meta = None
        self.views = {}
        self.out_rows = {}


class TriModalExtractor:
    """audio / visual / text: encoder objects with `extract_utterance` (HipHubertModel, HipCLIPModel, HipBertModel or
    stand-ins); any subset may be given.

    A batch is a dict:
      names            list of B clip ids
      audio            [B, L] float32 (already normalised) or int16 PCM (normalised on the GPU, `audio_do_normalize`)
      frames           [B*F, 3, H, W] float32 (processor output) or [B*F, H, W, 3] uint8 BGR (normalised on the GPU)
      frames_per_clip  list of B ints (sum = rows of `frames`)
      input_ids        [B, T] int64, `lengths` list of B ints (tokens incl. specials)
    `audio` / `frames` may also be LISTS of per-clip arrays of one shape each ([L] / [F, ...]): they are gathered straight into the
    pinned staging buffer (one host copy instead of stack + copy), on `copy_workers` threads (numpy's memcpy releases the GIL).
    """

    def __init__(self, audio=None, visual=None, text=None, device="cuda:0", text_strip=(1, -1), audio_do_normalize=True,
                 image_mean=None, image_std=None, copy_workers=4, save_workers=2):
        self.models = {"audio": audio, "visual": visual, "text": text}
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.text_strip = text_strip
        self.audio_do_normalize = audio_do_normalize
        self.image_mean, self.image_std = image_mean, image_std
        self.slots = [_Slot(), _Slot()]
        self.copy_workers, self.save_workers = max(1, int(copy_workers)), max(1, int(save_workers))
        self._copy_pool = None
        if self.cuda:
            self.copy_stream = torch.cuda.Stream(device=self.device)
            self.streams = {m: torch.cuda.Stream(device=self.device) for m in MODALITIES if self.models[m] is not None}
        else:
            self.copy_stream, self.streams = None, {}

    # ---- buffers -------------------------------------------------------------------------------------------
    def _pinned(self, store, key, like_shape, dtype):
        n = int(np.prod(like_shape))
        buf = store.get(key)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=dtype)
            if self.cuda:
                buf = buf.pin_memory()
            store[key] = buf
        return buf[:n].view(like_shape)

    def _device(self, slot, key, shape, dtype):
        n = int(np.prod(shape))
        buf = slot.dev_in.get(key)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            slot.dev_in[key] = buf
        return buf[:n].view(shape)

    # ---- pipeline stages -----------------------------------------------------------------------------------
    def _stage(self, slot, batch):
        slot.meta = {k: batch.get(k) for k in ("names", "frames_per_clip", "lengths")}
        views = {}
        for key in ("audio", "frames", "input_ids"):
            t = batch.get(key)
            if t is None:
                continue
            if isinstance(t, (list, tuple)) and key != "input_ids":
                host = self._gather(slot, key, t)                   # per-clip arrays -> their rows of the pinned buffer, in parallel
            else:
                t = torch.as_tensor(t)
                host = self._pinned(slot.pinned_in, key, tuple(t.shape), t.dtype)
                host.copy_(t)                                       # CPU memcpy into the pinned staging buffer
            views[key] = (host, self._device(slot, key, tuple(host.shape), host.dtype))
        if self.cuda:
            with torch.cuda.stream(self.copy_stream):
                for host, dev in views.values():
                    dev.copy_(host, non_blocking=True)
                slot.copied = torch.cuda.Event()
                slot.copied.record(self.copy_stream)
        else:
            for host, dev in views.values():
                dev.copy_(host)
        slot.views = {k: v[1] for k, v in views.items()}

    def _gather(self, slot, key, parts):
        parts = [np.asarray(x) for x in parts]
        first = parts[0]
        lead = [1 if key == "audio" else x.shape[0] for x in parts]
        tail = first.shape if key == "audio" else first.shape[1:]
        for x in parts:
            if (x.shape if key == "audio" else x.shape[1:]) != tail or x.dtype != first.dtype:
                raise ValueError(f"TriModalExtractor: the per-clip `{key}` arrays of a batch must share shape and dtype")
        host = self._pinned(slot.pinned_in, key, (sum(lead),) + tuple(tail), torch.from_numpy(first[:0]).dtype)
        dst = host.numpy()
        offs = np.concatenate([[0], np.cumsum(lead)])

        def put(i):
            if key == "audio":
                dst[offs[i]] = parts[i]
            else:
                dst[offs[i]:offs[i + 1]] = parts[i]
        if self.copy_workers > 1 and len(parts) > 1:
            if self._copy_pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._copy_pool = ThreadPoolExecutor(self.copy_workers, thread_name_prefix="mer-stage")
            list(self._copy_pool.map(put, range(len(parts))))
        else:
            for i in range(len(parts)):
                put(i)
        return host

    def _forward(self, m, slot):
        x = slot.views
        meta = slot.meta
        if m == "audio":
            a = x["audio"]
            if a.dtype == torch.int16:                              # PCM over PCIe, normalised on the GPU
                from .. import ops
                a = ops.wave_normalize(a, self.audio_do_normalize)
            return self.models[m].extract_utterance(a)
        if m == "visual":
            f = x["frames"]
            if f.dtype == torch.uint8:                              # BGR bytes over PCIe, processor arithmetic on the GPU
                from .. import ops
                from .visual import CLIP_MEAN, CLIP_STD
                f = ops.image_normalize_u8(f, self.image_mean or CLIP_MEAN, self.image_std or CLIP_STD, bgr=True)
            return self.models[m].extract_utterance(f, list(meta["frames_per_clip"]))
        start, end = self.text_strip
        return self.models[m].extract_utterance(x["input_ids"], list(meta["lengths"]), start, end)

    def _compute(self, slot):
        slot.done = {}
        need = {"audio": "audio", "visual": "frames", "text": "input_ids"}
        for m in MODALITIES:
            if self.models[m] is None or need[m] not in slot.views:
                continue
            if self.cuda:
                st = self.streams[m]
                st.wait_event(slot.copied)
                with torch.cuda.stream(st):
                    res = self._forward(m, slot)
                    out = self._pinned(slot.pinned_out, m, tuple(res.shape), res.dtype)
                    out.copy_(res, non_blocking=True)
                    res.record_stream(st)
                    ev = torch.cuda.Event()
                    ev.record(st)
            else:
                res = self._forward(m, slot)
                out = self._pinned(slot.pinned_out, m, tuple(res.shape), res.dtype)
                out.copy_(res)
                ev = None
            slot.done[m] = ev
            slot.out_rows[m] = out

    def _finish(self, slot):
        feats = {}
        for m, ev in slot.done.items():
            if ev is not None:
                ev.synchronize()
            feats[m] = slot.out_rows[m].numpy().copy()             # the pinned buffer is reused two batches later
        return slot.meta["names"], feats

    # ---- driver --------------------------------------------------------------------------------------------
    def run(self, batches):
        """Generator over (names, {"audio": [B, Da], "visual": [B, Dv], "text": [B, Dt]}) in input order."""
        pending = None
        for k, batch in enumerate(batches):
            slot = self.slots[k & 1]                                # batch k-2 (same slot) was finished one iteration ago
            self._stage(slot, batch)
            self._compute(slot)
            if pending is not None:
                yield self._finish(pending)
            pending = slot
        if pending is not None:
            yield self._finish(pending)

    def extract_to_dirs(self, batches, save_dirs):
        """Writes `<save_dirs[m]>/<clip>.npy` (UTT layout of the reference: float32 [D]) for every modality present."""
        from concurrent.futures import ThreadPoolExecutor
        from .pipeline import npy_save
        for d in save_dirs.values():
            os.makedirs(d, exist_ok=True)

        def save(names, feats):                                     # `feats` are copies (run()): nothing overwrites them
            for m, arr in feats.items():
                if m in save_dirs:
                    for name, row in zip(names, arr):
                        npy_save(os.path.join(save_dirs[m], f"{name}.npy"), row)   # np.save's bytes without its per-call overhead
        n, inflight = 0, []
        with ThreadPoolExecutor(self.save_workers, thread_name_prefix="mer-save") as pool:   # the feeding thread keeps feeding
            for names, feats in self.run(batches):
                inflight.append(pool.submit(save, names, feats))
                while len(inflight) > 2 * self.save_workers:       # bounded: a slow disk holds the pipeline back, not memory
                    inflight.pop(0).result()
                n += len(names)
            for f in inflight:
                f.result()                                          # a write error surfaces here
        return n
