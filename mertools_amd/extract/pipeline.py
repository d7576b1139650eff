"""Keeps the GPU-feeding thread of the extraction drivers free of everything that can wait.

The reference does, per clip and on one thread: forward -> `.cpu().numpy()` (blocks until the GPU is done) -> `np.save`
(extract_audio_huggingface.py:97-110).  With the encoders at ~2 k clips/s a batch is a few milliseconds of GPU time, so a
driver that blocks on every D2H copy and writes 64 files before it queues the next batch leaves the GPU idle most of the time.

AsyncWriter   results leave the device through pinned staging buffers with non-blocking copies; worker threads wait for the
              copy's event, take the bytes out of the staging buffer and run the caller's save function (np.save releases the
              GIL).  The bytes written are the bytes the synchronous path writes: same tensors, same np.save calls.
Uploader      host -> device copies on a side stream from pinned staging buffers, so that the next batch's inputs travel
              while the current batch computes; `ready()` orders the compute stream behind everything uploaded so far.
Both are bounded (a fixed number of staging slots): host memory stays O(slots x batch), and a slow disk back-pressures the
producer instead of growing a queue.
"""
import queue
import threading

import torch


class AsyncWriter:
    def __init__(self, device, workers=2, slots=4):
        self.device = torch.device(device)
        self.q = queue.Queue(maxsize=max(1, slots))
        self.err = None
        self.threads = [threading.Thread(target=self._run, name=f"mer-writer-{i}", daemon=True) for i in range(max(1, workers))]
        for t in self.threads:
            t.start()

    def _run(self):
        while True:
            item = self.q.get()
            try:
                if item is None:
                    return
                ev, pins, fn, keep = item
                if self.err is None:
                    ev.synchronize()
                    arrays = [p.numpy() for p in pins]   # views of the pinned buffers: alive until fn returns
                    fn(*arrays)
                del keep
            except BaseException as e:   # surfaced by the producer at the next submit() / close()
                self.err = self.err or e
            finally:
                self.q.task_done()

    def submit(self, tensors, fn):
        """tensors: device tensor or list of them; fn(*numpy_arrays) runs on a worker thread once the copies have landed.
        Called with the producing stream current.  Blocks only when every staging slot is in flight."""
        if self.err is not None:
            raise self.err
        single = torch.is_tensor(tensors)
        ts = [tensors] if single else list(tensors)
        pins = []
        for t in ts:
            p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            p.copy_(t, non_blocking=True)
            pins.append(p)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.q.put((ev, pins, fn, ts))   # `ts` keeps the device tensors alive until the copy has been waited for

    def close(self):
        """Drains the queue, stops the workers, re-raises the first error a save function raised."""
        self.q.join()
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err is not None:
            raise self.err

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:   # the producer failed: do not mask its exception
            try:
                self.close()
            except BaseException:
                pass
        return False


class SyncWriter:
    """The reference's behaviour (blocking D2H, save on the calling thread) behind the same interface — the drivers' async=False."""

    def __init__(self, device=None, **_):
        pass

    def submit(self, tensors, fn):
        ts = [tensors] if torch.is_tensor(tensors) else list(tensors)
        fn(*[t.cpu().numpy() for t in ts])

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def writer(device, asynchronous=True, workers=2, slots=4):
    dev = torch.device(device)
    if asynchronous and dev.type == "cuda":
        return AsyncWriter(dev, workers=workers, slots=slots)
    return SyncWriter()


class Uploader:
    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.inflight = []   # (event, pinned buffer): a staging buffer is reusable once its copy has completed

    def up(self, host_tensor):
        """Host tensor -> device tensor, copied on the upload stream (pinned staging, non-blocking)."""
        self.inflight = [(e, p) for e, p in self.inflight if not e.query()]
        pin = host_tensor if host_tensor.is_pinned() else torch.empty(host_tensor.shape, dtype=host_tensor.dtype, pin_memory=True).copy_(host_tensor)
        with torch.cuda.stream(self.stream):
            d = pin.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.inflight.append((ev, pin))
        return d

    def ready(self, *tensors):
        """The current (compute) stream waits for every upload issued so far; `tensors` were allocated on the upload stream and are
        about to be read on the compute stream — tell the caching allocator."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        for t in tensors:
            t.record_stream(cur)
