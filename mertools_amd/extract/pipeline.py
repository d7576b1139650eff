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
npy_save      np.save for the plain C-contiguous arrays the drivers write, without its per-call Python overhead (the .npy header is
              built by numpy's own header writer, once per dtype and shape): the same bytes in a third of the interpreter time —
              the writer threads share the GIL with the thread that feeds the GPU.
read_into_pinned   a .npy frame stack read straight into a pinned tensor on the calling (read-ahead) thread: no pageable copy, and
              no 1 MB memcpy per video on the GPU-feeding thread.
"""
import io
import os
import queue
import threading
import time

import numpy as np
import torch

_NPY_HEADERS = {}

# MER_EXTRACT_TRACE=1: wall time the drivers' GPU-feeding thread spends per stage (read_wait = blocked on the read-ahead threads,
# stage = batch assembly + upload, forward = the encoder call, submit = handing results to the writer, drain = waiting for the writer
# at the end), accumulated in TRACE and reported by bench.py --e2e.  Off: span() returns one shared do-nothing context manager.
TRACE = {}
_TRACE_ON = os.environ.get("MER_EXTRACT_TRACE", "") not in ("", "0")


class _Span:
    __slots__ = ("name", "t0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *a):
        TRACE[self.name] = TRACE.get(self.name, 0.0) + time.perf_counter() - self.t0
        return False


class _NoSpan:
    def __enter__(self):
        pass

    def __exit__(self, *a):
        return False


_NO_SPAN = _NoSpan()


def span(name):
    return _Span(name) if _TRACE_ON else _NO_SPAN


def trace_enable(on=True):
    """Switches the stage timers on / off and clears TRACE (bench.py --e2e; MER_EXTRACT_TRACE=1 switches them on at import)."""
    global _TRACE_ON
    _TRACE_ON = bool(on)
    TRACE.clear()


def npy_save(file, arr):
    """np.save(file, arr), byte for byte (tests/test_round3_cpu.py compares them), for str paths and plain ndarrays; anything else
    goes to np.save itself."""
    a = np.asanyarray(arr)
    if type(a) is not np.ndarray or a.dtype.hasobject or not a.flags.c_contiguous or not isinstance(file, (str, os.PathLike)):
        return np.save(file, arr)
    key = (a.dtype.str, a.shape)
    head = _NPY_HEADERS.get(key)
    if head is None:
        b = io.BytesIO()
        np.lib.format.write_array_header_1_0(b, np.lib.format.header_data_from_array_1_0(a))
        head = b.getvalue()
        if len(_NPY_HEADERS) < 4096:   # FRAME-level features: one entry per distinct frame count
            _NPY_HEADERS[key] = head
    file = os.fspath(file)
    if not file.endswith('.npy'):
        file = file + '.npy'
    with open(file, 'wb') as f:
        f.write(head)
        f.write(a.data if a.ndim else a.tobytes())


_NPY_PARSED = {}   # header bytes -> (shape, torch dtype) or None: stacks of one shape share one header, parsed once


def _npy_meta(version, len_bytes, hdr):
    key = (version, hdr)
    if key in _NPY_PARSED:
        return _NPY_PARSED[key]
    try:
        read = np.lib.format.read_array_header_1_0 if version == 1 else np.lib.format.read_array_header_2_0
        shape, fortran, dtype = read(io.BytesIO(len_bytes + hdr))
        if fortran or dtype.hasobject or dtype.byteorder == '>' or len(shape) == 0:
            meta = None
        else:
            meta = (tuple(shape), torch.from_numpy(np.empty(0, dtype=dtype)).dtype)   # TypeError: no torch equivalent
    except (ValueError, TypeError):
        meta = None
    if len(_NPY_PARSED) < 1024:
        _NPY_PARSED[key] = meta
    return meta


def read_into_pinned(path, pin=True):
    """A version-1/2 .npy file of a plain (non-object, C-order) array -> pinned CPU tensor holding its contents, or None when the file
    is anything else (the caller then takes np.load).  Called on read-ahead threads: unbuffered reads straight into the pinned pages
    (they release the GIL); the interpreter time per file is an open, a dictionary lookup of the header bytes and an allocation from
    torch's pinned-memory cache."""
    try:
        with open(path, 'rb', buffering=0) as f:
            head = f.read(10)
            if len(head) < 10 or head[:6] != b'\x93NUMPY' or head[6] not in (1, 2):
                return None
            len_bytes = head[8:10] if head[6] == 1 else head[8:10] + f.read(2)
            hlen = int.from_bytes(len_bytes, 'little')
            hdr = f.read(hlen)
            if len(hdr) != hlen:
                return None
            meta = _npy_meta(head[6], len_bytes, hdr)
            if meta is None:
                return None
            out = torch.empty(meta[0], dtype=meta[1], pin_memory=pin)   # (pin=False: the CPU tests)
            if out.numel() == 0:
                return out
            view = memoryview(out.numpy()).cast('B')
            got = 0
            while got < len(view):
                n = f.readinto(view[got:])
                if not n:
                    return None   # truncated file
                got += n
            return out
    except OSError:
        return None


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
        with span("drain"):
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

    def gather(self, host_tensors):
        """Host tensors of one trailing shape and dtype -> ONE device tensor (concatenated along dim 0), each piece copied on the upload
        stream from pinned memory: pieces that are already pinned (read_into_pinned) travel as they are, the others are staged."""
        self.inflight = [(e, p) for e, p in self.inflight if not e.query()]
        first = host_tensors[0]
        rows = sum(int(t.shape[0]) for t in host_tensors)
        with torch.cuda.stream(self.stream):
            d = torch.empty((rows,) + tuple(first.shape[1:]), dtype=first.dtype, device=self.device)
            r, pins = 0, []
            for t in host_tensors:
                pin = t if t.is_pinned() else torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t)
                d[r:r + pin.shape[0]].copy_(pin, non_blocking=True)
                r += pin.shape[0]
                pins.append(pin)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.inflight.append((ev, pins))
        return d

    def ready(self, *tensors):
        """The current (compute) stream waits for every upload issued so far; `tensors` were allocated on the upload stream and are
        about to be read on the compute stream — tell the caching allocator."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        for t in tensors:
            t.record_stream(cur)
