"""Ordered read-ahead for the extraction drivers' host stages (file reads, PIL resizing, waveform normalisation).

The reference reads and pre-processes one clip at a time in the main thread before each forward
(extract_vision_huggingface.py:108-116, extract_audio_huggingface.py:92-95) and uses `multiprocessing.Pool(8)` only for the
fusion loader's .npy reads (toolkit/utils/read_data.py:58).  With the encoders at ~1.8 k clips/s per GPU the host stages
decide the end-to-end rate, so the drivers can run them `workers` threads ahead of the GPU loop: numpy, PIL and file I/O
release the GIL, results come back in input order, exceptions surface at the item that raised them, and at most `depth`
results are alive at a time.  workers = 0 is the plain in-line loop.
"""
from collections import deque
from concurrent.futures import ThreadPoolExecutor

from .pipeline import span


class _Raised:
    __slots__ = ("exc",)

    def __init__(self, exc):
        self.exc = exc


def prefetch_map(fn, items, workers=0, depth=None, chunk=1):
    """Generator over fn(item) for item in items, in order, computed up to `depth` items ahead on `workers` threads.
    chunk: items per task — a task costs tens of microseconds of interpreter time on the consuming thread (future, queue, wake-up),
    which is what a 160 KB wav read costs; chunk > 1 amortises it.  Order, laziness and where an exception surfaces are unchanged."""
    if workers <= 0:
        for it in items:
            yield fn(it)
        return
    chunk = max(1, int(chunk))
    depth = depth or 4 * workers * chunk
    ntasks = max(1, -(-depth // chunk))

    def run(group):
        out = []
        for x in group:
            try:
                out.append(fn(x))
            except BaseException as e:   # delivered when the consumer reaches this item; the rest of the group is not started
                out.append(_Raised(e))
                break
        return out

    def drain(fut):
        with span("read_wait"):
            res = fut.result()
        for r in res:
            if isinstance(r, _Raised):
                raise r.exc
            yield r

    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="mer-prefetch") as pool:
        window = deque()
        it = iter(items)
        try:
            while True:
                group = []
                for x in it:
                    group.append(x)
                    if len(group) >= chunk:
                        break
                if not group:
                    break
                window.append(pool.submit(run, group))
                if len(window) >= ntasks:
                    yield from drain(window.popleft())
            while window:
                yield from drain(window.popleft())
        finally:
            for f in window:   # consumer stopped early or an item raised: drop what has not started yet
                f.cancel()
