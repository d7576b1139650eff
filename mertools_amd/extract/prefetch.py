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


def prefetch_map(fn, items, workers=0, depth=None):
    """Generator over fn(item) for item in items, in order, computed up to `depth` items ahead on `workers` threads."""
    if workers <= 0:
        for it in items:
            yield fn(it)
        return
    depth = depth or 4 * workers
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="mer-prefetch") as pool:
        window = deque()
        it = iter(items)
        try:
            for x in it:
                window.append(pool.submit(fn, x))
                if len(window) >= depth:
                    yield window.popleft().result()
            while window:
                yield window.popleft().result()
        finally:
            for f in window:   # consumer stopped early or an item raised: drop what has not started yet
                f.cancel()
