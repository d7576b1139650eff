"""Pillow-exact bicubic resize of 8-bit frames on the GPU (reference a6).

The reference resizes every frame on the host: `processor(images=...)` -> CLIPImageProcessor.resize -> PIL
`Image.resize(size, resample=BICUBIC)` on a uint8 image, then center_crop, then rescale / normalise
(MERBench/feature_extraction/visual/extract_vision_huggingface.py:116).  Pillow's 8-bit resampler is integer arithmetic:
two separable passes whose per-output-pixel windows and fixed-point coefficients (22 fractional bits) depend only on
(input size, output size).  This module builds those tables on the host, once per size pair (`pil_coeffs`), and hands
them to `mer_image_resize_crop_u8`, which runs the two passes on the GPU for exactly the cropped region — byte-identical
to Pillow (oracle: `oracle/host_ref.py:pil_resize_bicubic_u8`, itself pinned against the installed Pillow).
"""
import functools
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    """Pillow's bicubic_filter (a = -0.5), vectorised over a float64 array."""
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                    np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


@functools.lru_cache(maxsize=256)
def pil_coeffs(in_size, out_size):
    """-> (bounds int32 [out, 2] = (first input index, tap count), coeffs int32 [out, ksize], ksize) for one axis."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)        # astype truncates toward zero like a C cast
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = _bicubic((taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(taps < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                                                  # Pillow sums the taps left to right
        ww = ww + w[:, t]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
    fixed = np.where(taps < xmax[:, None], fixed, 0)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, fixed.astype(np.int32), ksize


def shortest_edge_geometry(h, w, size, crop=None):
    """CLIPImageProcessor geometry (HF:image_processing_clip.py resize(shortest_edge) + center_crop): the short side becomes
    `size`, the long side int(size * long / short); the crop window of `crop` (default `size`) is centred with floor division.
    -> (new_w, new_h, left, top, crop)"""
    crop = crop or size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    return new_w, new_h, (new_w - crop) // 2, (new_h - crop) // 2, crop


_DEVICE_TABLES = {}


def device_tables(in_size, out_size, device):
    """The axis tables as device int32 tensors, cached per (sizes, device)."""
    import torch
    key = (in_size, out_size, str(device))
    t = _DEVICE_TABLES.get(key)
    if t is None:
        b, k, ks = pil_coeffs(in_size, out_size)
        t = _DEVICE_TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), ks, b)
    return t


def resize_crop_u8(frames, size, crop=None):
    """Device uint8 [N, h, w, 3] -> device uint8 [N, crop, crop, 3]: Pillow-exact bicubic shortest-edge resize + centre crop."""
    import torch
    from .. import _lib
    from ..ops import stream
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
    N, h, w, _ = frames.shape
    new_w, new_h, left, top, crop = shortest_edge_geometry(h, w, size, crop)
    assert left >= 0 and top >= 0, "crop larger than the resized image is not supported"   # (HF pads in that case)
    xb, xk, xks, _ = device_tables(w, new_w, frames.device)
    yb, yk, yks, yb_host = device_tables(h, new_h, frames.device)
    y0 = int(yb_host[top, 0])
    y1 = int(yb_host[top + crop - 1, 0] + yb_host[top + crop - 1, 1])
    tmp = torch.empty((N, y1 - y0, crop, 3), dtype=torch.uint8, device=frames.device)
    out = torch.empty((N, crop, crop, 3), dtype=torch.uint8, device=frames.device)
    _lib.check(_lib.lib().mer_image_resize_crop_u8(frames.data_ptr(), N, h, w, left, top, crop, crop,
                                                   xb.data_ptr(), xk.data_ptr(), xks, yb.data_ptr(), yk.data_ptr(), yks, y0, y1,
                                                   tmp.data_ptr(), out.data_ptr(), stream()), "mer_image_resize_crop_u8")
    return out
