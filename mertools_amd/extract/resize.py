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


def _bilinear(x):
    """Pillow's bilinear_filter (triangle, support 1)."""
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


@functools.lru_cache(maxsize=256)
def pil_coeffs(in_size, out_size, filt="bicubic"):
    """-> (bounds int32 [out, 2] = (first input index, tap count), coeffs int32 [out, ksize], ksize) for one axis."""
    _filter, base_support = FILTERS[filt]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)        # astype truncates toward zero like a C cast
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = _filter((taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
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


def device_tables(in_size, out_size, device, filt="bicubic"):
    """The axis tables as device int32 tensors, cached per (sizes, filter, device)."""
    import torch
    key = (in_size, out_size, filt, str(device))
    t = _DEVICE_TABLES.get(key)
    if t is None:
        b, k, ks = pil_coeffs(in_size, out_size, filt)
        t = _DEVICE_TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), ks, b)
    return t


def resize_crop_u8(frames, size, crop=None, filt="bicubic", exact=False):
    """Device uint8 [N, h, w, 3] -> device uint8 [N, crop, crop, 3]: Pillow-exact resize + centre crop.
    exact=False: the shortest edge becomes `size` (CLIP / VideoMAE / DINOv2 processors); exact=True: resize to size x size
    without keeping the aspect ratio and without a crop (BEiT / data2vec-vision processor)."""
    import torch
    from .. import _lib
    from ..ops import stream
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
    N, h, w, _ = frames.shape
    if exact:
        new_w, new_h, left, top, crop = size, size, 0, 0, size
    else:
        new_w, new_h, left, top, crop = shortest_edge_geometry(h, w, size, crop)
    assert left >= 0 and top >= 0, "crop larger than the resized image is not supported"   # (HF pads in that case)
    xb, xk, xks, _ = device_tables(w, new_w, frames.device, filt)
    yb, yk, yks, yb_host = device_tables(h, new_h, frames.device, filt)
    y0 = int(yb_host[top, 0])
    y1 = int(yb_host[top + crop - 1, 0] + yb_host[top + crop - 1, 1])
    tmp = torch.empty((N, y1 - y0, crop, 3), dtype=torch.uint8, device=frames.device)
    out = torch.empty((N, crop, crop, 3), dtype=torch.uint8, device=frames.device)
    _lib.check(_lib.lib().mer_image_resize_crop_u8(frames.data_ptr(), N, h, w, left, top, crop, crop,
                                                   xb.data_ptr(), xk.data_ptr(), xks, yb.data_ptr(), yk.data_ptr(), yks, y0, y1,
                                                   tmp.data_ptr(), out.data_ptr(), stream()), "mer_image_resize_crop_u8")
    return out


# processor recipes of the reference's visual branches (extract_vision_huggingface.py:116,125,137,151): geometry, filter, mean / std
_IMAGENET = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
RECIPES = {
    "clip": dict(size=224, crop=None, filt="bicubic", exact=False, mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)),
    "videomae": dict(size=224, crop=None, filt="bilinear", exact=False, mean=_IMAGENET[0], std=_IMAGENET[1]),
    "dinov2": dict(size=256, crop=224, filt="bicubic", exact=False, mean=_IMAGENET[0], std=_IMAGENET[1]),
    "data2vec-vision": dict(size=224, crop=None, filt="bicubic", exact=True, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
}


def recipe_geometry(recipe, h, w, size=None):
    """-> dict(new_w, new_h, left, top, crop, filt, mean, std, size, exact) for frames of h x w under a branch's processor;
    `size` overrides the model resolution (default 224; DINOv2 keeps resizing to 256 and crops `size`, as dinov2_preprocess)."""
    r = dict(RECIPES[recipe])
    if size is not None:
        if r["crop"]:
            r["crop"] = size
        else:
            r["size"] = size
    if r["exact"]:
        new_w, new_h, left, top, crop = r["size"], r["size"], 0, 0, r["size"]
    else:
        new_w, new_h, left, top, crop = shortest_edge_geometry(h, w, r["size"], r["crop"])
    r.update(new_w=new_w, new_h=new_h, left=left, top=top, crop=crop)
    return r


def device_preprocess_u8(frames_bgr, device, recipe, size=None):
    """Host uint8 BGR frames [N, h, w, 3] (what the reference's readers return) -> device float32 [N, 3, S, S] exactly as the
    branch's HF image processor would produce: bytes over PCIe, Pillow-exact resize + centre crop and the rescale / normalise
    arithmetic on the GPU."""
    import torch
    from .. import ops
    px = torch.as_tensor(np.ascontiguousarray(frames_bgr)).to(device)
    g = recipe_geometry(recipe, px.shape[1], px.shape[2], size)
    if (g["new_w"], g["new_h"]) != (px.shape[2], px.shape[1]) or g["crop"] != px.shape[1] or g["crop"] != px.shape[2]:
        px = resize_crop_u8(px, g["size"], crop=g["crop"], filt=g["filt"], exact=g["exact"])
    return ops.image_normalize_u8(px, g["mean"], g["std"], bgr=True)
