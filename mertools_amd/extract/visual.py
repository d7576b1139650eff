"""Visual feature extraction — mirror of MERBench/feature_extraction/visual/extract_vision_huggingface.py:29-189
(CLIP branch).  Frame sampling / batching / save rules are the reference's; the vision tower runs on the HIP
encoder, frames of several videos share a batch, and the per-video frame mean is taken on the GPU."""
import math
import os

import numpy as np
import torch

from .pipeline import npy_save, span

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def func_read_frames(face_dir, vid):
    npy_path = os.path.join(face_dir, vid, f'{vid}.npy')
    assert os.path.exists(npy_path), f'Error: {vid} does not have frames.npy!'
    return np.load(npy_path)


def resample_frames_uniform(frames, nframe=16):
    """Uniform index pick, last index repeated when the video is short (reference :44-56; integer path, bit-exact)."""
    vlen = len(frames)
    n_frms_update = min(nframe, vlen)
    indices = np.arange(0, vlen, vlen / n_frms_update).astype(int).tolist()
    while len(indices) < nframe:
        indices.append(indices[-1])
    indices = indices[:nframe]
    assert len(indices) == nframe, f'{indices}, {vlen}, {nframe}'
    return frames[indices]


def split_into_batch(inputs, bsize=32):
    return [inputs[ii * bsize:(ii + 1) * bsize] for ii in range(math.ceil(len(inputs) / bsize))]


def clip_preprocess(frames_bgr, size=224):
    """func_opencv_to_image + CLIPImageProcessor (reference :29-31,116): BGR->RGB, resize shortest edge to `size`
    (PIL bicubic), centre crop, /255, normalise.  uint8 [N,h,w,3] -> float32 [N,3,size,size]."""
    from PIL import Image
    out = np.empty((len(frames_bgr), 3, size, size), dtype=np.float32)
    mean = np.array(CLIP_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(CLIP_STD, dtype=np.float32)[:, None, None]
    for i, f in enumerate(frames_bgr):
        img = Image.fromarray(np.ascontiguousarray(f[:, :, ::-1]))
        w, h = img.size
        if (w, h) != (size, size):
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = size, int(size * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
            img = img.resize((nw, nh), resample=Image.BICUBIC)
            left, top = (nw - size) // 2, (nh - size) // 2
            img = img.crop((left, top, left + size, top + size))
        arr = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
        out[i] = (arr - mean) / std
    return torch.from_numpy(out)


def save_embeddings(save_file, embeddings, feature_level, embedding_dim):
    """np.save rules of reference :171-189 (incl. the zero fallback for videos without frames)."""
    embeddings = np.array(embeddings).squeeze()
    if feature_level == 'FRAME':
        if len(embeddings) == 0:
            embeddings = np.zeros((1, embedding_dim))
        elif len(embeddings.shape) == 1:
            embeddings = embeddings[np.newaxis, :]
    else:
        if len(embeddings) == 0:
            embeddings = np.zeros((embedding_dim,))
        elif len(embeddings.shape) == 2:
            embeddings = np.mean(embeddings, axis=0)
    npy_save(save_file, embeddings)   # np.save's bytes (extract.pipeline)


def extract(model, face_dir, save_dir, feature_level='UTTERANCE', vids=None, frames_per_batch=512, reader=func_read_frames,
            device_preprocess=False, workers=0, rank=None, world=None, async_save=True, ramp_up=True):
    """CLIP branch of the reference main loop.  `model`: HipCLIPModel.  One .npy per video.
    device_preprocess: frames that already have the model's resolution go to the GPU as uint8 (a quarter of the fp32 bytes)
    and are rescaled / normalised there (mer_image_normalize_u8); other sizes keep the host PIL path —
    unless device_preprocess == "resize": then every uint8 video goes up as bytes and the Pillow-exact bicubic resize +
    centre crop runs on the GPU as well (extract.resize / mer_image_resize_crop_u8; no PIL on the host at all).
    workers: threads that read and pre-process videos ahead of the GPU loop (extract.prefetch; 0 = in line, as the reference).
    rank / world: this process's share of the videos (distributed.my_share: sorted(vids)[rank::world]; default = the
    torch.distributed rank / world size).  Videos are independent: no collective.  (The reference's EMBEDDING_DIM fallback for
    an empty video is the running maximum over the videos THIS process has seen before it.)
    ramp_up: the first two batches are cut at a quarter / half of `frames_per_batch`: the GPU starts while the read-ahead is still
    filling instead of after a full batch of frame stacks has been read (a fixed cost per run, visible on small corpora)."""
    from .prefetch import prefetch_map
    from ..distributed import my_share
    os.makedirs(save_dir, exist_ok=True)
    vids = my_share(vids if vids is not None else os.listdir(face_dir), rank, world)
    embedding_dim = -1
    pending, nframes, emitted = [], 0, 0
    size = model.config.vision_config.image_size

    up = None
    if device_preprocess:
        from .pipeline import Uploader
        up = Uploader(model.device)

    def flush():
        nonlocal pending, nframes, embedding_dim, emitted
        if not pending:
            return
        emitted += 1
        if all(p.dtype == torch.uint8 and not p.is_cuda for _, p in pending) and len({tuple(p.shape[1:]) for _, p in pending}) == 1:
            # device_preprocess fast path: the batch's uint8 frames (all of the model's resolution, or all of one size awaiting the GPU
            # resize) travel as ONE block on the upload stream and are resized / normalised by one launch each — no per-video H2D copy,
            # kernel launch or device allocation on this thread
            from .. import ops
            with span("stage"), torch.cuda.device(model.device):   # (launch on the model's device, whichever is current)
                block = up.gather([p for _, p in pending])
                up.ready(block)
                if tuple(block.shape[1:3]) != (size, size):
                    from .resize import resize_crop_u8
                    block = resize_crop_u8(block, size)
                px = ops.image_normalize_u8(block, CLIP_MEAN, CLIP_STD, bgr=True)
        else:
            with span("stage"):
                px = torch.cat([_to_px(p) for _, p in pending], 0)
        counts = [p.shape[0] for _, p in pending]
        with span("forward"):
            feats = model.get_image_features(px)  # [sum(frames), P], on the device
        embedding_dim = max(embedding_dim, feats.shape[-1])

        def save(arr, vids=[vid for vid, _ in pending], counts=counts, dim=embedding_dim):
            r = 0
            for vid, n in zip(vids, counts):
                save_embeddings(os.path.join(save_dir, f'{vid}.npy'), arr[r:r + n], feature_level, dim)
                r += n
        with span("submit"):
            out.submit(feats, save)   # async_save: pinned non-blocking D2H + np.save on worker threads (extract.pipeline)
        pending, nframes = [], 0

    def host_stage(vid):   # everything that needs no GPU: file read + (PIL) pre-processing
        frames = None
        if device_preprocess and reader is func_read_frames:   # the frame stack straight into pinned memory (no pageable copy)
            npy_path = os.path.join(face_dir, vid, f'{vid}.npy')
            assert os.path.exists(npy_path), f'Error: {vid} does not have frames.npy!'
            from .pipeline import read_into_pinned
            pin = read_into_pinned(npy_path)
            if pin is not None and pin.dtype == torch.uint8 and pin.dim() == 4 and len(pin) > 0 and (
                    tuple(pin.shape[1:3]) == (size, size) or device_preprocess == 'resize'):
                return vid, 'u8', pin
            frames = pin.numpy() if pin is not None else None
        if frames is None:
            frames = reader(face_dir, vid)
        if len(frames) == 0:
            return vid, None, None
        if device_preprocess and frames.dtype == np.uint8 and (frames.shape[1:3] == (size, size) or device_preprocess == 'resize'):
            return vid, 'u8', torch.from_numpy(np.ascontiguousarray(frames))
        return vid, 'f32', clip_preprocess(frames, size)

    from .pipeline import writer
    def _to_px(p):
        """one video's frames -> fp32 [n, 3, size, size] on the device (mixed batches: uint8 videos of different sizes, host-PIL ones)"""
        if p.dtype != torch.uint8:
            return p.to(model.device)
        from .. import ops
        p = p.to(model.device)
        with torch.cuda.device(model.device):
            if tuple(p.shape[1:3]) != (size, size):
                from .resize import resize_crop_u8
                p = resize_crop_u8(p, size)
            return ops.image_normalize_u8(p, CLIP_MEAN, CLIP_STD, bgr=True)

    with writer(model.device, async_save) as out:
        for vid, kind, px in prefetch_map(host_stage, vids, workers, chunk=2):
            if kind is None:
                flush()
                save_embeddings(os.path.join(save_dir, f'{vid}.npy'), np.zeros((0,)), feature_level, embedding_dim)
                continue
            cap = max(1, frames_per_batch >> max(0, 2 - emitted)) if ramp_up else frames_per_batch
            if nframes + len(px) > cap:
                flush()
            pending.append((vid, px))
            nframes += len(px)
        flush()



def _branch_preprocess(frames, recipe, host_fn, model, size, device_preprocess):
    """Host PIL path (default, as the reference) or, with device_preprocess == "resize" on uint8 frames, bytes up + the
    Pillow-exact resize / crop / normalise on the GPU (extract.resize recipes)."""
    if device_preprocess == 'resize' and frames.dtype == np.uint8:
        from .resize import device_preprocess_u8
        return device_preprocess_u8(frames, model.device, recipe, size)
    return host_fn(frames)


# ---- VideoMAE branch (reference :147-159) -------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def videomae_preprocess(frames_bgr, size=224):
    """func_opencv_to_numpy + VideoMAEImageProcessor (reference :29-35,151-153): BGR->RGB, resize shortest edge to
    `size` (PIL bilinear), centre crop, /255, ImageNet normalise.  uint8 [16,h,w,3] -> float32 [1,16,3,size,size]."""
    from PIL import Image
    out = np.empty((len(frames_bgr), 3, size, size), dtype=np.float32)
    mean = np.array(IMAGENET_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(IMAGENET_STD, dtype=np.float32)[:, None, None]
    for i, f in enumerate(frames_bgr):
        img = Image.fromarray(np.ascontiguousarray(f[:, :, ::-1]))
        w, h = img.size
        if (w, h) != (size, size):
            short, long = (w, h) if w <= h else (h, w)
            nw, nh = (size, int(size * long / short)) if w <= h else (int(size * long / short), size)
            img = img.resize((nw, nh), resample=Image.BILINEAR)
            left, top = (nw - size) // 2, (nh - size) // 2
            img = img.crop((left, top, left + size, top + size))
        arr = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
        out[i] = (arr - mean) / std
    return torch.from_numpy(out)[None]


def extract_videomae(model, face_dir, save_dir, feature_level='UTTERANCE', vids=None, videos_per_batch=8, reader=func_read_frames,
                     device_preprocess=False, rank=None, world=None):
    """VideoMAE branch of the reference loop: 16 uniformly resampled frames per video -> last_hidden_state ->
    view(8, 196, D).mean(1) -> [8, D] (FRAME) or its mean (UTTERANCE).  `model`: HipVideoMAEModel; videos are batched."""
    os.makedirs(save_dir, exist_ok=True)
    from ..distributed import my_share
    vids = my_share(vids if vids is not None else os.listdir(face_dir), rank, world)
    nseg = model.config.num_frames // model.config.tubelet_size
    pending = []

    def flush():
        if not pending:
            return
        seg = model.extract_segments(torch.cat([p for _, p in pending], 0)).cpu().numpy()
        for i, (vid, _) in enumerate(pending):
            save_embeddings(os.path.join(save_dir, f'{vid}.npy'), seg[i * nseg:(i + 1) * nseg], feature_level, seg.shape[-1])
        pending.clear()

    for vid in vids:
        frames = resample_frames_uniform(reader(face_dir, vid), nframe=model.config.num_frames)
        size = model.config.image_size
        px = _branch_preprocess(frames, 'videomae', lambda f: videomae_preprocess(f, size), model, size, device_preprocess)
        pending.append((vid, px if px.dim() == 5 else px[None]))
        if len(pending) >= videos_per_batch:
            flush()
    flush()


# ---- DINOv2 branch (reference :133-144) ---------------------------------------------------------------------------
def dinov2_preprocess(frames_bgr, size=224, resize_to=256):
    """func_opencv_to_image + the DINOv2 checkpoints' BitImageProcessor (reference :29-31,135): BGR->RGB, resize shortest
    edge to 256 (PIL bicubic), centre crop 224, /255, ImageNet normalise.  uint8 [N,h,w,3] -> float32 [N,3,size,size]."""
    from PIL import Image
    out = np.empty((len(frames_bgr), 3, size, size), dtype=np.float32)
    mean = np.array(IMAGENET_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(IMAGENET_STD, dtype=np.float32)[:, None, None]
    for i, f in enumerate(frames_bgr):
        img = Image.fromarray(np.ascontiguousarray(f[:, :, ::-1]))
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        nw, nh = (resize_to, int(resize_to * long / short)) if w <= h else (int(resize_to * long / short), resize_to)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        # centre crop (HF center_crop: top = (h - crop) // 2 ...; pads when the image is smaller, which cannot happen here)
        top, left = (nh - size) // 2, (nw - size) // 2
        img = img.crop((left, top, left + size, top + size))
        arr = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
        out[i] = (arr - mean) / std
    return torch.from_numpy(out)


def extract_dinov2(model, face_dir, save_dir, feature_level='UTTERANCE', vids=None, frames_per_batch=512, reader=func_read_frames, nframe=64,
                   device_preprocess=False, rank=None, world=None):
    """DINOv2 branch of the reference loop: 64 uniformly resampled frames per video (:134) -> token SUM of the last hidden
    state per frame (:141-142) -> [64, D] (FRAME) or its mean (UTTERANCE).  `model`: HipDinov2Model; videos share batches."""
    os.makedirs(save_dir, exist_ok=True)
    from ..distributed import my_share
    vids = my_share(vids if vids is not None else os.listdir(face_dir), rank, world)
    embedding_dim = -1
    pending, nframes = [], 0

    def flush():
        nonlocal pending, nframes, embedding_dim
        if not pending:
            return
        feats = model.extract_frames(torch.cat([p for _, p in pending], 0)).cpu().numpy()
        embedding_dim = max(embedding_dim, feats.shape[-1])
        r = 0
        for vid, p in pending:
            save_embeddings(os.path.join(save_dir, f'{vid}.npy'), feats[r:r + p.shape[0]], feature_level, embedding_dim)
            r += p.shape[0]
        pending, nframes = [], 0

    for vid in vids:
        frames = reader(face_dir, vid)
        if len(frames) == 0:
            flush()
            save_embeddings(os.path.join(save_dir, f'{vid}.npy'), np.zeros((0,)), feature_level, embedding_dim)
            continue
        size = model._cfg.image_size
        px = _branch_preprocess(resample_frames_uniform(frames, nframe=nframe), 'dinov2', lambda f: dinov2_preprocess(f, size), model, size, device_preprocess)
        if nframes + len(px) > frames_per_batch:
            flush()
        pending.append((vid, px))
        nframes += len(px)
    flush()


# ---- data2vec-vision branch (reference :123-131) ------------------------------------------------------------------
def data2vec_vision_preprocess(frames_bgr, size=224):
    """func_opencv_to_image + BeitImageProcessor of facebook/data2vec-vision-base (reference :29-31,125): BGR->RGB, resize to
    size x size (PIL bicubic, no aspect preservation, no crop), /255, normalise with mean = std = 0.5."""
    from PIL import Image
    out = np.empty((len(frames_bgr), 3, size, size), dtype=np.float32)
    for i, f in enumerate(frames_bgr):
        img = Image.fromarray(np.ascontiguousarray(f[:, :, ::-1]))
        if img.size != (size, size):
            img = img.resize((size, size), resample=Image.BICUBIC)
        arr = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
        out[i] = (arr - np.float32(0.5)) / np.float32(0.5)
    return torch.from_numpy(out)


def extract_data2vec_vision(model, face_dir, save_dir, feature_level='UTTERANCE', vids=None, frames_per_batch=512, reader=func_read_frames,
                            device_preprocess=False, rank=None, world=None):
    """data2vec-vision branch: ALL frames of a video (no resampling) -> token sum of the last hidden state per frame (:130-131)."""
    os.makedirs(save_dir, exist_ok=True)
    from ..distributed import my_share
    vids = my_share(vids if vids is not None else os.listdir(face_dir), rank, world)
    embedding_dim = -1
    pending, nframes = [], 0

    def flush():
        nonlocal pending, nframes, embedding_dim
        if not pending:
            return
        feats = model.extract_frames(torch.cat([p for _, p in pending], 0)).cpu().numpy()
        embedding_dim = max(embedding_dim, feats.shape[-1])
        r = 0
        for vid, p in pending:
            save_embeddings(os.path.join(save_dir, f'{vid}.npy'), feats[r:r + p.shape[0]], feature_level, embedding_dim)
            r += p.shape[0]
        pending, nframes = [], 0

    for vid in vids:
        frames = reader(face_dir, vid)
        if len(frames) == 0:
            flush()
            save_embeddings(os.path.join(save_dir, f'{vid}.npy'), np.zeros((0,)), feature_level, embedding_dim)
            continue
        size = model._cfg.image_size
        px = _branch_preprocess(frames, 'data2vec-vision', lambda f: data2vec_vision_preprocess(f, size), model, size, device_preprocess)
        if nframes + len(px) > frames_per_batch:
            flush()
        pending.append((vid, px))
        nframes += len(px)
    flush()


# ---- by name, as the reference's command line does (reference :18-26,83-101) ---------------------------------------
CLIP_VIT_BASE = 'clip-vit-base-patch32'
CLIP_VIT_LARGE = 'clip-vit-large-patch14'
DATA2VEC_VISUAL = 'data2vec-vision-base-ft1k'
VIDEOMAE_BASE = 'videomae-base'
VIDEOMAE_LARGE = 'videomae-large'
DINO2_LARGE = 'dinov2-large'
DINO2_GIANT = 'dinov2-giant'

# checkpoint architecture (config.model_type, what AutoModel dispatches on) -> (HIP encoder class, driver of that branch)
_BRANCHES = {'clip': ('HipCLIPModel', 'extract'), 'videomae': ('HipVideoMAEModel', 'extract_videomae'),
             'dinov2': ('HipDinov2Model', 'extract_dinov2'), 'data2vec-vision': ('HipData2VecVisionModel', 'extract_data2vec_vision'),
             'beit': ('HipData2VecVisionModel', 'extract_data2vec_vision')}


def load_model(model_name, gpu=0, precision="mean", model_dir=None):
    """`AutoModel.from_pretrained(PATH_TO_PRETRAINED_MODELS/transformers/<model_name>)` (reference :83-91) -> the HIP encoder of the
    checkpoint's architecture + the name of the driver that runs the reference's branch for it.  The branch follows the checkpoint's
    `config.model_type` — which is what the reference's name lists amount to (CLIP_VIT_* are CLIP checkpoints, DINO2_* DINOv2 ...) —
    so a differently named directory holding one of these architectures works too."""
    from transformers import AutoModel
    from .. import config, encoders
    from .._lib import MerError
    if model_dir is None:
        model_dir = os.path.join(config.PATH_TO_PRETRAINED_MODELS, f'transformers/{model_name}')
    hf = AutoModel.from_pretrained(model_dir)
    kind = getattr(hf.config, 'model_type', None)
    if kind not in _BRANCHES:
        raise MerError(f"{model_name}: architecture '{kind}' has no HIP visual branch (supported: {sorted(_BRANCHES)})")
    cls, driver = _BRANCHES[kind]
    torch.cuda.set_device(max(gpu, 0))   # reference :99-101
    return getattr(encoders, cls).from_hf(hf, device=f'cuda:{max(gpu, 0)}', precision=precision), driver


def extract_by_name(model_name, face_dir, save_dir, feature_level='UTTERANCE', gpu=0, precision="mean", **kw):
    """The reference's `python extract_vision_huggingface.py --model_name X --feature_level L --gpu G` for one face directory:
    loads the checkpoint by name and runs its branch; `kw` goes to that branch's driver (vids, workers, device_preprocess ...)."""
    model, driver = load_model(model_name, gpu, precision)
    return globals()[driver](model, face_dir, save_dir, feature_level, **kw)
