"""HIP drop-ins for AffectGPT's frozen front-ends (MER2025/MER2025_Track23/my_affectgpt/models/encoder.py):

  HIP_CLIP_VIT_LARGE  <- CLIP_VIT_LARGE (:177-209): raw frames [b,c,t,h,w] -> CLIPImageProcessor -> get_image_features
                         -> [b, t, projection_dim]
  HIP_HUBERT_LARGE    <- HUBERT_LARGE (:400-432): raw 2-s chunks [b,t,1,32000] -> feature extractor -> hidden states
                         -> MEAN of the last four layers -> mean over time -> [b, t, hidden]

Same `forward(x, raw_x)` signatures, same `.hidden_size`, registered under HIP_* names so that an AffectGPT config selects
them by name.  The frozen encoders hold no trainable parameters here (the reference freezes them, :191-194,412-415), so
they are plain objects around HipCLIPModel / HipHubertModel; outputs are fp32 CUDA tensors.

Behaviour preserved from the reference:
  * the audio feature extractor is called on the 2-D tensor [(b t), s] and indexed with [0] (:425-427): HF treats a
    tensor as ONE utterance, so `do_normalize` normalises with the mean/variance of the WHOLE (b t s) block, not per row;
  * frames go through PIL + CLIPImageProcessor (resize shortest edge bicubic, centre crop, /255, CLIP mean/std);
  * last-four-layer MEAN (the MERBench extractor uses the SUM).
"""
import numpy as np
import torch

from .registry import registry
from ..encoders import HipCLIPModel, HipHubertModel
from ..extract.visual import clip_preprocess


def joint_zero_mean_unit_var(raw_audio_2d):
    """Wav2Vec2FeatureExtractor(raw_audio [(b t), s] tensor).input_values[0] with do_normalize=True
    (HF:wav2vec2/feature_extraction_wav2vec2.py:78-97 on a single 2-D 'utterance'): float32, (x - mean) / sqrt(var + 1e-7)
    with the statistics of the whole block."""
    x = np.asarray(raw_audio_2d.detach().cpu() if torch.is_tensor(raw_audio_2d) else raw_audio_2d, dtype=np.float32)
    return torch.from_numpy(((x - x.mean()) / np.sqrt(x.var() + 1e-7)).astype(np.float32))


def _load(model, model_dir, hip_cls, device, precision):
    if model is None:
        if model_dir is None:
            raise ValueError("give either a HuggingFace model / HIP model (`model=`) or a checkpoint directory (`model_dir=`)")
        from transformers import AutoModel
        model = AutoModel.from_pretrained(model_dir)
    if isinstance(model, hip_cls):
        return model
    return hip_cls.from_hf(model, device=device, precision=precision)


@registry.register_visual_encoder("HIP_CLIP_VIT_LARGE")
class HIP_CLIP_VIT_LARGE:
    def __init__(self, model=None, model_dir=None, device="cuda:0", precision="mean", image_size=224):
        self.model = _load(model, model_dir, HipCLIPModel, device, precision)
        self.device = torch.device(device)
        self.image_size = image_size
        self.hidden_size = self.model.config.projection_dim     # 768 for ViT-L/14 (:196)

    def eval(self):
        return self

    # image encoding: [b c t h w] => [b t h]   (:200-208)
    def forward(self, image, raw_image):
        b, _, t, _, _ = raw_image.shape
        frames = raw_image.permute(0, 2, 3, 4, 1).reshape(b * t, raw_image.shape[3], raw_image.shape[4], raw_image.shape[1])
        frames = frames.detach().cpu().numpy().astype(np.uint8)               # func_VideoReader_to_Image (:34-39): RGB uint8
        pixel_values = clip_preprocess(frames[:, :, :, ::-1], self.image_size)  # clip_preprocess takes BGR (the MERBench reader's order)
        emb = self.model.get_image_features(pixel_values.to(self.device))      # [(b t), h]
        return emb.view(b, t, -1)

    __call__ = forward


@registry.register_acoustic_encoder("HIP_HUBERT_LARGE")
class HIP_HUBERT_LARGE:
    def __init__(self, model=None, model_dir=None, device="cuda:0", precision="mean", do_normalize=True):
        self.model = _load(model, model_dir, HipHubertModel, device, precision)
        self.device = torch.device(device)
        self.do_normalize = do_normalize     # Wav2Vec2FeatureExtractor.do_normalize of the checkpoint
        self.hidden_size = self.model.config.hidden_size

    def eval(self):
        return self

    # audio: [b, t, 1, 128, 204] mel (unused), raw_audio: [b, t, 1, 32000] samples   (:419-432)
    def forward(self, audio, raw_audio):
        raw = raw_audio[:, :, 0, :]
        b, t, s = raw.shape
        raw = raw.reshape(b * t, s)
        x = joint_zero_mean_unit_var(raw) if self.do_normalize else raw.detach().to("cpu", torch.float32)
        pooled = self.model.extract_utterance(x.to(self.device))     # sum of the last four layers, mean over time
        return (pooled * 0.25).view(b, t, -1)                        # mean over layers [-4,-3,-2,-1] (:428)

    __call__ = forward
