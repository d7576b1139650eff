#!/usr/bin/env python3
"""PCIe-inclusive rate of the tri-modal extraction step: inputs start in pinned host memory every step.
Variants: fp32 frames + fp32 audio (what the reference moves), uint8 frames + int16 PCM with the GPU pre-processing kernels."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import ops, synthetic as W
from mertools_amd.encoders import HipBertModel, HipCLIPModel, HipHubertModel
from mertools_amd.extract.visual import CLIP_MEAN, CLIP_STD

dev = torch.device("cuda:0")
B = 64
hc, cc, bc = W.hubert_config("base"), W.clip_config("base16"), W.bert_config("roberta-base")
ma = HipHubertModel(W.hubert_state_dict(hc, 0), hc, device=dev)
mv = HipCLIPModel(W.clip_state_dict(cc, 0), cc, device=dev)
mt = HipBertModel(W.bert_state_dict(bc, 0), bc, device=dev)
wav32 = W.synth_audio(B).pin_memory()
pcm16 = (torch.randn(B, 80000) * 3000).clamp(-32768, 32767).to(torch.int16).pin_memory()
px32 = W.synth_frames(B * 8).pin_memory()
fr8 = torch.randint(0, 256, (B * 8, 224, 224, 3), dtype=torch.uint8).pin_memory()
ids = W.synth_tokens(B).pin_memory()

def step_fp32():
    a = ma.extract_utterance(wav32.to(dev, non_blocking=True))
    v = mv.extract_utterance(px32.to(dev, non_blocking=True), [8] * B)
    t = mt.extract_utterance(ids.to(dev, non_blocking=True), [64] * B, 1, -1)
    return a, v, t

def step_compact():
    a = ma.extract_utterance(ops.wave_normalize(pcm16.to(dev, non_blocking=True)))
    v = mv.extract_utterance(ops.image_normalize_u8(fr8.to(dev, non_blocking=True), CLIP_MEAN, CLIP_STD, bgr=True), [8] * B)
    t = mt.extract_utterance(ids.to(dev, non_blocking=True), [64] * B, 1, -1)
    return a, v, t

def resident():
    wa, pv, ii = wav32.to(dev), px32.to(dev), ids.to(dev)
    def f():
        return ma.extract_utterance(wa), mv.extract_utterance(pv, [8] * B), mt.extract_utterance(ii, [64] * B, 1, -1)
    return f

out = {}
for name, fn in (("resident (bench.py, single stream)", resident()), ("fp32 over PCIe", step_fp32), ("uint8 frames + int16 PCM over PCIe", step_compact)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    out[name] = round(B / dt, 1)
out["bytes_per_step_MB"] = {"fp32": round((wav32.numel() * 4 + px32.numel() * 4 + ids.numel() * 8) / 1e6, 1),
                            "compact": round((pcm16.numel() * 2 + fr8.numel() + ids.numel() * 8) / 1e6, 1)}
print(json.dumps(out))
