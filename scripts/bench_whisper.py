#!/usr/bin/env python3
"""Whisper branch throughput on one GPU (synthetic whisper-base / large-v2-shaped weights, log-mel already resident):
clips/s of `extract_utterance` and the encoder / decoder split.  Usage: bench_whisper.py [base|large] [batch] [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mertools_amd import synthetic as W
from mertools_amd.whisper import HipWhisperModel

size = sys.argv[1] if len(sys.argv) > 1 else "base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
c = W.whisper_config(size)
m = HipWhisperModel(W.whisper_state_dict(c, 0), c, precision=os.environ.get("PRECISION", "mx"))
mel = (torch.randn(B, c.num_mel_bins, 2 * c.max_source_positions, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
ids = torch.full((B, 2), c.decoder_start_token_id, dtype=torch.long).cuda()
for _ in range(2):
    m(mel, decoder_input_ids=ids)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
te = td = 0.0
t0 = time.perf_counter()
for _ in range(iters):
    ev[0].record(); enc = m.encode(mel); ev[1].record(); out = m.decode(enc, ids); ev[2].record()
    torch.cuda.synchronize()
    te += ev[0].elapsed_time(ev[1]); td += ev[1].elapsed_time(ev[2])
wall = time.perf_counter() - t0
L, D, F = c.encoder_layers, c.d_model, c.encoder_ffn_dim
T = c.max_source_positions
flops = B * (2 * 2 * T * 3 * c.num_mel_bins * D + 2 * T * 3 * D * D + L * T * (2 * (4 * D * D + 2 * D * F) + 4 * T * D) + c.decoder_layers * T * 4 * D * D)
print(f"whisper-{size} B={B}: {B * iters / wall:.1f} clips/s wall | encoder {te / iters:.2f} ms, decoder {td / iters:.2f} ms per batch"
      f" | {flops / ((te + td) / iters * 1e-3) / 1e12:.0f} TFLOP/s on the MFMA-path GEMMs + attention")
