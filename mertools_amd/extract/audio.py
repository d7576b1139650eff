"""Audio feature extraction — mirror of MERBench/feature_extraction/audio/extract_audio_huggingface.py:40-113.

Same `extract(model_name, audio_files, save_dir, feature_level, gpu)` signature and the same
`<save_dir>/<clip>.npy` outputs (UTT: [D] float32; FRAME: [B*T, D] float32).  What differs is how the work
reaches the GPU: clips are bucketed by (chunked) length and pushed through the HIP HuBERT/wav2vec2 encoder
in batches; the last-4-layer sum and the utterance mean run on the GPU, so only [D] (or [T,D]) floats come
back per clip instead of 13 x [T,D] hidden states.
"""
import math
import os
import time
import wave

import numpy as np
import torch

MAXLEN = 16000 * 10
WHISPER_BASE = 'whisper-base'        # reference :35-36
WHISPER_LARGE = 'whisper-large-v2'


def split_into_batch(input_values, maxlen=MAXLEN):
    """[1, L] -> [ceil(L/maxlen), maxlen] zero-padded when L > maxlen, else unchanged (reference :40-50)."""
    if len(input_values[0]) <= maxlen:
        return input_values
    bs, wavlen = input_values.shape
    assert bs == 1
    tgtlen = math.ceil(wavlen / maxlen) * maxlen
    batches = torch.zeros((1, tgtlen))
    batches[:, :wavlen] = input_values
    return batches.view(-1, maxlen)


def wav2vec2_normalize(samples, do_normalize=True):
    """Wav2Vec2FeatureExtractor.__call__ on one utterance (reference :94; HF:wav2vec2/feature_extraction_wav2vec2.py:78-97,
    :207-215): zero-mean / unit-variance on the raw array, result as float32 [1, L]."""
    x = np.asarray(samples, dtype=np.float32)  # HF casts float64 -> float32 first, then normalises in float32
    if do_normalize:
        x = (x - x.mean()) / np.sqrt(x.var() + 1e-7)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))[None]


def read_audio(path):
    """(samples float64 in [-1,1), sample_rate) — soundfile when importable, else stdlib PCM16 WAV."""
    try:
        import soundfile as sf
        return sf.read(path)
    except ImportError:
        with wave.open(path, 'rb') as w:
            assert w.getsampwidth() == 2, 'only PCM16 wav supported without soundfile'
            raw = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float64) / 32768.0
            if w.getnchannels() > 1:
                raw = raw.reshape(-1, w.getnchannels())
            return raw, w.getframerate()


def save_feature(csv_file, feature, feature_level):
    """np.save rules of reference :103-110."""
    if feature_level == 'UTTERANCE':
        feature = np.array(feature).squeeze()
        if len(feature.shape) != 1:
            feature = np.mean(feature, axis=0)
    np.save(csv_file, feature)


def load_model(model_name, gpu, precision="mx"):
    """AutoModel checkpoint under config.PATH_TO_PRETRAINED_MODELS/transformers/<model_name> -> HIP encoder."""
    from transformers import AutoModel, Wav2Vec2FeatureExtractor
    from .. import config
    from ..encoders import HipHubertModel
    model_file = os.path.join(config.PATH_TO_PRETRAINED_MODELS, f'transformers/{model_name}')
    hf = AutoModel.from_pretrained(model_file)
    fe = Wav2Vec2FeatureExtractor.from_pretrained(model_file)
    return HipHubertModel.from_hf(hf, device=f'cuda:{max(gpu, 0)}', precision=precision), fe.do_normalize


def _slaney_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)


def _slaney_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)


def whisper_mel_filters(n_freq=201, n_mels=80, sr=16000):
    """[n_freq, n_mels] Slaney-scale, Slaney-normalised triangular bank over 0 .. sr/2 (what WhisperFeatureExtractor builds
    with mel_filter_bank(..., norm="slaney", mel_scale="slaney"); HF:whisper/feature_extraction_whisper.py:90-98)."""
    edges = _slaney_hz(np.linspace(_slaney_mel(0.0), _slaney_mel(sr / 2.0), n_mels + 2))
    fft_f = np.linspace(0.0, sr / 2.0, n_freq)
    width = np.diff(edges)
    slopes = edges[None, :] - fft_f[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / width[:-1], slopes[:, 2:] / width[1:]))
    return fb * (2.0 / (edges[2:] - edges[:-2]))[None, :]


def whisper_log_mel(samples, n_mels=80, n_samples=480000):
    """WhisperFeatureExtractor.__call__ on one utterance (reference :81; HF:whisper/feature_extraction_whisper.py:100-130,
    :260-330): zero-pad / cut to 30 s, |STFT|^2 (400-point periodic Hann, hop 160, reflect-centred), mel, log10 floored at
    1e-10, last frame dropped, clamp to max - 8, (x + 4) / 4 -> float32 [1, n_mels, 3000]."""
    x = np.zeros(n_samples, dtype=np.float32)
    w = np.asarray(samples, dtype=np.float32)[:n_samples]
    x[:len(w)] = w
    win = torch.hann_window(400, periodic=True, dtype=torch.float64)
    spec = torch.stft(torch.from_numpy(x).double(), 400, hop_length=160, window=win, center=True, pad_mode="reflect", return_complex=True)
    power = spec.abs().pow(2).numpy()                                             # [201, 3001]
    mel = whisper_mel_filters(201, n_mels).T @ power
    log_spec = np.log10(np.maximum(mel, 1e-10))[:, :-1]
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return torch.from_numpy(((log_spec + 4.0) / 4.0).astype(np.float32))[None]


def load_whisper(model_name, gpu, precision="mx"):
    from transformers import AutoModel
    from .. import config
    from ..whisper import HipWhisperModel
    model_file = os.path.join(config.PATH_TO_PRETRAINED_MODELS, f'transformers/{model_name}')
    return HipWhisperModel.from_hf(AutoModel.from_pretrained(model_file), device=f'cuda:{max(gpu, 0)}', precision=precision)


def extract_whisper(model_name, audio_files, save_dir, feature_level, gpu, model=None, batch_clips=16, reader=read_audio):
    """WHISPER_BASE / WHISPER_LARGE branch (reference :79-89): every clip is one fixed 30 s log-mel window, so clips batch
    without bucketing; the two decoder states come back as (2, D) per clip (FRAME) or their mean (UTTERANCE, reference :103-108)."""
    start_time = time.time()
    if model is None:
        model = load_whisper(model_name, gpu)
    os.makedirs(save_dir, exist_ok=True)
    for i in range(0, len(audio_files), batch_clips):
        group = audio_files[i:i + batch_clips]
        mels = []
        for audio_file in group:
            samples, sr = reader(audio_file)
            assert sr == 16000, 'currently, we only test on 16k audio'
            mels.append(whisper_log_mel(samples, model.config.num_mel_bins, 320 * model.config.max_source_positions))
        feats = model.extract_utterance(torch.cat(mels, 0)).cpu().numpy()       # [B, 2, D]
        for audio_file, feat in zip(group, feats):
            save_feature(os.path.join(save_dir, f'{os.path.basename(audio_file)[:-4]}.npy'), feat, feature_level)
    print(f'Total time used: {time.time() - start_time:.1f}s.')


def device_normalize(samples, do_normalize, device):
    """wav2vec2_normalize on the GPU: the utterance goes up as 16-bit PCM when it is exactly representable (what a PCM16 file
    holds: half the H2D bytes of fp32, a quarter of the float64 the reference moves), else as fp32; mer_wave_normalize."""
    from .. import ops
    x = np.asarray(samples, dtype=np.float64)
    pcm = np.round(x * 32768.0)
    if np.array_equal(pcm / 32768.0, x) and pcm.min() >= -32768 and pcm.max() <= 32767:
        t = torch.from_numpy(pcm.astype(np.int16))[None].to(device)
    else:
        t = torch.from_numpy(x.astype(np.float32))[None].to(device)
    return ops.wave_normalize(t, do_normalize).cpu()


def extract(model_name, audio_files, save_dir, feature_level, gpu, model=None, do_normalize=True, batch_rows=32,
            reader=read_audio, device_preprocess=False, workers=0):
    """device_preprocess: run the feature extractor's normalisation on the GPU instead of numpy.
    workers: threads that read and normalise the clips ahead of the bucketing loop (extract.prefetch; 0 = in line)."""
    from .prefetch import prefetch_map
    if model_name in (WHISPER_BASE, WHISPER_LARGE) or type(model).__name__ == 'HipWhisperModel':
        return extract_whisper(model_name, audio_files, save_dir, feature_level, gpu, model=model, reader=reader)   # reference :79-89
    start_time = time.time()
    if model is None:
        model, do_normalize = load_model(model_name, gpu)
    os.makedirs(save_dir, exist_ok=True)
    # bucket clips by the shape they have after split_into_batch: equal-length rows batch without any masking
    # (the reference never masks audio, and GroupNorm runs over the whole row, so padding would change results)
    buckets = {}

    def host_stage(audio_file):
        samples, sr = reader(audio_file)
        assert sr == 16000, 'currently, we only test on 16k audio'
        if device_preprocess:
            return audio_file, samples
        return audio_file, split_into_batch(wav2vec2_normalize(samples, do_normalize))

    for audio_file, iv in prefetch_map(host_stage, audio_files, workers):
        if device_preprocess:   # GPU work stays on the calling thread
            iv = split_into_batch(device_normalize(iv, do_normalize, model.device))
        buckets.setdefault(tuple(iv.shape), []).append((os.path.basename(audio_file)[:-4], iv))

    def flush(items):
        rows = torch.cat([iv for _, iv in items], 0)
        chunks = [iv.shape[0] for _, iv in items]
        T = model.out_frames(rows.shape[1])
        if feature_level == 'UTTERANCE':
            pooled = model.extract_utterance(rows, clip_chunks=chunks).cpu().numpy()
            for (vid, _), feat in zip(items, pooled):
                save_feature(os.path.join(save_dir, f'{vid}.npy'), feat, feature_level)
        else:
            _, frames, _ = model.forward_raw(rows, frames=True)
            frames = frames.cpu().numpy()
            r = 0
            for (vid, _), n in zip(items, chunks):
                save_feature(os.path.join(save_dir, f'{vid}.npy'), frames[r * T:(r + n) * T], feature_level)
                r += n

    for shape, items in buckets.items():
        cur, rows = [], 0
        for it in items:
            if cur and rows + it[1].shape[0] > batch_rows:
                flush(cur)
                cur, rows = [], 0
            cur.append(it)
            rows += it[1].shape[0]
        if cur:
            flush(cur)
    print(f'Total time used: {time.time() - start_time:.1f}s.')
