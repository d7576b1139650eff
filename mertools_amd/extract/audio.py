"""Audio feature extraction — mirror of MERBench/feature_extraction/audio/extract_audio_huggingface.py:40-113.

Same `extract(model_name, audio_files, save_dir, feature_level, gpu)` signature and the same
`<save_dir>/<clip>.npy` outputs (UTT: [D] float32; FRAME: [B*T, D] float32).  What differs is how the work
reaches the GPU: clips are bucketed by (chunked) length and pushed through the HIP HuBERT/wav2vec2 encoder
in batches; the last-4-layer sum and the utterance mean run on the GPU, so only [D] (or [T,D]) floats come
back per clip instead of 13 x [T,D] hidden states.
"""
import math
import os
import struct
import time
import wave

import numpy as np
import torch

from .pipeline import npy_save, span

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


_SF = []   # [soundfile module or None], resolved once (a failing import costs a sys.path scan per call)


def read_pcm16(path):
    """(int16 samples, sample_rate) of a mono PCM16 WAV without the float64 round trip, or None when the file is anything else (the
    caller then takes the float reader).  The RIFF chunks are walked here — one read of the file, no per-chunk Python objects: the
    stdlib `wave` reader costs more interpreter time per clip than the GPU needs for it — and the result is what `wave` returns
    (tests/test_round3_cpu.py: extra chunks, odd chunk sizes, truncated data)."""
    try:
        with open(path, 'rb') as f:
            buf = bytearray(os.fstat(f.fileno()).st_size)
            n = f.readinto(buf)
    except OSError:
        return None
    if n < 12 or buf[0:4] != b'RIFF' or buf[8:12] != b'WAVE':
        return None
    pos, rate = 12, None
    while pos + 8 <= n:
        size = int.from_bytes(buf[pos + 4:pos + 8], 'little')
        body = pos + 8
        tag = bytes(buf[pos:pos + 4])
        if tag == b'fmt ':
            if size < 16 or body + 16 > n:
                return None
            fmt, channels, rate, _, _, bits = struct.unpack_from('<HHIIHH', buf, body)
            if fmt != 1 or channels != 1 or bits != 16:   # (multi-channel files keep the reference's channel handling)
                return None
        elif tag == b'data':
            if rate is None:
                return None
            count = min(size, n - body) // 2
            return np.frombuffer(buf, dtype='<i2', count=count, offset=body), rate   # (a bytearray: writable, torch.from_numpy takes it)
        pos = body + size + (size & 1)
    return None


def read_audio(path):
    """(samples float64 in [-1,1), sample_rate) — soundfile when importable, else stdlib PCM16 WAV."""
    if not _SF:
        try:
            import soundfile as sf
            _SF.append(sf)
        except ImportError:
            _SF.append(None)
    if _SF[0] is not None:
        return _SF[0].read(path)
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
    npy_save(csv_file, feature)   # np.save's bytes (extract.pipeline)


def load_model(model_name, gpu, precision="mean"):
    """AutoModel checkpoint under config.PATH_TO_PRETRAINED_MODELS/transformers/<model_name> -> HIP encoder."""
    from transformers import AutoModel, Wav2Vec2FeatureExtractor
    from .. import config
    from ..encoders import HipHubertModel
    model_file = os.path.join(config.PATH_TO_PRETRAINED_MODELS, f'transformers/{model_name}')
    hf = AutoModel.from_pretrained(model_file)
    fe = Wav2Vec2FeatureExtractor.from_pretrained(model_file)
    torch.cuda.set_device(max(gpu, 0))
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


def to_pcm16_or_f32(samples):
    """int16 PCM when `samples` (float in [-1, 1)) is exactly what a PCM16 file holds — half the H2D bytes of fp32, a quarter of the
    float64 the reference moves — else float32."""
    x = np.asarray(samples)
    if x.dtype == np.int16:
        return x
    x = x.astype(np.float64, copy=False)
    pcm = np.round(x * 32768.0)
    if np.array_equal(pcm / 32768.0, x) and pcm.min() >= -32768 and pcm.max() <= 32767:
        return pcm.astype(np.int16)
    return x.astype(np.float32)


def device_normalize(samples, do_normalize, device):
    """wav2vec2_normalize on the GPU: the utterance goes up as 16-bit PCM when it is exactly representable (what a PCM16 file
    holds: half the H2D bytes of fp32, a quarter of the float64 the reference moves), else as fp32; mer_wave_normalize.
    The result STAYS on the device (the batch is assembled there)."""
    from .. import ops
    x = np.asarray(samples, dtype=np.float64)
    pcm = np.round(x * 32768.0)
    with torch.cuda.device(device):
        if np.array_equal(pcm / 32768.0, x) and pcm.min() >= -32768 and pcm.max() <= 32767:
            t = torch.from_numpy(pcm.astype(np.int16))[None].to(device)
        else:
            t = torch.from_numpy(x.astype(np.float32))[None].to(device)
        return ops.wave_normalize(t, do_normalize)


def constant_rows(arr):
    """Row indices of a host [rows, L] (or [L]) array / tensor that hold ONE value throughout — digital silence, a DC clip, an all-zero
    chunk of a long clip.  The degenerate input of the HuBERT family (every frame identical): the encoder routes such rows through its
    accurate twin (HipHubertModel.forward_raw, constant_rows).  O(1) per ordinary row: three samples are compared first."""
    a = arr.numpy() if torch.is_tensor(arr) else np.asarray(arr)
    if a.ndim == 1:
        a = a[None]
    out = []
    for r in range(a.shape[0]):
        row = a[r]
        if row[0] == row[len(row) // 2] == row[-1] and row.min() == row.max():
            out.append(r)
    return out


def constant_chunks(raw, do_normalize=True, maxlen=MAXLEN):
    """Chunk rows (split_into_batch) of a clip longer than `maxlen` that will hold ONE value, found on the host's RAW samples: the
    whole-clip normalisation is affine, so a constant stretch stays constant.  A full chunk is constant iff its raw samples are; the
    zero-padded last chunk only when its samples normalise to exactly 0 — the whole clip is one value (then every chunk is zero), or
    un-normalised digital silence.  (the device path used to skip the detection for chunked clips, leaving a silent
    chunk on the one-plane preset.)"""
    a = raw.numpy() if torch.is_tensor(raw) else np.asarray(raw)
    n = len(a)
    if n <= maxlen:
        return constant_rows(a)
    nfull, tail = divmod(n, maxlen)
    out = [r for r in constant_rows(a[:nfull * maxlen].reshape(nfull, maxlen))]
    if tail:
        t = a[nfull * maxlen:]
        if t[0] == t[len(t) // 2] == t[-1] and t.min() == t.max():
            whole = a[0] == a[n // 2] == a[-1] and a.min() == a.max()
            if (do_normalize and whole) or (not do_normalize and t[0] == 0):
                out.append(nfull)
    return out


def plan_batches(pending, batch_rows, ragged, final, max_stretch=1.5, keep_at_most=0):
    """Cuts the pending clips (dicts with 'rows', 'len') into batches; returns (batches, still_pending).
    ragged: clips are sorted by length and consecutive ones share a batch while the longest is at most `max_stretch` x the
    shortest (bounds the padded work) and the rows fit `batch_rows`; otherwise only clips of identical shape batch together.
    Unless `final`, groups that are not full stay pending so that later clips of similar length can join them — but never
    more than `keep_at_most` clips: beyond that the fullest partial groups are emitted too (bounded host memory)."""
    key = (lambda it: (it['len'],)) if ragged else (lambda it: (it['len'], it['rows']))
    groups, cur, rows = [], [], 0
    for it in sorted(pending, key=key):
        fits = cur and rows + it['rows'] <= batch_rows and (
            it['len'] <= max_stretch * cur[0]['len'] if ragged else key(it) == key(cur[0]))
        if cur and not fits:
            groups.append(cur)
            cur, rows = [], 0
        cur.append(it)
        rows += it['rows']
    if cur:
        groups.append(cur)
    if final:
        return groups, []
    nrows = lambda g: sum(it['rows'] for it in g)   # noqa: E731
    out = [g for g in groups if nrows(g) >= batch_rows]
    part = sorted((g for g in groups if nrows(g) < batch_rows), key=nrows)
    while part and sum(len(g) for g in part) > keep_at_most:
        out.append(part.pop())
    return out, [it for g in part for it in g]


def extract(model_name, audio_files, save_dir, feature_level, gpu, model=None, do_normalize=True, batch_rows=32,
            reader=read_audio, device_preprocess=False, workers=0, rank=None, world=None, window=256, ragged=True, async_save=True,
            ramp_up=True):
    """device_preprocess: run the feature extractor's normalisation on the GPU instead of numpy.
    workers: threads that read and normalise the clips ahead of the batching loop (extract.prefetch; 0 = in line).
    rank / world: this process's share of `audio_files` (distributed.shard: sorted(files)[rank::world]; default = the
    torch.distributed rank / world size, 0 / 1 when not initialised) — clips are independent, no collective.
    window: clips held on the host at most before batches are cut (host memory is O(window), not O(corpus)).
    ragged: batch clips of DIFFERENT lengths together (rows zero-padded to the longest, mer_hubert_forward_ragged makes each
    clip equal to its batch-of-one forward); False = only clips of identical length share a batch.
    async_save: features leave the GPU through pinned buffers without blocking this thread and are written by worker threads
    (extract.pipeline.AsyncWriter; the same bytes reach the same files); False = the reference's blocking copy + in-line np.save.
    ramp_up: the first two batches are cut at a quarter / half of `batch_rows` (and of `window`), so that the GPU starts while the
    read-ahead is still filling instead of after a full batch of files has been read — a fixed cost per run, visible on small corpora."""
    from .prefetch import prefetch_map
    from .. import distributed
    if rank is None:
        rank, world = distributed.rank_world()
    audio_files = distributed.my_share(audio_files, rank, world)
    if model_name in (WHISPER_BASE, WHISPER_LARGE) or type(model).__name__ == 'HipWhisperModel':
        return extract_whisper(model_name, audio_files, save_dir, feature_level, gpu, model=model, reader=reader)   # reference :79-89
    start_time = time.time()
    if model is None:
        model, do_normalize = load_model(model_name, gpu)
    os.makedirs(save_dir, exist_ok=True)

    def host_stage(audio_file):
        if device_preprocess and reader is read_audio:   # PCM16 files go up as the 16-bit samples they hold: no float64 round trip
            got = read_pcm16(audio_file)
            if got is not None:
                assert got[1] == 16000, 'currently, we only test on 16k audio'
                return audio_file, got[0], constant_chunks(got[0], do_normalize)
        samples, sr = reader(audio_file)
        assert sr == 16000, 'currently, we only test on 16k audio'
        if device_preprocess:   # (a worker thread: numpy releases the GIL) 16-bit PCM when the file holds exactly that, else fp32
            raw = to_pcm16_or_f32(samples)
            return audio_file, raw, constant_chunks(raw, do_normalize)
        iv = split_into_batch(wav2vec2_normalize(samples, do_normalize))
        return audio_file, iv, constant_rows(iv)

    up = None
    if device_preprocess:
        from .pipeline import Uploader
        up = Uploader(model.device)

    def flush_raw(items):
        """device_preprocess fast path: one-row clips of ONE length and sample type travel as a single pinned [B, L] block on the
        upload stream and are normalised by one kernel — no per-clip H2D copy, launch or device allocation on this thread."""
        from .. import ops
        with span("stage"):
            first = items[0]['raw']
            block = torch.empty((len(items), len(first)), dtype=torch.from_numpy(first[:1]).dtype, pin_memory=True)   # (cached by torch's host allocator)
            dst = block.numpy()
            for i, it in enumerate(items):   # straight into the pinned staging block: one host copy per clip
                dst[i] = it['raw']
            dev_block = up.up(block)
            # the kernel launches on the CURRENT device's stream (ops.stream()) while Uploader.ready() orders the stream of
            # model.device: make them the same device (a model on cuda:1 driven while cuda:0 is current)
            with torch.cuda.device(model.device):
                up.ready(dev_block)
                rows = ops.wave_normalize(dev_block, do_normalize)
        flush(items, rows)

    def flush(items, rows=None):
        L = max(it['len'] for it in items)
        same = all(it['len'] == L for it in items)
        ivs = [it.get('iv') for it in items]
        if rows is not None:   # already one normalised [B, L] block on the device (flush_raw)
            pass
        elif same:
            rows = torch.cat(ivs, 0)
        else:   # zero-padded rows; one-row clips only differ in length (chunked clips are exactly MAXLEN wide)
            rows = torch.zeros((sum(it['rows'] for it in items), L), dtype=torch.float32, device=ivs[0].device)
            r = 0
            for it in items:
                rows[r:r + it['rows'], :it['len']] = it['iv']
                r += it['rows']
        chunks = [it['rows'] for it in items]
        valid = None if same else [it['len'] for it in items for _ in range(it['rows'])]
        const, r0 = [], 0           # batch rows that hold one value throughout (found on the host, before the upload)
        for it in items:
            const += [r0 + r for r in it.get('const', ())]
            r0 += it['rows']
        ckw = {'constant_rows': const} if const else {}   # (only then: any object with the encoder's older call signature still works)
        T = model.out_frames(L)
        vids = [it['vid'] for it in items]
        if feature_level == 'UTTERANCE':
            with span("forward"):
                pooled = model.extract_utterance(rows, clip_chunks=chunks, valid_samples=valid, **ckw)

            def save_utt(arr, vids=vids):
                for vid, feat in zip(vids, arr):
                    save_feature(os.path.join(save_dir, f"{vid}.npy"), feat, feature_level)
            with span("submit"):
                out.submit(pooled, save_utt)
        else:
            starts, lens = model.clip_segments(L, chunks, valid)
            with span("forward"):
                _, frames, _ = model.forward_raw(rows, frames=True, valid_samples=valid, **ckw)

            def save_frames(arr, vids=vids, starts=starts, lens=lens):
                for vid, s0, n in zip(vids, starts, lens):
                    save_feature(os.path.join(save_dir, f"{vid}.npy"), arr[s0:s0 + n], feature_level)
            with span("submit"):
                out.submit(frames, save_frames)

    from .pipeline import writer
    pending = []
    emitted = 0

    def ramp(full):
        return max(1, full >> max(0, 2 - emitted)) if ramp_up else full

    with writer(model.device, async_save) as out:
        def emit(b):
            nonlocal emitted
            emitted += 1
            if any('raw' in it for it in b):
                same = len({(it['len'], it['raw'].dtype) for it in b if 'raw' in it}) == 1 and all('raw' in it for it in b)
                if same:
                    return flush_raw(b)
                from .. import ops
                for it in b:   # mixed lengths / sample types: normalise clip by clip (GPU work stays on the calling thread)
                    if 'raw' in it:
                        with torch.cuda.device(model.device):
                            it['iv'] = ops.wave_normalize(torch.from_numpy(it['raw'])[None].to(model.device), do_normalize)
            flush(b)

        for audio_file, iv, const in prefetch_map(host_stage, audio_files, workers, chunk=4):
            vid = os.path.basename(audio_file)[:-4]
            if device_preprocess:
                if len(iv) > MAXLEN:   # > 10 s: chunked after the normalisation (reference :40-50)
                    with torch.cuda.device(model.device):
                        from .. import ops
                        ivd = split_into_batch_any(ops.wave_normalize(torch.from_numpy(iv)[None].to(model.device), do_normalize))
                    pending.append(dict(vid=vid, iv=ivd, rows=ivd.shape[0], len=ivd.shape[1], const=const))
                else:
                    pending.append(dict(vid=vid, raw=iv, rows=1, len=len(iv), const=const))
            else:
                pending.append(dict(vid=vid, iv=iv, rows=iv.shape[0], len=iv.shape[1], const=const))
            # a full batch of clips of ONE length needs no sorting window: cut it as soon as it exists (a corpus of equal-length
            # clips would otherwise sit on the host until `window` files have been read, with the GPU idle)
            same = [it for it in pending if it['len'] == pending[-1]['len'] and it['rows'] == pending[-1]['rows'] and ('raw' in it) == ('raw' in pending[-1])]
            if sum(it['rows'] for it in same) >= ramp(batch_rows):
                ids = {id(it) for it in same}
                pending = [it for it in pending if id(it) not in ids]
                emit(same)
            elif len(pending) >= ramp(window):
                batches, pending = plan_batches(pending, batch_rows, ragged, final=False, keep_at_most=ramp(window) // 2)
                for b in batches:
                    emit(b)
        batches, pending = plan_batches(pending, batch_rows, ragged, final=True)
        for b in batches:
            emit(b)
    print(f'Total time used: {time.time() - start_time:.1f}s.')


def split_into_batch_any(input_values, maxlen=None):
    """split_into_batch for a tensor on any device (the reference's version allocates on the host)."""
    maxlen = maxlen if maxlen is not None else split_into_batch.__defaults__[0]
    if input_values.shape[1] <= maxlen:
        return input_values
    wavlen = input_values.shape[1]
    tgtlen = math.ceil(wavlen / maxlen) * maxlen
    out = torch.zeros((1, tgtlen), dtype=input_values.dtype, device=input_values.device)
    out[:, :wavlen] = input_values
    return out.view(-1, maxlen)
