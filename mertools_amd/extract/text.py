"""Text feature extraction — mirror of MERBench/feature_extraction/text/extract_text_huggingface.py:90-252 for the
encoder-only (BERT / RoBERTa family) models.  Special-token probing and .npy rules are the reference's;
sentences are right-padded into batches with per-row key-length masking on the HIP encoder."""
import os
import time

import numpy as np
import torch

try:     # at import time, like the reference script's own `import pandas` — not inside the first call of a cold process (0.1 s)
    import pandas as _pd
except ImportError:
    _pd = None

from .pipeline import npy_save, span

PROBE = '今天天气真好'


def find_start_end_pos(tokenizer):
    """(start, end) slice that strips the tokenizer's leading (0..2) / trailing (0..2) special tokens, found by decoding a probe
    sentence (behaviour of reference :90-114, bit-exact against it in tests/test_host_logic.py; integer path).  start = the
    first offset whose decoded tail begins with the probe (2 when none does); end = None when that tail IS the probe, else
    the first of -1, -2 that makes it so."""
    ids = tokenizer(PROBE, return_tensors='pt')['input_ids'][0]

    def text(a, b):
        return tokenizer.decode(ids[a:b]).replace(' ', '')

    start = next((s for s in range(3) if text(s, None).startswith(PROBE)), 2)
    end = None
    if text(start, None) != PROBE:
        end = next((e for e in (-1, -2) if text(start, e) == PROBE), None)
        assert end is not None, f'cannot isolate the probe sentence from {tokenizer.decode(ids)!r}'
    print(f'start: {start};  end: {end}')
    return start, end


def find_batchpos_embdim(tokenizer, model, gpu=-1):
    """(batch axis, feature dim) of `model`'s hidden states, probed with one sentence (reference :118-135: some of its LLMs are
    sequence-first).  The HIP encoders are batch-first."""
    probe = tokenizer(PROBE, return_tensors='pt')
    last = model(**probe, output_hidden_states=True).hidden_states[-1]
    shape = tuple(last.shape)
    assert 1 in shape[:2], shape
    batch_pos = 0 if shape[0] == 1 else 1
    print(f'batch_pos:{batch_pos}, feature_dim:{shape[2]}')
    return batch_pos, shape[2]


def save_embeddings(csv_file, embeddings, feature_level, feature_dim):
    """np.save rules of reference :235-249 (empty sentence -> float64 zeros)."""
    embeddings = np.array(embeddings).squeeze()
    if feature_level == 'FRAME':
        if len(embeddings) == 0:
            embeddings = np.zeros((1, feature_dim))
        elif len(embeddings.shape) == 1:
            embeddings = embeddings[np.newaxis, :]
    else:
        if len(embeddings) == 0:
            embeddings = np.zeros((feature_dim,))
        elif len(embeddings.shape) == 2:
            embeddings = np.mean(embeddings, axis=0)
    npy_save(csv_file, embeddings)   # np.save's bytes (extract.pipeline)


def extract_embedding(model_name, trans_dir, save_dir, feature_level, gpu=-1, punc_case=None, language='chinese',
                      model_dir=None, model=None, tokenizer=None, batch_size=64, rank=None, world=None, async_save=True):
    pd = _pd
    if pd is None:
        import pandas as pd
    print('=' * 30 + f' Extracting "{model_name}" ' + '=' * 30)
    start_time = time.time()
    if punc_case is None and language == 'chinese' and model_dir is None:
        save_dir = os.path.join(save_dir, f'{model_name}-{feature_level[:3]}')
    elif punc_case is not None:
        save_dir = os.path.join(save_dir, f'{model_name}-punc{punc_case}-{feature_level[:3]}')
    elif language == 'english':
        save_dir = os.path.join(save_dir, f'{model_name}-langeng-{feature_level[:3]}')
    elif model_dir is not None:
        save_dir = os.path.join(save_dir, f'{"-".join(model_dir.split("/")[-2:])}-{model_name}-{feature_level[:3]}')
    os.makedirs(save_dir, exist_ok=True)
    if model is None:
        from transformers import AutoModel, AutoTokenizer
        from .. import config
        from ..encoders import HipBertModel
        if model_dir is None:
            model_dir = os.path.join(config.PATH_TO_PRETRAINED_MODELS, f'transformers/{model_name}')
        torch.cuda.set_device(max(gpu, 0))   # reference: torch.cuda.set_device(gpu) (extract_text_huggingface.py:192-193)
        model = HipBertModel.from_hf(AutoModel.from_pretrained(model_dir), device=f'cuda:{max(gpu, 0)}')
        tokenizer = AutoTokenizer.from_pretrained(model_dir, use_fast=False)
    with span("probe"):
        start, end = find_start_end_pos(tokenizer)
        batch_pos, feature_dim = find_batchpos_embdim(tokenizer, model, gpu)
    pad_id = model.config.pad_token_id if model.config.pad_token_id is not None else 0
    with span("read_wait"):
        df = pd.read_csv(trans_dir)
    from ..distributed import rank_world
    if rank is None or world is None:
        rank, world = rank_world()
    if world > 1:   # this process's share of the sentences: sorted names [rank::world] (sentences are independent, no collective)
        mine = set(sorted(str(n) for n in df['name'])[rank::world])
        df = df[[str(n) in mine for n in df['name']]]
    # The reference's per-row loop (:216-233: `for idx, row in df.iterrows()`, one tokenizer call per sentence) as column operations:
    # the same test per sentence, the same ids — iterrows builds a Series per row and costs as much as the tokenizer.
    names = df['name'].tolist()
    sentences = (df['chinese'] if language == 'chinese' else df['english']).tolist()
    keep = [pd.isna(s) == False and len(s) > 0 for s in sentences]  # noqa: E712 (reference's test)
    for name, k in zip(names, keep):
        if not k:
            save_embeddings(os.path.join(save_dir, f"{name}.npy"), [], feature_level, feature_dim)
    kept = [s for s, k in zip(sentences, keep) if k]
    kept_names = [n for n, k in zip(names, keep) if k]
    # Tokenisation runs AHEAD of the GPU loop on one worker thread, chunk by chunk (the Rust backend releases the GIL and spreads a
    # chunk over the cores): the first batch is queued after the first chunk instead of after the whole corpus.  Every sentence is
    # tokenised on its own (no padding / truncation is requested), so the ids are those of the reference's per-row loop (:216-225);
    # sentences are sorted by length WITHIN a chunk — a sentence's features do not depend on its batch mates or on the padding
    # (tests/test_parity_hardening_gpu.py), so neither does any file.
    encode = batch_encoder(tokenizer, kept[:256])
    step = max(4 * batch_size, 256)
    chunks = [(kept_names[i:i + step], kept[i:i + step]) for i in range(0, len(kept), step)]
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mer-tokenize")
    pending = [pool.submit(encode, sents) for _, sents in chunks]
    from .pipeline import writer
    with writer(model.device, async_save) as out:   # pinned non-blocking D2H + np.save on worker threads (extract.pipeline)
        for (chunk_names, _), fut in zip(chunks, pending):
            with span("tokenize"):
                ids_all = fut.result()
            todo = sorted(zip(chunk_names, ids_all), key=lambda it: len(it[1]))
            for i in range(0, len(todo), batch_size):
                chunk = todo[i:i + batch_size]
                T = max(len(ids) for _, ids in chunk)
                batch = torch.full((len(chunk), T), pad_id, dtype=torch.int64)
                lens = []
                for r, (_, ids) in enumerate(chunk):
                    batch[r, :len(ids)] = torch.as_tensor(ids, dtype=torch.int64)
                    lens.append(len(ids))
                names = [name for name, _ in chunk]
                if feature_level == 'FRAME':
                    with span("forward"):
                        _, frames, _ = model.forward_raw(batch, lengths=lens, frames=True)

                    def save_frames(arr, names=names, lens=lens, T=T):
                        arr = arr.reshape(len(names), T, -1)
                        for r, name in enumerate(names):
                            e = lens[r] + end if end is not None else lens[r]
                            save_embeddings(os.path.join(save_dir, f'{name}.npy'), arr[r, start:e], feature_level, feature_dim)
                    with span("submit"):
                        out.submit(frames, save_frames)
                else:
                    with span("forward"):
                        pooled = model.extract_utterance(batch, lens, start, end)

                    def save_utt(arr, names=names, lens=lens):
                        for r, name in enumerate(names):
                            n_tok = (lens[r] + (end if end is not None else 0)) - start
                            save_embeddings(os.path.join(save_dir, f'{name}.npy'), arr[r] if n_tok > 0 else [], feature_level, feature_dim)
                    with span("submit"):
                        out.submit(pooled, save_utt)
    pool.shutdown(wait=True)
    print(f'Total {len(df)} files done! Time used ({model_name}): {time.time() - start_time:.1f}s.')


def batch_encoder(tokenizer, probe_sentences=(), allow_twin=None):
    """-> encode(list of sentences) == [tokenizer(s)['input_ids'] for s in sentences] (the reference's per-row call, :216-225), by the
    cheapest route that is the SAME tokenizer:
      1. a Rust-backed tokenizer's own backend `encode_batch` called directly — transformers' wrapper spends 4x the tokenisation time
         turning every Encoding into six Python lists of which the driver reads one; taken when a probe (the corpus' first sentences,
         up to 256, + the probe sentence) gives the ids of the reference's call;
      2. otherwise the tokenizer as given, one call over the list.
    A pure-Python ("slow": what use_fast=False gives under the reference's transformers 4.28) tokenizer is NOT silently replaced by
    the Rust twin of its directory: slow and fast tokenizers can part on rare input (control / format characters, unusual unicode,
    sentencepiece legacy modes) anywhere in a corpus, and index paths are to be bit-exact.  The twin route
    is opt-in — `allow_twin=True` or MER_TEXT_TWIN=1 — and then guarded three ways: the probe above, the twin's backend reset to
    no truncation / no padding (its `__call__` never runs), and a spot check of every chunk (8 sentences spread over it, the longest
    included) against the given tokenizer; the first mismatch switches to the tokenizer as given, for that chunk and for good."""
    if allow_twin is None:
        allow_twin = os.environ.get('MER_TEXT_TWIN', '0') == '1'
    probe = [s for s in list(probe_sentences)[:256] if s] + [PROBE]

    def as_given(sents):
        return tokenizer(sents)['input_ids'] if sents else []

    def raw_route(tok, reset=False):
        backend = getattr(tok, 'backend_tokenizer', None)
        if backend is None or not getattr(tok, 'is_fast', False):
            return None
        if reset:
            backend.no_truncation()
            backend.no_padding()

        def enc(sents):
            return [e.ids for e in backend.encode_batch(sents, add_special_tokens=True)] if sents else []
        return enc

    def guarded(enc):
        state = {'ok': True}

        def run(sents):
            if not state['ok'] or not sents:
                return as_given(sents)
            ids = enc(sents)
            n = len(sents)
            pick = sorted({0, n - 1, max(range(n), key=lambda i: len(sents[i]))} | {(j * n) // 8 for j in range(8)})
            if any(ids[i] != tokenizer(sents[i])['input_ids'] for i in pick):
                state['ok'] = False
                return as_given(sents)
            return ids
        return run

    want = None
    for cand_name in ('raw', 'twin'):
        try:
            if cand_name == 'raw':
                enc = raw_route(tokenizer)
            else:
                if not allow_twin or getattr(tokenizer, 'is_fast', False):
                    break
                path = getattr(tokenizer, 'name_or_path', '')
                if not (path and os.path.isdir(path)):
                    break
                from transformers import AutoTokenizer
                enc = raw_route(AutoTokenizer.from_pretrained(path, use_fast=True), reset=True)
            if enc is None:
                continue
            if want is None:
                want = [tokenizer(s)['input_ids'] for s in probe]      # the reference's call, sentence by sentence
            if enc(probe) == want:
                return enc if cand_name == 'raw' else guarded(enc)
        except Exception:
            continue
    return as_given


def merge_subword_embeddings(tokens, output, sentence, combine_type='mean'):
    """Sub-word -> word alignment of MER2023/feature_extraction/text/extract_text_embedding_LZ.py:254-292 (English pipeline,
    SURVEY §8f row 4): `tokens` are the tokenizer's pieces of one sentence (special tokens already stripped), `output` their
    embeddings [T, D], `sentence` the words that were tokenised (is_split_into_words=True).  Pieces are glued until they spell
    the current word ('▁' of ALBERT/XLNet and 'Ġ' of RoBERTa/GPT stripped first, '##' of BERT when gluing); a piece equal
    to the word, or '[UNK]', maps one-to-one.  combine_type: 'sum' | 'mean' | 'last'.  Returns a list of len(sentence) vectors."""
    n_tokens, n_words = len(output), len(sentence)
    if n_tokens == n_words:            # :252-254 every sub-word is a word
        return list(output)
    sentence_embedding, pointer, word, word_embedding = [], 0, '', []
    for j, token in enumerate(tokens):
        token_embedding = output[j]
        current_word = sentence[pointer]
        token = token.replace('▁', '').replace('Ġ', '')
        if token == current_word or token == '[UNK]':
            sentence_embedding.append(token_embedding)
            pointer += 1
        else:
            word_embedding.append(token_embedding)
            word = word + token.replace('##', '')
            if word == current_word:
                if combine_type == 'sum':
                    merged = np.sum(np.vstack(word_embedding), axis=0)
                elif combine_type == 'mean':
                    merged = np.mean(np.vstack(word_embedding), axis=0)
                elif combine_type == 'last':
                    merged = word_embedding[-1]
                else:
                    raise Exception('Error: not supported type to combine subword embedding.')
                sentence_embedding.append(merged)
                word, word_embedding = '', []
                pointer += 1
    assert len(sentence) == len(sentence_embedding), f'{len(sentence)} words but {len(sentence_embedding)} merged embeddings: {tokens} / {sentence}'
    return sentence_embedding
