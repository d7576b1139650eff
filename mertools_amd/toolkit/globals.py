"""Emotion <-> index tables (mirror of MERBench/toolkit/globals.py:1-5 — the label/index path must be bit-exact)."""
emos_mer = ['neutral', 'angry', 'happy', 'sad', 'worried', 'surprise']
emo2idx_mer = {emo: ii for ii, emo in enumerate(emos_mer)}
idx2emo_mer = {ii: emo for ii, emo in enumerate(emos_mer)}
