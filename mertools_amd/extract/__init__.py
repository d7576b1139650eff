"""Batched MI355X drivers with the call signatures / .npy layout of MERBench's feature_extraction scripts
(SURVEY.md §8 a1-a11): extract.audio.extract, extract.visual.extract, extract.text.extract_embedding; and
extract.trimodal.TriModalExtractor — the three encoders of a clip batch on three streams with the next batch's
H2D copies overlapped (the deployable form of bench.py's step)."""
