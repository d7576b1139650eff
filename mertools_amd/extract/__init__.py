"""Batched MI355X drivers with the call signatures / .npy layout of MERBench's feature_extraction scripts
(SURVEY.md §8 a1-a11): extract.audio.extract, extract.visual.extract, extract.text.extract_embedding."""
