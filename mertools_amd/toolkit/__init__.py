"""Host-side mirror of MERBench's `toolkit` package for the fusion stage of the hot path.
Same module / class / function names and argument meaning as the reference so main-release.py-style drivers
run unchanged; the model arithmetic goes through the HIP kernels (mertools_amd.fusion_ops)."""
