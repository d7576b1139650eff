"""mertools_amd — MI355X (gfx950) implementation of the MERTools feature-extraction + fusion hot path.

Python here is orchestration only; the compute lives in libmer_hip.so (HIP kernels behind the C ABI
of include/mer_hip.h).  Importing the package does not load the library; the first op does, and it
raises if the library has not been built (`python -m mertools_amd.build`).
"""
__version__ = "0.1.0"
