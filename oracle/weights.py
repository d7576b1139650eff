"""Seeded synthetic checkpoints / inputs — shared with the product so both sides see identical tensors."""
from mertools_amd.synthetic import *  # noqa: F401,F403
