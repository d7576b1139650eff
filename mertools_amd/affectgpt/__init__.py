"""AffectGPT frozen front-ends on the HIP encoders: `registry` mirrors the reference's encoder
registry, `encoder` holds the drop-in classes."""
from .registry import registry  # noqa: F401
from . import encoder  # noqa: F401
