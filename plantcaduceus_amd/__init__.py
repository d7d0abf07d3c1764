"""plantcaduceus_amd — MI355X-native PlantCaduceus inference / embedding engine (hot path only).

Python host code mirroring the reference's HF `Auto*` surface; the arithmetic runs in `libpcad.so`
(hand-written HIP for gfx950) behind the C ABI declared in `include/pcad.h`.
"""
from .configuration_caduceus import CaduceusConfig  # noqa: F401

__version__ = "0.1.0"


def register():
    """Register the `caduceus` model type with the HF Auto classes (idempotent)."""
    from .modeling_caduceus import register_auto_classes
    register_auto_classes()
