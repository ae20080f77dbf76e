"""cafe_amd -- MI355X-native implementation of CAFE's sketch-driven compact embedding layer.

The compute lives in libcafe_hip.so (hand-written HIP for gfx950, C ABI in include/cafe_hip.h);
this package is the thin host side that mirrors the reference's Python surface
(tricks/sk_embedding_bag.py, tricks/sketch_ml_embed.py)."""
from ._lib import LIB_PATH, CafeError, load  # noqa: F401

__all__ = ["CafeError", "load", "LIB_PATH"]
