"""lili_om_amd — MI355X (gfx950) native hot path of LiLi-OM: feature extraction + scan-to-map matcher.

The product is liblili_hip.so (hand-written HIP kernels behind the C ABI of include/lili_hip.h);
this package is the thin host-side binding used by tests and bench.py.  No CPU fallback exists.
"""
from . import api  # noqa: F401
from .api import (Context, ScanToMapMatcher, RotExtractor, LivoxExtractor, LocalMap, FrontendOdometry, RotFrontendOdometry, BackendKeyframes, LiliError, make_params, load_library,  # noqa: F401
                  KIND_SURF, KIND_EDGE, MASK_SURF, MASK_EDGE)

__all__ = ["api", "Context", "ScanToMapMatcher", "RotExtractor", "LivoxExtractor", "LocalMap", "FrontendOdometry", "RotFrontendOdometry", "BackendKeyframes", "LiliError", "make_params", "load_library",
           "KIND_SURF", "KIND_EDGE", "MASK_SURF", "MASK_EDGE"]
