"""ucfp_amd -- MI355X (gfx950) fingerprinting + brute-force ANN core behind UCFP's operator surface.

Host-side mirror of the reference's hot-path modules (src/modality/{image,audio,text}.rs,
src/index/mod.rs); all arithmetic runs in hand-written HIP kernels reached through the C ABI
of include/ucfp_hip.h (libucfp_hip.so). There is no CPU fallback.
"""
from . import core, errors  # noqa: F401
from .core import FORMAT_VERSION, Hit, HitSource, Modality, Record  # noqa: F401

__all__ = ["core", "errors", "Record", "Hit", "HitSource", "Modality", "FORMAT_VERSION"]
