"""MI355X-native batched entropy-coding core behind the Stanford Compression Library's coder API.

Scope: the per-symbol inner loops of SCL's rANS / tANS / range / arithmetic coders,
as hand-written gfx950 kernels behind ``include/scl_hip.h``; the Python classes mirror
``scl.compressors`` / ``scl.core`` / ``scl.utils`` for that path only.
"""
__version__ = "0.1.0"
