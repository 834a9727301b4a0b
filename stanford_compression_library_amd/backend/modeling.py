"""Model construction for real (non-synthetic) byte data -- row f3 of the scope table.

The reference offers ``DataBlock.get_counts`` / ``get_empirical_distribution`` (scl/core/data_block.py:37-94,
Python loops) and *no* frequency normaliser, although tANS asserts a power-of-two total (tANS.py:42-44) and the
range coder needs ``total <= 2**16``.  This module adds the two missing pieces:

* ``histogram_u8`` / ``histogram_u16`` -- symbol counts on the device (``scl_histogram_u8`` for bytes, ``scl_histogram_u16``
  for alphabets up to 65536 symbols); equal ``DataBlock.get_counts()``;
* ``normalize_counts`` -- a deterministic quantiser to a given total M (every present symbol keeps f >= 1,
  largest-remainder rounding, ties broken by symbol index), so encoder and decoder rebuild identical tables
  from the same counts.
"""
from __future__ import annotations

import numpy as np

from ..core.prob_dist import Frequencies
from . import lib as _lib


def histogram_u8(sym) -> np.ndarray:
    """uint8 CUDA tensor (any shape, contiguous) -> int64[256] counts."""
    import torch

    assert sym.is_cuda and sym.dtype == torch.uint8 and sym.is_contiguous()
    counts = torch.zeros(256, dtype=torch.int64, device=sym.device)
    rc = _lib.load().scl_histogram_u8(sym.data_ptr(), sym.numel(), counts.data_ptr(),
                                      torch.cuda.current_stream(sym.device).cuda_stream)
    _lib.check(rc, "scl_histogram_u8")
    return counts.cpu().numpy()


def histogram_u16(sym, K: int) -> np.ndarray:
    """uint16 (or int16) CUDA tensor of alphabet indices < K (any shape, contiguous) -> int64[K] counts
    (``scl_histogram_u16``: alphabets up to 65536 symbols); an index >= K raises ``KeyError`` like
    ``Frequencies.frequency`` (prob_dist.py:207-208)."""
    import torch

    assert sym.is_cuda and sym.dtype in (torch.uint16, torch.int16) and sym.is_contiguous()
    counts = torch.zeros(int(K), dtype=torch.int64, device=sym.device)
    bad = torch.zeros(1, dtype=torch.int32, device=sym.device)
    rc = _lib.load().scl_histogram_u16(sym.data_ptr(), sym.numel(), int(K), counts.data_ptr(), bad.data_ptr(),
                                       torch.cuda.current_stream(sym.device).cuda_stream)
    _lib.check(rc, "scl_histogram_u16")
    if int(bad.item()):
        raise KeyError(f"{int(bad.item())} symbol indices outside the alphabet of {K}")
    return counts.cpu().numpy()


def normalize_counts(counts, total: int = 4096) -> np.ndarray:
    """Scale non-negative integer counts to integers summing to ``total``; symbols with a non-zero count keep
    f >= 1, symbols with count 0 stay 0.  Deterministic (pure integer arithmetic)."""
    counts = np.asarray(counts, dtype=np.int64)
    assert counts.min() >= 0 and counts.sum() > 0
    present = counts > 0
    n_present = int(present.sum())
    assert total >= n_present, "total too small for the alphabet"
    n = int(counts.sum())
    # every present symbol gets 1 up front; the remaining mass is shared proportionally (largest remainder)
    spare = total - n_present
    scaled = counts * spare                      # exact integers
    base = scaled // n
    rem = scaled - base * n
    f = np.where(present, base + 1, 0)
    short = total - int(f.sum())
    if short > 0:
        order = np.lexsort((np.arange(counts.size), -rem))  # largest remainder first, then lowest index
        order = order[present[order]]
        f[order[:short]] += 1
    assert int(f.sum()) == total
    return f


def frequencies_from_counts(counts, total: int = 4096) -> Frequencies:
    """Frequencies over the symbols 0..255 that occur (insertion order = byte value), normalised to ``total``."""
    f = normalize_counts(counts, total)
    return Frequencies({int(s): int(v) for s, v in enumerate(f) if v > 0})
