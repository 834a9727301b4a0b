"""Shared glue of the four drop-in coder pairs: symbol <-> alphabet-index mapping and BitArray packing."""
from __future__ import annotations

import numpy as np

from ..core.data_block import DataBlock
from ..utils.bitarray_utils import BitArray


def index_dtype(alphabet_size: int):
    """uint8 indices for alphabets up to 256 symbols (the tuned kernels), uint16 up to 65536 (the *_u16 entry points)"""
    return np.uint16 if alphabet_size > 256 else np.uint8


def symbols_to_indices(data_block: DataBlock, index_of: dict) -> np.ndarray:
    """Map a block's symbols to alphabet indices (uint8, or uint16 for alphabets above 256 symbols); an unknown symbol
    raises ``KeyError`` exactly like ``Frequencies.frequency`` in the reference (prob_dist.py:207-208)."""
    data = data_block.data_list
    if isinstance(data, np.ndarray):
        data = data.tolist()
    # (map over the dict's own __getitem__: no generator frame per symbol -- half the time of a generator expression)
    return np.fromiter(map(index_of.__getitem__, data), dtype=index_dtype(len(index_of)), count=len(data))


def indices_to_block(idx: np.ndarray, alphabet: list) -> DataBlock:
    return DataBlock(list(map(alphabet.__getitem__, idx.tolist())))


def bitarray_to_packed(bits: BitArray):
    return bits.packed(), len(bits)


def check_alphabet(alphabet):
    if len(alphabet) > 65536:
        raise NotImplementedError(
            f"alphabet of {len(alphabet)} symbols: the MI355X kernels carry symbols as uint8 / uint16 indices (<= 65536)")
