"""Carry-less byte-wise range coder with the reference's class API, executed by the gfx950 kernels.

Drop-in for reference scl/compressors/range_coder.py: ``RangeCoderParams`` (:55-76), ``RangeEncoder``
(:79-207), ``RangeDecoder`` (:217-317).  Kernels: ``csrc/scl_range.hip``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from ..backend.models import RangeModel
from ..core.data_block import DataBlock
from ..core.data_encoder_decoder import DataDecoder, DataEncoder
from ..core.prob_dist import Frequencies
from ..utils.bitarray_utils import BitArray
from ._common import check_alphabet, indices_to_block, symbols_to_indices
from ._stream_batch import BatchedStreamDecoderMixin, BatchedStreamEncoderMixin

__all__ = ["RangeCoderParams", "RangeEncoder", "RangeDecoder"]


@dataclass
class RangeCoderParams:
    DATA_BLOCK_SIZE_BITS: int = 32
    PRECISION: int = 32

    def __post_init__(self):
        assert self.PRECISION % 8 == 0
        self.TOP = 1 << (self.PRECISION - 8)      # one byte below the full precision
        self.BOTTOM = 1 << (self.PRECISION - 16)  # two bytes below: bound on total_freq and on range
        self.MASK = (1 << self.PRECISION) - 1


class _RangeBase:
    def __init__(self, params: RangeCoderParams, freqs: Frequencies):
        self.params = params
        self.freqs = freqs
        assert min(self.freqs.freq_dict.values()) > 0
        assert self.freqs.total_freq <= self.params.BOTTOM
        self._model = None

    def _device_model(self) -> RangeModel:
        if self._model is None:
            check_alphabet(self.freqs.alphabet)
            if self.params.PRECISION > 64:
                raise NotImplementedError("PRECISION > 64: the gfx950 kernels keep low/range in 64 bits")
            self._model = RangeModel(self.freqs.freq_list, self.params.PRECISION, self.params.DATA_BLOCK_SIZE_BITS)
            self._index_of = self.freqs.symbol_index()
            self._alphabet = self.freqs.alphabet
        return self._model


class RangeEncoder(BatchedStreamEncoderMixin, _RangeBase, DataEncoder):
    def _batch_model(self):
        return self._device_model(), self._index_of

    def encode_block(self, data_block: DataBlock) -> BitArray:
        """[size | renormalisation bytes | PRECISION/8 flush bytes] -- range_coder.py:188-207."""
        model = self._device_model()
        idx = symbols_to_indices(data_block, self._index_of)
        assert data_block.size < (1 << self.params.DATA_BLOCK_SIZE_BITS), "block size does not fit its header"
        packed, nbits = model.encode_host(idx)
        return BitArray.from_packed(packed, nbits)


class RangeDecoder(BatchedStreamDecoderMixin, _RangeBase, DataDecoder):
    def _batch_model(self):
        self._size_bits = self.params.DATA_BLOCK_SIZE_BITS
        return self._device_model(), self._alphabet

    def decode_block(self, encoded_bitarray: BitArray) -> Tuple[DataBlock, int]:
        """-> (DataBlock, num_bits_consumed incl. the size header) -- range_coder.py:269-317."""
        model = self._device_model()
        idx, used = model.decode_host(encoded_bitarray.packed(), len(encoded_bitarray),
                                      self.params.DATA_BLOCK_SIZE_BITS,
                                          max_block_size=getattr(self, "max_block_size", None))
        return indices_to_block(idx, self._alphabet), used
