"""rANS with the reference's class API, executed by the gfx950 kernels.

Drop-in for reference scl/compressors/rANS.py: ``rANSParams`` (:78-120), ``rANSEncoder`` (:123-210),
``rANSDecoder`` (:213-297).  ``encode_block`` / ``decode_block`` keep their signatures and produce /
consume bit-identical streams; the per-symbol loops run in
``stanford_compression_library_amd/csrc/scl_rans.hip`` (one lane per chunk).  There is no CPU
implementation behind these classes: without ``libscl_hip.so`` and an MI355X they raise.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from ..backend.models import RansModel
from ..core.data_block import DataBlock
from ..core.data_encoder_decoder import DataDecoder, DataEncoder
from ..core.prob_dist import Frequencies
from ..utils.bitarray_utils import BitArray, get_bit_width
from ._common import check_alphabet, indices_to_block, symbols_to_indices
from ._stream_batch import BatchedStreamDecoderMixin, BatchedStreamEncoderMixin

__all__ = ["rANSParams", "rANSEncoder", "rANSDecoder"]


@dataclass
class rANSParams:
    """Same fields, defaults and derived attributes as the reference dataclass (rANS.py:78-120)."""

    freqs: Frequencies
    DATA_BLOCK_SIZE_BITS: int = 32  # bits used for the block-size header
    NUM_BITS_OUT: int = 1           # bits streamed out per shrink step
    RANGE_FACTOR: int = 1 << 16     # state range is [RANGE_FACTOR*M, 2^NUM_BITS_OUT*RANGE_FACTOR*M - 1]

    def __post_init__(self):
        self.M = self.freqs.total_freq
        self.L = self.RANGE_FACTOR * self.M
        self.H = self.L * (1 << self.NUM_BITS_OUT) - 1
        # reference state arithmetic is numpy.int64: larger H silently overflows there (quirk Q7)
        assert self.H < (1 << 63), "H >= 2**63 overflows the reference's int64 state"
        scale = self.RANGE_FACTOR
        self.min_shrunk_state = {s: scale * f for s, f in self.freqs.freq_dict.items()}
        self.max_shrunk_state = {s: scale * f * (1 << self.NUM_BITS_OUT) - 1 for s, f in self.freqs.freq_dict.items()}
        self.INITIAL_STATE = self.L
        self.NUM_STATE_BITS = get_bit_width(self.H)
        self.BITS_OUT_MASK = 1 << self.NUM_BITS_OUT

    # -- device side -------------------------------------------------------------------------------
    def _device_model(self) -> RansModel:
        model = self.__dict__.get("_model")
        if model is None:
            check_alphabet(self.freqs.alphabet)
            model = RansModel(self.freqs.freq_list, self.RANGE_FACTOR, self.NUM_BITS_OUT, self.DATA_BLOCK_SIZE_BITS)
            self.__dict__["_model"] = model
            self.__dict__["_index_of"] = self.freqs.symbol_index()
            self.__dict__["_alphabet"] = self.freqs.alphabet
        return model


class rANSEncoder(BatchedStreamEncoderMixin, DataEncoder):
    def __init__(self, rans_params: rANSParams):
        self.params = rans_params

    def _batch_model(self):
        return self.params._device_model(), self.params._index_of

    def encode_block(self, data_block: DataBlock) -> BitArray:
        """[size | final state | per-symbol fields, last symbol first] -- rANS.py:186-210."""
        model = self.params._device_model()
        idx = symbols_to_indices(data_block, self.params._index_of)
        assert data_block.size < (1 << self.params.DATA_BLOCK_SIZE_BITS), "block size does not fit its header"
        packed, nbits = model.encode_host(idx)
        return BitArray.from_packed(packed, nbits)


class rANSDecoder(BatchedStreamDecoderMixin, DataDecoder):
    def __init__(self, rans_params: rANSParams):
        self.params = rans_params
        self._size_bits = rans_params.DATA_BLOCK_SIZE_BITS

    def _batch_model(self):
        return self.params._device_model(), self.params._alphabet

    def decode_block(self, encoded_bitarray: BitArray) -> Tuple[DataBlock, int]:
        """-> (DataBlock, num_bits_consumed); trailing bits are ignored -- rANS.py:270-297.
        A final state different from INITIAL_STATE raises AssertionError like the reference (:295)."""
        from ..backend.lib import E_CHUNK, SclHipError

        model = self.params._device_model()
        try:
            idx, used = model.decode_host(encoded_bitarray.packed(), len(encoded_bitarray),
                                          self.params.DATA_BLOCK_SIZE_BITS,
                                          max_block_size=getattr(self, "max_block_size", None))
        except SclHipError as e:
            if e.code == E_CHUNK and "STATE" in e.message:
                raise AssertionError("final rANS state != INITIAL_STATE") from e
            raise
        return indices_to_block(idx, self.params._alphabet), used
