"""Finite-precision arithmetic coder with the reference's class API, executed by the gfx950 kernels.

Drop-in for reference scl/compressors/arithmetic_coding.py: ``AECParams`` (:20-38), ``ArithmeticEncoder``
(:41-161), ``ArithmeticDecoder`` (:164-287).  Kernels: ``csrc/scl_aec.hip``.

Like the reference, an encoder / decoder object OWNS its ``freq_model`` and never resets it
(arithmetic_coding.py:52-56,118; quirk Q4): every ``encode_block`` / ``decode_block`` starts from the state the
model object is in -- counts and, for order-k, the last k symbols -- and leaves the advanced state in it, so
block 2+ of ``encode()`` / ``encode_file()`` is bit-identical to the reference's as well
(``scl_aec_{encode,decode}_host_resume``; fixture ``tests/golden/golden_stream.npz``).  The batched device API
(``backend.models.AecModel.encode_batch``) keeps the other meaning: one chunk = one fresh coder -- which is what a
``FixedFreqModel`` stream is, so ``encode()`` / ``decode()`` of a static model run as batched launches (round 5).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from ..backend.models import AecModel
from ..core.data_block import DataBlock
from ..core.data_encoder_decoder import DataDecoder, DataEncoder
from ..utils.bitarray_utils import BitArray
from ._common import check_alphabet, indices_to_block, symbols_to_indices
from ._stream_batch import BatchedStreamDecoderMixin, BatchedStreamEncoderMixin
from .probability_models import KIND_FIXED, FreqModelBase

__all__ = ["AECParams", "ArithmeticEncoder", "ArithmeticDecoder"]


@dataclass
class AECParams:
    DATA_BLOCK_SIZE_BITS: int = 32
    PRECISION: int = 32

    def __post_init__(self):
        self.FULL: int = 1 << self.PRECISION
        self.HALF: int = 1 << (self.PRECISION - 1)
        self.QTR: int = 1 << (self.PRECISION - 2)
        self.MAX_ALLOWED_TOTAL_FREQ: int = self.QTR
        self.MAX_BLOCK_SIZE: int = 1 << self.DATA_BLOCK_SIZE_BITS


class _AecBase:
    def __init__(self, params: AECParams, freq_model: FreqModelBase):
        self.params = params
        self.freq_model = freq_model
        self._model = None

    def _device_model(self) -> AecModel:
        if self._model is None:
            spec = self.freq_model.device_spec()
            check_alphabet(spec["alphabet"])
            if not 8 <= self.params.PRECISION <= 62:
                # 33..62 run the any-parameter kernels with low / high in 128 bits.  The reference's dataclass takes any
                # value but its arithmetic does not: PRECISION = 64 dies with a TypeError in `low << 1`, and range * count
                # wraps in numpy int64 once PRECISION + bit_length(total) exceeds 63 (oracle/gen_goldens.py, G7wide)
                raise NotImplementedError("PRECISION outside 8..62")
            self._model = AecModel(spec["kind"], spec["freq_init"], spec["K"], spec["k"], spec["max_total"],
                                   self.params.PRECISION, self.params.DATA_BLOCK_SIZE_BITS)
            self._alphabet = list(spec["alphabet"])
            self._index_of = {a: i for i, a in enumerate(self._alphabet)}
            self._stateful = spec["kind"] != KIND_FIXED  # adaptive models: state is carried by the model object
        return self._model


class ArithmeticEncoder(BatchedStreamEncoderMixin, _AecBase, DataEncoder):
    def _batch_model(self):
        return self._device_model(), self._index_of

    def encode(self, data_stream, block_size: int, encode_writer):
        """A FixedFreqModel makes the blocks independent: one batched launch per slab (compressors/_stream_batch.py).  An
        adaptive model carries its state from block to block (quirk Q4): the reference's block loop, one block at a time."""
        self._device_model()
        if self._stateful:
            return DataEncoder.encode(self, data_stream, block_size, encode_writer)
        return super().encode(data_stream, block_size, encode_writer)

    def encode_block(self, data_block: DataBlock) -> BitArray:
        """[size | renormalisation bits | termination bits] -- arithmetic_coding.py:80-161.
        The reference's ``size < 1 << MAX_BLOCK_SIZE`` assert (a 2^32-bit integer, quirk Q3) is replaced by
        the intended check against DATA_BLOCK_SIZE_BITS."""
        from ..backend.lib import E_CHUNK, SclHipError

        model = self._device_model()
        idx = symbols_to_indices(data_block, self._index_of)
        assert data_block.size < self.params.MAX_BLOCK_SIZE, \
            "choose a larger DATA_BLOCK_SIZE_BITS, as data_block.size is too big"
        try:
            if self._stateful:
                counts, past = self.freq_model.export_state()
                packed, nbits = model.encode_host_resume(idx, counts, past)
                self.freq_model.import_state(counts, past)
            else:
                packed, nbits = model.encode_host(idx)
        except SclHipError as e:
            if e.code == E_CHUNK and "TOTAL" in e.message:
                raise AssertionError("the frequency total is too large (>= MAX_ALLOWED_TOTAL_FREQ)") from e
            raise
        return BitArray.from_packed(packed, nbits)


class ArithmeticDecoder(BatchedStreamDecoderMixin, _AecBase, DataDecoder):
    def _batch_model(self):
        self._size_bits = self.params.DATA_BLOCK_SIZE_BITS
        return self._device_model(), self._alphabet

    def decode(self, encode_reader, output_stream):
        """batched for a FixedFreqModel, the block loop for adaptive models (see ``ArithmeticEncoder.encode``)"""
        self._device_model()
        if self._stateful:
            return DataDecoder.decode(self, encode_reader, output_stream)
        return super().decode(encode_reader, output_stream)

    def decode_block(self, encoded_bitarray: BitArray) -> Tuple[DataBlock, int]:
        """-> (DataBlock, num_bits_consumed); trailing bits are tolerated -- arithmetic_coding.py:203-287."""
        model = self._device_model()
        if self._stateful:
            counts, past = self.freq_model.export_state()
            idx, used = model.decode_host_resume(encoded_bitarray.packed(), len(encoded_bitarray),
                                                 self.params.DATA_BLOCK_SIZE_BITS, counts, past,
                                                 max_block_size=getattr(self, "max_block_size", None))
            self.freq_model.import_state(counts, past)
        else:
            idx, used = model.decode_host(encoded_bitarray.packed(), len(encoded_bitarray),
                                          self.params.DATA_BLOCK_SIZE_BITS,
                                          max_block_size=getattr(self, "max_block_size", None))
        return indices_to_block(idx, self._alphabet), used
