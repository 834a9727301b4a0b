"""tANS (table ANS = cached rANS) with the reference's class API, executed by the gfx950 kernels.

Drop-in for reference scl/compressors/tANS.py: ``tANSParams`` (:31-53), ``tANSEncoder`` (:56-193),
``tANSDecoder`` (:196-279).  The lookup tables are built on the device
(``csrc/scl_tans.hip``: ``tans_build_tables``) and exposed under the reference's attribute names as
dicts, because the reference's own test compares them as dicts (tANS.py:285-337).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from ..backend.models import TansModel
from ..core.data_block import DataBlock
from ..core.data_encoder_decoder import DataDecoder, DataEncoder
from ..utils.bitarray_utils import BitArray
from ..utils.misc_utils import is_power_of_two
from ._common import check_alphabet, indices_to_block, symbols_to_indices
from ._stream_batch import BatchedStreamDecoderMixin, BatchedStreamEncoderMixin
from .rANS import rANSParams

__all__ = ["tANSParams", "tANSEncoder", "tANSDecoder"]


@dataclass
class tANSParams(rANSParams):
    """rANSParams restricted to what makes rANS cachable (tANS.py:31-53)."""

    def __post_init__(self):
        super().__post_init__()
        assert is_power_of_two(self.M), \
            "Please normalize self.M parameter (sum of frequencies) to be a power of two"
        assert self.NUM_BITS_OUT == 1, "only NUM_OUT_BITS = 1 supported for now"
        if self.RANGE_FACTOR > (1 << 16):
            print("WARNING: RANGE_FACTOR > 2^16 --> the lookup tables could be huge")

    def _device_model(self) -> TansModel:
        model = self.__dict__.get("_model")
        if model is None:
            check_alphabet(self.freqs.alphabet)
            model = TansModel(self.freqs.freq_list, self.RANGE_FACTOR, self.DATA_BLOCK_SIZE_BITS)
            self.__dict__["_model"] = model
            self.__dict__["_index_of"] = self.freqs.symbol_index()
            self.__dict__["_alphabet"] = self.freqs.alphabet
        return model

    def _device_tables(self):
        tabs = self.__dict__.get("_tables")
        if tabs is None:
            tabs = self._device_model().tables()
            self.__dict__["_tables"] = tabs
        return tabs


class tANSEncoder(BatchedStreamEncoderMixin, DataEncoder):
    """Table-driven encoder; the three encoder tables of the reference are views of the device tables."""

    def __init__(self, tans_params: tANSParams):
        self.params = tans_params

    def _batch_model(self):
        return self.params._device_model(), self.params._index_of

    @property
    def base_encode_step_table(self) -> dict:
        """{(s, x_shrunk): next_state} (tANS.py:88-99)."""
        p, enc = self.params, self.params._device_tables()["enc"]
        out, cum, RF = {}, 0, p.RANGE_FACTOR
        for s, f in p.freqs.freq_dict.items():
            base = RF * cum
            for j in range(RF * f):
                out[(s, RF * f + j)] = int(enc[base + j])
            cum += f
        return out

    @property
    def shrink_state_num_out_bits_base_table(self) -> dict:
        nb = self.params._device_tables()["nbits"]
        return {s: int(nb[i]) for i, s in enumerate(self.params.freqs.alphabet)}

    @property
    def shrink_state_thresh_table(self) -> dict:
        th = self.params._device_tables()["thresh"]
        return {s: int(th[i]) for i, s in enumerate(self.params.freqs.alphabet)}

    def encode_block(self, data_block: DataBlock) -> BitArray:
        """Same stream layout as rANS (tANS.py:159-193)."""
        model = self.params._device_model()
        idx = symbols_to_indices(data_block, self.params._index_of)
        assert data_block.size < (1 << self.params.DATA_BLOCK_SIZE_BITS), "block size does not fit its header"
        packed, nbits = model.encode_host(idx)
        return BitArray.from_packed(packed, nbits)


class tANSDecoder(BatchedStreamDecoderMixin, DataDecoder):
    def __init__(self, tans_params: tANSParams):
        self.params = tans_params
        self._size_bits = tans_params.DATA_BLOCK_SIZE_BITS

    def _batch_model(self):
        return self.params._device_model(), self.params._alphabet

    @property
    def base_decode_step_table(self) -> dict:
        """{state: (s, x_shrunk)} for state in [L, H] (tANS.py:208-215)."""
        p, tabs = self.params, self.params._device_tables()
        alphabet = p.freqs.alphabet
        return {p.L + j: (alphabet[int(s)], int(xs)) for j, (s, xs) in enumerate(zip(tabs["dec_sym"], tabs["dec_xs"]))}

    @property
    def expand_state_num_bits_table(self) -> dict:
        """{x_shrunk: NUM_STATE_BITS - bit_width(x_shrunk)} (tANS.py:217-226); the kernel uses clz."""
        p, out = self.params, {}
        for s in p.freqs.alphabet:
            for xs in range(p.min_shrunk_state[s], p.max_shrunk_state[s] + 1):
                out[xs] = p.NUM_STATE_BITS - xs.bit_length()
        return out

    def decode_block(self, encoded_bitarray: BitArray) -> Tuple[DataBlock, int]:
        from ..backend.lib import E_CHUNK, SclHipError

        model = self.params._device_model()
        try:
            idx, used = model.decode_host(encoded_bitarray.packed(), len(encoded_bitarray),
                                          self.params.DATA_BLOCK_SIZE_BITS,
                                          max_block_size=getattr(self, "max_block_size", None))
        except SclHipError as e:
            if e.code == E_CHUNK and "STATE" in e.message:
                raise AssertionError("final tANS state != INITIAL_STATE") from e
            raise
        return indices_to_block(idx, self.params._alphabet), used
