"""The reference's test harness, by name and signature (reference scl/utils/test_utils.py:17-212), so that a test written
against the reference runs against this package after changing the import prefix (tests/test_gpu_reference_sweeps.py does
exactly that with the reference's own hot-path sweeps).

What the helpers check is dictated by the reference -- round-trip equality, ``num_bits_consumed == len(encoded)`` in the
presence of trailing garbage, code length within a tolerance of the empirical ``-log2 p`` or of a known rate -- the code is
this package's own.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from ..core.data_block import DataBlock
from ..core.data_encoder_decoder import DataDecoder, DataEncoder
from ..core.prob_dist import Frequencies, ProbabilityDist, get_avg_neg_log_prob
from .bitarray_utils import BitArray, get_random_bitarray

__all__ = ["get_random_data_block", "are_blocks_equal", "try_lossless_compression", "lossless_entropy_coder_test",
           "lossless_test_against_expected_bitrate"]

_MAX_TRAILING_BITS = 100  # the reference appends np.random.randint(100) random bits (test_utils.py:98-101)


def get_random_data_block(prob_dist: ProbabilityDist, size: int, seed: int = None) -> DataBlock:
    """``size`` i.i.d. draws from ``prob_dist`` (reference :17-28).  Same generator and call as the reference, hence the
    same block for the same seed -- the golden vectors rely on it."""
    symbols = np.random.default_rng(seed).choice(prob_dist.alphabet, size=size, p=prob_dist.prob_list)
    return DataBlock(symbols.tolist())


def are_blocks_equal(data_block_1: DataBlock, data_block_2: DataBlock) -> bool:
    """same length and the same symbol at every position (reference :31-46)"""
    a, b = data_block_1.data_list, data_block_2.data_list
    return len(a) == len(b) and not any(x != y for x, y in zip(a, b))


def _bits_per_symbol(n_bits: int, block: DataBlock) -> float:
    return n_bits / block.size


def try_lossless_compression(data_block: DataBlock, encoder: DataEncoder, decoder: DataDecoder,
                             add_extra_bits_to_encoder_output: bool = False,
                             verbose: bool = False) -> Tuple[bool, int, BitArray]:
    """encode, optionally glue 0..99 random bits behind the code (numpy's global generator, like the reference: seed it for
    a reproducible run), decode; the decoder must report exactly the encoder's length as consumed (reference :73-108).
    Returns (round trip equal?, bits consumed, the code)."""
    code = encoder.encode_block(data_block)
    fed = BitArray(code)
    if add_extra_bits_to_encoder_output:
        fed += get_random_bitarray(int(np.random.randint(_MAX_TRAILING_BITS)))
    back, consumed = decoder.decode_block(fed)
    if verbose:
        print(f"{data_block.size} symbols -> {len(code)} bits (+{len(fed) - len(code)} trailing), consumed {consumed}")
    assert consumed == len(code), "Decoder did not consume all bits"
    return are_blocks_equal(data_block, back), consumed, code


def lossless_entropy_coder_test(encoder: DataEncoder, decoder: DataDecoder, freq: Frequencies, data_size: int,
                                encoding_optimality_precision: float = None, seed: int = 0):
    """a random block from ``freq``'s distribution must survive the round trip (with trailing garbage) and, if a
    precision is given, cost within that many bits per symbol of its empirical -log2 p (reference :138-180)"""
    dist = freq.get_prob_dist()
    block = get_random_data_block(dist, data_size, seed=seed)
    ideal = get_avg_neg_log_prob(dist, block)
    same, n_bits, _ = try_lossless_compression(block, encoder, decoder, add_extra_bits_to_encoder_output=True)
    rate = _bits_per_symbol(n_bits, block)
    print(f" avg_log_prob={ideal:.3f}, avg_codelen: {rate:.3f}")
    if encoding_optimality_precision is not None:
        assert abs(rate - ideal) < encoding_optimality_precision, (
            f"code length {rate:.4f} bit/symbol is not within {encoding_optimality_precision} of -log2 p = {ideal:.4f}")
    assert same, "round trip changed the data"


def lossless_test_against_expected_bitrate(encoder: DataEncoder, decoder: DataDecoder, data_block: DataBlock,
                                           expected_bitrate: float, encoding_optimality_precision: float):
    """the given block must survive the round trip (with trailing garbage) at a rate within the given precision of
    ``expected_bitrate`` (reference :183-212)"""
    same, n_bits, _ = try_lossless_compression(data_block, encoder, decoder, add_extra_bits_to_encoder_output=True)
    rate = _bits_per_symbol(n_bits, data_block)
    print(f" expected_bitrate={expected_bitrate:.3f}, avg_codelen: {rate:.3f}")
    assert abs(rate - expected_bitrate) < encoding_optimality_precision, (
        f"code length {rate:.4f} bit/symbol is not within {encoding_optimality_precision} of the expected "
        f"{expected_bitrate:.4f}")
    assert same, "round trip changed the data"
