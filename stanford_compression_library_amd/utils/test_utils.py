"""The reference's test harness, by name and signature (reference scl/utils/test_utils.py:17-212), so that a test written
against the reference runs against this package after changing the import prefix (tests/test_gpu_reference_sweeps.py does
exactly that with the reference's own hot-path sweeps).

What the helpers check is dictated by the reference -- round-trip equality, ``num_bits_consumed == len(encoded)`` in the
presence of trailing garbage, code length within a tolerance of the empirical ``-log2 p`` or of a known rate -- the code is
this package's own.
"""
from __future__ import annotations

import filecmp
import os
import tempfile
from typing import Tuple

import numpy as np

from ..core.data_block import DataBlock
from ..core.data_encoder_decoder import DataDecoder, DataEncoder
from ..core.data_stream import TextFileDataStream, Uint8FileDataStream
from ..core.prob_dist import Frequencies, ProbabilityDist, get_avg_neg_log_prob
from .bitarray_utils import BitArray, get_random_bitarray

__all__ = ["get_random_data_block", "create_random_text_file", "create_random_binary_file", "are_blocks_equal",
           "try_lossless_compression", "try_file_lossless_compression", "lossless_entropy_coder_test",
           "lossless_test_against_expected_bitrate"]

_MAX_TRAILING_BITS = 100  # the reference appends np.random.randint(100) random bits (test_utils.py:98-101)


def get_random_data_block(prob_dist: ProbabilityDist, size: int, seed: int = None) -> DataBlock:
    """``size`` i.i.d. draws from ``prob_dist`` (reference :17-28).  Same generator and call as the reference, hence the
    same block for the same seed -- the golden vectors rely on it."""
    symbols = np.random.default_rng(seed).choice(prob_dist.alphabet, size=size, p=prob_dist.prob_list)
    return DataBlock(symbols.tolist())


def create_random_text_file(file_path: str, file_size: int, prob_dist: ProbabilityDist) -> None:
    """``file_size`` i.i.d. characters from ``prob_dist`` written as a text file (reference :31-41; unseeded, like there)"""
    with TextFileDataStream(file_path, "w") as fds:
        fds.write_block(get_random_data_block(prob_dist, file_size))


def create_random_binary_file(file_path: str, file_size: int, prob_dist: ProbabilityDist) -> None:
    """the same for a binary file: the distribution's alphabet must be byte values 0..255 (reference :44-55)"""
    with Uint8FileDataStream(file_path, "wb") as fds:
        fds.write_block(get_random_data_block(prob_dist, file_size))


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


def try_file_lossless_compression(input_file_path: str, encoder: DataEncoder, decoder: DataDecoder,
                                  encode_block_size=1000) -> bool:
    """``encode_file`` into a temporary directory, ``decode_file`` back, compare the bytes with the input (reference
    :111-135).  Text files, like there: ``encode_file`` / ``decode_file`` open text streams (data_encoder_decoder.py:71-83,
    :146-158).  With the HIP-backed coders this is the streaming driver's path end to end (row f2)."""
    with tempfile.TemporaryDirectory() as tmp:
        coded, back = os.path.join(tmp, "encoded_file.bin"), os.path.join(tmp, "reconst_file.txt")
        encoder.encode_file(input_file_path, coded, block_size=encode_block_size)
        decoder.decode_file(coded, back)
        return filecmp.cmp(input_file_path, back, shallow=False)


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
