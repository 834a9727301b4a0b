"""Test harness with the reference's names and checks (reference scl/utils/test_utils.py:17-212)."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from ..core.data_block import DataBlock
from ..core.data_encoder_decoder import DataDecoder, DataEncoder
from ..core.prob_dist import Frequencies, ProbabilityDist, get_avg_neg_log_prob
from .bitarray_utils import BitArray, get_random_bitarray


def get_random_data_block(prob_dist: ProbabilityDist, size: int, seed: int = None) -> DataBlock:
    """i.i.d. block from ``prob_dist`` (reference :17-28; same numpy generator, so same data per seed)."""
    rng = np.random.default_rng(seed)
    data = rng.choice(prob_dist.alphabet, size=size, p=prob_dist.prob_list)
    return DataBlock(data.tolist())


def are_blocks_equal(data_block_1: DataBlock, data_block_2: DataBlock) -> bool:
    if data_block_1.size != data_block_2.size:
        return False
    return all(a == b for a, b in zip(data_block_1.data_list, data_block_2.data_list))


def try_lossless_compression(data_block: DataBlock, encoder: DataEncoder, decoder: DataDecoder,
                             add_extra_bits_to_encoder_output: bool = False,
                             verbose: bool = False) -> Tuple[bool, int, BitArray]:
    """Round trip with optional random trailing bits; asserts the decoder consumed exactly the encoder's
    bits (reference :73-108)."""
    encoded_bitarray = encoder.encode_block(data_block)
    encoded_bitarray_extra = BitArray(encoded_bitarray)
    if add_extra_bits_to_encoder_output:
        encoded_bitarray_extra += get_random_bitarray(int(np.random.randint(100)))
    decoded_block, num_bits_consumed = decoder.decode_block(encoded_bitarray_extra)
    assert num_bits_consumed == len(encoded_bitarray), "Decoder did not consume all bits"
    return are_blocks_equal(data_block, decoded_block), num_bits_consumed, encoded_bitarray


def lossless_entropy_coder_test(encoder: DataEncoder, decoder: DataDecoder, freq: Frequencies, data_size: int,
                                encoding_optimality_precision: float = None, seed: int = 0):
    """losslessness + optional closeness of the code length to the empirical -log2 p (reference :138-180)."""
    prob_dist = freq.get_prob_dist()
    data_block = get_random_data_block(prob_dist, data_size, seed=seed)
    avg_log_prob = get_avg_neg_log_prob(prob_dist, data_block)
    is_lossless, encode_len, _ = try_lossless_compression(data_block, encoder, decoder,
                                                          add_extra_bits_to_encoder_output=True)
    avg_codelen = encode_len / data_block.size
    print(f" avg_log_prob={avg_log_prob:.3f}, avg_codelen: {avg_codelen:.3f}")
    if encoding_optimality_precision is not None:
        assert np.abs(avg_codelen - avg_log_prob) < encoding_optimality_precision, \
            f"avg_codelen={avg_codelen} is not {encoding_optimality_precision} close to avg_log_prob={avg_log_prob}"
    assert is_lossless


def lossless_test_against_expected_bitrate(encoder: DataEncoder, decoder: DataDecoder, data_block: DataBlock,
                                           expected_bitrate: float, encoding_optimality_precision: float):
    """losslessness + closeness to a known bitrate (reference :183-212)."""
    is_lossless, encode_len, _ = try_lossless_compression(data_block, encoder, decoder,
                                                          add_extra_bits_to_encoder_output=True)
    avg_codelen = encode_len / data_block.size
    print(f" expected_bitrate={expected_bitrate:.3f}, avg_codelen: {avg_codelen:.3f}")
    assert np.abs(avg_codelen - expected_bitrate) < encoding_optimality_precision, \
        f"avg_codelen={avg_codelen} is not {encoding_optimality_precision} close to expected_bitrate={expected_bitrate}"
    assert is_lossless
