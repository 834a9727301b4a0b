"""CPU: ``scl_framed_index_host`` (C ABI, host-only) against the reference-shaped reader -- ``EncodedBlockReader.get_block`` +
``Padder.remove_byte_padding`` (encoded_stream.py:196-225, :48-58) as mirrored in core/encoded_stream.py, which
tests/test_streams_framing.py pins on the reference's own fixtures."""
import os

import numpy as np
import pytest

from stanford_compression_library_amd.backend.models import framed_index_host
from stanford_compression_library_amd.core.encoded_stream import EncodedBlockReader, EncodedBlockWriter
from stanford_compression_library_amd.utils.bitarray_utils import BitArray, bitarray_to_uint, uint_to_bitarray


def _blocks(rng, n, size_bits):
    out = []
    for _ in range(n):
        size = int(rng.integers(0, 1 << min(size_bits, 20)))
        body = int(rng.integers(0, 300))
        out.append(uint_to_bitarray(size, size_bits) + BitArray("".join(rng.choice(["0", "1"], size=body).tolist())))
    return out


@pytest.mark.parametrize("size_bits", [1, 8, 13, 32, 40, 64])
def test_index_equals_block_reader(tmp_path, size_bits):
    rng = np.random.default_rng(size_bits)
    blocks = _blocks(rng, 200, size_bits)
    path = os.path.join(tmp_path, "f.bin")
    with EncodedBlockWriter(path) as w:
        for b in blocks:
            w.write_block(b)
    raw = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    offs, nbits, sizes, used = framed_index_host(raw, size_bits)
    assert used == raw.size and len(offs) == len(blocks)
    bits = np.unpackbits(raw)
    with EncodedBlockReader(path) as r:
        for i in range(len(blocks)):
            got = r.get_block()
            assert len(got) == int(nbits[i]) == len(blocks[i])
            assert bits[int(offs[i]): int(offs[i]) + int(nbits[i])].tolist() == [int(b) for b in got.to01()]
            assert int(sizes[i]) == bitarray_to_uint(got[:size_bits])
        assert r.get_block() is None


def test_partial_buffers_and_limits(tmp_path):
    rng = np.random.default_rng(5)
    blocks = _blocks(rng, 50, 32)
    path = os.path.join(tmp_path, "f.bin")
    with EncodedBlockWriter(path) as w:
        for b in blocks:
            w.write_block(b)
    raw = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    full = framed_index_host(raw, 32)
    # every cut: the records that end before it are indexed, `consumed` is where the crossing one starts
    ends = np.cumsum([4 + (len(b) + 3 + 7) // 8 for b in blocks])
    for cut in [0, 1, 3, 4, 5, int(ends[0]) - 1, int(ends[0]), int(ends[0]) + 3, int(ends[10]) + 4, raw.size - 1]:
        offs, nbits, sizes, used = framed_index_host(np.ascontiguousarray(raw[:cut]), 32)
        k = int(np.searchsorted(ends, cut, side="right"))
        assert len(offs) == k and used == (int(ends[k - 1]) if k else 0)
        assert (offs == full[0][:k]).all() and (nbits == full[1][:k]).all() and (sizes == full[2][:k]).all()
    offs, nbits, sizes, used = framed_index_host(raw, 32, max_records=7)
    assert len(offs) == 7 and used == int(ends[6])


def test_malformed_records_raise():
    good = (1).to_bytes(4, "big") + bytes([0b00000000])  # one payload byte: pad count 0, five stream bits
    offs, nbits, sizes, used = framed_index_host(np.frombuffer(good, dtype=np.uint8), 5)
    assert nbits.tolist() == [5] and offs.tolist() == [35] and used == 5
    with pytest.raises(AssertionError, match="shorter than"):  # the same record cannot hold a 6-bit size header
        framed_index_host(np.frombuffer(good, dtype=np.uint8), 6)
    with pytest.raises(AssertionError, match="empty payload"):
        framed_index_host(np.frombuffer(good + (0).to_bytes(4, "big") + b"\x00", dtype=np.uint8), 5)
    with pytest.raises(AssertionError, match="shorter than"):  # pad count 7 does not fit one byte with 3 + 7 + 1 bits
        framed_index_host(np.frombuffer((1).to_bytes(4, "big") + bytes([0b11100000]), dtype=np.uint8), 1)
    assert framed_index_host(np.zeros(0, dtype=np.uint8), 32)[3] == 0
