"""Host-side block framing and data streams (rows f1/f2), no GPU: the known-answer checks of the reference's own
tests (core/encoded_stream.py:61-75,115-131,231-266; core/data_stream.py tests) restated on our classes."""
import os

import numpy as np

from stanford_compression_library_amd.core.data_block import DataBlock
from stanford_compression_library_amd.core.data_stream import (ListDataStream, TextFileDataStream,
                                                                Uint8FileDataStream)
from stanford_compression_library_amd.core.encoded_stream import (EncodedBlockReader, EncodedBlockWriter,
                                                                   HeaderHandler, Padder)
from stanford_compression_library_amd.utils.bitarray_utils import BitArray, get_random_bitarray


def test_padder_known_answers():
    # payload of 13 bits: 3 + 13 = 16 -> no pad bits; payload of 12 bits -> 1 pad bit
    assert Padder.add_byte_padding(BitArray("1" * 13)).to01() == "000" + "1" * 13
    assert Padder.add_byte_padding(BitArray("1" * 12)).to01() == "001" + "0" + "1" * 12
    assert Padder.add_byte_padding(BitArray("")).to01() == "101" + "00000"
    for n in range(0, 40):
        payload = get_random_bitarray(n)
        padded = Padder.add_byte_padding(payload)
        assert len(padded) % 8 == 0 and Padder.remove_byte_padding(padded) == payload


def test_header_roundtrip():
    padded = Padder.add_byte_padding(BitArray("1" * 23))
    framed = HeaderHandler.add_header(padded)
    data = framed.tobytes()
    assert HeaderHandler.get_payload_size(data[:4]) == len(padded) // 8 == 4
    assert data[:4] == bytes([0, 0, 0, 4])


def test_block_writer_reader_roundtrip(tmp_path):
    path = os.path.join(tmp_path, "enc.bin")
    blocks = [get_random_bitarray(n) for n in (0, 1, 7, 8, 13, 64, 1000)]
    with EncodedBlockWriter(path) as w:
        for b in blocks:
            w.write_block(b)
    with EncodedBlockReader(path) as r:
        got = []
        while True:
            blk = r.get_block()
            if blk is None:
                break
            got.append(blk)
    assert got == blocks
    # exact file layout of the first two records
    raw = open(path, "rb").read()
    assert raw[:5] == bytes([0, 0, 0, 1, 0b10100000])                       # empty block: pad count 5
    assert raw[5:10] == bytes([0, 0, 0, 1]) + bytes([(4 << 5) | blocks[1][0]])  # 1-bit block: pad count 4


def test_list_and_file_streams(tmp_path):
    s = ListDataStream(list(range(10)))
    assert s.get_block(4).data_list == [0, 1, 2, 3] and s.get_block(4).data_list == [4, 5, 6, 7]
    assert s.get_block(4).data_list == [8, 9] and s.get_block(4) is None
    s.seek(0)
    assert s.get_symbol() == 0
    out = ListDataStream([])
    out.write_block(DataBlock([5, 6]))
    assert out.input_list == [5, 6]
    tpath = os.path.join(tmp_path, "t.txt")
    with TextFileDataStream(tpath, "w") as f:
        f.write_block(DataBlock(list("hello world")))
    with TextFileDataStream(tpath, "r") as f:
        assert f.get_block(5).data_list == list("hello") and f.get_symbol() == " "
        assert f.get_block(100).data_list == list("world") and f.get_block(1) is None
    bpath = os.path.join(tmp_path, "b.bin")
    data = np.random.default_rng(0).integers(0, 256, 1000).tolist()
    with Uint8FileDataStream(bpath, "wb") as f:
        f.write_block(DataBlock(data))
    with Uint8FileDataStream(bpath, "rb") as f:
        assert f.get_block(1000).data_list == data and f.get_block(1) is None


def test_framing_equals_reference_fixture(tmp_path):
    """G10 of oracle/gen_goldens.py: bytes the reference's EncodedBlockWriter (core/encoded_stream.py:150-175)
    wrote for twelve bit strings of awkward lengths, reproduced by our writer and read back by our reader"""
    from conftest import load_golden

    case = [c for c in load_golden("stream") if c.kind == "framing"][0]
    nbits, packed = case.arr("block_nbits").tolist(), case.arr("block_out")
    blocks, pos = [], 0
    for nb in nbits:
        blocks.append(BitArray.from_packed(packed[pos:pos + (nb + 7) // 8], nb))
        pos += (nb + 7) // 8
    path = os.path.join(tmp_path, "enc.bin")
    with EncodedBlockWriter(path) as w:
        for b in blocks:
            w.write_block(b)
    assert np.array_equal(np.fromfile(path, dtype=np.uint8), case.arr("file"))
    ref_path = os.path.join(tmp_path, "ref.bin")
    case.arr("file").tofile(ref_path)
    with EncodedBlockReader(ref_path) as r:
        for b in blocks:
            assert r.get_block() == b
        assert r.get_block() is None
