"""GPU: the BULK shape of the streaming driver (row f2; compressors/_stream_batch.py) -- file streams that move symbols as
arrays of codes -- must write the same file bytes, decode to the same symbols and raise the same exceptions as the LIST
shape (DataBlock lists, the reference's contract), whose bytes tests/test_gpu_stream_goldens.py pins on the reference's own
files."""
import os

import numpy as np
import pytest

from stanford_compression_library_amd.backend import lib as backend_lib
from stanford_compression_library_amd.backend.modeling import frequencies_from_counts
from stanford_compression_library_amd.compressors import _stream_batch
from stanford_compression_library_amd.compressors.arithmetic_coding import AECParams, ArithmeticDecoder, ArithmeticEncoder
from stanford_compression_library_amd.compressors.probability_models import AdaptiveIIDFreqModel, FixedFreqModel
from stanford_compression_library_amd.compressors.range_coder import RangeCoderParams, RangeDecoder, RangeEncoder
from stanford_compression_library_amd.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
from stanford_compression_library_amd.compressors.tANS import tANSDecoder, tANSEncoder, tANSParams
from stanford_compression_library_amd.core.data_block import DataBlock
from stanford_compression_library_amd.core.data_stream import ListDataStream, TextFileDataStream, Uint8FileDataStream
from stanford_compression_library_amd.core.encoded_stream import EncodedBlockReader, EncodedBlockWriter
from stanford_compression_library_amd.core.prob_dist import Frequencies

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _coders(coder, fr):
    if coder == "rans":
        p = rANSParams(fr)
        return rANSEncoder(p), rANSDecoder(p)
    if coder == "tans":
        p = tANSParams(fr, RANGE_FACTOR=1)
        return tANSEncoder(p), tANSDecoder(p)
    if coder == "aec":
        pa = AECParams()
        return (ArithmeticEncoder(pa, FixedFreqModel(fr, pa.MAX_ALLOWED_TOTAL_FREQ)),
                ArithmeticDecoder(pa, FixedFreqModel(fr, pa.MAX_ALLOWED_TOTAL_FREQ)))
    return RangeEncoder(RangeCoderParams(), fr), RangeDecoder(RangeCoderParams(), fr)


def _encode_list(enc, symbols, block_size, path):
    with EncodedBlockWriter(path) as w:
        enc.encode(ListDataStream(list(symbols)), block_size, w)
    return open(path, "rb").read()


@pytest.mark.parametrize("coder", ["rans", "tans", "range", "aec"])
@pytest.mark.parametrize("block_size,slab", [(777, 1 << 26), (1024, 1 << 26), (4096, 10_000), (500, 1500), (48, 64)])
def test_byte_file_bulk_equals_list_shape(coder, block_size, slab, tmp_path, monkeypatch):
    """several slabs (two alternating stages, records crossing the decoder's buffers, buffers that have to grow) or one"""
    backend_lib.require_device()
    monkeypatch.setattr(_stream_batch, "SLAB_BYTES", slab)
    rng = np.random.default_rng(block_size)
    data = rng.choice(256, size=23_456, p=np.r_[np.full(128, 0.006), np.full(128, 0.0018125)]).astype(np.uint8)
    fr = frequencies_from_counts(np.bincount(data, minlength=256), 4096)
    enc, dec = _coders(coder, fr)
    src, a, b, out = (os.path.join(tmp_path, n) for n in ("in.bin", "a.bin", "b.bin", "out.bin"))
    data.tofile(src)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, block_size, w)
    assert open(a, "rb").read() == _encode_list(enc, data.tolist(), block_size, b)
    if slab == 1 << 26:  # and both equal the block loop over the one-block entry points (encode_block)
        with EncodedBlockWriter(b) as w:
            for i in range(0, data.size, block_size):
                w.write_block(enc.encode_block(DataBlock(data[i:i + block_size].tolist())))
        assert open(a, "rb").read() == open(b, "rb").read()
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s:
        dec.decode(r, s)
    assert open(out, "rb").read() == data.tobytes()
    lst = ListDataStream([])
    with EncodedBlockReader(a) as r:
        dec.decode(r, lst)
    assert lst.input_list == data.tolist()


def test_sparse_alphabet_lut_and_unknown_symbol(tmp_path):
    """an alphabet that is not 0..255 in order goes through the code table; a byte outside it raises the KeyError of the
    per-block loop, naming the first such symbol"""
    backend_lib.require_device()
    alphabet = [200, 7, 10, 3, 255, 0]
    rng = np.random.default_rng(2)
    data = rng.choice(alphabet, size=9_999, p=[.4, .3, .1, .1, .05, .05]).astype(np.uint8)
    fr = Frequencies(dict(zip(alphabet, [40, 30, 10, 10, 5, 5])))
    enc, dec = rANSEncoder(rANSParams(fr)), rANSDecoder(rANSParams(fr))
    src, a, b, out = (os.path.join(tmp_path, n) for n in ("in.bin", "a.bin", "b.bin", "out.bin"))
    data.tofile(src)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 1000, w)
    assert open(a, "rb").read() == _encode_list(enc, data.tolist(), 1000, b)
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s:
        dec.decode(r, s)
    assert open(out, "rb").read() == data.tobytes()
    bad = data.copy()
    bad[4321], bad[7000] = 9, 11
    bad.tofile(src)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w, pytest.raises(KeyError) as e:
        enc.encode(s, 1000, w)
    assert e.value.args == (9,)
    with pytest.raises(KeyError) as e2, EncodedBlockWriter(b) as w:
        enc.encode(ListDataStream(bad.tolist()), 1000, w)
    assert e2.value.args == e.value.args
    # ... and, like the reference's block loop, every block in front of the offending one (block 4) has been written by
    # then: both shapes leave the first four records of the good file behind, whatever the slab size
    good = _encode_list(enc, data[:4000].tolist(), 1000, os.path.join(tmp_path, "g.bin"))
    assert open(a, "rb").read() == good and open(b, "rb").read() == good
    for slab in (1000, 3000, 5000):
        with pytest.MonkeyPatch.context() as mp:
            mp.setattr(_stream_batch, "SLAB_BYTES", slab)
            with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w, pytest.raises(KeyError):
                enc.encode(s, 1000, w)
        assert open(a, "rb").read() == good, slab


@pytest.mark.parametrize("alphabet", ["abcdefgh \n", "abéÿ \n", "abλ中 \n"])
def test_text_file_bulk_equals_list_shape(alphabet, tmp_path):
    """characters: one byte each when they all fit latin-1, else code points; encode_file / decode_file go this way"""
    backend_lib.require_device()
    rng = np.random.default_rng(len(alphabet))
    chars = list(alphabet)
    text = "".join(rng.choice(chars, size=12_345))
    fr = Frequencies(dict(zip(chars, frequencies_from_counts(np.array([text.count(c) for c in chars]), 1024).freq_list)))
    enc, dec = rANSEncoder(rANSParams(fr)), rANSDecoder(rANSParams(fr))
    src, a, b, out = (os.path.join(tmp_path, n) for n in ("in.txt", "a.bin", "b.bin", "out.txt"))
    open(src, "w").write(text)
    enc.encode_file(src, a, block_size=1000)
    assert open(a, "rb").read() == _encode_list(enc, list(text), 1000, b)
    dec.decode_file(a, out)
    assert open(out).read() == text
    open(src, "w").write(text[:5000] + "Z" + text[5000:])
    with pytest.raises(KeyError) as e:
        enc.encode_file(src, a, block_size=1000)
    assert e.value.args == ("Z",)


def test_wide_alphabet_text(tmp_path):
    """more than 256 symbols: uint16 indices through the *_u16 entry points, code points on the host"""
    backend_lib.require_device()
    chars = [chr(0x100 + i) for i in range(300)]
    rng = np.random.default_rng(9)
    text = "".join(rng.choice(chars, size=4_000))
    fr = Frequencies({ch: 1 + text.count(ch) for ch in chars})
    enc, dec = RangeEncoder(RangeCoderParams(), fr), RangeDecoder(RangeCoderParams(), fr)
    src, a, b, out = (os.path.join(tmp_path, n) for n in ("in.txt", "a.bin", "b.bin", "out.txt"))
    open(src, "w").write(text)
    enc.encode_file(src, a, block_size=333)
    assert open(a, "rb").read() == _encode_list(enc, list(text), 333, b)
    dec.decode_file(a, out)
    assert open(out).read() == text


def test_blocks_of_different_sizes_in_one_file(tmp_path):
    """a framed file is any sequence of records: rows of unequal length are packed on the device"""
    backend_lib.require_device()
    rng = np.random.default_rng(4)
    data = rng.choice(6, size=7_000, p=[.4, .3, .1, .1, .05, .05]).astype(np.uint8)
    fr = Frequencies(dict(zip(range(6), [40, 30, 10, 10, 5, 5])))
    enc, dec = rANSEncoder(rANSParams(fr)), rANSDecoder(rANSParams(fr))
    a, b, cat, out = (os.path.join(tmp_path, n) for n in ("a.bin", "b.bin", "cat.bin", "out.bin"))
    _encode_list(enc, data[:3_000].tolist(), 700, a)
    _encode_list(enc, data[3_000:].tolist(), 450, b)
    open(cat, "wb").write(open(a, "rb").read() + open(b, "rb").read())
    with EncodedBlockReader(cat) as r, Uint8FileDataStream(out, "wb") as s:
        dec.decode(r, s)
    assert open(out, "rb").read() == data.tobytes()


def test_empty_and_truncated_files(tmp_path):
    backend_lib.require_device()
    fr = Frequencies(dict(zip(range(4), [4, 2, 1, 1])))
    enc, dec = rANSEncoder(rANSParams(fr)), rANSDecoder(rANSParams(fr))
    src, a, out = (os.path.join(tmp_path, n) for n in ("in.bin", "a.bin", "out.bin"))
    open(src, "wb").close()
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 100, w)
    assert os.path.getsize(a) == 0
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s:
        dec.decode(r, s)
    assert os.path.getsize(out) == 0
    np.random.default_rng(1).choice(4, size=1000).astype(np.uint8).tofile(src)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 100, w)
    blob = open(a, "rb").read()
    for cut in (len(blob) - 1, len(blob) - 40, 3):
        open(a, "wb").write(blob[:cut])
        with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s, pytest.raises(AssertionError, match="truncated"):
            dec.decode(r, s)
    # a header that announces more bytes than the file has never sizes a buffer
    open(a, "wb").write((0xFFFFFFF0).to_bytes(4, "big") + blob[4:200])
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s, pytest.raises(AssertionError, match="truncated"):
        dec.decode(r, s)


def test_adaptive_arithmetic_coder_keeps_the_block_loop(tmp_path):
    """an adaptive model carries its state across blocks (quirk Q4): the file must equal the per-block loop's, whose bytes
    tests/test_gpu_stream_goldens.py pins on the reference's own multi-block files"""
    backend_lib.require_device()
    rng = np.random.default_rng(6)
    data = rng.choice(4, size=3_000, p=[.5, .25, .15, .1]).astype(np.uint8)
    pa = AECParams()
    fr = Frequencies({i: 1 for i in range(4)})
    src, a, b, out = (os.path.join(tmp_path, n) for n in ("in.bin", "a.bin", "b.bin", "out.bin"))
    data.tofile(src)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        ArithmeticEncoder(pa, AdaptiveIIDFreqModel(fr, pa.MAX_ALLOWED_TOTAL_FREQ)).encode(s, 500, w)
    enc = ArithmeticEncoder(pa, AdaptiveIIDFreqModel(fr, pa.MAX_ALLOWED_TOTAL_FREQ))
    with EncodedBlockWriter(b) as w:
        for i in range(0, data.size, 500):
            w.write_block(enc.encode_block(DataBlock(data[i:i + 500].tolist())))
    assert open(a, "rb").read() == open(b, "rb").read()
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s:
        ArithmeticDecoder(pa, AdaptiveIIDFreqModel(fr, pa.MAX_ALLOWED_TOTAL_FREQ)).decode(r, s)
    assert open(out, "rb").read() == data.tobytes()


_SEEDS = int(os.environ.get("SCL_STREAM_SEEDS", "12"))


@pytest.mark.parametrize("seed", range(_SEEDS))
def test_random_stream_campaign(seed, tmp_path, monkeypatch):
    """randomised: coder, alphabet (size, order, which bytes), block size, slab size, file length -- the bulk shape's file
    equals the list shape's, decodes back through both output shapes, and survives being decoded with another slab size"""
    backend_lib.require_device()
    rng = np.random.default_rng(1000 + seed)
    coder = ["rans", "tans", "range", "aec"][int(rng.integers(4))]
    K = int(rng.choice([2, 3, 16, 97, 256]))
    symbols = rng.permutation(256)[:K].astype(np.uint8)
    n = int(rng.choice([1, 17, 1000, 5_000, 40_000]))
    block_size = int(rng.choice([1, 7, 64, 100, 1024, 4096, 50_000]))
    p = rng.dirichlet(np.full(K, 0.5))
    data = symbols[rng.choice(K, size=n, p=p)]
    counts = np.maximum(1, np.round(p * 1000)).astype(np.int64)
    total = 1 << int(np.ceil(np.log2(max(int(counts.sum()), 2))))
    counts[int(np.argmax(counts))] += total - int(counts.sum())   # tANS wants a power-of-two total
    fr = Frequencies({int(s): int(c) for s, c in zip(symbols.tolist(), counts.tolist())})
    enc, dec = _coders(coder, fr)
    monkeypatch.setattr(_stream_batch, "SLAB_BYTES", int(rng.choice([64, 700, 10_000, 1 << 26])))
    src, a, b, out = (os.path.join(tmp_path, x) for x in ("in.bin", "a.bin", "b.bin", "out.bin"))
    data.tofile(src)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, block_size, w)
    assert open(a, "rb").read() == _encode_list(enc, data.tolist(), block_size, b), (coder, K, n, block_size)
    monkeypatch.setattr(_stream_batch, "SLAB_BYTES", int(rng.choice([64, 333, 10_000, 1 << 26])))
    # The arithmetic decoder's num_bits_consumed can fall short of a block's length (its rule for the bits read ahead,
    # arithmetic_coding.py:272-279, may give back one more bit than the encoder's termination wrote -- seen on one-symbol
    # blocks): the reference's block loop then fails its own assert (data_encoder_decoder.py:141), and so must both shapes.
    short = False
    if coder == "aec":
        with EncodedBlockReader(a) as r:
            while (blk := r.get_block()) is not None:
                short |= dec.decode_block(blk)[1] != len(blk)
    if short:
        with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s, pytest.raises(AssertionError, match="num_bits"):
            dec.decode(r, s)
        with EncodedBlockReader(a) as r, pytest.raises(AssertionError, match="num_bits"):
            dec.decode(r, ListDataStream([]))
        return
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s:
        dec.decode(r, s)
    assert open(out, "rb").read() == data.tobytes()
    lst = ListDataStream([])
    with EncodedBlockReader(a) as r:
        dec.decode(r, lst)
    assert lst.input_list == data.tolist()


def test_subclassed_streams_keep_the_list_shape(tmp_path):
    """a stream class that overrides the per-symbol / per-block methods (here: counting, and mapping on the way out) must
    see every symbol: no bulk shortcut around it"""
    backend_lib.require_device()

    class Counting(Uint8FileDataStream):
        seen = 0

        def get_block(self, block_size):
            blk = super().get_block(block_size)
            if blk is not None:
                Counting.seen += blk.size
            return blk

    class Upper(Uint8FileDataStream):
        def write_block(self, data_block):
            super().write_block(DataBlock([s ^ 0x20 for s in data_block.data_list]))

    rng = np.random.default_rng(3)
    data = rng.choice(np.arange(97, 123), size=5_000).astype(np.uint8)
    fr = Frequencies({int(c): 2 if c != 97 else 14 for c in range(97, 123)})   # total 64: a power of two for every coder
    enc, dec = _coders("rans", fr)
    src, a, out = (os.path.join(tmp_path, x) for x in ("in.bin", "a.bin", "out.bin"))
    data.tofile(src)
    with Counting(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 512, w)
    assert Counting.seen == data.size
    with EncodedBlockReader(a) as r, Upper(out, "wb") as s:
        dec.decode(r, s)
    assert open(out, "rb").read() == bytes(int(b) ^ 0x20 for b in data)


def test_malformed_record_after_valid_ones_decodes_them_first(tmp_path):
    """the reference's reader raises at the malformed record, after the records in front of it were decoded and written"""
    backend_lib.require_device()
    rng = np.random.default_rng(3)
    data = rng.integers(0, 4, size=6000, dtype=np.uint8)
    enc, dec = _coders("rans", Frequencies({0: 5, 1: 1, 2: 1, 3: 1}))
    src, a, out = (os.path.join(tmp_path, n) for n in ("in.bin", "a.bin", "out.bin"))
    data.tofile(src)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 1000, w)
    blob = bytearray(open(a, "rb").read())
    # walk to record 3 and make its payload shorter than a size header (a 1-byte payload, 8 - pad - 3 bits of it)
    pos = 0
    for _ in range(3):
        pos += 4 + int.from_bytes(blob[pos:pos + 4], "big")
    bad = bytes(blob[:pos]) + (1).to_bytes(4, "big") + b"\x00" + bytes(blob[pos:])
    open(a, "wb").write(bad)
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s, pytest.raises(AssertionError):
        dec.decode(r, s)
    assert open(out, "rb").read() == data[:3000].tobytes()


def test_gzip_like_file_object_and_read_codes_override(tmp_path):
    """a file object whose ``mode`` is not a string (gzip.GzipFile: an int) and a stream subclass that overrides
    only ``read_codes`` both go through ``read_codes``, not the direct ``readinto`` path"""
    import gzip

    backend_lib.require_device()
    rng = np.random.default_rng(4)
    data = rng.integers(0, 4, size=5000, dtype=np.uint8)
    enc, dec = _coders("rans", Frequencies({0: 5, 1: 1, 2: 1, 3: 1}))
    src, gz, a, b = (os.path.join(tmp_path, n) for n in ("in.bin", "in.gz", "a.bin", "b.bin"))
    data.tofile(src)
    with gzip.open(gz, "wb") as f:
        f.write(data.tobytes())
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 777, w)
    s = Uint8FileDataStream(src, "rb")
    s.__enter__()
    s.file_obj.close()
    s.file_obj = gzip.GzipFile(gz, "rb")
    assert not isinstance(s.file_obj.mode, str)
    with EncodedBlockWriter(b) as w:
        enc.encode(s, 777, w)
    s.file_obj.close()
    assert open(a, "rb").read() == open(b, "rb").read()

    class Inverted(Uint8FileDataStream):  # only read_codes overridden: its view of the file must be what gets encoded
        def read_codes(self, n):
            c = super().read_codes(n)
            return None if c is None else (3 - c).astype(np.uint8)

    with Inverted(src, "rb") as s, EncodedBlockWriter(b) as w:
        enc.encode(s, 777, w)
    out = ListDataStream([])
    with EncodedBlockReader(b) as r:
        dec.decode(r, out)
    assert out.input_list == (3 - data).tolist()


def test_blocks_above_16Mi_symbols_decode_by_default(tmp_path, monkeypatch):
    """a 2^24 default cap on the size header refused valid blocks -- plain ``encode_block`` ->
    ``decode_block`` above 16 Mi symbols, and the file decoder's one-block path (blocks above ``MAX_BLOCK_SYMBOLS``).  The
    default now accepts whatever the reference's 32-bit header announces; ``max_block_size`` is an opt-in guard."""
    backend_lib.require_device()
    n = (1 << 24) + 3
    rng = np.random.default_rng(24)
    data = rng.integers(0, 4, size=n, dtype=np.uint8)
    fr = Frequencies({0: 5, 1: 1, 2: 1, 3: 1})
    enc, dec = _coders("rans", fr)
    assert dec.max_block_size is None
    bits = enc.encode_block(DataBlock(data.tolist()))
    block, used = dec.decode_block(bits)
    assert used == len(bits) and block.size == n
    assert np.array_equal(np.asarray(block.data_list, dtype=np.uint8), data)
    dec.max_block_size = 1 << 24  # opt-in guard: the same block is refused, before anything is allocated
    with pytest.raises(AssertionError, match="max_block_size"):
        dec.decode_block(bits)
    # the file decoder's one-block path with BOTH limits in play: a block above MAX_BLOCK_SYMBOLS is decoded through
    # decode_block, whose cap is the decoder's own (None = accept)
    small = data[:5000]
    a, out = os.path.join(tmp_path, "a.bin"), os.path.join(tmp_path, "out.bin")
    small.tofile(os.path.join(tmp_path, "in.bin"))
    with Uint8FileDataStream(os.path.join(tmp_path, "in.bin"), "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 1000, w)
    monkeypatch.setattr(_stream_batch, "MAX_BLOCK_SYMBOLS", 400)
    dec.max_block_size = None
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s:
        dec.decode(r, s)
    assert open(out, "rb").read() == small.tobytes()
    dec.max_block_size = 999
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s, pytest.raises(AssertionError, match="max_block_size"):
        dec.decode(r, s)


@pytest.mark.parametrize("coder", ["rans", "range"])
def test_parallel_positional_reads_change_nothing(coder, tmp_path, monkeypatch):
    """utils/fileio.py: slabs of a regular file are read by several ``os.preadv`` calls at once (round 6).  Same file bytes
    out of ``encode``, same symbols out of ``decode``, as with the file object's own ``readinto`` -- slab boundaries in the
    middle of parts, records that cross the decoder's buffers, a last short slab"""
    from stanford_compression_library_amd.utils import fileio

    backend_lib.require_device()
    monkeypatch.setattr(_stream_batch, "SLAB_BYTES", 300_000)
    monkeypatch.setattr(fileio, "MIN_BYTES", 10_000)
    monkeypatch.setattr(fileio, "_ALIGN", 4096)
    rng = np.random.default_rng(8)
    data = rng.choice(256, size=2_345_678, p=np.r_[np.full(128, 0.006), np.full(128, 0.0018125)]).astype(np.uint8)
    fr = frequencies_from_counts(np.bincount(data, minlength=256), 4096)
    enc, dec = _coders(coder, fr)
    src, a, b, out = (os.path.join(tmp_path, n) for n in ("in.bin", "a.bin", "b.bin", "out.bin"))
    data.tofile(src)
    calls = []
    real = fileio.read_into

    def counted(fobj, view, n):
        got = real(fobj, view, n)
        calls.append(got)
        return got

    monkeypatch.setattr(fileio, "read_into", counted)
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(a) as w:
        enc.encode(s, 1000, w)
    with EncodedBlockReader(a) as r, Uint8FileDataStream(out, "wb") as s:
        dec.decode(r, s)
    assert open(out, "rb").read() == data.tobytes()
    assert sum(1 for c in calls if c) >= 8, calls  # the positional path really ran, on both sides
    monkeypatch.setattr(fileio, "read_into", lambda fobj, view, n: None)  # ... and the file object's own readinto
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(b) as w:
        enc.encode(s, 1000, w)
    assert open(a, "rb").read() == open(b, "rb").read()
