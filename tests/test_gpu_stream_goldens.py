"""GPU parity, rows a18 / f1 / f2 against REFERENCE-GENERATED stream fixtures (tests/golden/golden_stream.npz, written by
oracle/gen_goldens.py from the reference's own DataEncoder.encode / encode_file / EncodedBlockWriter):

* G9      three-block streams coded by ONE coder object: the framed file bytes, every block's bits, and -- for the
          arithmetic coder -- the adaptive model carried from block to block (quirk Q4: the reference never resets
          ``freq_model``, arithmetic_coding.py:52-56,118; core/data_encoder_decoder.py:57-69) plus the state the model
          object is left in;
* G9file  ``encode_file`` / ``decode_file`` on a text file;
* G10     ``EncodedBlockWriter`` bytes for bit strings of awkward lengths, reproduced by the device framing pass;
* the batched ``scl_aec_{encode,decode}_batch_resume`` entry points against the oracle carrying the same state.
Bit-exact everywhere (integer / bit work)."""
import os

import numpy as np
import pytest

import scl_oracle as orc
from conftest import golden_ids, load_golden, stream_blocks
from stanford_compression_library_amd.backend import models
from stanford_compression_library_amd.backend.models import EncodedBatch
from stanford_compression_library_amd.compressors.arithmetic_coding import (AECParams, ArithmeticDecoder,
                                                                             ArithmeticEncoder)
from stanford_compression_library_amd.compressors.probability_models import (AdaptiveIIDFreqModel,
                                                                               AdaptiveOrderKFreqModel,
                                                                               FixedFreqModel)
from stanford_compression_library_amd.compressors.range_coder import RangeCoderParams, RangeDecoder, RangeEncoder
from stanford_compression_library_amd.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
from stanford_compression_library_amd.compressors.tANS import tANSDecoder, tANSEncoder, tANSParams
from stanford_compression_library_amd.core.data_block import DataBlock
from stanford_compression_library_amd.core.data_stream import ListDataStream
from stanford_compression_library_amd.core.encoded_stream import EncodedBlockReader, EncodedBlockWriter
from stanford_compression_library_amd.core.prob_dist import Frequencies
from stanford_compression_library_amd.utils.bitarray_utils import BitArray

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ALL = load_golden("stream")
STREAM = [c for c in ALL if c.kind in ("rans", "tans", "range", "aec")]
FILES = [c for c in ALL if c.kind == "file"]
FRAMING = [c for c in ALL if c.kind == "framing"]
MODEL = {"fixed": orc.MODEL_FIXED, "iid": orc.MODEL_IID, "orderk": orc.MODEL_ORDERK}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def _freq_model(meta, alphabet):
    if meta["model"] == "orderk":
        return AdaptiveOrderKFreqModel(alphabet, meta["k"], meta["max_total"])
    fr = Frequencies(dict(zip(alphabet, meta["freq"])))
    return (FixedFreqModel if meta["model"] == "fixed" else AdaptiveIIDFreqModel)(fr, meta["max_total"])


def _coder_pair(meta, alphabet, kind):
    """-> (make_encoder, make_decoder) from a fixture's parameters"""
    if kind == "aec":
        p = AECParams(DATA_BLOCK_SIZE_BITS=meta["size_bits"], PRECISION=meta["precision"])
        return (lambda: ArithmeticEncoder(p, _freq_model(meta, alphabet)),
                lambda: ArithmeticDecoder(p, _freq_model(meta, alphabet)))
    fr = Frequencies(dict(zip(alphabet, meta["freq"])))
    if kind == "rans":
        p = rANSParams(fr, DATA_BLOCK_SIZE_BITS=meta["size_bits"], NUM_BITS_OUT=meta["b"], RANGE_FACTOR=meta["RF"])
        return (lambda: rANSEncoder(p)), (lambda: rANSDecoder(p))
    if kind == "tans":
        p = tANSParams(fr, DATA_BLOCK_SIZE_BITS=meta["size_bits"], RANGE_FACTOR=meta["RF"])
        return (lambda: tANSEncoder(p)), (lambda: tANSDecoder(p))
    p = RangeCoderParams(DATA_BLOCK_SIZE_BITS=meta["size_bits"], PRECISION=meta["precision"])
    return (lambda: RangeEncoder(p, fr)), (lambda: RangeDecoder(p, fr))


def _check_model_state(freq_model, case, tag):
    counts, past = case.arr(f"{tag}_counts"), case.arr(f"{tag}_past_k").tolist()
    if isinstance(freq_model, AdaptiveOrderKFreqModel):
        assert np.array_equal(np.asarray(freq_model.freqs_kplus1_tuple).ravel(), counts)
        assert list(freq_model.past_k) == past
    else:
        assert list(freq_model.freqs_current.freq_list) == counts.tolist()


@pytest.mark.parametrize("case", STREAM, ids=golden_ids(STREAM))
def test_stream_equals_reference_file(case, tmp_path, dev):
    K = len(case.freq)
    alphabet = [f"s{i}" for i in range(K)]
    data = [alphabet[i] for i in case.arr("sym").tolist()]
    make_enc, make_dec = _coder_pair(case.meta, alphabet, case.kind)
    # (1) the block loop: one object, encode() -> file bytes of the reference
    path = os.path.join(tmp_path, "enc.bin")
    encoder = make_enc()
    with EncodedBlockWriter(path) as w:
        encoder.encode(ListDataStream(list(data)), block_size=case.block_size, encode_writer=w)
    assert np.array_equal(np.fromfile(path, dtype=np.uint8), case.arr("file"))
    # (2) encode_block per block on a second object -> the reference's per-block bits
    enc2 = make_enc()
    for i, (sym, packed, nb) in enumerate(stream_blocks(case)):
        bits = enc2.encode_block(DataBlock([alphabet[j] for j in sym.tolist()]))
        assert len(bits) == nb and np.array_equal(bits.packed(), packed), f"block {i}"
    # (3) decode() of the REFERENCE's file with one decoder object
    ref_path = os.path.join(tmp_path, "ref.bin")
    case.arr("file").tofile(ref_path)
    decoder = make_dec()
    out = ListDataStream([])
    with EncodedBlockReader(ref_path) as r:
        decoder.decode(r, out)
    assert out.input_list == data
    # (4) the model objects end in the state the reference's objects end in
    if case.kind == "aec" and case.model != "fixed":
        _check_model_state(encoder.freq_model, case, "enc")
        _check_model_state(enc2.freq_model, case, "enc")
        _check_model_state(decoder.freq_model, case, "dec")


@pytest.mark.parametrize("case", FILES, ids=golden_ids(FILES))
def test_encode_file_equals_reference(case, tmp_path, dev):
    text = case.arr("text").tobytes().decode("ascii")
    chars = list(case.alphabet)
    src, dst, back = (os.path.join(tmp_path, n) for n in ("in.txt", "out.bin", "back.txt"))
    with open(src, "w") as f:
        f.write(text)
    make_enc, make_dec = _coder_pair(case.meta, chars, case.coder)
    make_enc().encode_file(src, dst, block_size=case.block_size)
    assert np.array_equal(np.fromfile(dst, dtype=np.uint8), case.arr("file"))
    ref = os.path.join(tmp_path, "ref.bin")
    case.arr("file").tofile(ref)
    make_dec().decode_file(ref, back)
    assert open(back).read() == text


@pytest.mark.parametrize("case", FRAMING, ids=golden_ids(FRAMING))
@pytest.mark.parametrize("back_to_front", [False, True])
def test_device_framing_equals_reference_writer(case, back_to_front, dev):
    """scl_streams_compact(SCL_COMPACT_FRAMED) on the fixture's bit strings == the bytes the reference's
    EncodedBlockWriter wrote (core/encoded_stream.py:150-175).  Streams are laid out the way the encoders leave them:
    front-aligned slots (range / arithmetic) or ending at the slot end (rANS / tANS)."""
    nbits = case.arr("block_nbits").astype(np.int64)
    packed = case.arr("block_out")
    stride = 384
    n = len(nbits)
    buf = np.zeros(n * stride + 16, np.uint8)
    offs = np.zeros(n, np.int64)
    pos = 0
    for c, nb in enumerate(nbits.tolist()):
        bits = np.unpackbits(packed[pos:pos + (nb + 7) // 8])[:nb]
        pos += (nb + 7) // 8
        start = 8 * (c + 1) * stride - nb if back_to_front else 8 * c * stride
        slot_bits = np.unpackbits(buf[c * stride:(c + 1) * stride])
        slot_bits[start - 8 * c * stride: start - 8 * c * stride + nb] = bits
        buf[c * stride:(c + 1) * stride] = np.packbits(slot_bits)
        offs[c] = start
    enc = EncodedBatch(torch.from_numpy(buf).to(dev), stride, torch.from_numpy(offs).to(dev),
                       torch.from_numpy(nbits.astype(np.int32)).to(dev), torch.zeros(n, dtype=torch.int32, device=dev), n)
    framed, offsets = models.compact(enc, framed=True)
    total = int(offsets[-1].item())
    assert np.array_equal(framed[:total].cpu().numpy(), case.arr("file"))
    dense, offsets = models.compact(enc, framed=False)
    assert np.array_equal(dense[: int(offsets[-1].item())].cpu().numpy(), packed)


RESUME_MODELS = [("iid", 6, 0, [34, 35, 546, 1, 13, 245], 1 << 30), ("iid", 4, 0, [1, 1, 1, 1], 1 << 10),
                 ("orderk", 16, 1, None, 1 << 30), ("orderk", 3, 2, None, 1 << 30), ("orderk", 256, 1, None, 1 << 30),
                 ("orderk", 40, 1, None, 1 << 30), ("orderk", 5, 0, None, 1 << 30)]


@pytest.mark.parametrize("model,K,k,freq,max_total", RESUME_MODELS, ids=[f"{m[0]}_K{m[1]}_k{m[2]}" for m in RESUME_MODELS])
def test_batch_resume_vs_oracle(model, K, k, freq, max_total, dev):
    """N coder objects x 3 ragged blocks through scl_aec_encode_batch_resume / decode_batch_resume: every block of every
    coder equals the oracle carrying the same state (which test_oracle_goldens.py pins on the reference's multi-block
    streams), and the downloaded device state equals the oracle's."""
    kind = MODEL[model]
    n_coders, width = 37, 300
    rng = np.random.default_rng(K * 10 + k)
    m = models.AecModel(kind, freq, K, k, max_total, 32, 32)
    st_enc = m.alloc_state(n_coders, dev)
    st_dec = m.alloc_state(n_coders, dev)
    o_enc = [orc.aec_fresh_state(kind, K, k, freq) for _ in range(n_coders)]
    o_dec = [s.copy() for s in o_enc]
    kw = dict(model_kind=kind, K=K, k=k, f_init=freq, max_total=max_total)
    for blk in range(4):
        # block 3 continues only the first 11 coders of the 37-coder state (the state's layout is that of
        # the n_coders it was reset with, whatever the length of the batch that resumes it)
        n_now = n_coders if blk < 3 else 11
        sym = rng.integers(0, K, (n_now, width)).astype(np.uint8)
        lens = rng.integers(0 if blk else 1, width + 1, n_now).astype(np.int32)
        lens[0] = width
        enc = m.encode_batch_resume(torch.from_numpy(sym).to(dev), st_enc, lens=torch.from_numpy(lens).to(dev))
        torch.cuda.synchronize()
        assert not enc.status.cpu().numpy().any()
        data, offs, nbits = enc.data.cpu().numpy(), enc.bit_offset.cpu().numpy(), enc.nbits.cpu().numpy()
        keep = lens > 0  # the reference's decoder never terminates on an empty block (quirk Q5)
        for c in range(n_now):
            ref, rn = orc.aec_encode(sym[c, :lens[c]], state=o_enc[c], **kw)
            assert rn == int(nbits[c]), (blk, c)
            got = np.unpackbits(data[offs[c] // 8: offs[c] // 8 + (rn + 7) // 8])[:rn]
            assert np.array_equal(got, np.unpackbits(ref)[:rn]), (blk, c)
        dsym, dlens, used, status = m.decode_batch_resume(enc.data, enc.bit_offset, enc.nbits, width, st_dec)
        torch.cuda.synchronize()
        assert not status.cpu().numpy().any()
        dsym, dlens, used = dsym.cpu().numpy(), dlens.cpu().numpy(), used.cpu().numpy()
        for c in range(n_now):
            assert dlens[c] == lens[c]
            assert np.array_equal(dsym[c, :lens[c]], sym[c, :lens[c]])
            if keep[c]:
                # num_bits_consumed as the reference computes it (arithmetic_coding.py:277-285): equal to the stream
                # length except for rare one-symbol blocks, where the oracle -- pinned on the reference -- says so too
                first = int(offs[c]) // 8
                back, ref_used = orc.aec_decode(data[first: first + (int(nbits[c]) + 7) // 8], int(nbits[c]),
                                                state=o_dec[c], **kw)
                assert used[c] == ref_used and np.array_equal(back, sym[c, :lens[c]])
                assert ref_used == nbits[c] or lens[c] <= 2
    with pytest.raises(AssertionError):  # more chunks than the state has coders
        m.encode_batch_resume(torch.zeros((n_coders + 1, 16), dtype=torch.uint8, device=dev), st_enc)
    for c in (0, 1, 10, 11, n_coders - 1):
        for st, oracle_state in ((st_enc, o_enc[c]), (st_dec, o_dec[c])):
            counts, past = m.state_download(st, n_coders, c)
            assert np.array_equal(counts.astype(np.uint64), oracle_state[:-1])
            ctx = 0
            for s in past[:k].tolist():
                ctx = ctx * K + s
            assert ctx == int(oracle_state[-1])
