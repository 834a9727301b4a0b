"""Generate golden vectors for the hot path from the *imported reference*.

Runs ONLY in the build container (the reference never travels):

    cd /tmp && PYTHONPATH=/root/reference /opt/conda/bin/python3.9 -W ignore \
        /root/repo/oracle/gen_goldens.py /root/repo/tests/golden

It imports ``scl`` (reference), encodes/decodes seeded inputs with the reference's own
Encoder/Decoder classes and stores *data only*: parameters, input symbol indices, the packed
output bits, their length, and ``num_bits_consumed`` for 0 / 3 / 61 stored trailing garbage bits.
The fixtures pin oracle/scl_oracle.c (tests/test_oracle_goldens.py) and, through it and directly,
the HIP kernels (tests/test_gpu_goldens.py).  Groups follow SURVEY.md section 8c (G1..G8); G9 / G10 (golden_stream.npz) are the reference's block loop
(multi-block streams with ONE coder object, encode_file) and its EncodedBlockWriter framing.
"""
import copy
import json
import os
import sys
import tempfile

import numpy as np

from scl.compressors.arithmetic_coding import AECParams, ArithmeticDecoder, ArithmeticEncoder
from scl.compressors.probability_models import (
    AdaptiveIIDFreqModel,
    AdaptiveOrderKFreqModel,
    FixedFreqModel,
)
from scl.compressors.range_coder import RangeCoderParams, RangeDecoder, RangeEncoder
from scl.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
from scl.compressors.tANS import tANSDecoder, tANSEncoder, tANSParams
from scl.core.data_block import DataBlock
from scl.core.data_stream import ListDataStream
from scl.core.encoded_stream import EncodedBlockReader, EncodedBlockWriter
from scl.core.prob_dist import Frequencies
from scl.utils.bitarray_utils import BitArray
from scl.utils.test_utils import get_random_data_block

GARBAGE = (0, 3, 61)


def t256_table():
    """S2 frequency table of SURVEY.md 8d: 256 symbols, M = 4096, every f >= 1."""
    w = np.random.default_rng(1).dirichlet(np.ones(256))
    f = np.maximum(1, np.floor(4096 * w)).astype(np.int64)
    f[int(np.argmax(f))] += 4096 - int(f.sum())
    assert f.sum() == 4096 and f.min() >= 1
    return f


def pack(bits: BitArray):
    return np.frombuffer(bits.tobytes(), dtype=np.uint8).copy(), len(bits)


class Collector:
    def __init__(self):
        self.cases = []
        self.arrays = {}

    def add(self, meta, **arrays):
        idx = len(self.cases)
        meta = dict(meta)
        meta["id"] = idx
        self.cases.append(meta)
        for k, v in arrays.items():
            self.arrays[f"c{idx}_{k}"] = np.asarray(v)

    def save(self, path):
        np.savez_compressed(path, manifest=np.array(json.dumps(self.cases)), **self.arrays)
        print(f"{path}: {len(self.cases)} cases")


def trailing_checks(make_decoder, enc_bits, rng, expect_syms, alphabet):
    """decode with 0/3/61 stored garbage bits appended; return garbage arrays + consumed counts."""
    garb, consumed = [], []
    for g in GARBAGE:
        extra = rng.integers(0, 2, g).astype(np.uint8)
        bits = BitArray(enc_bits)
        bits.extend("".join(str(int(b)) for b in extra))
        block, used = make_decoder().decode_block(bits)
        assert [alphabet.index(s) for s in block.data_list] == list(expect_syms), "reference round trip failed"
        assert used == len(enc_bits)
        garb.append(extra)
        consumed.append(used)
    return garb, consumed


def add_coder_case(col, kind, meta, alphabet, idx_syms, make_encoder, make_decoder, rng, decode=True):
    data = DataBlock([alphabet[i] for i in idx_syms])
    enc_bits = make_encoder().encode_block(data)
    packed, nbits = pack(enc_bits)
    arrays = dict(sym=np.asarray(idx_syms, dtype=np.uint16 if len(alphabet) > 256 else np.uint8), out=packed)
    meta = dict(meta, kind=kind, n=len(idx_syms), nbits=nbits)
    if decode:
        garb, consumed = trailing_checks(make_decoder, enc_bits, rng, idx_syms, alphabet)
        meta["consumed"] = consumed
        meta["garbage_lens"] = list(GARBAGE)
        for g, arr in zip(GARBAGE, garb):
            arrays[f"garbage{g}"] = arr
    col.add(meta, **arrays)
    return enc_bits


def iid_indices(freq_list, n, seed):
    p = np.asarray(freq_list, dtype=np.float64)
    p = p / p.sum()
    return np.random.default_rng(seed).choice(len(freq_list), size=n, p=p).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
def gen_rans(col, rng):
    # G1: the reference's own known-answer test (rANS.py:303-360)
    fr = Frequencies({"A": 3, "B": 3, "C": 2})
    p = rANSParams(fr, DATA_BLOCK_SIZE_BITS=5, NUM_BITS_OUT=1, RANGE_FACTOR=1)
    bits = add_coder_case(col, "rans", dict(group="G1", freq=[3, 3, 2], RF=1, b=1, size_bits=5), ["A", "B", "C"],
                          [0, 2, 1], lambda: rANSEncoder(p), lambda: rANSDecoder(p), rng)
    assert bits == BitArray("00011101110010")

    # G3: the reference's parameter sweep (rANS.py:366-379) at several lengths incl. empty
    sweep = [
        ([1, 1, 2], {}),
        ([12, 34, 1, 45], {}),
        ([34, 35, 546, 1, 13, 245], dict(NUM_BITS_OUT=8)),
        ([5, 5, 5, 5, 5, 5], dict(RANGE_FACTOR=1 << 12)),
        ([1, 3], dict(RANGE_FACTOR=1 << 4)),
    ]
    for fl, kw in sweep:
        alphabet = list("ABCDEF"[: len(fl)])
        fr = Frequencies(dict(zip(alphabet, fl)))
        p = rANSParams(fr, **kw)
        for n in (0, 1, 2, 17, 1000):
            syms = iid_indices(fl, n, seed=0)
            add_coder_case(col, "rans", dict(group="G3", freq=fl, RF=p.RANGE_FACTOR, b=p.NUM_BITS_OUT,
                                             size_bits=p.DATA_BLOCK_SIZE_BITS), alphabet, syms,
                           lambda: rANSEncoder(p), lambda: rANSDecoder(p), rng)

    # G3b: K = 256, M = 4096 (the headline model) at several (b, RF)
    f256 = t256_table().tolist()
    alphabet = list(range(256))
    fr = Frequencies(dict(zip(alphabet, f256)))
    for b, RF, n in ((1, 1 << 16, 4096), (8, 1 << 8, 2048), (16, 1, 1024), (1, 1, 1024), (3, 1 << 5, 512)):
        p = rANSParams(fr, NUM_BITS_OUT=b, RANGE_FACTOR=RF)
        syms = iid_indices(f256, n, seed=2)
        add_coder_case(col, "rans", dict(group="G3b", freq=f256, RF=RF, b=b, size_bits=32), alphabet, syms,
                       lambda: rANSEncoder(p), lambda: rANSDecoder(p), rng)

    # G3c: uniform 256-symbol table (f = 16, M = 4096), default params
    fu = [16] * 256
    fr = Frequencies(dict(zip(alphabet, fu)))
    p = rANSParams(fr)
    syms = np.random.default_rng(3).integers(0, 256, 2048, dtype=np.uint8)
    add_coder_case(col, "rans", dict(group="G3c", freq=fu, RF=p.RANGE_FACTOR, b=1, size_bits=32), alphabet, syms,
                   lambda: rANSEncoder(p), lambda: rANSDecoder(p), rng)

    # G4: BASELINE.json configs[0] -- 4 KiB Bernoulli(0.8) block, default params
    fr = Frequencies({0: 1, 1: 4})
    p = rANSParams(fr)
    block = get_random_data_block(fr.get_prob_dist(), 4096, seed=0)
    bits = add_coder_case(col, "rans", dict(group="G4", freq=[1, 4], RF=p.RANGE_FACTOR, b=1, size_bits=32), [0, 1],
                          np.asarray(block.data_list, dtype=np.uint8), lambda: rANSEncoder(p), lambda: rANSDecoder(p), rng)
    assert len(bits) == 3068


def gen_tans(col, rng):
    # G2: lookup tables of the reference's KAT model (tANS.py:285-337) + its bitstream (:340-415)
    fl = [3, 3, 2]
    alphabet = ["A", "B", "C"]
    fr = Frequencies(dict(zip(alphabet, fl)))
    p = tANSParams(fr, RANGE_FACTOR=1, NUM_BITS_OUT=1, DATA_BLOCK_SIZE_BITS=5)
    enc, dec = tANSEncoder(p), tANSDecoder(p)
    enc_tab = np.array([[alphabet.index(s), xs, v] for (s, xs), v in enc.base_encode_step_table.items()], dtype=np.int64)
    dec_tab = np.array([[x, alphabet.index(s), xs] for x, (s, xs) in dec.base_decode_step_table.items()], dtype=np.int64)
    nb_tab = np.array([enc.shrink_state_num_out_bits_base_table[s] for s in alphabet], dtype=np.int64)
    th_tab = np.array([enc.shrink_state_thresh_table[s] for s in alphabet], dtype=np.int64)
    ex_tab = np.array(sorted(dec.expand_state_num_bits_table.items()), dtype=np.int64)
    col.add(dict(kind="tans_tables", group="G2", freq=fl, RF=1), enc_tab=enc_tab, dec_tab=dec_tab, nbits_tab=nb_tab,
            thresh_tab=th_tab, expand_tab=ex_tab)
    bits = add_coder_case(col, "tans", dict(group="G2", freq=fl, RF=1, size_bits=5), alphabet, [0, 2, 1],
                          lambda: tANSEncoder(p), lambda: tANSDecoder(p), rng)
    assert bits == BitArray("00011101110010")

    # G5: the reference's tANS sweep (tANS.py:421-430; the third set uses RF=2^8 / 2^10 instead of the
    # default 2^16, whose 2^20-entry tables take minutes to build in the reference)
    sweep = [([1, 1, 2], 1), ([1, 3], 1 << 4), ([3, 4, 9], 1 << 8), ([3, 4, 9], 1 << 10)]
    for fl, RF in sweep:
        alphabet = list("ABCDEF"[: len(fl)])
        fr = Frequencies(dict(zip(alphabet, fl)))
        p = tANSParams(fr, RANGE_FACTOR=RF)
        enc_obj, dec_obj = tANSEncoder(p), tANSDecoder(p)
        for n in (0, 1, 17, 1000):
            syms = iid_indices(fl, n, seed=0)
            add_coder_case(col, "tans", dict(group="G5", freq=fl, RF=p.RANGE_FACTOR, size_bits=32), alphabet, syms,
                           lambda: enc_obj, lambda: dec_obj, rng)

    # G5b: K = 256, M = 4096, RF = 1 and RF = 16; the stream must equal rANS with equal params
    f256 = t256_table().tolist()
    alphabet = list(range(256))
    fr = Frequencies(dict(zip(alphabet, f256)))
    for RF, n in ((1, 4096), (16, 1024)):
        p = tANSParams(fr, RANGE_FACTOR=RF)
        enc_obj, dec_obj = tANSEncoder(p), tANSDecoder(p)
        syms = iid_indices(f256, n, seed=2)
        bits = add_coder_case(col, "tans", dict(group="G5b", freq=f256, RF=RF, size_bits=32), alphabet, syms,
                              lambda: enc_obj, lambda: dec_obj, rng)
        rp = rANSParams(fr, RANGE_FACTOR=RF)
        assert bits == rANSEncoder(rp).encode_block(DataBlock(syms.tolist()))


def gen_range(col, rng):
    # G6: reference sweep (range_coder.py:336-341) + edge cases (:351-374) + K=256 uniform
    sweep = [[1, 1, 2], [12, 34, 1, 45], [34, 35, 546, 1, 13, 245], [5, 5, 5, 5, 5, 5], [1, 3], [1, 65534]]
    for fl in sweep:
        alphabet = list("ABCDEF"[: len(fl)])
        fr = Frequencies(dict(zip(alphabet, fl)))
        p = RangeCoderParams()
        syms = iid_indices(fl, 1000, seed=0)
        add_coder_case(col, "range", dict(group="G6", freq=fl, precision=32, size_bits=32), alphabet, syms,
                       lambda: RangeEncoder(p, fr), lambda: RangeDecoder(p, fr), rng)
    p = RangeCoderParams()
    edge = [([1, 65535], [0, 1] * 500), ([1, 1, 65534], [0, 1, 2] * 300), ([1, 1, 65534], [0] * 700),
            ([1, 1, 65534], [2] * 700), ([1, 1, 65534], [1] * 33)]
    for fl, pattern in edge:
        alphabet = list("ABC"[: len(fl)])
        fr = Frequencies(dict(zip(alphabet, fl)))
        add_coder_case(col, "range", dict(group="G6edge", freq=fl, precision=32, size_bits=32), alphabet,
                       pattern, lambda: RangeEncoder(p, fr), lambda: RangeDecoder(p, fr), rng)
    # every length 0..49 (flush correctness, empty block; range_coder.py:369-374)
    fl = [12, 34, 1, 45]
    alphabet = list("ABCD")
    fr = Frequencies(dict(zip(alphabet, fl)))
    base = iid_indices(fl, 5000, seed=0)
    for n in range(0, 50):
        add_coder_case(col, "range", dict(group="G6len", freq=fl, precision=32, size_bits=32), alphabet,
                       base[:n], lambda: RangeEncoder(p, fr), lambda: RangeDecoder(p, fr), rng)
    fu = [1] * 256
    alphabet = list(range(256))
    fr = Frequencies(dict(zip(alphabet, fu)))
    p = RangeCoderParams()
    syms = np.random.default_rng(3).integers(0, 256, 4096, dtype=np.uint8)
    bits = add_coder_case(col, "range", dict(group="G6u", freq=fu, precision=32, size_bits=32), alphabet, syms,
                          lambda: RangeEncoder(p, fr), lambda: RangeDecoder(p, fr), rng)
    print("  range K=256 uniform n=4096:", len(bits), "bits")
    f256 = t256_table().tolist()
    fr = Frequencies(dict(zip(alphabet, f256)))
    for prec, sb in ((32, 32), (24, 13), (40, 32)):
        if sum(f256) > (1 << (prec - 16)):
            continue
        p = RangeCoderParams(DATA_BLOCK_SIZE_BITS=sb, PRECISION=prec)
        syms = iid_indices(f256, 1500, seed=4)
        add_coder_case(col, "range", dict(group="G6p", freq=f256, precision=prec, size_bits=sb), alphabet, syms,
                       lambda: RangeEncoder(p, fr), lambda: RangeDecoder(p, fr), rng)


def markov1(K, n, seed=4):
    """S4 source of SURVEY.md 8d: order-1 Markov chain with Dirichlet(0.3) rows, x_0 drawn from row 0."""
    rng = np.random.default_rng(seed)
    P = rng.dirichlet(0.3 * np.ones(K), size=K)
    x = np.zeros(n, dtype=np.uint8)
    prev = 0
    cdf = np.cumsum(P, axis=1)
    u = rng.random(n)
    for t in range(n):
        prev = min(int(np.searchsorted(cdf[prev], u[t], side="right")), K - 1)
        x[t] = prev
    return x


def markov2_ref(n, seed=0):
    """the reference's test source (arithmetic_coding.py:384-402), restated as index data"""
    rng = np.random.default_rng(seed)
    random_bits = rng.choice(2, size=n - 2)
    x = np.zeros(n, dtype=np.uint8)
    x[0] = rng.choice(3)
    x[1] = rng.choice(3)
    for i in range(2, n):
        x[i] = (x[i - 1] + x[i - 2] + random_bits[i - 2]) % 3
    return x


def gen_aec(col, rng):
    # G7: fixed + adaptive-iid models (arithmetic_coding.py:305-317, :346-358), incl. PRECISION=16
    sweep = [([1, 1, 2], {}), ([12, 34, 1, 45], {}), ([34, 35, 546, 1, 13, 245], dict(DATA_BLOCK_SIZE_BITS=12)),
             ([5, 5, 5, 5, 5, 5], dict(DATA_BLOCK_SIZE_BITS=12, PRECISION=16))]
    for fl, kw in sweep:
        alphabet = list("ABCDEF"[: len(fl)])
        fr = Frequencies(dict(zip(alphabet, fl)))
        p = AECParams(**kw)
        syms = iid_indices(fl, 1000, seed=0)
        for model_name, init in (("fixed", fl), ("iid", fl), ("iid", [1] * len(fl))):
            fr_init = Frequencies(dict(zip(alphabet, init)))
            if model_name == "fixed":
                mk = lambda: FixedFreqModel(fr_init, p.MAX_ALLOWED_TOTAL_FREQ)
            else:
                mk = lambda: AdaptiveIIDFreqModel(fr_init, p.MAX_ALLOWED_TOTAL_FREQ)
            add_coder_case(col, "aec", dict(group="G7", model=model_name, freq=init, K=len(fl), k=0,
                                            max_total=p.MAX_ALLOWED_TOTAL_FREQ, precision=p.PRECISION,
                                            size_bits=p.DATA_BLOCK_SIZE_BITS), alphabet, syms,
                           lambda: ArithmeticEncoder(p, mk()), lambda: ArithmeticDecoder(p, mk()), rng)
    # PRECISION=16 with long input exercises the halving rule of the adaptive model
    fl = [3, 1, 7, 2]
    alphabet = list("ABCD")
    p = AECParams(PRECISION=16)
    syms = iid_indices(fl, 40000, seed=5)
    fr_init = Frequencies(dict(zip(alphabet, [1] * 4)))
    mk = lambda: AdaptiveIIDFreqModel(fr_init, p.MAX_ALLOWED_TOTAL_FREQ)
    add_coder_case(col, "aec", dict(group="G7halve", model="iid", freq=[1] * 4, K=4, k=0, max_total=p.MAX_ALLOWED_TOTAL_FREQ,
                                    precision=16, size_bits=32), alphabet, syms,
                   lambda: ArithmeticEncoder(p, mk()), lambda: ArithmeticDecoder(p, mk()), rng)
    # G7wide: PRECISION above 32.  The reference's dataclass takes any value, its arithmetic does not: low / high are
    # Python integers but the cumulative counts are numpy int64, so range * count silently wraps once PRECISION +
    # bit_length(total) exceeds 63, and PRECISION = 64 dies with a TypeError in `low << 1` (probed here; like quirk Q7).
    # 40 and 48 with small totals are inside its sound region.
    for prec in (40, 48):
        fl = [12, 34, 1, 45]
        alphabet = list("ABCD")
        p = AECParams(PRECISION=prec)
        syms = iid_indices(fl, 600, seed=prec)
        fr_init = Frequencies(dict(zip(alphabet, fl)))
        for model_name, mk, freq, k in (
                ("fixed", lambda: FixedFreqModel(fr_init, p.MAX_ALLOWED_TOTAL_FREQ), fl, 0),
                ("iid", lambda: AdaptiveIIDFreqModel(fr_init, p.MAX_ALLOWED_TOTAL_FREQ), fl, 0),
                ("orderk", lambda: AdaptiveOrderKFreqModel(alphabet, 1, p.MAX_ALLOWED_TOTAL_FREQ), [1] * 4, 1)):
            add_coder_case(col, "aec", dict(group="G7wide", model=model_name, freq=freq, K=4, k=k,
                                            max_total=p.MAX_ALLOWED_TOTAL_FREQ, precision=prec, size_bits=32),
                           alphabet, syms, lambda: ArithmeticEncoder(p, mk()), lambda: ArithmeticDecoder(p, mk()), rng)
    # short blocks incl. length 1 (length 0 never terminates in the reference decoder, quirk Q5)
    for n in (1, 2, 3, 9):
        fl = [2, 1, 5]
        alphabet = list("ABC")
        p = AECParams()
        fr_init = Frequencies(dict(zip(alphabet, fl)))
        mk = lambda: AdaptiveIIDFreqModel(fr_init, p.MAX_ALLOWED_TOTAL_FREQ)
        add_coder_case(col, "aec", dict(group="G7short", model="iid", freq=fl, K=3, k=0, max_total=p.MAX_ALLOWED_TOTAL_FREQ,
                                        precision=32, size_bits=32), alphabet, iid_indices(fl, n, seed=n),
                       lambda: ArithmeticEncoder(p, mk()), lambda: ArithmeticDecoder(p, mk()), rng)
    # empty block: encoder only
    p = AECParams()
    fr_init = Frequencies({"A": 1, "B": 2})
    add_coder_case(col, "aec", dict(group="G7empty", model="fixed", freq=[1, 2], K=2, k=0, max_total=p.MAX_ALLOWED_TOTAL_FREQ,
                                    precision=32, size_bits=32), ["A", "B"], [],
                   lambda: ArithmeticEncoder(p, FixedFreqModel(fr_init, p.MAX_ALLOWED_TOTAL_FREQ)), None, rng, decode=False)

    # G8: order-k on the reference's 2nd-order Markov source, k = 0..3; k = 0 equals adaptive iid
    x = markov2_ref(10000)
    p = AECParams()
    alphabet = [0, 1, 2]
    streams = {}
    for k in (0, 1, 2, 3):
        mk = lambda: AdaptiveOrderKFreqModel(alphabet, k, p.MAX_ALLOWED_TOTAL_FREQ)
        streams[k] = add_coder_case(col, "aec", dict(group="G8", model="orderk", freq=[1, 1, 1], K=3, k=k,
                                                     max_total=p.MAX_ALLOWED_TOTAL_FREQ, precision=32, size_bits=32),
                                    alphabet, x, lambda: ArithmeticEncoder(p, mk()), lambda: ArithmeticDecoder(p, mk()), rng)
    iid = AdaptiveIIDFreqModel(Frequencies({0: 1, 1: 1, 2: 1}), p.MAX_ALLOWED_TOTAL_FREQ)
    assert streams[0] == ArithmeticEncoder(p, iid).encode_block(DataBlock(x.tolist()))
    # G8b: order-1 on Markov-1 chunks (BASELINE.json configs[3]) for K = 4, 16, 256
    for K, n in ((4, 4096), (16, 4096), (256, 2048)):
        x = markov1(K, n)
        alphabet = list(range(K))
        mk = lambda: AdaptiveOrderKFreqModel(alphabet, 1, p.MAX_ALLOWED_TOTAL_FREQ)
        add_coder_case(col, "aec", dict(group="G8b", model="orderk", freq=[1] * K, K=K, k=1, max_total=p.MAX_ALLOWED_TOTAL_FREQ,
                                        precision=32, size_bits=32), alphabet, x,
                       lambda: ArithmeticEncoder(p, mk()), lambda: ArithmeticDecoder(p, mk()), rng)


# ------------------------------------------------------------------------------------------------
# G9 / G10: the block loop and the on-disk framing, generated by the reference's own
# DataEncoder.encode / encode_file (core/data_encoder_decoder.py:43-86) and EncodedBlockWriter
# (core/encoded_stream.py:137-175).  Every coder object is used for ALL blocks of its stream, so the
# adaptive arithmetic-coder models carry their state from block to block (quirk Q4).
def model_state(freq_model):
    """(counts row-major, past_k) of a reference freq model, for checking the host mirror afterwards"""
    if isinstance(freq_model, AdaptiveOrderKFreqModel):
        return np.asarray(freq_model.freqs_kplus1_tuple, dtype=np.int64).ravel(), list(freq_model.past_k)
    return np.asarray(freq_model.freqs_current.freq_list, dtype=np.int64), []


class _Sink:
    """output stream for DataDecoder.decode: the reference's ListDataStream.write_symbol never advances its
    position, so it cannot collect more than one symbol"""

    def __init__(self):
        self.data = []

    def write_block(self, block):
        self.data.extend(block.data_list)


def add_stream_case(col, kind, meta, alphabet, idx_syms, block_size, make_encoder, make_decoder, tmp):
    data = [alphabet[i] for i in idx_syms]
    path = os.path.join(tmp, "enc.bin")
    encoder = make_encoder()
    with EncodedBlockWriter(path) as writer:
        encoder.encode(ListDataStream(data), block_size=block_size, encode_writer=writer)
    file_bytes = np.fromfile(path, dtype=np.uint8)
    # the same blocks once more through encode_block on a second object: per-block streams
    enc2 = make_encoder()
    block_bits = [enc2.encode_block(DataBlock(data[i:i + block_size])) for i in range(0, len(data), block_size)]
    with EncodedBlockReader(path) as reader:
        for bits in block_bits:
            assert reader.get_block() == bits
        assert reader.get_block() is None
    # decode with ONE decoder object through the reference's loop
    decoder = make_decoder()
    sink = _Sink()
    with EncodedBlockReader(path) as reader:
        decoder.decode(reader, sink)
    assert sink.data == data, "reference stream round trip failed"
    arrays = dict(sym=np.asarray(idx_syms, dtype=np.uint8), file=file_bytes,
                  block_nbits=np.asarray([len(b) for b in block_bits], dtype=np.int64),
                  block_out=np.concatenate([pack(b)[0] for b in block_bits]) if block_bits else np.zeros(0, np.uint8))
    meta = dict(meta, kind=kind, n=len(idx_syms), block_size=block_size)
    if kind == "aec":
        for tag, obj in (("enc", encoder), ("dec", decoder)):
            counts, past = model_state(obj.freq_model)
            arrays[f"{tag}_counts"] = counts
            arrays[f"{tag}_past_k"] = np.asarray(past, dtype=np.int64)
    col.add(meta, **arrays)


def gen_stream(col, rng):
    tmp = tempfile.mkdtemp()
    # G10: EncodedBlockWriter on bare bit strings of awkward lengths (incl. empty and byte multiples)
    lens = [0, 1, 5, 7, 8, 13, 16, 61, 64, 1000, 3, 2045]
    blocks = [BitArray("".join(str(int(b)) for b in rng.integers(0, 2, n))) for n in lens]
    path = os.path.join(tmp, "framing.bin")
    with EncodedBlockWriter(path) as writer:
        for b in blocks:
            writer.write_block(b)
    with EncodedBlockReader(path) as reader:
        for b in blocks:
            assert reader.get_block() == b
    col.add(dict(kind="framing", group="G10"), file=np.fromfile(path, dtype=np.uint8),
            block_nbits=np.asarray(lens, dtype=np.int64), block_out=np.concatenate([pack(b)[0] for b in blocks]))

    # G9: three-block streams (last block partial) through encode()/decode()
    fl = [34, 35, 546, 1, 13, 245]
    alphabet = list("ABCDEF")
    fr = Frequencies(dict(zip(alphabet, fl)))
    syms = iid_indices(fl, 2500, seed=9)
    rp = rANSParams(fr)
    add_stream_case(col, "rans", dict(group="G9", freq=fl, RF=rp.RANGE_FACTOR, b=1, size_bits=32), alphabet, syms, 1000,
                    lambda: rANSEncoder(rp), lambda: rANSDecoder(rp), tmp)
    fl2 = [3, 4, 9]
    fr2 = Frequencies(dict(zip("ABC", fl2)))
    tp = tANSParams(fr2, RANGE_FACTOR=1 << 4)
    add_stream_case(col, "tans", dict(group="G9", freq=fl2, RF=tp.RANGE_FACTOR, size_bits=32), list("ABC"),
                    iid_indices(fl2, 2500, seed=9), 1000, lambda: tANSEncoder(tp), lambda: tANSDecoder(tp), tmp)
    gp = RangeCoderParams()
    add_stream_case(col, "range", dict(group="G9", freq=fl, precision=32, size_bits=32), alphabet, syms, 1000,
                    lambda: RangeEncoder(gp, fr), lambda: RangeDecoder(gp, fr), tmp)
    # arithmetic coder: the model object lives across the blocks
    p = AECParams()
    mt = p.MAX_ALLOWED_TOTAL_FREQ
    fixed = dict(group="G9", model="fixed", freq=fl, K=6, k=0, max_total=mt, precision=32, size_bits=32)
    add_stream_case(col, "aec", fixed, alphabet, syms, 1000, lambda: ArithmeticEncoder(p, FixedFreqModel(fr, mt)),
                    lambda: ArithmeticDecoder(p, FixedFreqModel(fr, mt)), tmp)
    for init in (fl, [1] * 6):
        fr_init = Frequencies(dict(zip(alphabet, init)))
        add_stream_case(col, "aec", dict(fixed, model="iid", freq=init), alphabet, syms, 1000,
                        lambda: ArithmeticEncoder(p, AdaptiveIIDFreqModel(fr_init, mt)),
                        lambda: ArithmeticDecoder(p, AdaptiveIIDFreqModel(fr_init, mt)), tmp)
    # halving rule reached in the second block (PRECISION = 16: cap 2^14)
    p16 = AECParams(PRECISION=16)
    fr4 = Frequencies(dict(zip("ABCD", [1] * 4)))
    add_stream_case(col, "aec", dict(group="G9halve", model="iid", freq=[1] * 4, K=4, k=0, max_total=p16.MAX_ALLOWED_TOTAL_FREQ,
                                     precision=16, size_bits=32), list("ABCD"), iid_indices([3, 1, 7, 2], 36000, seed=5),
                    12000, lambda: ArithmeticEncoder(p16, AdaptiveIIDFreqModel(fr4, p16.MAX_ALLOWED_TOTAL_FREQ)),
                    lambda: ArithmeticDecoder(p16, AdaptiveIIDFreqModel(fr4, p16.MAX_ALLOWED_TOTAL_FREQ)), tmp)
    # order-k: counts AND the context (past_k) cross the block boundary
    for K, k, n, bs in ((4, 1, 2500, 1000), (16, 1, 2500, 1000), (3, 2, 2500, 1000), (256, 1, 1200, 500), (3, 0, 700, 300)):
        x = markov2_ref(n) if K == 3 else markov1(K, n)
        alph = list(range(K))
        add_stream_case(col, "aec", dict(group="G9", model="orderk", freq=[1] * K, K=K, k=k, max_total=mt, precision=32,
                                         size_bits=32), alph, x, bs,
                        lambda: ArithmeticEncoder(p, AdaptiveOrderKFreqModel(alph, k, mt)),
                        lambda: ArithmeticDecoder(p, AdaptiveOrderKFreqModel(alph, k, mt)), tmp)

    # G9file: encode_file / decode_file on a text file (TextFileDataStream, one character per symbol)
    text = "".join(np.random.default_rng(11).choice(list("ab cd\ne"), size=1700, p=[.3, .2, .2, .1, .1, .05, .05]))
    src, dst, back = (os.path.join(tmp, n) for n in ("in.txt", "out.bin", "back.txt"))
    with open(src, "w") as f:
        f.write(text)
    chars = sorted(set(text))
    for name in ("rans", "aec_iid", "aec_order1"):
        if name == "rans":
            frt = Frequencies({c: text.count(c) for c in chars})
            params = rANSParams(frt)
            enc, dec = rANSEncoder(params), rANSDecoder(params)
            meta = dict(coder="rans", freq=[text.count(c) for c in chars], RF=params.RANGE_FACTOR, b=1, size_bits=32)
        elif name == "aec_iid":
            fri = Frequencies({c: 1 for c in chars})
            enc = ArithmeticEncoder(p, AdaptiveIIDFreqModel(fri, mt))
            dec = ArithmeticDecoder(p, AdaptiveIIDFreqModel(fri, mt))
            meta = dict(coder="aec", model="iid", freq=[1] * len(chars), K=len(chars), k=0, max_total=mt, precision=32, size_bits=32)
        else:
            enc = ArithmeticEncoder(p, AdaptiveOrderKFreqModel(chars, 1, mt))
            dec = ArithmeticDecoder(p, AdaptiveOrderKFreqModel(chars, 1, mt))
            meta = dict(coder="aec", model="orderk", freq=[1] * len(chars), K=len(chars), k=1, max_total=mt, precision=32, size_bits=32)
        enc.encode_file(src, dst, block_size=600)
        dec.decode_file(dst, back)
        assert open(back).read() == text
        col.add(dict(meta, kind="file", group="G9file", block_size=600, alphabet="".join(chars)),
                text=np.frombuffer(text.encode("ascii"), dtype=np.uint8), file=np.fromfile(dst, dtype=np.uint8))


# ------------------------------------------------------------------------------------------------
# G11: alphabets of more than 256 symbols (golden_wide.npz; symbol indices stored as uint16).  The reference codes any
# hashable alphabet; these pin the uint16 entry points (the *_u16 twins of include/scl_hip.h, the *_w16 oracle functions).
def gen_wide(col, rng):
    def table(K, lo, hi, seed):
        return [int(v) for v in np.random.default_rng(seed).integers(lo, hi, K)]

    # rANS: K = 300 with the default parameters, K = 1000 with NUM_BITS_OUT = 2 and a small RANGE_FACTOR
    for K, kw, n in ((300, {}, 600), (1000, dict(NUM_BITS_OUT=2, RANGE_FACTOR=1 << 4), 700)):
        fl = table(K, 1, 30, K)
        alphabet = list(range(K))
        fr = Frequencies(dict(zip(alphabet, fl)))
        p = rANSParams(fr, **kw)
        syms = np.random.default_rng(K + 1).integers(0, K, n)
        add_coder_case(col, "rans", dict(group="G11", freq=fl, RF=p.RANGE_FACTOR, b=p.NUM_BITS_OUT,
                                         size_bits=p.DATA_BLOCK_SIZE_BITS), alphabet, syms,
                       lambda: rANSEncoder(p), lambda: rANSDecoder(p), rng)
    # tANS: K = 512, M = 1024 (a power of two), RANGE_FACTOR = 2; the stream equals rANS with the same parameters
    K = 512
    fl = [1] * K
    for i in np.random.default_rng(7).integers(0, K, 512):
        fl[int(i)] += 1
    alphabet = list(range(K))
    fr = Frequencies(dict(zip(alphabet, fl)))
    tp = tANSParams(fr, RANGE_FACTOR=2)
    syms = iid_indices(fl, 600, seed=8).astype(np.int64)
    bits = add_coder_case(col, "tans", dict(group="G11", freq=fl, RF=2, size_bits=tp.DATA_BLOCK_SIZE_BITS), alphabet,
                          syms, lambda: tANSEncoder(tp), lambda: tANSDecoder(tp), rng)
    assert bits == rANSEncoder(rANSParams(fr, RANGE_FACTOR=2)).encode_block(DataBlock([int(v) for v in syms]))
    # range coder: K = 700
    K = 700
    fl = table(K, 1, 40, 9)
    alphabet = list(range(K))
    fr = Frequencies(dict(zip(alphabet, fl)))
    rp = RangeCoderParams()
    syms = np.random.default_rng(10).integers(0, K, 600)
    add_coder_case(col, "range", dict(group="G11", freq=fl, precision=rp.PRECISION, size_bits=rp.DATA_BLOCK_SIZE_BITS),
                   alphabet, syms, lambda: RangeEncoder(rp, fr), lambda: RangeDecoder(rp, fr), rng)
    # arithmetic coder: fixed / adaptive i.i.d. / order-1 on K = 300
    K = 300
    fl = table(K, 1, 25, 11)
    alphabet = list(range(K))
    ap = AECParams()
    syms = np.random.default_rng(12).integers(0, K, 400)
    fr = Frequencies(dict(zip(alphabet, fl)))
    ones = Frequencies(dict(zip(alphabet, [1] * K)))
    models = (("fixed", fl, 0, lambda: FixedFreqModel(fr, ap.MAX_ALLOWED_TOTAL_FREQ)),
              ("iid", [1] * K, 0, lambda: AdaptiveIIDFreqModel(ones, ap.MAX_ALLOWED_TOTAL_FREQ)),
              ("orderk", [1] * K, 1, lambda: AdaptiveOrderKFreqModel(alphabet, 1, ap.MAX_ALLOWED_TOTAL_FREQ)))
    for model_name, init, k, mk in models:
        add_coder_case(col, "aec", dict(group="G11", model=model_name, freq=init, K=K, k=k,
                                        max_total=ap.MAX_ALLOWED_TOTAL_FREQ, precision=ap.PRECISION,
                                        size_bits=ap.DATA_BLOCK_SIZE_BITS), alphabet, syms,
                       lambda: ArithmeticEncoder(ap, mk()), lambda: ArithmeticDecoder(ap, mk()), rng)


def gen_counts(col, rng):
    """G12 (row f3): ``DataBlock.get_counts`` / ``get_empirical_distribution`` of the reference (core/data_block.py:37-94)
    on byte blocks and on blocks over wider alphabets: the symbols that occur (sorted) and their counts, plus the
    probabilities the reference derives from them.  What the device histograms (scl_histogram_u8 / _u16) must equal."""
    cases = [("bytes_empty", 256, 0), ("bytes_1", 256, 1), ("bytes_17", 256, 17), ("bytes_4096", 256, 4096),
             ("bytes_hot", 256, 100003), ("bytes_few", 7, 5000), ("u16_1000", 1000, 4097), ("u16_65536", 65536, 30001)]
    for name, K, n in cases:
        data = rng.integers(0, K, n)
        if name == "bytes_hot":
            data[: n // 2] = 7  # one symbol takes half the block
        block = DataBlock(data.tolist())
        counts = block.get_counts() if n else {}
        syms = np.array(sorted(counts), dtype=np.int64)
        cnt = np.array([counts[int(s)] for s in syms], dtype=np.int64)
        probs = np.zeros(0)
        if n:
            pd = block.get_empirical_distribution().prob_dict
            probs = np.array([pd[int(s)] for s in syms], dtype=np.float64)
        assert int(cnt.sum()) == n
        col.add(dict(kind="counts", group="G12", name=name, K=K, n=n),
                data=data.astype(np.uint16 if K > 256 else np.uint8), symbols=syms, counts=cnt, probs=probs)


def main(out_dir):
    for name, fn in (("rans", gen_rans), ("tans", gen_tans), ("range", gen_range), ("aec", gen_aec),
                     ("stream", gen_stream), ("wide", gen_wide), ("counts", gen_counts)):
        col = Collector()
        fn(col, np.random.default_rng(12345))
        col.save(f"{out_dir}/golden_{name}.npz")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/repo/tests/golden")
