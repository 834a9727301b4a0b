"""Pure-Python, per-symbol restatements of the reference's four coders -- the "reference-style" CPU baseline.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by bench.py's ``cpu_baseline`` leg,
never by the package's product path.

Why it exists: the reference is pure Python and cannot travel to the GPU box, and the C oracle (scl_oracle.c) is a far
stronger CPU baseline than anything a user of the reference ever ran.  This file restates the reference's coders in
their own ALGORITHMIC SHAPE -- one Python-level step per symbol, the cumulative table rebuilt from the frequency dict
on every access, every released bit group turned into a bit string and PREPENDED (rANS / tANS) or appended (range /
arithmetic coder) to the growing stream, the decoders re-slicing their input per symbol and searching the cumulative
counts with numpy per symbol, the adaptive models updated through the host model objects -- so that its speed relates
to the reference's by a measured ratio instead of by guesswork.  Written against this
package's own ``Frequencies`` / ``BitArray`` / ``FreqModel`` host classes; it shares no code with the reference.

  rans_encode_block  <->  rANSEncoder.encode_block     scl/compressors/rANS.py:186-210 (shrink_state :149-161,
                                                        rans_base_encode_step :138-147)
  rans_decode_block  <->  rANSDecoder.decode_block     scl/compressors/rANS.py:270-297 (rans_base_decode_step :234-249,
                                                        expand_state :251-260)
  TansSetup          <->  tANSEncoder / tANSDecoder table builders   scl/compressors/tANS.py:74-110, :214-226
  tans_encode_block  <->  tANSEncoder.encode_block     scl/compressors/tANS.py:126-193
  tans_decode_block  <->  tANSDecoder.decode_block     scl/compressors/tANS.py:239-279
  range_encode_block <->  RangeEncoder.encode_block    scl/compressors/range_coder.py:88-207
  range_decode_block <->  RangeDecoder.decode_block    scl/compressors/range_coder.py:225-317
  aec_encode_block   <->  ArithmeticEncoder.encode_block   scl/compressors/arithmetic_coding.py:80-161
  aec_decode_block   <->  ArithmeticDecoder.decode_block   scl/compressors/arithmetic_coding.py:203-287
                          (models: scl/compressors/probability_models.py:57-160 through the package's host mirrors)
Bit-exact against the reference-generated goldens (tests/test_oracle_goldens.py::test_restatement_*).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from stanford_compression_library_amd.core.prob_dist import Frequencies  # noqa: E402
from stanford_compression_library_amd.utils.bitarray_utils import (BitArray, bitarray_to_uint,  # noqa: E402
                                                                    get_bit_width, uint_to_bitarray)


class RansSetup:
    """derived constants of rANSParams (rANS.py:78-120)"""

    def __init__(self, freqs: Frequencies, size_bits=32, num_bits_out=1, range_factor=1 << 16):
        self.freqs = freqs
        self.size_bits, self.b, self.RF = size_bits, num_bits_out, range_factor
        self.M = freqs.total_freq
        self.L = self.RF * self.M
        self.H = self.L * (1 << self.b) - 1
        self.max_shrunk = {s: self.RF * f * (1 << self.b) - 1 for s, f in freqs.freq_dict.items()}
        self.nsb = get_bit_width(self.H)


def rans_encode_block(p: RansSetup, symbols) -> BitArray:
    state = p.L
    stream = BitArray("")
    for s in symbols:
        # shrink_state: release NUM_BITS_OUT bits at a time until the state fits the symbol's interval
        out = BitArray("")
        while state > p.max_shrunk[s]:
            out = uint_to_bitarray(state % (1 << p.b), bit_width=p.b) + out
            state >>= p.b
        # rans_base_encode_step, with the cumulative table rebuilt per symbol like the reference's property
        f = p.freqs.frequency(s)
        c = p.freqs.cumulative_freq_dict[s]
        state = (state // f) * p.M + c + state % f
        stream = out + stream  # the later symbol's bits go in front
    stream = uint_to_bitarray(state, bit_width=p.nsb) + stream
    return uint_to_bitarray(len(symbols), bit_width=p.size_bits) + stream


def rans_decode_block(p: RansSetup, bits: BitArray):
    n = bitarray_to_uint(bits[: p.size_bits])
    used = p.size_bits
    state = bitarray_to_uint(bits[used: used + p.nsb])
    used += p.nsb
    alphabet = p.freqs.alphabet
    out = []
    for _ in range(n):
        rest = bits[used:]  # the reference re-slices its input for every symbol
        # rans_base_decode_step: slot -> symbol by a search over the cumulative counts, rebuilt per symbol
        cum = list(p.freqs.cumulative_freq_dict.values())
        slot = state % p.M
        idx = int(np.searchsorted(cum, slot, side="right")) - 1
        s = alphabet[idx]
        state = (state // p.M) * p.freqs.frequency(s) + slot - cum[idx]
        # expand_state
        k = 0
        while state < p.L:
            state = (state << p.b) + bitarray_to_uint(rest[k: k + p.b])
            k += p.b
        used += k
        out = [s] + out
    assert state == p.L
    return out, used


# ---- tANS: cached rANS (tANS.py) --------------------------------------------------------------------------
class TansSetup(RansSetup):
    """the five lookup tables, built the way the reference builds them: Python dicts filled by loops over the state
    range (tANS.py:74-110 encoder side, :214-226 decoder side); NUM_BITS_OUT is 1 by definition (tANS.py:39-47)"""

    def __init__(self, freqs: Frequencies, size_bits=32, range_factor=1 << 16):
        super().__init__(freqs, size_bits, 1, range_factor)
        assert self.M & (self.M - 1) == 0, "tANS needs a power-of-two total"
        cum = freqs.cumulative_freq_dict
        self.min_shrunk = {s: self.RF * f for s, f in freqs.freq_dict.items()}
        # encoder: (symbol, shrunk state) -> next state; bits released = base[s] (+1 from thresh[s] upwards)
        self.enc_step, self.nbits_base, self.thresh = {}, {}, {}
        for s, f in freqs.freq_dict.items():
            for xs in range(self.min_shrunk[s], self.max_shrunk[s] + 1):
                self.enc_step[(s, xs)] = (xs // f) * self.M + cum[s] + xs % f
            base = self.nsb - get_bit_width(self.max_shrunk[s])
            self.nbits_base[s] = base
            self.thresh[s] = (self.max_shrunk[s] + 1) << base
        # decoder: state -> (symbol, shrunk state); shrunk state -> bits to read back
        alphabet, cum_list = freqs.alphabet, list(cum.values())
        self.dec_step, self.expand_bits = {}, {}
        for x in range(self.L, self.H + 1):
            slot = x % self.M
            idx = int(np.searchsorted(cum_list, slot, side="right")) - 1
            s = alphabet[idx]
            self.dec_step[x] = (s, (x // self.M) * freqs.frequency(s) + slot - cum_list[idx])
        for s in alphabet:
            for xs in range(self.min_shrunk[s], self.max_shrunk[s] + 1):
                self.expand_bits[xs] = self.nsb - get_bit_width(xs)


def tans_encode_block(p: TansSetup, symbols) -> BitArray:
    state = p.L
    stream = BitArray("")
    for s in symbols:
        k = p.nbits_base[s] + (1 if state >= p.thresh[s] else 0)
        out = uint_to_bitarray(state)[-k:] if k else BitArray("")  # the low k bits of the state (quirk Q9)
        state = p.enc_step[(s, state >> k)]
        stream = out + stream
    stream = uint_to_bitarray(state, bit_width=p.nsb) + stream
    return uint_to_bitarray(len(symbols), bit_width=p.size_bits) + stream


def tans_decode_block(p: TansSetup, bits: BitArray):
    n = bitarray_to_uint(bits[: p.size_bits])
    used = p.size_bits
    state = bitarray_to_uint(bits[used: used + p.nsb])
    used += p.nsb
    out = []
    for _ in range(n):
        rest = bits[used:]  # re-sliced per symbol, as in the reference
        s, xs = p.dec_step[state]
        k = p.expand_bits[xs]
        state = (xs << k) + (bitarray_to_uint(rest[:k]) if k else 0)
        used += k
        out = [s] + out
    assert state == p.L
    return out, used


# ---- range coder (range_coder.py) -------------------------------------------------------------------------
class RangeSetup:
    """RangeCoderParams (range_coder.py:52-76) + the model"""

    def __init__(self, freqs: Frequencies, precision=32, size_bits=32):
        assert precision % 8 == 0
        self.freqs, self.precision, self.size_bits = freqs, precision, size_bits
        self.top, self.bottom = 1 << (precision - 8), 1 << (precision - 16)
        self.mask = (1 << precision) - 1
        assert min(freqs.freq_dict.values()) > 0 and freqs.total_freq <= self.bottom


def _range_shrink(freqs: Frequencies, s, low, rng):
    # every derived view is recomputed on access, like the reference's properties (range_coder.py:88-107)
    c = freqs.cumulative_freq_dict[s]
    d = c + freqs.frequency(s)
    rng = rng // freqs.total_freq
    return low + c * rng, rng * (d - c)


def range_encode_block(p: RangeSetup, symbols) -> BitArray:
    low, rng = 0, p.mask
    bits = uint_to_bitarray(len(symbols), p.size_bits)
    for s in symbols:
        low, rng = _range_shrink(p.freqs, s, low, rng)
        # normalize (range_coder.py:109-179): release the top byte while it is settled; a range below BOTTOM that still
        # straddles a byte boundary is cut back to the boundary first (carry-less)
        while (low ^ (low + rng)) < p.top or rng < p.bottom:
            if (low ^ (low + rng)) >= p.top:
                rng = (p.mask + 1 - low) & (p.bottom - 1)
            bits.frombytes(bytes([low >> (p.precision - 8)]))
            low = (low << 8) & p.mask
            rng <<= 8
    for _ in range(p.precision // 8):  # flush (range_coder.py:181-186)
        bits.frombytes(bytes([low >> (p.precision - 8)]))
        low = (low << 8) & p.mask
    return bits


def range_decode_block(p: RangeSetup, bits: BitArray):
    n = bitarray_to_uint(bits[: p.size_bits])
    body = bits[p.size_bits:]
    used = 0
    low, rng, state = 0, p.mask, 0
    for _ in range(p.precision // 8):
        state = (state << 8) | bitarray_to_uint(body[used: used + 8])
        used += 8
    out = []
    alphabet = p.freqs.alphabet
    while len(out) < n:
        # decode_symbol (range_coder.py:233-247): a vector of all interval starts, searched with numpy, per symbol
        starts = low + np.array(list(p.freqs.cumulative_freq_dict.values())) * (rng // p.freqs.total_freq)
        s = alphabet[int(np.searchsorted(starts, state, side="right")) - 1]
        out.append(s)
        low, rng = _range_shrink(p.freqs, s, low, rng)
        while (low ^ (low + rng)) < p.top or rng < p.bottom:
            if (low ^ (low + rng)) >= p.top:
                rng = (p.mask + 1 - low) & (p.bottom - 1)
            state = ((state << 8) | bitarray_to_uint(body[used: used + 8])) & p.mask
            used += 8
            low = (low << 8) & p.mask
            rng <<= 8
    return out, used + p.size_bits


# ---- arithmetic coder (arithmetic_coding.py) + the three models ----------------------------------------------
class AecSetup:
    """AECParams (arithmetic_coding.py:20-45).  The reference's `1 << MAX_BLOCK_SIZE` assert (quirk Q3: a 512 MiB integer
    per call) is not restated -- BASELINE.md excludes it from the reference's own timings as well."""

    def __init__(self, precision=32, size_bits=32):
        self.precision, self.size_bits = precision, size_bits
        self.full, self.half, self.qtr = 1 << precision, 1 << (precision - 1), 1 << (precision - 2)
        self.max_total = self.qtr


def _aec_shrink(freqs: Frequencies, s, low, high):
    rng = high - low
    c = freqs.cumulative_freq_dict[s]
    d = c + freqs.frequency(s)
    return low + (rng * c) // freqs.total_freq, low + (rng * d) // freqs.total_freq


def aec_encode_block(p: AecSetup, model, symbols) -> BitArray:
    """`model`: a host model object (compressors/probability_models.py); it is advanced like the reference's
    (shrink, THEN update, THEN renormalise: quirk Q2)"""
    low, high = 0, p.full
    bits = uint_to_bitarray(len(symbols), p.size_bits)
    pending = 0
    for s in symbols:
        assert model.freqs_current.total_freq < p.max_total
        low, high = _aec_shrink(model.freqs_current, s, low, high)
        model.update_model(s)
        while high < p.half or low > p.half:  # strict comparisons (quirk Q1)
            if high < p.half:
                bits.extend("0" + "1" * pending)
                low, high = low << 1, high << 1
            else:
                bits.extend("1" + "0" * pending)
                low, high = (low - p.half) << 1, (high - p.half) << 1
            pending = 0
        while low > p.qtr and high < 3 * p.qtr:
            pending += 1
            low, high = (low - p.qtr) << 1, (high - p.qtr) << 1
    pending += 1
    bits.extend(("0" + "1" * pending) if low <= p.qtr else ("1" + "0" * pending))
    return bits


def aec_decode_block(p: AecSetup, model, bits: BitArray):
    n = bitarray_to_uint(bits[: p.size_bits])
    body = bits[p.size_bits:]
    size = len(body)
    low, high, state = 0, p.full, 0
    used = 0
    while used < p.precision and used < size:
        if body[used]:
            state += 1 << (p.precision - used - 1)
        used += 1
    used = p.precision
    out = []
    while True:  # (an empty block decodes one symbol before it looks at the count: quirk Q5 -- callers pass n >= 1)
        freqs = model.freqs_current
        rng = high - low
        starts = low + (np.array(list(freqs.cumulative_freq_dict.values())) * rng) // freqs.total_freq
        s = freqs.alphabet[int(np.searchsorted(starts, state, side="right")) - 1]
        low, high = _aec_shrink(model.freqs_current, s, low, high)
        out.append(s)
        model.update_model(s)
        if len(out) == n:
            break
        while high < p.half or low > p.half:
            if high < p.half:
                low, high, state = low << 1, high << 1, state << 1
            else:
                low, high, state = (low - p.half) << 1, (high - p.half) << 1, (state - p.half) << 1
            if used < size:
                state += body[used]
            used += 1
        while low > p.qtr and high < 3 * p.qtr:
            low, high, state = (low - p.qtr) << 1, (high - p.qtr) << 1, (state - p.qtr) << 1
            if used < size:
                state += body[used]
            used += 1
    # bits the decoder read ahead of what the encoder wrote (arithmetic_coding.py:263-282)
    extra = 0
    for extra in range(p.precision):
        lo = (state >> extra) << extra
        if lo < low or lo + (1 << extra) > high:
            break
    return out, used - (extra - 1) + p.size_bits


class _NumpyOrderK:
    """the package's order-k host model, handing out its context row the way the reference does: a fresh Frequencies
    per access whose counts are numpy integers (probability_models.py:134-142) -- the package's own mirror converts the
    row to Python ints, which makes every product and division of the coder ~1.7x cheaper than the reference's"""

    def __init__(self, model):
        self.m = model

    @property
    def freqs_current(self):
        m = self.m
        row = m.freqs_kplus1_tuple[tuple(m.past_k)] if m.k > 0 else m.freqs_kplus1_tuple
        return Frequencies(dict(zip(m.alphabet, np.ravel(row))))

    def update_model(self, s):
        self.m.update_model(s)


# ---- timed baseline (bench.py) --------------------------------------------------------------------------
def make_codec(spec: dict):
    """spec -> (encode(symbol list) -> BitArray, decode(BitArray) -> (symbol list, bits used)).
    spec: coder = rans | tans | range | aec; freq (static models); range_factor, num_bits_out (rANS / tANS);
    model = fixed | iid | orderk, K, k (arithmetic coder).  Adaptive models start fresh for every chunk, which is what
    the batched device entry points do (one chunk = one new coder)."""
    coder = spec["coder"]
    if coder in ("rans", "tans", "range") or spec.get("model") in ("fixed", "iid"):
        freqs = Frequencies(dict(enumerate(int(f) for f in spec["freq"])))
    if coder == "rans":
        p = RansSetup(freqs, 32, spec.get("num_bits_out", 1), spec.get("range_factor", 1 << 16))
        return (lambda row: rans_encode_block(p, row)), (lambda bits: rans_decode_block(p, bits))
    if coder == "tans":
        p = TansSetup(freqs, 32, spec.get("range_factor", 1))
        return (lambda row: tans_encode_block(p, row)), (lambda bits: tans_decode_block(p, bits))
    if coder == "range":
        p = RangeSetup(freqs, 32, 32)
        return (lambda row: range_encode_block(p, row)), (lambda bits: range_decode_block(p, bits))
    assert coder == "aec"
    from stanford_compression_library_amd.compressors.probability_models import (AdaptiveIIDFreqModel,
                                                                                 AdaptiveOrderKFreqModel, FixedFreqModel)
    p = AecSetup(32, 32)

    def fresh():
        if spec["model"] == "fixed":
            return FixedFreqModel(freqs, p.max_total)
        if spec["model"] == "iid":
            return AdaptiveIIDFreqModel(freqs, p.max_total)
        return _NumpyOrderK(AdaptiveOrderKFreqModel(list(range(int(spec["K"]))), int(spec.get("k", 1)), p.max_total))

    return (lambda row: aec_encode_block(p, fresh(), row)), (lambda bits: aec_decode_block(p, fresh(), bits))


def _worker(args):
    spec, rows = args
    enc, dec = make_codec(spec)
    t0 = time.perf_counter()
    streams = [enc(row) for row in rows]
    t1 = time.perf_counter()
    ok = True
    for row, st in zip(rows, streams):
        back, used = dec(st)
        ok = ok and back == row and used == len(st)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, ok, [(len(st), bytes(st.tobytes())) for st in streams]


def timed_baseline(spec: dict, sym2d: np.ndarray, workers=None, chunks_per_worker=4):
    """encode + decode ``chunks_per_worker`` chunks per worker process on ``workers`` host cores (default: all).
    Returns per-core and aggregate MB/s and the produced streams (for a parity check against the GPU).  Table
    construction (tANS) happens inside the workers, outside the timed sections."""
    import multiprocessing as mp

    workers = workers or len(os.sched_getaffinity(0))
    n = min(sym2d.shape[0], workers * chunks_per_worker)
    workers = max(1, n // chunks_per_worker)
    n = workers * chunks_per_worker
    jobs = [(spec, [row.tolist() for row in sym2d[w * chunks_per_worker:(w + 1) * chunks_per_worker]])
            for w in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_worker, jobs)
    wall = time.perf_counter() - t0
    nbytes = n * sym2d.shape[1]
    per_chunk_bytes = chunks_per_worker * sym2d.shape[1]
    enc_core = float(np.mean([per_chunk_bytes / r[0] for r in res])) / 1e6
    dec_core = float(np.mean([per_chunk_bytes / r[1] for r in res])) / 1e6
    rt_core = float(np.mean([per_chunk_bytes / (r[0] + r[1]) for r in res])) / 1e6
    # aggregate = what the cores deliver together while all of them run (sum of the per-worker rates): the pool's start-up
    # and the (untimed) table construction are not the coders' time
    rt_sum = float(np.sum([per_chunk_bytes / (r[0] + r[1]) for r in res])) / 1e6
    streams = [s for r in res for s in r[3]]
    return dict(ok=all(r[2] for r in res), workers=workers, chunks=n, bytes=nbytes, wall_s=wall,
                encode_MBps_per_core=enc_core, decode_MBps_per_core=dec_core, round_trip_MBps_per_core=rt_core,
                round_trip_MBps_aggregate=rt_sum, round_trip_MBps_wall=nbytes / wall / 1e6, streams=streams)
