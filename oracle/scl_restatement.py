"""Pure-Python, per-symbol restatement of the reference's rANS coder -- the "reference-style" CPU baseline.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by bench.py's ``cpu_baseline`` leg,
never by the package's product path.

Why it exists: the reference is pure Python and cannot travel to the GPU box, and the C oracle (scl_oracle.c) is a far
stronger CPU baseline than anything a user of the reference ever ran.  This file restates the reference's rANS
encoder / decoder in its own ALGORITHMIC SHAPE -- one Python-level step per symbol, the cumulative table rebuilt from
the frequency dict on every step, every released bit group turned into a bit string and PREPENDED to the growing
stream, the decoder re-slicing its input per symbol -- so that its speed relates to the reference's by a measured
ratio instead of by guesswork.  Written against this package's own ``Frequencies`` /
``BitArray`` host classes; it shares no code with the reference.

  rans_encode_block  <->  rANSEncoder.encode_block   scl/compressors/rANS.py:186-210 (shrink_state :149-161,
                                                      rans_base_encode_step :138-147)
  rans_decode_block  <->  rANSDecoder.decode_block   scl/compressors/rANS.py:270-297 (rans_base_decode_step :234-249,
                                                      expand_state :251-260)
Bit-exact against the reference-generated goldens (tests/test_oracle_goldens.py::test_restatement_rans).
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


# ---- timed baseline (bench.py) --------------------------------------------------------------------------
def _worker(args):
    freq_list, rows, rf, b = args
    p = RansSetup(Frequencies(dict(enumerate(freq_list))), 32, b, rf)
    t0 = time.perf_counter()
    streams = [rans_encode_block(p, row) for row in rows]
    t1 = time.perf_counter()
    ok = True
    for row, st in zip(rows, streams):
        back, used = rans_decode_block(p, st)
        ok = ok and back == row and used == len(st)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, ok, [(len(st), bytes(st.tobytes())) for st in streams]


def timed_baseline(freq, sym2d: np.ndarray, range_factor=1 << 16, num_bits_out=1, workers=None, chunks_per_worker=8):
    """encode + decode ``chunks_per_worker`` chunks per worker process on ``workers`` host cores (default: all).
    Returns per-core and aggregate MB/s and the produced streams (for a parity check against the GPU)."""
    import multiprocessing as mp

    workers = workers or len(os.sched_getaffinity(0))
    n = min(sym2d.shape[0], workers * chunks_per_worker)
    workers = max(1, n // chunks_per_worker)
    n = workers * chunks_per_worker
    fl = [int(f) for f in freq]
    jobs = [(fl, [row.tolist() for row in sym2d[w * chunks_per_worker:(w + 1) * chunks_per_worker]], range_factor,
             num_bits_out) for w in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_worker, jobs)
    wall = time.perf_counter() - t0
    nbytes = n * sym2d.shape[1]
    per_chunk_bytes = chunks_per_worker * sym2d.shape[1]
    enc_core = float(np.mean([per_chunk_bytes / r[0] for r in res])) / 1e6
    dec_core = float(np.mean([per_chunk_bytes / r[1] for r in res])) / 1e6
    rt_core = float(np.mean([per_chunk_bytes / (r[0] + r[1]) for r in res])) / 1e6
    streams = [s for r in res for s in r[3]]
    return dict(ok=all(r[2] for r in res), workers=workers, chunks=n, bytes=nbytes, wall_s=wall,
                encode_MBps_per_core=enc_core, decode_MBps_per_core=dec_core, round_trip_MBps_per_core=rt_core,
                round_trip_MBps_aggregate=nbytes / wall / 1e6, streams=streams)
