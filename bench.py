#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X entropy-coding core.

Metric (BASELINE.json): MB/s encode+decode of 1 GiB i.i.d. bytes with 256-symbol static rANS
(reference defaults NUM_BITS_OUT=1, RANGE_FACTOR=2^16, M=4096), plus achieved HBM GB/s vs peak.

One "step" = one full encode pass + one full decode pass over the per-GPU batch
(262 144 chunks x 4 KiB = 1 GiB, one wavefront lane per chunk), inputs resident in HBM.  The slots of the batch are laid
out as `--layout` says: auto = wave-striped slots (C ABI 8) for rANS / tANS batches that fill the chip, linear otherwise;
the same streams at the same logical bit positions either way (`config.slot_layout` names what ran).
value = (bytes of all ranks * steps) / wall time of the timed region / 1e6, i.e. N / (t_enc + t_dec).

    python bench.py                      # 1 GPU, finishes in a few minutes incl. the CPU baselines and other_configs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Warm-up: `--warmup W` untimed steps AND at least `--min-warm-ms` (default 300 ms) of them -- five steps of this workload
are 6 ms, far less than the clocks of a fresh box need to settle (round 3: the same binary read 7-10 % lower behind a
5-step warm-up).  Then EXACTLY `--steps` timed steps between barrier + synchronize on both sides.

Per-kernel figures come from HIP events around every encode and every decode of the timed steps (on the stream the
kernels are launched on): `avg_launch_ms` is their mean (what `roofline.achieved` is computed from, and what the
rocprofv3 summaries under profiles/ report), `median_launch_ms` their median, with `frac_median` beside `frac`.

Multi-GPU: chunks are independent, so every rank encodes/decodes its own 1 GiB shard (weak scaling,
no data-path collective); the only communication is the barrier / max-reduction of the timing.
`--gather` additionally times the optional final gather of the compacted streams to rank 0
(BASELINE.json configs[4]) and reports it separately; it never enters `value`.

The ONE line printed is the COMPACT contract line (< 6 KB: the contract keys, roofline{,_encode,_decode,_dense}, both CPU
baselines, `summary` last -- round 5's 21.6 KB line left the driver's record unparsed); the full record goes to
gpurun_out/bench_full.json (`--full-json`), `--full-line` prints it instead.

The default 1-GPU run (no workload flags) also measures BASELINE.json's other single-GPU configurations and attaches them
as `other_configs` to the full record (their numbers ride in the line's `summary`): configs[1] at its literal size (65 536 chunks), configs[2] (range coder, 1 GiB
of uniform bytes), configs[3] (order-1 adaptive arithmetic coder on a Markov-1 source: K = 16 at 1 GiB, K = 256 at 256 MiB)
-- each with kernel times, `roofline` (frac + PMC traffic where a stamped pass exists) and both CPU baselines.
`stream_file` (same run, static-model coders): the reference-API FILE path end to end on 256 MiB of the batch -- file I/O,
PCIe both ways, kernels, framing (row f2; `summary.headline.file` = [encode, decode] MB/s); reported, never `value`.
"""
import argparse
import copy
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def host_cores():
    """(cores this process may actually use, logical CPUs it may be scheduled on): the affinity mask, cut down to the
    cgroup's CPU quota when the container has one (a 256-thread box that grants 64 CPUs' worth of time runs 256 workers
    slower than 64)"""
    logical = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / p
        except (OSError, ValueError):
            pass
    usable = logical if quota is None else max(1, min(logical, int(quota + 0.5)))
    return usable, logical


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
WORKLOAD_FLAGS = ("chunks", "chunk_len", "table", "coder", "aec_K", "aec_model", "num_bits_out", "range_factor", "source",
                  "sym_pad", "layout")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-warm-ms", type=float, default=300.0,
                    help="warm-up lasts at least this long (and at least --warmup steps); 0 = by count only")
    ap.add_argument("--chunks", type=int, default=262144, help="chunks per GPU")
    ap.add_argument("--chunk-len", type=int, default=4096)
    ap.add_argument("--table", choices=["t256", "uniform", "uniform1"], default="t256",
                    help="t256: Dirichlet table M=4096; uniform: f=16, M=4096; uniform1: f=1, M=256 (configs[2])")
    ap.add_argument("--coder", choices=["rans", "tans", "range", "aec"], default="rans")
    ap.add_argument("--aec-K", type=int, default=16, help="alphabet of the order-1 adaptive arithmetic coder (configs[3])")
    ap.add_argument("--aec-model", choices=["order1", "fixed", "iid"], default="order1",
                    help="arithmetic coder: order-1 adaptive model on a Markov-1 source, FixedFreqModel(--table) on i.i.d. "
                         "symbols, or AdaptiveIIDFreqModel (all-ones start, 256 symbols) on the same i.i.d. symbols")
    ap.add_argument("--num-bits-out", type=int, default=1, help="rANS NUM_BITS_OUT (reference default 1)")
    ap.add_argument("--range-factor", type=int, default=1 << 16, help="rANS RANGE_FACTOR (reference default 2^16)")
    ap.add_argument("--layout", choices=["auto", "linear", "striped"], default="auto",
                    help="slot layout of the rANS / tANS batches: wave-striped slots (ABI 8: four waves per SIMD in the "
                         "encoder, row-by-row stores) for batches that fill the chip, linear slots otherwise (auto); the "
                         "other coders have linear slots only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-pipeline", action="store_true",
                    help="skip the two-sub-batch pipelined dense encode (roofline_dense.pipelined_ms): it launches the encode "
                         "kernel on HALF batches, which a profiler's per-kernel means would mix with the full-size launches")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="headline only (the default 1-GPU run also measures configs[1..3], see the docstring)")
    ap.add_argument("--gather", action="store_true", help="also time compaction + gather to rank 0")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL record (tens of KB: other_configs, stream_file, every roofline field) as the one "
                         "line instead of the compact contract line; the full record always goes to --full-json")
    ap.add_argument("--full-json", default=None,
                    help="where the full record is written (default: gpurun_out/bench_full.json under the repo root)")
    ap.add_argument("--watchdog-s", type=float, default=None,
                    help="a phase that makes no progress for this long ends the process with exit code 3 and names the "
                         "phase (default: 600 s for --gpus > 1, off otherwise)")
    ap.add_argument("--sym-pad", type=int, default=0, help="experiment: extra bytes between input rows")
    ap.add_argument("--source", choices=["iid", "markov1"], default="iid",
                    help="static-model coders: i.i.d. symbols with p = f/M of --table (the headline), or an order-1 Markov "
                         "byte source (north_star's second source) coded with the table its own histogram gives "
                         "(scl_histogram_u8 + normalize_counts, M = 4096)")
    args = ap.parse_args(argv)
    defaults = ap.parse_args([])
    args.default_workload = all(getattr(args, k) == getattr(defaults, k) for k in WORKLOAD_FLAGS)
    return args


# ---- watchdog: a hang (a peer that never arrives, a collective that never completes) must cost minutes, not the
# driver's whole time limit -----------------------------------------------------------------------------------------
class Watchdog:
    def __init__(self, seconds, rank):
        self.seconds, self.rank = seconds, rank
        self.phase, self.t = "start", time.monotonic()
        if seconds and seconds > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def enter(self, phase):
        self.phase, self.t = phase, time.monotonic()

    def _run(self):
        while True:
            time.sleep(1.0)
            if time.monotonic() - self.t > self.seconds:
                sys.stderr.write(f"bench.py watchdog: rank {self.rank} made no progress for {self.seconds:.0f} s in phase "
                                 f"'{self.phase}' -- giving up (exit code 3)\n")
                sys.stderr.flush()
                os._exit(3)


def spawn_ranks(args):
    """`python bench.py --gpus N` launched like the N = 1 headline (no RANK in the environment): start the N ranks
    ourselves, one process per GPU, by re-executing this file under torch.distributed.run; rank 0 of the children prints
    the one JSON line, which passes through.  Returns the children's exit code."""
    import socket
    import subprocess

    import torch

    shared = os.environ.get("SCL_BENCH_SHARED_GPU") == "1"
    have = torch.cuda.device_count()
    if not shared and have < args.gpus:
        sys.stderr.write(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this node\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


# ---- workload ------------------------------------------------------------------------------------------------------
def make_model(w, freq):
    """-> (device model, coder parameters for the JSON line, spec of the same coder for the CPU baselines)"""
    from stanford_compression_library_amd.backend import lib, models

    fl = [int(f) for f in freq]
    if w.coder == "rans":
        return (models.RansModel(fl, w.range_factor, w.num_bits_out, 32),
                dict(NUM_BITS_OUT=w.num_bits_out, RANGE_FACTOR=w.range_factor),
                dict(coder="rans", freq=fl, range_factor=w.range_factor, num_bits_out=w.num_bits_out))
    if w.coder == "tans":
        return (models.TansModel(fl, 1, 32), dict(NUM_BITS_OUT=1, RANGE_FACTOR=1),
                dict(coder="tans", freq=fl, range_factor=1))
    if w.coder == "aec" and w.aec_model == "fixed":
        return (models.AecModel(lib.MODEL_FIXED, fl, len(fl), 0, 1 << 30, 32, 32),
                dict(PRECISION=32, model=f"FixedFreqModel({w.table})"), dict(coder="aec", model="fixed", freq=fl))
    if w.coder == "aec" and w.aec_model == "iid":
        ones = [1] * len(fl)
        return (models.AecModel(lib.MODEL_IID, ones, len(fl), 0, 1 << 30, 32, 32),
                dict(PRECISION=32, model=f"AdaptiveIIDFreqModel(all ones, K={len(fl)}) on {w.table} symbols"),
                dict(coder="aec", model="iid", freq=ones))
    if w.coder == "aec":
        K = w.aec_K
        return (models.AecModel(lib.MODEL_ORDERK, None, K, 1, 1 << 30, 32, 32),
                dict(PRECISION=32, model=f"AdaptiveOrderKFreqModel(k=1, K={K})"),
                dict(coder="aec", model="orderk", K=K, k=1))
    return models.RangeModel(fl, 32, 32), dict(PRECISION=32), dict(coder="range", freq=fl)


def csrc_sha(coder):
    """identifies the kernel sources a PMC pass was taken on: sha256 over the csrc files the coder's kernels compile from
    (profiles/traffic.json entries carry it; an entry from other sources is not quoted)"""
    d = os.path.join(ROOT, "stanford_compression_library_amd", "csrc")
    fam = {"rans": ("scl_rans",), "tans": ("scl_rans", "scl_tans"), "range": ("scl_range",), "aec": ("scl_aec",)}[coder]
    names = sorted(n for n in os.listdir(d) if n.endswith((".hip", ".h")) and
                   (n.startswith(fam) or n in ("scl_common.h", "scl_ans_fast_io.h")))
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode() + b"\0" + open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def core_sha():
    """the same for the compaction kernels (scl_core.hip): a traffic entry's `compact` figure is quoted only for these"""
    d = os.path.join(ROOT, "stanford_compression_library_amd", "csrc")
    h = hashlib.sha256()
    for n in ("scl_common.h", "scl_core.hip"):
        h.update(n.encode() + b"\0" + open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def stream_file_rate(w, res, n_bytes=1 << 28):
    """Row f2 next to the kernels (rank 0, N = 1, after the timed region; never `value`): the reference-API file path end to
    end -- ``encoder.encode(Uint8FileDataStream, chunk_len, EncodedBlockWriter)`` and ``decoder.decode(...)`` back -- on the
    first 256 MiB of this batch written to a file: file I/O, PCIe both ways, symbol lookup, kernels, framing."""
    import tempfile
    import time

    import numpy as np

    from stanford_compression_library_amd.compressors.range_coder import RangeCoderParams, RangeDecoder, RangeEncoder
    from stanford_compression_library_amd.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
    from stanford_compression_library_amd.compressors.tANS import tANSDecoder, tANSEncoder, tANSParams
    from stanford_compression_library_amd.core.data_stream import Uint8FileDataStream
    from stanford_compression_library_amd.core.encoded_stream import EncodedBlockReader, EncodedBlockWriter
    from stanford_compression_library_amd.core.prob_dist import Frequencies

    sym = res["sym"]
    chunk_len = int(sym.shape[1])
    rows = min(int(sym.shape[0]), max(1, n_bytes // chunk_len))
    data = sym[:rows].contiguous().cpu().numpy().reshape(-1)
    fr = Frequencies({i: int(f) for i, f in enumerate(np.asarray(res["freq"]).tolist())})
    if w.coder == "rans":
        p = rANSParams(fr, NUM_BITS_OUT=w.num_bits_out, RANGE_FACTOR=w.range_factor)
        enc, dec = rANSEncoder(p), rANSDecoder(p)
    elif w.coder == "tans":
        p = tANSParams(fr, RANGE_FACTOR=1)
        enc, dec = tANSEncoder(p), tANSDecoder(p)
    else:
        enc, dec = RangeEncoder(RangeCoderParams(), fr), RangeDecoder(RangeCoderParams(), fr)
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    src, mid, out = (os.path.join(d, f"scl_bench_{os.getpid()}_{n}.bin") for n in ("in", "enc", "out"))
    try:
        data.tofile(src)
        te = td = float("inf")
        for _ in range(3):  # the first repetition creates the model and the page-locked buffers
            # (a fresh output file every time: opening an existing 1 GB file for writing truncates it first, 0.08-0.13 s of
            # page freeing in /dev/shm that belongs to the PREVIOUS repetition's output, not to encode / decode)
            for stale in (mid, out):
                if os.path.exists(stale):
                    os.remove(stale)
            t0 = time.perf_counter()
            with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(mid) as wr:
                enc.encode(s, chunk_len, wr)
            t1 = time.perf_counter()
            with EncodedBlockReader(mid) as rd, Uint8FileDataStream(out, "wb") as s:
                dec.decode(rd, s)
            t2 = time.perf_counter()
            te, td = min(te, t1 - t0), min(td, t2 - t1)
        assert np.array_equal(np.fromfile(out, dtype=np.uint8), data), "file round trip differs from its input"
        framed = os.path.getsize(mid)
    finally:
        for pth in (src, mid, out):
            if os.path.exists(pth):
                os.remove(pth)
    return {"what": "encoder.encode(Uint8FileDataStream, chunk_len, EncodedBlockWriter) / decoder.decode(EncodedBlockReader, "
                    "Uint8FileDataStream): file I/O + PCIe + kernels + framing, best of 3; not `value`",
            "bytes": int(data.size), "framed_bytes": int(framed), "dir": d, "encode_MBps": round(data.size / te / 1e6, 1),
            "decode_MBps": round(data.size / td / 1e6, 1), "round_trip_verified": True}


def stream_bits(data_np, offs, nbits, c):
    import numpy as np

    nb, o = int(nbits[c]), int(offs[c])
    got = np.unpackbits(data_np[o // 8:(o + nb + 7) // 8 + 1])
    return got[o % 8:o % 8 + nb]


def cpu_baseline(w, spec, sym_dev, enc, target_seconds=10.0):
    """Times the CPU oracle (C restatement, oracle/scl_oracle.c, one thread per host core) on a bounded sample of the SAME
    workload and, as a by-product, checks the GPU streams of those chunks bit-for-bit against it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    import scl_oracle as orc

    kind = {"fixed": orc.MODEL_FIXED, "iid": orc.MODEL_IID, "orderk": orc.MODEL_ORDERK}.get(spec.get("model"), 0)
    kw = dict(RF=spec.get("range_factor", 1 << 16), b=spec.get("num_bits_out", 1), model_kind=kind,
              K=spec.get("K"), k=spec.get("k", 0))
    freq = spec.get("freq")
    chunk_len = int(sym_dev.shape[1])
    # the C oracle runs a slice of chunks per call; ctypes releases the GIL, so one Python thread per host core scales
    cores, logical = host_cores()  # every core the container may use
    n_probe = min(32, sym_dev.shape[0])
    sym = sym_dev[:n_probe].cpu().numpy()
    t0 = time.perf_counter()
    streams, nbits = orc.encode_batch(w.coder, sym, freq, **kw)
    orc.decode_batch(w.coder, streams, nbits, freq, chunk_len, **kw)
    per_chunk = (time.perf_counter() - t0) / n_probe
    single_thread = sym.size / (per_chunk * n_probe) / 1e6
    n = int(max(n_probe, min(sym_dev.shape[0], 65536, cores * target_seconds / max(per_chunk, 1e-9))))
    sym = sym_dev[:n].cpu().numpy()
    bounds = [n * i // cores for i in range(cores + 1)]
    parts = [(bounds[i], bounds[i + 1]) for i in range(cores) if bounds[i + 1] > bounds[i]]
    with ThreadPoolExecutor(max_workers=len(parts)) as pool:
        t0 = time.perf_counter()
        enc_parts = list(pool.map(lambda ab: orc.encode_batch(w.coder, sym[ab[0]:ab[1]], freq, **kw), parts))
        t1 = time.perf_counter()
        dec_parts = list(pool.map(lambda i: orc.decode_batch(w.coder, enc_parts[i][0], enc_parts[i][1], freq, chunk_len, **kw),
                                  range(len(parts))))
        t2 = time.perf_counter()
    nbits = np.concatenate([e[1] for e in enc_parts])
    dec = np.concatenate([d[0] for d in dec_parts])
    used = np.concatenate([d[1] for d in dec_parts])
    assert np.array_equal(dec, sym) and np.array_equal(used, nbits)
    # parity by-product: GPU streams of the sampled chunks == oracle streams
    g_nbits = enc.nbits[:n].cpu().numpy().astype(np.uint64)
    assert np.array_equal(g_nbits, nbits), "GPU/oracle stream lengths differ"
    checked = 0
    if n * enc.stride < (1 << 31):
        data = linear_slots(enc, n).cpu().numpy()
        offs = enc.bit_offset[:n].cpu().numpy()
        streams = [e[0] for e in enc_parts]
        for c in range(0, n, max(1, n // 64)):
            pi = max(i for i, (a, _) in enumerate(parts) if a <= c)
            ref = np.unpackbits(streams[pi][c - parts[pi][0]])[:int(nbits[c])]
            assert np.array_equal(stream_bits(data, offs, nbits, c), ref), f"chunk {c}: GPU != oracle"
            checked += 1
    nbytes = sym.size
    return {
        "value": round(nbytes / (t2 - t0) / 1e6, 3), "unit": "MB/s", "cores": len(parts), "kind": "port",
        # which baseline this is: a C restatement of the reference's algorithm -- NOT the reference's speed (the reference is
        # pure Python and runs ~25 000 x slower per core: cpu_baseline_restatement below, calibrated in BASELINE.md 4.2-4.3)
        "kind_note": "C port of the algorithm (oracle/scl_oracle.c), a 'reasonable CPU' line; the reference itself is pure "
                     "Python: see cpu_baseline_restatement for its speed",
        "sample": f"{n} chunks x {chunk_len} B of the same batch ({nbytes / 2**20:.1f} MiB), oracle/scl_oracle.c "
                  f"-O2, {len(parts)} threads (one per usable host core; {logical} logical CPUs), encode {nbytes / (t1 - t0) / 1e6:.2f} MB/s + decode "
                  f"{nbytes / (t2 - t1) / 1e6:.2f} MB/s aggregate; one thread alone: {single_thread:.2f} MB/s round trip",
        "single_thread_MBps": round(single_thread, 3),
        "encode_MBps": round(nbytes / (t1 - t0) / 1e6, 3), "decode_MBps": round(nbytes / (t2 - t1) / 1e6, 3),
        "gpu_streams_checked_against_oracle": checked,
    }


RESTATEMENT_MAX_WORKERS = 64


def restatement_baseline(w, spec, sym_dev, enc):
    """The reference-style baseline SURVEY 8d asks for: oracle/scl_restatement.py -- pure Python, one step per symbol,
    the reference's algorithmic shape -- on the host cores with multiprocessing.  Its streams are compared with the GPU's
    for the same chunks.  BASELINE.md 4.2-4.3: the imported reference runs at 0.7-1.25 x this speed, coder by coder."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import scl_restatement as rst

    # bounded: one worker PROCESS per host core x `per` chunks (a 4 KiB chunk takes 0.2 - 3 s per direction pair in pure
    # Python), at most RESTATEMENT_MAX_WORKERS of them: forking more copies of this process costs more wall time than the
    # sample they would add (256 workers: 31 s per configuration on the round-6 box) -- the cap is stated in `sample`
    usable, logical = host_cores()
    cores = min(usable, RESTATEMENT_MAX_WORKERS)
    slow = w.coder == "aec" and (spec.get("model") != "orderk" or spec.get("K", 0) > 64)
    per = (1 if slow else 4) if w.chunk_len <= 4096 else 1
    n = min(sym_dev.shape[0], cores * per)
    sym = sym_dev[:n].cpu().numpy()
    r = rst.timed_baseline(spec, sym, workers=cores, chunks_per_worker=per)
    assert r["ok"], "restatement round trip failed"
    checked = 0
    if n * enc.stride < (1 << 31):
        data = linear_slots(enc, n).cpu().numpy()
        offs, nbits = enc.bit_offset[:r["chunks"]].cpu().numpy(), enc.nbits[:r["chunks"]].cpu().numpy()
        for c, (nb, payload) in enumerate(r["streams"]):
            assert nb == int(nbits[c]), f"chunk {c}: GPU/restatement stream lengths differ"
            assert np.array_equal(stream_bits(data, offs, nbits, c),
                                  np.unpackbits(np.frombuffer(payload, np.uint8))[:nb]), f"chunk {c}: GPU != restatement"
            checked += 1
    return {
        "value": round(r["round_trip_MBps_aggregate"], 4), "unit": "MB/s", "cores": r["workers"], "kind": "restatement",
        "sample": f"{r['chunks']} chunks x {sym.shape[1]} B of the same batch, oracle/scl_restatement.py (pure Python, per-symbol, "
                  f"the reference's algorithmic shape), {r['workers']} worker processes x {per} chunks ({usable} usable cores, {logical} logical CPUs; "
                  f"at most {RESTATEMENT_MAX_WORKERS} workers), wall {r['wall_s']:.1f} s",
        "per_core_MBps": {"encode": round(r["encode_MBps_per_core"], 5), "decode": round(r["decode_MBps_per_core"], 5),
                          "round_trip": round(r["round_trip_MBps_per_core"], 5)},
        "reference_over_restatement": "0.7-1.25 by coder (BASELINE.md 4.2 / 4.3, measured in the build container against the "
                                      "imported reference)",
        "gpu_streams_checked_against_restatement": checked,
    }


def linear_slots(enc, n):
    """the first n logical slots of an encoded batch as one linear byte tensor (+16): what `bit_offset` indexes.  A copy
    for striped batches (a permutation of 16-byte pieces) -- the CPU baselines compare sampled streams bit for bit"""
    import torch

    if enc.layout == "linear":
        return enc.data[: n * enc.stride + 16]
    n64 = (n + 63) // 64
    body = enc.data[: n64 * 64 * enc.stride].view(n64, enc.stride // 16, 64, 16).permute(0, 2, 1, 3).reshape(-1)
    return torch.cat([body[: n * enc.stride], body.new_zeros(16)])


def rocprof_kernel_names(w, model, layout="linear"):
    """the names rocprofv3 prints for the two kernels of the timed step, from the library itself (C ABI 6:
    scl_rans_kernel_names / scl_tans_kernel_names report the instantiation the launch code picks for this model and batch
    size), so that the line can be matched mechanically with profiles/*_kernel_trace_summary.txt
    (tests/test_bench_contract.py does)"""
    import ctypes as C

    if w.coder in ("rans", "tans"):
        try:
            return model.kernel_names(int(w.chunks), layout)
        except Exception:
            pass
    return f"{w.coder}_encode", f"{w.coder}_decode"


def traffic_key(w, freq, layout="linear"):
    """what a PMC pass must have been taken on to be quoted for this run (tools/make_traffic_json.py stores it)"""
    key = {"coder": w.coder, "chunks": w.chunks, "chunk_len": w.chunk_len}
    if layout != "linear":
        key["layout"] = layout
    if w.coder == "aec" and w.aec_model == "order1":
        key.update(model="order1", K=w.aec_K)
    else:
        key.update(table=w.table, source=w.source, M=int(freq.sum()))
        if w.coder == "rans":
            key.update(num_bits_out=w.num_bits_out, range_factor=w.range_factor)
        if w.coder == "aec":
            key.update(model=w.aec_model)
    return key


def load_traffic_note(key, sha):
    """HBM traffic per launch measured with rocprofv3 PMC passes (committed under profiles/traffic.json): the entry
    taken on exactly this workload AND on exactly these kernel sources (`csrc_sha`), or None -- a pass is never attached to
    another workload or to another build."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        for e in json.load(open(path)).get("entries", []):
            if e.get("key") == key and e.get("csrc_sha") == sha:
                return e
    except Exception:
        pass
    return None


class Dist:
    """the few things bench.py needs from torch.distributed, no-ops on one rank"""

    def __init__(self, world, rank, local_rank, shared_gpu, dev):
        self.world, self.rank, self.local_rank, self.shared_gpu, self.dev = world, rank, local_rank, shared_gpu, dev

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist

            if self.shared_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[self.local_rank])

    def reduce(self, values, op="max"):
        """[floats] -> [floats], reduced over ranks"""
        if self.world == 1:
            return list(values)
        import torch
        import torch.distributed as dist

        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if self.shared_gpu else self.dev)
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
        return [float(v) for v in t.tolist()]

    def gather_all_i64(self, values):
        """[ints] per rank -> [[ints] of rank 0, ...] on every rank, exact (int64)"""
        if self.world == 1:
            return [list(values)]
        import torch
        import torch.distributed as dist

        t = torch.tensor(list(values), dtype=torch.int64, device="cpu" if self.shared_gpu else self.dev)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [[int(v) for v in o.tolist()] for o in out]

    def gather_all(self, values):
        """[floats] per rank -> [[floats] of rank 0, [floats] of rank 1, ...] on every rank"""
        if self.world == 1:
            return [list(values)]
        import torch
        import torch.distributed as dist

        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if self.shared_gpu else self.dev)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [[float(v) for v in o.tolist()] for o in out]


def payload_checksum(t):
    """two 64-bit checksums (wrapping sums of the 8-byte words and of word x position) of a uint8 device tensor: what every
    rank publishes about its dense payload OUT OF BAND (torch.distributed), so that the root can check what arrived over
    the C ABI's own RCCL exchange without a second copy of the data"""
    import torch

    n = t.numel()
    pad = (-n) % 8
    x = torch.cat([t, torch.zeros(pad, dtype=torch.uint8, device=t.device)]) if (pad or t.storage_offset() % 8) else t
    if x.data_ptr() % 8:
        x = x.clone()
    wds = x.view(torch.int64)
    idx = torch.arange(1, wds.numel() + 1, dtype=torch.int64, device=t.device)
    return [int(n), int(wds.sum().item()), int((wds * idx).sum().item())]


def measure(w, D, wd, steps, warmup, min_warm_ms, with_cpu, with_restatement, with_dense=True):
    """one workload on this rank's GPU -> the fields of its JSON object (rank 0 assembles the line)"""
    import numpy as np
    import torch
    import torch.distributed as dist

    from stanford_compression_library_amd import bench_data

    dev, rank, world = D.dev, D.rank, D.world
    freq = {"t256": bench_data.t256_table, "uniform": bench_data.uniform256_table,
            "uniform1": lambda: np.ones(256, dtype=np.int64)}[w.table]()
    n_chunks, chunk_len = w.chunks, w.chunk_len
    static_model = w.coder != "aec" or w.aec_model in ("fixed", "iid")
    source_note = f"256-symbol static table {w.table} (M={int(freq.sum())}), i.i.d. symbols p=f/M"
    wd.enter(f"{w.coder}: input generation")
    if not static_model:
        # configs[3]: Markov-1 source (S4 of SURVEY 8d), every chunk its own chain, generated on the device: the whole
        # batch is distinct data (a tiled batch would be served from L2 and is not an HBM measurement)
        sym = bench_data.markov1_chunks_device(w.aec_K, n_chunks, chunk_len, seed=4000 + rank, device=dev)
        source_note = f"order-1 adaptive model, K={w.aec_K}, Markov-1 source (all chunks distinct)"
    elif w.source == "markov1":
        # north_star's second source for the static coders: Markov-1 bytes; the table is what the data's own histogram
        # gives (row f3: scl_histogram_u8 + the deterministic normaliser), every symbol present so that f >= 1
        from stanford_compression_library_amd.backend.modeling import histogram_u8, normalize_counts

        sym = bench_data.markov1_chunks_device(256, n_chunks, chunk_len, seed=4000 + rank, device=dev)
        counts = histogram_u8(sym) + 1
        if world > 1:  # every rank codes with the same table: the histogram of all shards
            t = torch.from_numpy(counts).to("cpu" if D.shared_gpu else dev)
            dist.all_reduce(t)
            counts = t.cpu().numpy()
        freq = normalize_counts(counts, 65536 if w.coder == "range" and w.table == "uniform1" else 4096)
        source_note = (f"order-1 Markov byte source (Dirichlet(0.3) rows, all chunks distinct), static table = normalised "
                       f"histogram of the data (M={int(freq.sum())})")
    else:
        sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=5000 + rank, device=dev)
    model, coder_params, spec = make_model(w, freq)
    layout = model.pick_layout(getattr(w, "layout", "auto"), n_chunks) if w.coder in ("rans", "tans", "range") else "linear"
    kernels = rocprof_kernel_names(w, model, layout)
    if w.sym_pad:
        padded = torch.zeros((n_chunks, chunk_len + w.sym_pad), dtype=torch.uint8, device=dev)
        padded[:, :chunk_len] = sym
        sym = padded[:, :chunk_len]
    enc = model.alloc_encoded(n_chunks, chunk_len, dev, layout=layout)
    dec_out = model.alloc_decoded(n_chunks, chunk_len, dev)

    def step(events=None):
        if events is not None:
            events[0].record()
        model.encode_batch(sym, out=enc)
        if events is not None:
            events[1].record()
        model.decode_encoded(enc, chunk_len, out=dec_out)
        if events is not None:
            events[2].record()

    # ---- warm-up: at least `warmup` steps and at least `min_warm_ms` of them -------------------------------------
    wd.enter(f"{w.coder}: warm-up")
    t_w = time.perf_counter()
    done = 0
    while done < warmup or (time.perf_counter() - t_w) * 1e3 < min_warm_ms:
        for _ in range(max(1, min(5, warmup - done))):
            step()
            done += 1
        torch.cuda.synchronize()
        if done >= 100000:
            break
    warm_ms = (time.perf_counter() - t_w) * 1e3
    torch.cuda.synchronize()
    wd.enter(f"{w.coder}: barrier before the timed region")
    D.barrier()
    torch.cuda.synchronize()
    wd.enter(f"{w.coder}: timed region")
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        step(evs[i])  # events sit on torch's current stream, the stream the kernels are launched on
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.reduce([elapsed], "max")[0]

    # ---- correctness of what was just timed (outside the timed region) ------------------------------
    wd.enter(f"{w.coder}: verification")
    dec_sym, dec_lens, dec_used, dec_status = dec_out
    ok = (int(enc.status.abs().sum()) == 0 and int(dec_status.abs().sum()) == 0
          and torch.equal(dec_sym[:, :chunk_len], sym) and torch.equal(dec_used, enc.nbits)
          and int(dec_lens.min()) == chunk_len and int(dec_lens.max()) == chunk_len)
    if not ok:
        raise SystemExit(f"rank {rank}: round trip FAILED (status enc={int(enc.status.abs().sum())} "
                         f"dec={int(dec_status.abs().sum())})")

    enc_t = np.array([e[0].elapsed_time(e[1]) for e in evs])
    dec_t = np.array([e[1].elapsed_time(e[2]) for e in evs])
    # the decode kernel again, this time behind ANOTHER DECODE instead of behind the encode that has just written ~1 GB
    # (the write-back of those lines is paid by whoever runs next: VERDICT r4 weak #9 -- the spread is in the line now)
    wd.enter(f"{w.coder}: decode after decode")
    n_dd = max(3, min(steps, 20))
    model.decode_encoded(enc, chunk_len, out=dec_out)
    dd = [torch.cuda.Event(enable_timing=True) for _ in range(n_dd + 1)]
    dd[0].record()
    for i in range(n_dd):
        model.decode_encoded(enc, chunk_len, out=dec_out)
        dd[i + 1].record()
    torch.cuda.synchronize()
    dec_after_dec_ms = float(np.mean([dd[i].elapsed_time(dd[i + 1]) for i in range(n_dd)]))
    enc_ms, dec_ms = float(enc_t.mean()), float(dec_t.mean())
    enc_med, dec_med = float(np.median(enc_t)), float(np.median(dec_t))
    in_bytes = n_chunks * chunk_len
    stream_bytes = int(((enc.nbits.to(torch.int64) + 7) // 8).sum().item())
    alg_bytes = in_bytes + stream_bytes  # encode = n read + ceil(bits/8) written (decode mirrors it)
    bits_per_symbol = float(enc.nbits.to(torch.float64).mean().item()) / chunk_len

    # The encoders leave every stream in its own slot, described by (bit_offset, nbits) -- the form the decoders read.
    # SURVEY 8d counts the left-align / compaction pass with the encode; it is timed here (HIP events, data resident) and
    # reported beside `value` (`value_dense`, `roofline_dense`), which it never enters.
    compact_ms = dense_pipelined_ms = None
    if with_dense:
        from stanford_compression_library_amd.backend import models as _m

        wd.enter(f"{w.coder}: compaction")
        # caller-owned worst-case buffers (scl_streams_compact never waits for the host)
        c_dense = torch.empty(_m.compact_capacity(n_chunks, enc.stride), dtype=torch.uint8, device=dev)
        c_offs = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
        c_scratch = torch.empty(_m.compact_scratch_bytes(n_chunks), dtype=torch.uint8, device=dev)
        for _ in range(3):
            _m.compact_into(enc, c_dense, c_offs, c_scratch)  # warm
        cev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        cev[0].record()
        for _ in range(10):
            _m.compact_into(enc, c_dense, c_offs, c_scratch)
        cev[1].record()
        torch.cuda.synchronize()
        compact_ms = cev[0].elapsed_time(cev[1]) / 10
        assert int(c_offs[-1].item()) == stream_bytes, "compaction total differs from the sum of the stream sizes"
        # the same result as ONE pipelined operation: two sub-batches on two streams, the compaction of the first running
        # while the second is encoded (backend/models.py DensePipeline; scl_streams_compact_at keeps the offsets on the
        # device).  Timed as a whole -- encode included -- and checked byte for byte against the sequential result.
        if not model._needs_scratch and n_chunks >= 4096 and not getattr(w, "no_dense_pipeline", False):
            wd.enter(f"{w.coder}: pipelined dense encode")
            pipe = _m.DensePipeline(model, n_chunks, chunk_len, dev, n_sub=2, layout=layout)
            p_dense, p_offs = pipe.run(sym)
            torch.cuda.synchronize()
            assert torch.equal(p_offs, c_offs) and torch.equal(p_dense[:stream_bytes], c_dense[:stream_bytes]), \
                "pipelined dense output differs from encode + scl_streams_compact"
            pev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            pev[0].record()
            for _ in range(10):
                pipe.run(sym)
            pev[1].record()
            torch.cuda.synchronize()
            dense_pipelined_ms = pev[0].elapsed_time(pev[1]) / 10
            del pipe, p_dense, p_offs
        del c_dense, c_offs, c_scratch

    res = dict(w=w, freq=freq, sym=sym, enc=enc, model=model, kernels=kernels, layout=layout, spec=spec, coder_params=coder_params, source_note=source_note,
               static_model=static_model, elapsed=elapsed, own_elapsed=own_elapsed, enc_ms=enc_ms, dec_ms=dec_ms,
               enc_med=enc_med, dec_med=dec_med, enc_min=float(enc_t.min()), dec_min=float(dec_t.min()), in_bytes=in_bytes,
               stream_bytes=stream_bytes, alg_bytes=alg_bytes, bits_per_symbol=bits_per_symbol, compact_ms=compact_ms,
               dense_pipelined_ms=dense_pipelined_ms, dec_after_dec_ms=dec_after_dec_ms,
               warm_ms=warm_ms, warm_steps=done, steps=steps)
    if rank == 0 and world == 1:
        if with_cpu:
            wd.enter(f"{w.coder}: cpu_baseline (C oracle)")
            res["cpu_baseline"] = cpu_baseline(w, spec, sym, enc, target_seconds=10.0 if w.default_workload else 5.0)
        if with_restatement:
            wd.enter(f"{w.coder}: cpu_baseline_restatement (pure Python)")
            res["cpu_baseline_restatement"] = restatement_baseline(w, spec, sym, enc)
    return res


def rooflines(res):
    """roofline objects of one measured workload (encode kernel, decode kernel, encode + compaction)"""
    w, freq = res["w"], res["freq"]
    tkey, sha = traffic_key(w, freq, res.get("layout", "linear")), csrc_sha(w.coder)
    traffic = load_traffic_note(tkey, sha)
    alg, in_bytes = res["alg_bytes"], res["in_bytes"]

    def roof(ms, med, side, kernel):
        gbs = alg / (ms * 1e-3) / 1e9
        t = traffic.get(side) if traffic else None
        return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": t,
                # not a live counter: HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/, quoted
                # only when the pass was taken on this workload and on these kernel sources
                "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate run, same "
                                  "workload, same csrc_sha)" if t else None,
                "traffic_over_algorithmic": round(t / alg, 3) if t else None,
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(ms, 4), "median_launch_ms": round(med, 4),
                "frac_median": round(alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "read_only_frac": round(in_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if side == "encode" else None}

    k_enc, k_dec = res["kernels"]
    r_enc = roof(res["enc_ms"], res["enc_med"], "encode", k_enc)
    r_dec = roof(res["dec_ms"], res["dec_med"], "decode", k_dec)
    # `avg_launch_ms` is the decode behind the step's encode (what a round trip pays); the same kernel behind another decode:
    r_dec["after_decode_ms"] = round(res["dec_after_dec_ms"], 4)
    r_dec["frac_after_decode"] = round(alg / (res["dec_after_dec_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    r_dense = None
    if res["compact_ms"] is not None:
        seq_ms = res["enc_ms"] + res["compact_ms"]
        pipe_ms = res.get("dense_pipelined_ms")
        ms = min(seq_ms, pipe_ms) if pipe_ms else seq_ms
        gbs = alg / (ms * 1e-3) / 1e9
        # SURVEY 8d's reading of the encode: symbols in, left-aligned dense streams out (encode kernel + scl_streams_compact),
        # priced at the same algorithmic bytes (symbols read + stream bytes written once).  `sequential_ms` = the two one
        # after the other; `pipelined_ms` = the same result in two sub-batches on two streams (DensePipeline), the whole
        # operation timed; `frac` is on the faster of the two
        cp = "cp_copy_striped" if res.get("layout") == "striped" else "cp_copy"
        r_dense = {"bound": "hbm", "kernels": [k_enc, "cp_scan_tiles + cp_scan_sums + cp_add_base + " + cp],
                   "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                   "algorithmic_bytes_per_launch": alg, "encode_ms": round(res["enc_ms"], 4),
                   "compact_ms": round(res["compact_ms"], 4), "sequential_ms": round(seq_ms, 4),
                   "pipelined_ms": round(pipe_ms, 4) if pipe_ms else None, "sub_batches": 2 if pipe_ms else 1,
                   "frac_sequential": round(alg / (seq_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None}
        # HBM bytes of encode + compaction from the PMC passes (cp_copy's entry is quoted for the same workload and the
        # same scl_core.hip): two transfers more than the algorithmic bytes -- the slots are written, read back and written
        # again -- which is what bounds this figure ("dense")
        t_cp = traffic.get("compact") if traffic and traffic.get("core_sha") == core_sha() else None
        if t_cp and r_enc["traffic"]:
            r_dense["traffic"] = int(r_enc["traffic"] + t_cp)
            r_dense["traffic_over_algorithmic"] = round(r_dense["traffic"] / alg, 3)
            r_dense["traffic_source"] = r_enc["traffic_source"]
    return r_enc, r_dec, r_dense, tkey, sha


def config_entry(name, res):
    """one element of `other_configs`"""
    w = res["w"]
    r_enc, r_dec, r_dense, tkey, sha = rooflines(res)
    total = res["in_bytes"]
    return {
        "config": name,
        "workload": f"batched {w.coder}: {w.chunks} independent {w.chunk_len} B chunks ({total / 2**30:.3f} GiB), one lane per "
                    f"chunk, " + res["source_note"],
        "coder": w.coder, **res["coder_params"], "chunks": w.chunks, "chunk_len": w.chunk_len, "slot_layout": res["layout"],
        "steps": res["steps"], "warmup_steps": res["warm_steps"], "warmup_ms": round(res["warm_ms"], 1),
        "value": round(total * res["steps"] / res["elapsed"] / 1e6, 2), "unit": "MB/s",
        "ms_per_step": round(res["elapsed"] / res["steps"] * 1e3, 4),
        "encode_ms": round(res["enc_ms"], 4), "decode_ms": round(res["dec_ms"], 4),
        "bits_per_symbol_out": round(res["bits_per_symbol"], 4),
        "roofline": r_enc if res["enc_ms"] >= res["dec_ms"] else r_dec,
        "roofline_encode": r_enc, "roofline_decode": r_dec, "roofline_dense": r_dense,
        "round_trip_verified": True, "traffic_key": tkey, "csrc_sha": sha, "core_sha": core_sha(),
        "cpu_baseline": res.get("cpu_baseline"), "cpu_baseline_restatement": res.get("cpu_baseline_restatement"),
        "_summary": summary_entry(r_enc, r_dec, r_dense, res.get("cpu_baseline"), res.get("cpu_baseline_restatement")),
    }


def summary_entry(r_enc, r_dec, r_dense, cpu, rst):
    """the few numbers of one configuration the driver's record must keep: times, fractions of the 8 TB/s
    roofline, PMC traffic over algorithmic bytes, both CPU baselines"""
    def f3(x):
        return None if x is None else round(x, 3)

    return {"enc": f3(r_enc["avg_launch_ms"]), "dec": f3(r_dec["avg_launch_ms"]), "dec_dd": f3(r_dec.get("after_decode_ms")),
            "f_enc": f3(r_enc["frac"]), "f_dec": f3(r_dec["frac"]), "f_dense": f3(r_dense["frac"]) if r_dense else None,
            "tx": [r_enc.get("traffic_over_algorithmic"), r_dec.get("traffic_over_algorithmic"),
                   r_dense.get("traffic_over_algorithmic") if r_dense else None],
            "cpu_c": round(cpu["value"], 1) if cpu else None, "cpu_py": round(rst["value"], 3) if rst else None}


LINE_LIMIT = 6144  # bytes: the driver's record must be able to hold and parse the whole line (round 5's 21.6 KB did not)


def write_full_record(out, path=None):
    """the full record (every roofline field, other_configs, stream_file ...) goes to a FILE, never to stdout / stderr: the
    driver keeps a bounded tail of both and parses the one line out of it.  Returns the repo-relative path or None."""
    path = path or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            f.write(json.dumps(out) + "\n")
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def _compact_roof(r):
    if not r:
        return None
    keep = ("bound", "kernel", "kernels", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
            "read_only_frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "after_decode_ms", "encode_ms", "compact_ms")
    c = {k: r[k] for k in keep if k in r and r[k] is not None or k == "traffic" and k in r}
    if r.get("traffic") is not None:
        c["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc, same csrc_sha)"
    return c


def _compact_cpu(c):
    if not c:
        return None
    keep = ("value", "unit", "cores", "kind", "single_thread_MBps", "encode_MBps", "decode_MBps")
    o = {k: c[k] for k in keep if k in c}
    o["sample"] = c.get("sample", "")[:200]
    checked = c.get("gpu_streams_checked_against_oracle", c.get("gpu_streams_checked_against_restatement"))
    if checked is not None:
        o["gpu_streams_checked"] = checked
    return o


def compact_line(out, full_path=None):
    """the ONE line of the contract, < LINE_LIMIT bytes: the contract keys, the rooflines (dominant kernel, encode, decode,
    dense) with their PMC traffic, both CPU baselines, multi_gpu / gather when present and `summary` (one short entry per
    BASELINE configuration) as the LAST key.  Everything else is in the full record (`full_record`)."""
    keys = ("metric", "value", "value_definition", "value_dense", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "encode_MBps", "decode_MBps")
    line = {k: out[k] for k in keys if k in out}
    line["roofline"] = _compact_roof(out.get("roofline"))
    for k in ("roofline_encode", "roofline_decode", "roofline_dense"):
        line[k] = _compact_roof(out.get(k))
    line["round_trip_verified"] = out.get("round_trip_verified")
    line["csrc_sha"] = out.get("csrc_sha")
    if "multi_gpu" in out:
        line["multi_gpu"] = out["multi_gpu"]
    if "gather" in out:
        g = dict(out["gather"])
        g["verified"] = bool(g.get("verified"))  # the sentence is in the full record
        for k in ("per_rank_encode_ms", "per_rank_compact_ms"):
            g.pop(k, None)
        line["gather"] = g
    line["cpu_baseline"] = _compact_cpu(out.get("cpu_baseline"))
    if "cpu_baseline_restatement" in out:
        line["cpu_baseline_restatement"] = _compact_cpu(out["cpu_baseline_restatement"])
    line["full_record"] = full_path
    line["summary"] = out.get("summary")
    # never exceed the limit: shed the optional objects, least important first (summary carries their numbers anyway)
    for k in ("roofline_encode", "roofline_decode", "cpu_baseline_restatement", "roofline_dense", "gather"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def other_workloads(args):
    """BASELINE.json's other single-GPU configurations, as (name, flags, steps)"""
    def wl(**kw):
        w = copy.copy(args)
        for k, v in kw.items():
            setattr(w, k, v)
        w.default_workload = False
        return w

    return [
        ("configs[1]: 256-symbol static rANS, 65 536 x 4 KiB chunks", wl(chunks=65536), 50),
        ("configs[2]: 32-bit range coder, 1 GiB uniform bytes", wl(coder="range", table="uniform1"), 20),
        ("configs[3]: order-1 adaptive arithmetic coder, K=16, 1 GiB Markov-1", wl(coder="aec", aec_K=16), 5),
        ("configs[3] on bytes: order-1 adaptive arithmetic coder, K=256, 256 MiB Markov-1",
         wl(coder="aec", aec_K=256, chunks=65536), 3),
    ]


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    from stanford_compression_library_amd.backend import lib

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = Watchdog(args.watchdog_s if args.watchdog_s is not None else (600.0 if world > 1 else 0.0), rank)
    lib.require_device()
    # SCL_BENCH_SHARED_GPU=1 (testing the N > 1 code path on a one-GPU box): every rank uses cuda:0 and the ranks
    # talk over gloo -- RCCL refuses two ranks on one device.  Never set for a measurement.
    shared_gpu = os.environ.get("SCL_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        wd.enter("init_process_group")
        import datetime

        pg_timeout = datetime.timedelta(seconds=max(60.0, (args.watchdog_s or 600.0)))
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)),
                                    timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)),
                                    device_id=dev, timeout=pg_timeout)
        world = dist.get_world_size()
    D = Dist(world, rank, local_rank, shared_gpu, dev)

    want_cpu = world == 1 and not args.no_cpu_baseline
    single = None
    if world > 1:
        # the N = 1 figure of THIS run: rank 0 alone, the others parked at a barrier (same binary, same box, same clocks)
        if rank == 0:
            solo = Dist(1, 0, local_rank, shared_gpu, dev)
            r1 = measure(args, solo, wd, args.steps, args.warmup, args.min_warm_ms, False, False, with_dense=False)
            single = r1["in_bytes"] * r1["steps"] / r1["elapsed"] / 1e6
            del r1
            torch.cuda.empty_cache()
        wd.enter("barrier after rank 0's solo run")
        D.barrier()
    res = measure(args, D, wd, args.steps, args.warmup, args.min_warm_ms, want_cpu, want_cpu)
    n_chunks, chunk_len = args.chunks, args.chunk_len
    enc, sym, model = res["enc"], res["sym"], res["model"]
    elapsed, in_bytes, stream_bytes = res["elapsed"], res["in_bytes"], res["stream_bytes"]
    enc_ms, dec_ms, compact_ms = res["enc_ms"], res["dec_ms"], res["compact_ms"]

    def barrier():
        D.barrier()

    gather_info = None
    if args.gather:
        # BASELINE.json configs[4]: per-GPU encode, then the gather of the per-block streams to rank 0 over RCCL
        # (scl_streams_gather_rccl of the C ABI; torch.distributed point-to-point on the gloo test path).  Reported
        # beside `value`, never inside it: (a) the three phases one after the other, (b) the same work as a pipeline of
        # sub-batches in which sub-batch i travels while sub-batch i + 1 is encoded and compacted.
        from stanford_compression_library_amd.backend.models import compact
        from stanford_compression_library_amd.backend.sharded import (GatherWorkspace, RcclGather, block_offsets,
                                                                      encode_gather_overlapped, gather_streams_to_root)

        wd.enter("gather: communicator + untimed first exchange")
        comm = RcclGather(world, rank, dev) if not shared_gpu else None
        dense, offsets = compact(enc)  # untimed: the output buffer comes from the allocator's cache afterwards
        gather_streams_to_root(dense, offsets, world, rank, dev, comm=comm)  # untimed: connections, buffers
        del dense, offsets
        torch.cuda.synchronize()
        barrier()
        wd.enter("gather: sequential encode / compact / gather")
        g0 = time.perf_counter()
        model.encode_batch(sym, out=enc)
        torch.cuda.synchronize()
        ga = time.perf_counter()
        dense, offsets = compact(enc)
        torch.cuda.synchronize()
        g1 = time.perf_counter()
        total, gathered, goffs = gather_streams_to_root(dense, offsets, world, rank, dev, return_data=True, comm=comm)
        torch.cuda.synchronize()
        g_own = time.perf_counter()
        barrier()
        g2 = time.perf_counter()
        # ---- the exchange validates itself: every rank publishes size + two checksums of its payload out
        # of band; the root recomputes them on the segments it RECEIVED, checks the communicator really has `world` ranks
        # and the chunk offsets are one increasing table over all ranks' chunks -- or the run fails naming the phase
        wd.enter("gather: verification")
        mine = payload_checksum(dense[:int(offsets[-1].item())])
        sums = D.gather_all_i64(mine)
        n_blocks, verdict = None, "ok"
        if comm is not None and comm.nranks != world:
            verdict = f"the RCCL communicator reports {comm.nranks} ranks, bench.py was started with {world}"
        if rank == 0 and verdict == "ok":
            pos = 0
            for r, (nb, c1, c2) in enumerate(sums):
                got = payload_checksum(gathered[pos:pos + nb])
                if got != [nb, c1, c2]:
                    verdict = f"payload of rank {r} ({nb} bytes at offset {pos}) differs from what that rank sent"
                    break
                pos += nb
            if verdict == "ok" and pos != int(total):
                verdict = f"gathered {int(total)} bytes, the ranks sent {pos}"
            if verdict == "ok":
                go = goffs.cpu().numpy()
                if go.size != world * n_chunks + 1 or int(go[0]) != 0 or int(go[-1]) != pos or (np.diff(go) <= 0).any():
                    verdict = "the gathered per-chunk offset table is not one increasing table over all ranks' chunks"
            if verdict == "ok":
                n_blocks = int(block_offsets(goffs, max(1, (1 << 20) // chunk_len)).numel()) - 1
        flags = D.gather_all_i64([0 if verdict == "ok" else 1])
        if any(f[0] for f in flags):
            sys.stderr.write(f"bench.py --gather: rank {rank}: gather verification FAILED: {verdict}\n")
            raise SystemExit(4)
        del gathered, goffs
        barrier()
        wd.enter("gather: overlapped pipeline")
        ws = GatherWorkspace(model, n_chunks, chunk_len, world, dev)  # caller-owned buffers, streams, pinned counts
        encode_gather_overlapped(model, sym, world, rank, comm=comm, workspace=ws)  # untimed: connections, allocator
        torch.cuda.synchronize()
        barrier()
        timings, _ = encode_gather_overlapped(model, sym, world, rank, comm=comm, workspace=ws)
        barrier()
        t_ov = D.reduce([timings["overlapped_ms"]], "max")[0]
        per_rank = D.gather_all([(ga - g0) * 1e3, (g1 - ga) * 1e3, (g_own - g1) * 1e3])

        def mmm(i):
            v = sorted(r[i] for r in per_rank)
            return {"min": round(v[0], 3), "median": round(float(np.median(v)), 3), "max": round(v[-1], 3)}

        gather_info = {"transport": "rccl (scl_streams_gather_rccl)" if comm is not None else "torch.distributed p2p",
                       "rccl_ranks": comm.nranks if comm is not None else None,
                       "encode_ms": round((ga - g0) * 1e3, 3), "compact_ms": round((g1 - ga) * 1e3, 3),
                       "gather_ms": round((g2 - g1) * 1e3, 3), "sequential_ms": round((g2 - g0) * 1e3, 3),
                       "per_rank_encode_ms": mmm(0), "per_rank_compact_ms": mmm(1), "per_rank_gather_ms": mmm(2),
                       "overlapped_ms": round(float(t_ov), 3), "sub_batches": timings["sub_batches"],
                       "gathered_bytes": int(total), "blocks_1MiB": n_blocks,
                       "verified": "per-rank size + two 64-bit checksums exchanged out of band == the segments the root "
                                   "received; offset table increasing over all chunks; communicator rank count == --gpus"}
        if comm is not None:
            comm.close()

    # per-rank step times of the timed region (what a straggler looks like from the one JSON line)
    wd.enter("per-rank statistics")
    per_rank_steps = D.gather_all([res["own_elapsed"] / args.steps * 1e3, enc_ms, dec_ms])

    file_rate = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.coder in ("rans", "tans", "range"):
        wd.enter("stream_file (reference-API file path, end to end)")
        try:
            file_rate = stream_file_rate(args, res)
        except Exception as exc:  # an extra: it must not take the line with it
            file_rate = {"error": f"{type(exc).__name__}: {exc}"}
    others = []
    if rank == 0 and world == 1 and args.default_workload and not args.no_other_configs:
        del res["sym"], res["enc"], res["model"]
        del enc, sym, model
        torch.cuda.empty_cache()
        for name, w, steps in other_workloads(args):
            try:
                r = measure(w, D, wd, steps, 3, args.min_warm_ms, want_cpu, want_cpu)
                others.append(config_entry(name, r))
                del r
            except Exception as exc:  # a failing extra configuration must not take the headline line with it
                others.append({"config": name, "error": f"{type(exc).__name__}: {exc}"})
            torch.cuda.empty_cache()

    if rank == 0:
        wd.enter("report")
        total_bytes = in_bytes * world
        value = total_bytes * args.steps / elapsed / 1e6
        r_enc, r_dec, r_dense, tkey, sha = rooflines(res)
        dense_extra_ms = compact_ms
        if res.get("dense_pipelined_ms"):  # what the dense form adds to a step: the pipelined operation replaces the encode
            dense_extra_ms = min(compact_ms, max(res["dense_pipelined_ms"] - enc_ms, 0.0))
        value_dense = total_bytes * args.steps / (elapsed + args.steps * dense_extra_ms * 1e-3) / 1e6
        w = args
        out = {
            "metric": "MB/s encode+decode, 1 GiB i.i.d. bytes, 256-sym rANS; achieved HBM GB/s %peak",
            "value": round(value, 2),
            # what `value` is: N / (t_encode + t_decode) with every stream left in its own slot, (bit_offset, nbits) --
            # the form the decoders read.  value_dense: the same with the left-align / compaction pass of SURVEY 8d
            # (scl_streams_compact, BitArray.tobytes() of every stream back to back) counted into the step.
            "value_definition": "slots", "value_dense": round(value_dense, 2),
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_actual": {"steps": res["warm_steps"], "ms": round(res["warm_ms"], 1), "min_ms": args.min_warm_ms},
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"batched {w.coder}: {n_chunks} independent "
                                   f"{chunk_len} B chunks per GPU ({in_bytes / 2**30:.3f} GiB/GPU), one lane per chunk, "
                                   + res["source_note"], "source": w.source if res["static_model"] else "markov1",
                       "coder": w.coder, **res["coder_params"], "chunks_per_gpu": n_chunks, "chunk_len": chunk_len,
                       "slot_layout": res["layout"],
                       "bits_per_symbol_out": round(res["bits_per_symbol"], 4), "sharding": f"{world} x independent shards"},
            "encode_MBps": round(total_bytes / (enc_ms * 1e-3) / 1e6, 2),
            "decode_MBps": round(total_bytes / (dec_ms * 1e-3) / 1e6, 2),
            "roofline": r_enc if enc_ms >= dec_ms else r_dec,
            "roofline_encode": r_enc, "roofline_decode": r_dec, "roofline_dense": r_dense,
            "round_trip_verified": True, "traffic_key": tkey, "csrc_sha": sha, "core_sha": core_sha(),
            "dense_output": {"compact_ms": round(compact_ms, 4), "compacted_bytes": stream_bytes,
                             "value_incl_compaction_MBps":
                                 round(total_bytes / ((enc_ms + compact_ms + dec_ms) * 1e-3) / 1e6, 2)},
        }
        if world > 1:
            def mmm(i):
                v = sorted(r[i] for r in per_rank_steps)
                return {"min": round(v[0], 4), "median": round(float(np.median(v)), 4), "max": round(v[-1], 4)}

            out["multi_gpu"] = {
                "ranks": world, "backend": "gloo (shared-GPU test mode)" if shared_gpu else "nccl (RCCL)",
                "per_rank_ms_per_step": mmm(0), "per_rank_encode_ms": mmm(1), "per_rank_decode_ms": mmm(2),
                # rank 0 alone on the same box in the same run, then all ranks together: value / (N x that)
                "single_rank_value_same_run": round(single, 2) if single else None,
                "scaling_efficiency": round(value / (world * single), 4) if single else None,
            }
        if gather_info:
            out["gather"] = gather_info
        out["cpu_baseline"] = res.get("cpu_baseline")
        if "cpu_baseline_restatement" in res:
            out["cpu_baseline_restatement"] = res["cpu_baseline_restatement"]
        if file_rate:
            out["stream_file"] = file_rate
        if others:
            out["other_configs"] = others
        # LAST key, <= 1 KB: the driver's record keeps the tail of the line.  One entry per configuration -- kernel times
        # (enc, dec: ms per launch; dec_dd = the decode kernel behind another decode instead of behind the encode), fractions
        # of the 8 TB/s roofline on algorithmic bytes (f_dense: encode + compaction), tx = PMC HBM traffic over algorithmic
        # bytes [encode, decode, dense] where a stamped pass exists, cpu_c / cpu_py = C port / pure-Python restatement (MB/s),
        # MBps = the configuration's round-trip value.
        summ = {"headline": summary_entry(r_enc, r_dec, r_dense, res.get("cpu_baseline"), res.get("cpu_baseline_restatement"))}
        summ["headline"]["MBps"] = round(value)
        if file_rate and "encode_MBps" in file_rate:  # end-to-end file path [encode, decode] MB/s
            summ["headline"]["file"] = [round(file_rate["encode_MBps"]), round(file_rate["decode_MBps"])]
        for o in others:
            tag = o["config"].split(":")[0].replace("configs", "c").replace(" on bytes", "b")
            if "_summary" in o:
                summ[tag] = dict(o.pop("_summary"), MBps=round(o["value"]))
            else:
                summ[tag] = {"error": o.get("error", "?")[:60]}
        out["summary"] = summ
        full_path = write_full_record(out, args.full_json)
        if not args.full_line:
            out = compact_line(out, full_path)
        # the ONE line of the contract -- at the start of a line of its own even if a library (RCCL prints warnings and its
        # version banner to stdout without a trailing newline) left the cursor elsewhere
        sys.stdout.flush()
        sys.stdout.write("\n" + json.dumps(out) + "\n")
        sys.stdout.flush()
    if world > 1:
        wd.enter("destroy_process_group")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
