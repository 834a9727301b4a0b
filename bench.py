#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X entropy-coding core.

Metric (BASELINE.json): MB/s encode+decode of 1 GiB i.i.d. bytes with 256-symbol static rANS
(reference defaults NUM_BITS_OUT=1, RANGE_FACTOR=2^16, M=4096), plus achieved HBM GB/s vs peak.

One "step" = one full encode pass + one full decode pass over the per-GPU batch
(262 144 chunks x 4 KiB = 1 GiB, one wavefront lane per chunk), inputs resident in HBM.
value = (bytes of all ranks * steps) / wall time of the timed region / 1e6, i.e. N / (t_enc + t_dec).

    python bench.py                      # 1 GPU, finishes in a few minutes incl. the CPU baseline
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: chunks are independent, so every rank encodes/decodes its own 1 GiB shard (weak scaling,
no data-path collective); the only communication is the barrier / max-reduction of the timing.
`--gather` additionally times the optional final gather of the compacted streams to rank 0
(BASELINE.json configs[4]) and reports it separately; it never enters `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the clocks to settle (10 steps after 3 read ~5 % low: 840 vs 885 GB/s), still < 0.1 s
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time compaction + gather to rank 0")
    ap.add_argument("--sym-pad", type=int, default=0, help="experiment: extra bytes between input rows")
    ap.add_argument("--source", choices=["iid", "markov1"], default="iid",
                    help="static-model coders: i.i.d. symbols with p = f/M of --table (the headline), or an order-1 Markov "
                         "byte source (north_star's second source) coded with the table its own histogram gives "
                         "(scl_histogram_u8 + normalize_counts, M = 4096)")
    return ap.parse_args()


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


def make_model(args, freq):
    from stanford_compression_library_amd.backend import models

    if args.coder == "rans":
        return (models.RansModel(freq.tolist(), args.range_factor, args.num_bits_out, 32),
                dict(NUM_BITS_OUT=args.num_bits_out, RANGE_FACTOR=args.range_factor))
    if args.coder == "tans":
        return models.TansModel(freq.tolist(), 1, 32), dict(NUM_BITS_OUT=1, RANGE_FACTOR=1)
    if args.coder == "aec" and args.aec_model == "fixed":
        return (models.AecModel(0, freq.tolist(), int(freq.size), 0, 1 << 30, 32, 32),
                dict(PRECISION=32, model=f"FixedFreqModel({args.table})"))
    if args.coder == "aec" and args.aec_model == "iid":
        return (models.AecModel(1, [1] * int(freq.size), int(freq.size), 0, 1 << 30, 32, 32),
                dict(PRECISION=32, model=f"AdaptiveIIDFreqModel(all ones, K={int(freq.size)}) on {args.table} symbols"))
    if args.coder == "aec":
        K = args.aec_K
        return (models.AecModel(backend_lib_consts()["MODEL_ORDERK"], None, K, 1, 1 << 30, 32, 32),
                dict(PRECISION=32, model=f"AdaptiveOrderKFreqModel(k=1, K={K})"))
    return models.RangeModel(freq.tolist(), 32, 32), dict(PRECISION=32)


def backend_lib_consts():
    from stanford_compression_library_amd.backend import lib

    return {"MODEL_ORDERK": lib.MODEL_ORDERK}


def cpu_baseline(args, freq, sym_dev, enc, target_seconds=10.0):
    """Times the CPU oracle (C restatement, one thread) on a bounded sample of the SAME workload and, as a
    by-product, checks the GPU streams of those chunks bit-for-bit against it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import scl_oracle as orc

    if args.coder != "rans":
        return None
    from concurrent.futures import ThreadPoolExecutor

    # the C oracle runs one chunk per call; ctypes releases the GIL, so one Python thread per host core scales
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    n_probe = min(128, sym_dev.shape[0])
    sym = sym_dev[:n_probe].cpu().numpy()
    t0 = time.perf_counter()
    kw = dict(RF=args.range_factor, b=args.num_bits_out)
    streams, nbits = orc.rans_encode_batch(sym, freq, **kw)
    orc.rans_decode_batch(streams, nbits, freq, sym.shape[1], **kw)
    per_chunk = (time.perf_counter() - t0) / n_probe
    single_thread = sym.size / (per_chunk * n_probe) / 1e6
    n = int(max(n_probe, min(sym_dev.shape[0], 65536, cores * target_seconds / max(per_chunk, 1e-9))))
    sym = sym_dev[:n].cpu().numpy()
    bounds = [n * i // cores for i in range(cores + 1)]
    parts = [(bounds[i], bounds[i + 1]) for i in range(cores) if bounds[i + 1] > bounds[i]]
    with ThreadPoolExecutor(max_workers=len(parts)) as pool:
        t0 = time.perf_counter()
        enc_parts = list(pool.map(lambda ab: orc.rans_encode_batch(sym[ab[0]:ab[1]], freq, **kw), parts))
        t1 = time.perf_counter()
        dec_parts = list(pool.map(lambda i: orc.rans_decode_batch(enc_parts[i][0], enc_parts[i][1], freq, sym.shape[1], **kw),
                                  range(len(parts))))
        t2 = time.perf_counter()
    streams = np.concatenate([e[0] for e in enc_parts])
    nbits = np.concatenate([e[1] for e in enc_parts])
    dec = np.concatenate([d[0] for d in dec_parts])
    used = np.concatenate([d[1] for d in dec_parts])
    assert np.array_equal(dec, sym) and np.array_equal(used, nbits)
    # parity by-product: GPU streams of the sampled chunks == oracle streams
    g_nbits = enc.nbits[:n].cpu().numpy().astype(np.uint64)
    assert np.array_equal(g_nbits, nbits), "GPU/oracle stream lengths differ"
    data = enc.data.cpu().numpy() if n * enc.stride < (1 << 31) else None
    if data is not None:
        offs = enc.bit_offset[:n].cpu().numpy()
        for c in range(0, n, max(1, n // 64)):
            nb = int(nbits[c])
            got = np.unpackbits(data[int(offs[c]) // 8:(int(offs[c]) + nb + 7) // 8 + 1])
            lo = int(offs[c]) % 8
            assert np.array_equal(got[lo:lo + nb], np.unpackbits(streams[c])[:nb]), f"chunk {c}: GPU != oracle"
    nbytes = sym.size
    return {
        "value": round(nbytes / (t2 - t0) / 1e6, 3), "unit": "MB/s", "cores": len(parts), "kind": "port",
        # which baseline this is: a C restatement of the reference's algorithm -- NOT the reference's speed (the reference is
        # pure Python and runs ~25 000 x slower per core: cpu_baseline_restatement below, calibrated in BASELINE.md 4.2)
        "kind_note": "C port of the algorithm (oracle/scl_oracle.c), a 'reasonable CPU' line; the reference itself is pure "
                     "Python: see cpu_baseline_restatement for its speed",
        "sample": f"{n} chunks x {sym.shape[1]} B of the same batch ({nbytes / 2**20:.1f} MiB), oracle/scl_oracle.c "
                  f"-O2, {len(parts)} threads (one per host core), encode {nbytes / (t1 - t0) / 1e6:.2f} MB/s + decode "
                  f"{nbytes / (t2 - t1) / 1e6:.2f} MB/s aggregate; one thread alone: {single_thread:.2f} MB/s round trip",
        "single_thread_MBps": round(single_thread, 3),
        "encode_MBps": round(nbytes / (t1 - t0) / 1e6, 3), "decode_MBps": round(nbytes / (t2 - t1) / 1e6, 3),
        "gpu_streams_checked_against_oracle": True,
    }


def restatement_baseline(args, freq, sym_dev, enc):
    """The reference-style baseline SURVEY 8d asks for: oracle/scl_restatement.py -- pure Python, one step per symbol,
    the reference's algorithmic shape -- on every host core with multiprocessing, 8 chunks per worker.  Its streams are
    compared with the GPU's for the same chunks.  BASELINE.md 4.2: the imported reference runs 0.96-1.0x as fast."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import scl_restatement as rst

    if args.coder != "rans":
        return None
    # bounded: <= 32 workers x 4 chunks (a 4 KiB chunk takes ~0.4 s per direction-pair in pure Python)
    cores = min(len(os.sched_getaffinity(0)), 32)
    per = 4 if args.chunk_len <= 4096 else 1
    n = min(sym_dev.shape[0], cores * per)
    sym = sym_dev[:n].cpu().numpy()
    r = rst.timed_baseline(freq, sym, args.range_factor, args.num_bits_out, workers=cores, chunks_per_worker=per)
    assert r["ok"], "restatement round trip failed"
    data = enc.data.cpu().numpy() if n * enc.stride < (1 << 31) else None
    if data is not None:
        offs, nbits = enc.bit_offset[:r["chunks"]].cpu().numpy(), enc.nbits[:r["chunks"]].cpu().numpy()
        for c, (nb, payload) in enumerate(r["streams"]):
            assert nb == int(nbits[c]), f"chunk {c}: GPU/restatement stream lengths differ"
            got = np.unpackbits(data[int(offs[c]) // 8:(int(offs[c]) + nb + 7) // 8 + 1])
            lo = int(offs[c]) % 8
            assert np.array_equal(got[lo:lo + nb], np.unpackbits(np.frombuffer(payload, np.uint8))[:nb]), \
                f"chunk {c}: GPU != restatement"
    return {
        "value": round(r["round_trip_MBps_aggregate"], 4), "unit": "MB/s", "cores": r["workers"], "kind": "restatement",
        "sample": f"{r['chunks']} chunks x {sym.shape[1]} B of the same batch, oracle/scl_restatement.py (pure Python, per-symbol, "
                  f"the reference's algorithmic shape), {r['workers']} worker processes x {per} chunks, wall {r['wall_s']:.1f} s",
        "per_core_MBps": {"encode": round(r["encode_MBps_per_core"], 5), "decode": round(r["decode_MBps_per_core"], 5),
                          "round_trip": round(r["round_trip_MBps_per_core"], 5)},
        "reference_over_restatement": "0.96-1.0 (BASELINE.md 4.2, measured in the build container against the imported reference)",
        "gpu_streams_checked_against_restatement": data is not None,
    }


def rocprof_kernel_names(args, freq):
    """the names rocprofv3 prints for the two kernels of the timed step (so that the line can be matched mechanically
    with profiles/*_kernel_trace_summary.txt); mirrors the launch rules of csrc/scl_rans_fast.hip"""
    if args.coder == "rans" and args.num_bits_out == 1:
        K, M = int(freq.size), int(freq.sum())
        r = int(args.range_factor).bit_length() - 1
        default_shape = (M == 4096 and r == 16)
        check = 0 if K == 256 else (1 if K <= 128 else 2)
        # writer: scl_rans_fast.hip rf_use_slot_writer (fewer rounds with three workgroups per CU, no lockstep table)
        import torch
        w = -(-args.chunks // 256)
        cus = torch.cuda.get_device_properties(0).multi_processor_count if torch.cuda.is_available() else 256
        lockstep = int(freq.max()) == int(freq.min())
        slots = (not lockstep) and 3 * (-(-w // (3 * cus))) < 2 * (-(-w // (2 * cus)))
        if os.environ.get("SCL_RANS_ENC_WRITER", "")[:1].upper() in ("L", "S"):
            slots = os.environ["SCL_RANS_ENC_WRITER"][:1].upper() == "S"
        enc = (f"rans_encode_fast_kernel<AnsBackWriter{'S' if slots else 'L'}<256>, {check}, "
               f"{'10, 16' if default_shape else '0, 0'}>")
        threads = 1024 if args.chunks > 2 * 256 * 256 else 256
        if M & (M - 1):
            dec = f"rans_decode_fast_kernel<-1, 0, {threads}>"
        else:
            dec = f"rans_decode_fast_kernel<{'12, 3' if default_shape else '0, 0'}, {threads}>"
        return enc, dec
    return f"{args.coder}_encode", f"{args.coder}_decode"


def traffic_key(args, freq):
    """what a PMC pass must have been taken on to be quoted for this run (tools/make_traffic_json.py stores it)"""
    key = {"coder": args.coder, "chunks": args.chunks, "chunk_len": args.chunk_len}
    if args.coder == "aec" and args.aec_model == "order1":
        key.update(model="order1", K=args.aec_K)
    else:
        key.update(table=args.table, source=args.source, M=int(freq.sum()))
        if args.coder == "rans":
            key.update(num_bits_out=args.num_bits_out, range_factor=args.range_factor)
        if args.coder == "aec":
            key.update(model=args.aec_model)
    return key


def load_traffic_note(key):
    """HBM traffic per launch measured with rocprofv3 PMC passes (committed under profiles/traffic.json): the entry
    taken on exactly this workload, or None -- a pass is never attached to another workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        for e in json.load(open(path)).get("entries", []):
            if e.get("key") == key:
                return e
    except Exception:
        pass
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend import lib

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)),
                                    device_id=dev)
        world = dist.get_world_size()

    freq = {"t256": bench_data.t256_table, "uniform": bench_data.uniform256_table,
            "uniform1": lambda: np.ones(256, dtype=np.int64)}[args.table]()
    n_chunks, chunk_len = args.chunks, args.chunk_len
    static_model = args.coder != "aec" or args.aec_model in ("fixed", "iid")
    source_note = f"256-symbol static table {args.table} (M={int(freq.sum())}), i.i.d. symbols p=f/M"
    if not static_model:
        # configs[3]: Markov-1 source (S4 of SURVEY 8d), every chunk its own chain, generated on the device: the whole
        # batch is distinct data (a tiled batch would be served from L2 and is not an HBM measurement)
        sym = bench_data.markov1_chunks_device(args.aec_K, n_chunks, chunk_len, seed=4000 + rank, device=dev)
        source_note = f"order-1 adaptive model, K={args.aec_K}, Markov-1 source (all chunks distinct)"
    elif args.source == "markov1":
        # north_star's second source for the static coders: Markov-1 bytes; the table is what the data's own histogram
        # gives (row f3: scl_histogram_u8 + the deterministic normaliser), every symbol present so that f >= 1
        from stanford_compression_library_amd.backend.modeling import histogram_u8, normalize_counts

        sym = bench_data.markov1_chunks_device(256, n_chunks, chunk_len, seed=4000 + rank, device=dev)
        counts = histogram_u8(sym) + 1
        if world > 1:  # every rank codes with the same table: the histogram of all shards
            t = torch.from_numpy(counts).to("cpu" if shared_gpu else dev)
            dist.all_reduce(t)
            counts = t.cpu().numpy()
        freq = normalize_counts(counts, 65536 if args.coder == "range" and args.table == "uniform1" else 4096)
        source_note = (f"order-1 Markov byte source (Dirichlet(0.3) rows, all chunks distinct), static table = normalised "
                       f"histogram of the data (M={int(freq.sum())})")
    else:
        sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=5000 + rank, device=dev)
    model, coder_params = make_model(args, freq)
    if args.sym_pad:
        padded = torch.zeros((n_chunks, chunk_len + args.sym_pad), dtype=torch.uint8, device=dev)
        padded[:, :chunk_len] = sym
        sym = padded[:, :chunk_len]
    enc = model.alloc_encoded(n_chunks, chunk_len, dev)
    dec_out = model.alloc_decoded(n_chunks, chunk_len, dev)

    def step(events=None):
        if events is not None:
            events[0].record()
        model.encode_batch(sym, out=enc)
        if events is not None:
            events[1].record()
        model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec_out)
        if events is not None:
            events[2].record()

    def barrier():
        if world > 1:
            if shared_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(evs[i])  # events sit on torch's current stream, the stream the kernels are launched on
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was just timed (outside the timed region) ------------------------------
    dec_sym, dec_lens, dec_used, dec_status = dec_out
    ok = (int(enc.status.abs().sum()) == 0 and int(dec_status.abs().sum()) == 0
          and torch.equal(dec_sym[:, :chunk_len], sym) and torch.equal(dec_used, enc.nbits)
          and int(dec_lens.min()) == chunk_len and int(dec_lens.max()) == chunk_len)
    if not ok:
        raise SystemExit(f"rank {rank}: round trip FAILED (status enc={int(enc.status.abs().sum())} "
                         f"dec={int(dec_status.abs().sum())})")

    enc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    dec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    in_bytes = n_chunks * chunk_len
    stream_bytes = int(((enc.nbits.to(torch.int64) + 7) // 8).sum().item())
    alg_bytes = in_bytes + stream_bytes  # encode = n read + ceil(bits/8) written (decode mirrors it)
    bits_per_symbol = float(enc.nbits.to(torch.float64).mean().item()) / chunk_len

    # The encoders leave every stream in its own slot, described by (bit_offset, nbits) -- the form the decoders read.
    # SURVEY 8d counts the optional left-align / compaction pass with the encode; it is timed here (HIP events, data
    # resident) and reported beside `value`, which it never enters.
    from stanford_compression_library_amd.backend import models as _m

    # caller-owned worst-case buffers (scl_streams_compact never waits for the host; the convenience wrapper
    # models.compact() sizes its output from the bit counts first, which costs a reduction and a host round trip)
    c_dense = torch.empty(_m.compact_capacity(n_chunks, enc.stride), dtype=torch.uint8, device=dev)
    c_offs = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
    c_scratch = torch.empty(_m.compact_scratch_bytes(n_chunks), dtype=torch.uint8, device=dev)
    _m.compact_into(enc, c_dense, c_offs, c_scratch)  # warm
    cev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cev[0].record()
    for _ in range(5):
        _m.compact_into(enc, c_dense, c_offs, c_scratch)
    cev[1].record()
    torch.cuda.synchronize()
    compact_ms = cev[0].elapsed_time(cev[1]) / 5
    assert int(c_offs[-1].item()) == stream_bytes, "compaction total differs from the sum of the stream sizes"
    del c_dense, c_offs, c_scratch

    gather_info = None
    if args.gather:
        # BASELINE.json configs[4]: per-GPU encode, then the gather of the per-block streams to rank 0 over RCCL
        # (scl_streams_gather_rccl of the C ABI; torch.distributed point-to-point on the gloo test path).  Reported
        # beside `value`, never inside it: (a) the three phases one after the other, (b) the same work as a pipeline of
        # sub-batches in which sub-batch i travels while sub-batch i + 1 is encoded and compacted.
        from stanford_compression_library_amd.backend.models import compact
        from stanford_compression_library_amd.backend.sharded import (RcclGather, block_offsets, encode_gather_overlapped,
                                                                      gather_streams_to_root)

        comm = RcclGather(world, rank, dev) if not shared_gpu else None
        dense, offsets = compact(enc)  # untimed: the output buffer comes from the allocator's cache afterwards
        gather_streams_to_root(dense, offsets, world, rank, dev, comm=comm)  # untimed: connections, buffers
        del dense, offsets
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        model.encode_batch(sym, out=enc)
        torch.cuda.synchronize()
        ga = time.perf_counter()
        dense, offsets = compact(enc)
        torch.cuda.synchronize()
        g1 = time.perf_counter()
        total, gathered, goffs = gather_streams_to_root(dense, offsets, world, rank, dev, return_data=True, comm=comm)
        torch.cuda.synchronize()
        barrier()
        g2 = time.perf_counter()
        n_blocks = None
        if rank == 0:
            n_blocks = int(block_offsets(goffs, max(1, (1 << 20) // chunk_len)).numel()) - 1
        del gathered, goffs
        barrier()
        from stanford_compression_library_amd.backend.sharded import GatherWorkspace

        ws = GatherWorkspace(model, n_chunks, chunk_len, world, dev)  # caller-owned buffers, streams, pinned counts
        encode_gather_overlapped(model, sym, world, rank, comm=comm, workspace=ws)  # untimed: connections, allocator
        torch.cuda.synchronize()
        barrier()
        timings, _ = encode_gather_overlapped(model, sym, world, rank, comm=comm, workspace=ws)
        barrier()
        t_ov = torch.tensor([timings["overlapped_ms"]], dtype=torch.float64, device="cpu" if shared_gpu or world == 1 else dev)
        if world > 1:
            dist.all_reduce(t_ov, op=dist.ReduceOp.MAX)
        gather_info = {"transport": "rccl (scl_streams_gather_rccl)" if comm is not None else "torch.distributed p2p",
                       "encode_ms": round((ga - g0) * 1e3, 3), "compact_ms": round((g1 - ga) * 1e3, 3),
                       "gather_ms": round((g2 - g1) * 1e3, 3), "sequential_ms": round((g2 - g0) * 1e3, 3),
                       "overlapped_ms": round(float(t_ov.item()), 3), "sub_batches": timings["sub_batches"],
                       "gathered_bytes": int(total), "blocks_1MiB": n_blocks}
        if comm is not None:
            comm.close()

    if rank == 0:
        total_bytes = in_bytes * world
        value = total_bytes * args.steps / elapsed / 1e6
        tkey = traffic_key(args, freq)
        traffic = load_traffic_note(tkey)

        def roof(ms, name, kernel):
            gbs = alg_bytes / (ms * 1e-3) / 1e9
            t = traffic.get(name.split("_")[-1]) if traffic else None
            return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": t,
                    # not a live counter: HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/
                    "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate run)" if t else None,
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(ms, 4),
                    "read_only_frac": round(in_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if "encode" in name else None}

        k_enc, k_dec = rocprof_kernel_names(args, freq)
        r_enc, r_dec = roof(enc_ms, f"{args.coder}_encode", k_enc), roof(dec_ms, f"{args.coder}_decode", k_dec)
        value_dense = total_bytes * args.steps / (elapsed + args.steps * compact_ms * 1e-3) / 1e6
        out = {
            "metric": "MB/s encode+decode, 1 GiB i.i.d. bytes, 256-sym rANS; achieved HBM GB/s %peak",
            "value": round(value, 2),
            # what `value` is: N / (t_encode + t_decode) with every stream left in its own slot, (bit_offset, nbits) --
            # the form the decoders read.  value_dense: the same with the left-align / compaction pass of SURVEY 8d
            # (scl_streams_compact, BitArray.tobytes() of every stream back to back) counted into the step.
            "value_definition": "slots", "value_dense": round(value_dense, 2),
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"batched {args.coder}: {n_chunks} independent "
                                   f"{chunk_len} B chunks per GPU ({in_bytes / 2**30:.3f} GiB/GPU), one lane per chunk, "
                                   + source_note, "source": args.source if static_model else "markov1",
                       "coder": args.coder, **coder_params, "chunks_per_gpu": n_chunks, "chunk_len": chunk_len,
                       "bits_per_symbol_out": round(bits_per_symbol, 4), "sharding": f"{world} x independent shards"},
            "encode_MBps": round(total_bytes / (enc_ms * 1e-3) / 1e6, 2),
            "decode_MBps": round(total_bytes / (dec_ms * 1e-3) / 1e6, 2),
            "roofline": r_enc if enc_ms >= dec_ms else r_dec,
            "roofline_encode": r_enc, "roofline_decode": r_dec,
            "round_trip_verified": True, "traffic_key": tkey,
            "dense_output": {"compact_ms": round(compact_ms, 4), "compacted_bytes": stream_bytes,
                             "value_incl_compaction_MBps":
                                 round(total_bytes / ((enc_ms + compact_ms + dec_ms) * 1e-3) / 1e6, 2)},
        }
        if gather_info:
            out["gather"] = gather_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, freq, sym, enc)
            out["cpu_baseline_restatement"] = restatement_baseline(args, freq, sym, enc)
        else:
            out["cpu_baseline"] = None
        # the ONE line of the contract -- at the start of a line of its own even if a library (RCCL prints warnings and its
        # version banner to stdout without a trailing newline) left the cursor elsewhere
        sys.stdout.flush()
        sys.stdout.write("\n" + json.dumps(out) + "\n")
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
