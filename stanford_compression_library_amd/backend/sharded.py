"""Multi-GPU layer of the path: shard, encode locally, (optionally) gather the streams.

Chunks share nothing but read-only model tables, so the path shards embarrassingly
(SURVEY.md section 8e): rank r owns a block-contiguous range of chunks, runs the same single-GPU
kernels on it, and no collective sits on the data path.  The only exchange step is the optional final
*variable-length gather* of the compacted per-chunk streams to one rank (BASELINE.json configs[4]):
the byte counts travel first, then every sender's payload goes straight into the root's buffer at the prefix
offsets -- over RCCL each sender uses its own xGMI link to the root, so no ring is involved.

Two transports, same result:
* ``RcclGather`` -- the C ABI's ``scl_rccl_*`` / ``scl_streams_gather_rccl`` (csrc/scl_gather.hip: grouped
  ncclSend / ncclRecv on a HIP stream of our choice, which is what lets a gather overlap the next encode);
  used when the process group runs on RCCL ("nccl" backend);
* plain ``torch.distributed`` point-to-point ops -- any backend, "gloo" in the CPU tests.
``encode_gather_overlapped`` is configs[4] end to end: the rank's shard is cut into sub-batches, and sub-batch i
travels on the communication stream while sub-batch i + 1 is encoded and compacted on the compute stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np


_TRACE = None  # tools/time_gather.py: list of (label, ms since the pipeline started) when set


def _mark(label, t0):
    if _TRACE is not None:
        import time

        _TRACE.append((label, (time.perf_counter() - t0) * 1e3))


def shard_range(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Block-contiguous partition of ``n_units`` (chunks or blocks) over ``world`` ranks; the first
    ``n_units % world`` ranks get one extra unit."""
    base, extra = divmod(int(n_units), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def block_offsets(chunk_offsets, chunks_per_block: int):
    """Byte offset of every block of ``chunks_per_block`` chunks (configs[4]: 1 MiB blocks = 256 chunks of 4 KiB)
    from the per-chunk offsets ``scl_streams_compact`` returns (n + 1 entries); the last entry stays the total."""
    n = int(chunk_offsets.numel()) - 1
    idx = list(range(0, n, int(chunks_per_block))) + [n]
    return chunk_offsets[idx]


class RcclGather:
    """Communicator of ``scl_rccl_*`` over the ranks of the default ``torch.distributed`` group.  The 128-byte id is
    created on rank 0 and handed out through the group (any backend); the communicator belongs to ``device``."""

    def __init__(self, world: int, rank: int, device):
        import torch
        import torch.distributed as dist

        from . import lib as _lib

        self._L = _lib.load()
        self.world, self.rank, self.device = int(world), int(rank), device
        ident = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            _lib.check(self._L.scl_rccl_unique_id(_lib.u8_ptr(ident)), "scl_rccl_unique_id")
        if world > 1:
            box = [ident.tobytes() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ident = np.frombuffer(box[0], dtype=np.uint8).copy()
        self._h = C.c_void_p()
        with torch.cuda.device(device):
            rc = self._L.scl_rccl_comm_create(_lib.u8_ptr(ident), self.rank, self.world, C.byref(self._h))
        _lib.check(rc, "scl_rccl_comm_create")
        self._check = _lib.check

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.scl_rccl_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        """what the communicator reports about itself (``ncclCommUserRank`` / ``ncclCommCount``) and its device"""
        r, n, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        self._check(self._L.scl_rccl_comm_info(self._h, C.byref(r), C.byref(n), C.byref(d)), "scl_rccl_comm_info")
        return dict(rank=r.value, nranks=n.value, device=d.value)

    @property
    def nranks(self) -> int:
        return self.info()["nranks"]

    def counts(self, value: int, stream=None) -> np.ndarray:
        """one u64 of every rank, in rank order (collective; synchronises the stream)"""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        counts = np.zeros(self.world, dtype=np.uint64)
        with torch.cuda.device(self.device):
            rc = self._L.scl_rccl_allgather_u64(self._h, int(value), counts.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                st.cuda_stream)
        self._check(rc, "scl_rccl_allgather_u64")
        return counts

    def sizes(self, nbytes: int, stream=None) -> np.ndarray:
        """all ranks' byte counts as exclusive prefix offsets [world + 1] (collective; synchronises the stream)"""
        return np.concatenate([[0], np.cumsum(self.counts(nbytes, stream))]).astype(np.uint64)

    def allgather_async(self, d_in, d_out, stream=None):
        """asynchronous on ``stream``, device to device: every rank's int64 tensor ``d_in`` [n] -> ``d_out`` [world, n]"""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        if d_in.dtype != torch.int64 or d_out.dtype != torch.int64 or d_out.numel() != self.world * d_in.numel():
            # (not an assert: `python -O` strips those, and a wrong size here is a write past a device buffer)
            raise ValueError(f"allgather_async: int64 tensors of n and world * n elements wanted, got {d_in.dtype} "
                             f"[{d_in.numel()}] and {d_out.dtype} [{d_out.numel()}] for world = {self.world}")
        with torch.cuda.device(self.device):
            rc = self._L.scl_rccl_allgather_async(self._h, d_in.data_ptr(), d_out.data_ptr(), d_in.numel(), st.cuda_stream)
        self._check(rc, "scl_rccl_allgather_async")

    def gatherv(self, parts, root: int = 0, stream=None):
        """several variable-length gathers in ONE grouped exchange, asynchronous on ``stream``.  ``parts``: list of
        (payload uint8 tensor or None, nbytes, offsets uint64 [world + 1], out uint8 tensor or None)."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        n = len(parts)
        send = (C.c_void_p * n)(*[p[0].data_ptr() if (p[0] is not None and p[1]) else None for p in parts])
        recv = (C.c_void_p * n)(*[p[3].data_ptr() if p[3] is not None else None for p in parts])
        nbytes = (C.c_uint64 * n)(*[int(p[1]) for p in parts])
        offs = np.ascontiguousarray(np.stack([np.asarray(p[2], dtype=np.uint64) for p in parts]))
        if offs.shape != (n, self.world + 1):
            raise ValueError(f"gatherv: every part needs world + 1 = {self.world + 1} offsets, got an array of shape {offs.shape}")
        with torch.cuda.device(self.device):
            rc = self._L.scl_streams_gatherv_rccl(self._h, int(root), n, send, nbytes, recv,
                                                  offs.ctypes.data_as(C.POINTER(C.c_uint64)), st.cuda_stream)
        self._check(rc, "scl_streams_gatherv_rccl")

    def gather(self, payload, nbytes: int, offsets: np.ndarray, out=None, root: int = 0, stream=None):
        """asynchronous on ``stream``: this rank's ``payload[:nbytes]`` (uint8, device) -> ``out[offsets[rank]:]`` on the root"""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        with torch.cuda.device(self.device):
            rc = self._L.scl_streams_gather_rccl(self._h, int(root), payload.data_ptr() if nbytes else None, int(nbytes),
                                                 out.data_ptr() if out is not None else None,
                                                 offs.ctypes.data_as(C.POINTER(C.c_uint64)), st.cuda_stream)
        self._check(rc, "scl_streams_gather_rccl")


def _gather_torch(payload, nbytes, world, rank, device, dst, counts):
    """torch.distributed transport of one variable-length gather; returns the root's buffer (None elsewhere)"""
    import torch
    import torch.distributed as dist

    base = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    out = None
    if rank == dst:
        out = torch.empty(int(base[-1]), dtype=payload.dtype, device=device)
        ops = []
        for r in range(world):
            if r == dst:
                out[base[r]:base[r + 1]] = payload[:nbytes]
            elif counts[r]:
                ops.append(dist.P2POp(dist.irecv, out[base[r]:base[r + 1]], r))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    elif nbytes:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, payload[:nbytes].contiguous(), dst)]):
            req.wait()
    return out


def gather_streams_to_root(dense, offsets, world: int, rank: int, device=None, dst: int = 0,
                           return_data: bool = False, comm: Optional[RcclGather] = None):
    """Gather every rank's compacted stream buffer (``dense[:offsets[-1]]``) and per-chunk byte offsets to
    ``dst``.

    Returns the total gathered byte count on every rank; with ``return_data`` the root additionally gets
    ``(bytes_tensor, global_offsets)`` where ``global_offsets`` has one entry per chunk of every rank (rank
    order) plus the grand total, i.e. exactly what a single process would have produced.
    ``comm``: an :class:`RcclGather` -- the transfers then run through the C ABI on the current HIP stream.
    """
    import torch
    import torch.distributed as dist

    device = device if device is not None else dense.device
    n_local = int(offsets.numel()) - 1
    my_bytes = int(offsets[-1].item())
    if comm is not None:
        # one collective for both numbers: payload bytes in the low 40 bits, chunk count above
        if my_bytes >= (1 << 40) or n_local >= (1 << 24):
            raise ValueError(f"gather_streams_to_root: {my_bytes} bytes / {n_local} chunks per rank exceed the packed count "
                             "(40 bits of bytes, 24 bits of chunks)")
        packed = comm.counts(my_bytes | (n_local << 40))
        sizes, counts = (packed & np.uint64((1 << 40) - 1)).astype(np.int64), (packed >> np.uint64(40)).astype(np.int64)
        base = np.concatenate([[0], np.cumsum(sizes)])
        cbase = np.concatenate([[0], np.cumsum(counts)])
    else:
        meta = torch.tensor([my_bytes, n_local], dtype=torch.int64, device=device)
        all_meta = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(all_meta, meta)
        sizes = np.array([int(m[0].item()) for m in all_meta], dtype=np.int64)
        counts = np.array([int(m[1].item()) for m in all_meta], dtype=np.int64)
        base = np.concatenate([[0], np.cumsum(sizes)])
        cbase = np.concatenate([[0], np.cumsum(counts)])
    total = int(base[-1])
    out = goffs = None
    local_offs = offsets[:n_local].contiguous()
    if comm is not None:
        if rank == dst:
            out = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
            goffs = torch.empty(int(cbase[-1]) + 1, dtype=torch.int64, device=device)
        comm.gatherv([(dense, my_bytes, base.astype(np.uint64), out),
                      (local_offs.view(torch.uint8), 8 * n_local, (8 * cbase).astype(np.uint64),
                       goffs.view(torch.uint8) if goffs is not None else None)], root=dst)
        torch.cuda.current_stream(device).synchronize()
        if out is not None:
            out = out[:total]
    else:
        out = _gather_torch(dense, my_bytes, world, rank, device, dst, sizes)
        g = _gather_torch(local_offs, n_local, world, rank, device, dst, counts)
        if rank == dst:
            goffs = torch.empty(int(cbase[-1]) + 1, dtype=torch.int64, device=device)
            goffs[:-1] = g
    if rank == dst:
        # offsets arrive relative to their own rank's buffer
        for r in range(world):
            goffs[cbase[r]:cbase[r + 1]] += int(base[r])
        goffs[-1] = total
    if return_data:
        return total, out, goffs
    return total


class GatherWorkspace:
    """Everything :func:`encode_gather_overlapped` needs that does not depend on the data: the two streams, one
    worst-case buffer per kind for the whole shard (sub-batch i owns a slice of each), the pinned read-back area for
    the counts.  Build it once per (model, shard shape) and hand it to every call -- the pipeline itself then allocates
    nothing but the root's exactly-sized receive buffers."""

    def __init__(self, model, n_chunks: int, chunk_len: int, world: int, device, n_sub: Optional[int] = None,
                 framed: bool = False):
        import torch

        from .models import compact_capacity, compact_scratch_bytes

        n_sub = default_sub_batches(n_chunks) if n_sub is None else int(n_sub)
        self.n_chunks, self.chunk_len, self.world, self.n_sub, self.framed = n_chunks, chunk_len, world, n_sub, framed
        self.device = dev = device
        self.bounds = b = [n_chunks * i // n_sub for i in range(n_sub + 1)]
        self.sizes = [b[i + 1] - b[i] for i in range(n_sub)]
        self.comp, self.comm_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.stride = model.slot_bytes(chunk_len)
        self.enc = model.alloc_encoded(n_chunks, chunk_len, dev, self.stride)
        self.caps = [(compact_capacity(n, self.stride, framed) + 255) // 256 * 256 for n in self.sizes]
        self.cap_base = [0]
        for c in self.caps:
            self.cap_base.append(self.cap_base[-1] + c)
        self.dense = torch.empty(self.cap_base[-1], dtype=torch.uint8, device=dev)
        # offsets of sub-batch i: entries [p_i, p_i + n_i] (n_i + 1 of them), then ONE slot holding n_i, so that
        # {payload bytes, chunks} are two consecutive u64 -- the all-gather's input, no copy kernel
        self.pos = [b[i] + 2 * i for i in range(n_sub)]
        offs = torch.zeros(n_chunks + 2 * n_sub, dtype=torch.int64)
        for i in range(n_sub):
            offs[self.pos[i] + self.sizes[i] + 1] = self.sizes[i]
        self.offs = offs.to(dev)
        self.scr = (compact_scratch_bytes(max(self.sizes + [1])) + 255) // 256 * 256
        self.scratch = torch.empty(self.scr * n_sub, dtype=torch.uint8, device=dev)
        self.all_meta = torch.empty((n_sub, world, 2), dtype=torch.int64, device=dev)
        self.h_meta = torch.empty((n_sub, world, 2), dtype=torch.int64, pin_memory=True)
        torch.cuda.current_stream(dev).synchronize()

    def matches(self, model, sym, world, n_sub, framed) -> bool:
        return (tuple(sym.shape) == (self.n_chunks, self.chunk_len) and sym.device == self.device and world == self.world
                and n_sub == self.n_sub and framed == self.framed and model.slot_bytes(self.chunk_len) == self.stride)


def default_sub_batches(n_chunks: int) -> int:
    """sub-batches of at least 128 Ki chunks (two waves per SIMD on 256 CUs: below that the lane-per-chunk encoders run
    at half their rate, profiles/r02_bench_config2_64Ki.json), at most 4"""
    return max(1, min(4, int(n_chunks) // (128 * 1024)))


def encode_gather_overlapped(model, sym, world: int, rank: int, n_sub: Optional[int] = None, dst: int = 0,
                             comm: Optional[RcclGather] = None, framed: bool = False,
                             workspace: Optional[GatherWorkspace] = None):
    """configs[4] end to end for this rank's shard ``sym`` (uint8 [n_chunks, chunk_len] on the device): the shard is
    cut into ``n_sub`` sub-batches; ALL of them are queued on the compute stream up front (encode + compaction into
    the workspace's worst-case buffers: nothing there ever waits for the host), and sub-batch i travels to ``dst`` on
    the communication stream while the later ones are still being coded.

    Host traffic per sub-batch with an :class:`RcclGather`: ONE asynchronous device-to-device all-gather carrying
    ``{payload bytes, chunks}`` of every rank (read straight out of the compaction's offset table), ONE wait on the event
    behind its read-back (the root must know the counts to post its receives), ONE call that moves the payload and the
    per-chunk offset table in one grouped send / receive and globalises the offsets on the root
    (``scl_streams_gather_blocks_rccl``).  No ``.item()``, no stream synchronisation inside the loop.  The all-gather of
    sub-batch i + 1 is queued BEFORE the exchange of sub-batch i, so its counts are already on the host when the
    exchange of i ends and the communication stream never idles on a host round trip.  Default: sub-batches of at least
    128 Ki chunks, at most 4 (``default_sub_batches``) -- smaller ones leave the encoder one wave per SIMD, and over
    xGMI half a 1 GiB shard (~0.5 GiB of streams, ~3 ms per link) already hides the ~0.5 ms it takes to code the
    other half; every sub-batch costs the host ~0.1 ms (profiles/r03_gather_host_trace.txt).
    Without an RcclGather (gloo in the CPU tests, or a single process) the per-sub-batch exchange goes through
    :func:`gather_streams_to_root`.  ``workspace``: a :class:`GatherWorkspace` to reuse (else one is built, untimed).

    Returns (timings dict, on the root a list of per-sub-batch (bytes, global_offsets) pairs in sub-batch order -- the
    root's buffer for sub-batch i holds the ranks' sub-batch-i payloads in rank order --, None elsewhere).  The pairs
    of a non-RCCL run are views of the workspace: valid until its next use.
    """
    import time

    import torch

    from . import lib as _lib

    L = _lib.load()
    dev = sym.device
    n_chunks, chunk_len = sym.shape
    n_sub = default_sub_batches(n_chunks) if n_sub is None else int(n_sub)
    if n_chunks and (sym.stride(0) % 16 or sym.data_ptr() % 16):  # rows on 16-byte boundaries (encode_rows_into)
        padded = torch.empty((n_chunks, (chunk_len + 15) // 16 * 16), dtype=torch.uint8, device=dev)
        padded[:, :chunk_len] = sym
        sym = padded[:, :chunk_len]
    ws = workspace
    if ws is None or not ws.matches(model, sym, world, n_sub, framed):
        ws = GatherWorkspace(model, n_chunks, chunk_len, world, dev, n_sub, framed)
    bounds, pos, enc, stride = ws.bounds, ws.pos, ws.enc, ws.stride
    comp, comm_stream = ws.comp, ws.comm_stream
    mode = _lib.COMPACT_FRAMED if framed else _lib.COMPACT_DENSE
    t0 = time.perf_counter()
    comp.wait_stream(torch.cuda.current_stream(dev))
    comm_stream.wait_stream(torch.cuda.current_stream(dev))

    # ---- compute stream: everything, now ----------------------------------------------------------------------
    done = []
    with torch.cuda.device(dev):
        for i in range(n_sub):
            a, b = bounds[i], bounds[i + 1]
            model.encode_rows_into(sym, a, b, enc, comp.cuda_stream)
            rc = L.scl_streams_compact(enc.data.data_ptr() + a * stride, enc.bit_offset.data_ptr() + 8 * a,
                                       enc.nbits.data_ptr() + 4 * a, b - a, mode, ws.dense.data_ptr() + ws.cap_base[i],
                                       ws.caps[i], ws.offs.data_ptr() + 8 * pos[i], ws.scratch.data_ptr() + ws.scr * i,
                                       comp.cuda_stream)
            _lib.check(rc, "scl_streams_compact")
            ev = torch.cuda.Event()
            ev.record(comp)
            done.append(ev)
            _mark(f"compute {i} queued", t0)

    results = []
    if comm is not None:
        # ---- RCCL: all-gather(i + 1) is queued ahead of exchange(i) ------------------------------------------
        sizes_ready = []

        def queue_sizes(i):
            comm_stream.wait_event(done[i])
            with torch.cuda.device(dev):
                rc = L.scl_rccl_allgather_async(comm._h, ws.offs.data_ptr() + 8 * (pos[i] + ws.sizes[i]),
                                                ws.all_meta[i].data_ptr(), 2, comm_stream.cuda_stream)
            _lib.check(rc, "scl_rccl_allgather_async")
            with torch.cuda.stream(comm_stream):
                ws.h_meta[i].copy_(ws.all_meta[i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(comm_stream)
            sizes_ready.append(ev)

        queue_sizes(0)
        u64p = C.POINTER(C.c_uint64)
        for i in range(n_sub):
            if i + 1 < n_sub:
                queue_sizes(i + 1)
            _mark(f"sizes {i + 1} queued", t0)
            sizes_ready[i].synchronize()  # the one host wait of this sub-batch: the counts of every rank
            _mark(f"sizes {i} on host", t0)
            counts = ws.h_meta[i].numpy().astype(np.uint64)
            nbytes_r, chunks_r = np.ascontiguousarray(counts[:, 0]), np.ascontiguousarray(counts[:, 1])
            total = int(nbytes_r.sum())
            out = goffs = None
            if rank == dst:
                with torch.cuda.stream(comm_stream):
                    out = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
                    goffs = torch.empty(int(chunks_r.sum()) + 1, dtype=torch.int64, device=dev)
            with torch.cuda.device(dev):
                rc = L.scl_streams_gather_blocks_rccl(
                    comm._h, int(dst), ws.dense.data_ptr() + ws.cap_base[i], int(nbytes_r[rank]),
                    ws.offs.data_ptr() + 8 * pos[i], ws.sizes[i], out.data_ptr() if out is not None else None,
                    goffs.data_ptr() if goffs is not None else None, nbytes_r.ctypes.data_as(u64p),
                    chunks_r.ctypes.data_as(u64p), comm_stream.cuda_stream)
            _lib.check(rc, "scl_streams_gather_blocks_rccl")
            results.append((total, out[:total] if out is not None else None, goffs))
            _mark(f"exchange {i} queued", t0)
    else:
        # ---- any torch.distributed backend (gloo in the tests) / a single process ---------------------------
        for i in range(n_sub):
            dense = ws.dense[ws.cap_base[i]:ws.cap_base[i] + ws.caps[i]]
            offs = ws.offs[pos[i]:pos[i] + ws.sizes[i] + 1]
            done[i].synchronize()
            comm_stream.wait_event(done[i])
            with torch.cuda.stream(comm_stream):
                if world > 1:
                    results.append(gather_streams_to_root(dense, offs, world, rank, dev, dst, return_data=True))
                else:
                    n = int(offs[-1].item())
                    results.append((n, dense[:n], offs))
    comm_stream.synchronize()
    comp.synchronize()
    total_ms = (time.perf_counter() - t0) * 1e3
    cur = torch.cuda.current_stream(dev)
    cur.wait_stream(comm_stream)
    # the root's receive tensors were allocated on the communication stream and are handed to a caller who works on the
    # current one: tell the caching allocator, or a block freed by the caller could be handed out again on the
    # communication stream while the caller's kernels still read it
    for r in results:
        for t in r[1:]:
            if t is not None and t.is_cuda:
                t.record_stream(cur)
    timings = {"overlapped_ms": round(total_ms, 3), "sub_batches": n_sub,
               "gathered_bytes": int(sum(r[0] for r in results))}
    return timings, ([(r[1], r[2]) for r in results] if rank == dst else None)
