"""Multi-GPU layer of the path: shard, encode locally, (optionally) gather the streams.

Chunks share nothing but read-only model tables, so the path shards embarrassingly
(SURVEY.md section 8e): rank r owns a block-contiguous range of chunks, runs the same single-GPU
kernels on it, and no collective sits on the data path.  The only exchange step is the optional final
*variable-length gather* of the compacted per-chunk streams to one rank (BASELINE.json configs[4]):
an all_gather of byte counts followed by direct point-to-point transfers into the root's buffer at the
prefix offsets -- over RCCL each sender uses its own xGMI link to the root, so no ring is involved.

Everything here is backend-agnostic ``torch.distributed`` (backend "nccl" == RCCL on ROCm; "gloo" in the
CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Block-contiguous partition of ``n_units`` (chunks or blocks) over ``world`` ranks; the first
    ``n_units % world`` ranks get one extra unit."""
    base, extra = divmod(int(n_units), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_streams_to_root(dense, offsets, world: int, rank: int, device=None, dst: int = 0,
                           return_data: bool = False):
    """Gather every rank's compacted stream buffer (``dense[:offsets[-1]]``) and per-chunk byte offsets to
    ``dst``.

    Returns the total gathered byte count on every rank; with ``return_data`` the root additionally gets
    ``(bytes_tensor, global_offsets)`` where ``global_offsets`` has one entry per chunk of every rank (rank
    order) plus the grand total, i.e. exactly what a single process would have produced.
    """
    import torch
    import torch.distributed as dist

    device = device if device is not None else dense.device
    n_local = int(offsets.numel()) - 1
    my_bytes = int(offsets[-1].item())
    meta = torch.tensor([my_bytes, n_local], dtype=torch.int64, device=device)
    all_meta = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_meta, meta)
    sizes = [int(m[0].item()) for m in all_meta]
    counts = [int(m[1].item()) for m in all_meta]
    total = sum(sizes)
    byte_base = [sum(sizes[:r]) for r in range(world)]
    out = goffs = None
    if rank == dst:
        out = torch.empty(total, dtype=torch.uint8, device=device)
        goffs = torch.empty(sum(counts) + 1, dtype=torch.int64, device=device)
        ops, pos = [], 0
        for r in range(world):
            if r == dst:
                out[byte_base[r]:byte_base[r] + sizes[r]] = dense[:sizes[r]]
                goffs[pos:pos + counts[r]] = offsets[:counts[r]] + byte_base[r]
            else:
                if sizes[r]:
                    ops.append(dist.P2POp(dist.irecv, out[byte_base[r]:byte_base[r] + sizes[r]], r))
                ops.append(dist.P2POp(dist.irecv, goffs[pos:pos + counts[r]], r))
            pos += counts[r]
        goffs[-1] = total
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        # offsets of remote ranks arrive relative to their own buffers
        pos = 0
        for r in range(world):
            if r != dst:
                goffs[pos:pos + counts[r]] += byte_base[r]
            pos += counts[r]
    else:
        ops = []
        if my_bytes:
            ops.append(dist.P2POp(dist.isend, dense[:my_bytes].contiguous(), dst))
        ops.append(dist.P2POp(dist.isend, offsets[:n_local].contiguous(), dst))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if return_data:
        return total, out, goffs
    return total
