"""GPU: the exchange step of BASELINE.json configs[4] on HIP-produced streams.

* one rank, RCCL transport (``scl_rccl_*`` / ``scl_streams_gather_rccl`` of the C ABI; a one-rank communicator is all
  a one-GPU box can build): the gathered buffer and the per-chunk / per-block offsets equal the local compaction, also
  through the overlapped sub-batch pipeline;
* two ranks SHARING the GPU over gloo (RCCL refuses two ranks on one device; the gather then runs on the
  torch.distributed transport): every rank encodes its block-contiguous shard with the HIP kernels, the root's
  gathered buffers equal what one process produces for the whole batch.
The 1 -> 8 GPU curve over xGMI is the driver's to measure."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_rccl_transport_one_rank():
    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend import lib, models
    from stanford_compression_library_amd.backend.sharded import (RcclGather, block_offsets, encode_gather_overlapped,
                                                                  gather_streams_to_root)

    lib.require_device()
    dev = torch.device("cuda:0")
    freq = bench_data.t256_table()
    model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
    n_chunks, chunk_len = 1024, 512
    sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=77, device=dev)
    enc = model.encode_batch(sym)
    dense, offs = models.compact(enc)
    total = int(offs[-1])
    comm = RcclGather(1, 0, dev)
    try:
        got_total, out, goffs = gather_streams_to_root(dense, offs, 1, 0, dev, return_data=True, comm=comm)
        assert got_total == total and torch.equal(out, dense[:total]) and torch.equal(goffs, offs)
        # per-block offsets of configs[4]: one block = 256 chunks
        blocks = block_offsets(goffs, 256)
        assert blocks.numel() == 5 and int(blocks[0]) == 0 and int(blocks[-1]) == total
        assert torch.equal(blocks[:-1], offs[0:n_chunks:256])
        # overlapped pipeline: 4 sub-batches, each gathered while the next one encodes
        timings, parts = encode_gather_overlapped(model, sym, 1, 0, n_sub=4, comm=comm)
        assert timings["gathered_bytes"] == total and len(parts) == 4
        cat = torch.cat([p[0] for p in parts])
        assert torch.equal(cat, dense[:total])
        base = 0
        for i, (data, o) in enumerate(parts):
            lo = n_chunks * i // 4
            assert torch.equal(o[:-1] + base, offs[lo:lo + o.numel() - 1])
            base += int(o[-1])
    finally:
        comm.close()


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend import models
    from stanford_compression_library_amd.backend.sharded import shard_range, encode_gather_overlapped, gather_streams_to_root
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    freq = bench_data.t256_table()
    model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
    n_chunks, chunk_len = 1000, 384
    sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=31, device=dev)   # same on both ranks
    a, b = shard_range(n_chunks, world, rank)
    # (1) plain gather of the rank's whole shard
    enc = model.encode_batch(sym[a:b])
    dense, offs = models.compact(enc)
    total, out, goffs = gather_streams_to_root(dense, offs, world, rank, dev, return_data=True)
    # (2) the overlapped pipeline, 3 sub-batches per rank
    timings, parts = encode_gather_overlapped(model, sym[a:b], world, rank, n_sub=3)
    if rank == 0:
        ref_enc = model.encode_batch(sym)
        ref, ref_offs = models.compact(ref_enc)
        n = int(ref_offs[-1])
        assert total == n and torch.equal(out, ref[:n]) and torch.equal(goffs, ref_offs), "plain gather differs"
        # sub-batch i of the pipeline holds [rank 0 sub i | rank 1 sub i]: check every chunk against the reference
        ref_cpu, ref_o = ref.cpu().numpy(), ref_offs.cpu().numpy()
        seen = 0
        for i, (data, o) in enumerate(parts):
            data, o = data.cpu().numpy(), o.cpu().numpy()
            idx = []
            for r in range(world):
                ra, rb = shard_range(n_chunks, world, r)
                m = rb - ra
                idx += list(range(ra + m * i // 3, ra + m * (i + 1) // 3))
            assert len(idx) == o.size - 1
            for k, c in enumerate(idx):
                assert np.array_equal(data[o[k]:o[k + 1]], ref_cpu[ref_o[c]:ref_o[c + 1]]), (i, k, c)
            seen += len(idx)
        assert seen == n_chunks and timings["gathered_bytes"] == n
        print("SHARDED_GPU_OK", n, timings)
    dist.destroy_process_group()
""")


def test_two_ranks_share_the_gpu_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDED_GPU_OK" in outs[0]
