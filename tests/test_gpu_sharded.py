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


def test_config5_one_rank_leg_at_full_size():
    """BASELINE.json configs[4], the part one GPU can run AT ITS SIZE (S5 of SURVEY 8d: rank 0's shard = 1024 blocks of
    1 MiB = 262 144 chunks of 4 KiB, seed 5000): per-GPU rANS encode, compaction, the RCCL exchange through the C ABI
    (one-rank communicator) -- size-independent properties: the gathered payload is the compaction byte for byte, the
    per-block offset table has 1024 + 1 increasing entries ending at the total, every chunk decodes back from the
    GATHERED buffer (decoder reading dense streams by their gathered offsets), consumed == produced, and a fixed sample
    of chunks equals the oracle bit for bit."""
    import scl_oracle as orc
    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend import lib, models
    from stanford_compression_library_amd.backend.sharded import RcclGather, block_offsets, gather_streams_to_root

    lib.require_device()
    dev = torch.device("cuda:0")
    freq = bench_data.t256_table()
    model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
    n_chunks, chunk_len = 262144, 4096
    sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=5000, device=dev)
    enc = model.encode_batch(sym)
    dense, offs = models.compact(enc)
    total = int(offs[-1])
    comm = RcclGather(1, 0, dev)
    try:
        assert comm.nranks == 1
        got_total, out, goffs = gather_streams_to_root(dense, offs, 1, 0, dev, return_data=True, comm=comm)
    finally:
        comm.close()
    torch.cuda.synchronize()
    assert got_total == total and torch.equal(goffs, offs) and torch.equal(out[:total], dense[:total])
    blocks = block_offsets(goffs, 256)  # 1 MiB blocks
    assert blocks.numel() == 1025 and int(blocks[0]) == 0 and int(blocks[-1]) == total
    assert bool((blocks[1:] > blocks[:-1]).all())
    # decode from the gathered buffer: stream c = the nbits[c] bits from byte goffs[c] on (DENSE records are left-aligned)
    buf = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
    buf[:total] = out[:total]
    dec, dlens, used, status = model.decode_batch(buf, goffs[:-1] * 8, enc.nbits, chunk_len)
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0 and torch.equal(dec, sym) and torch.equal(used, enc.nbits)
    assert int(dlens.min()) == chunk_len == int(dlens.max())
    g_np, o_np, nb = out.cpu().numpy(), goffs.cpu().numpy(), enc.nbits.cpu().numpy()
    for c in (0, 1, 255, 256, 4095, 131072, n_chunks - 1):
        rb, rn = orc.rans_encode(sym[c].cpu().numpy(), freq)
        assert int(nb[c]) == rn and int(o_np[c + 1] - o_np[c]) == (rn + 7) // 8
        assert np.array_equal(g_np[o_np[c]:o_np[c + 1]], rb[:(rn + 7) // 8]), f"chunk {c}: gathered bytes != oracle"


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend import models
    from stanford_compression_library_amd.backend.sharded import shard_range, encode_gather_overlapped, gather_streams_to_root
    from stanford_compression_library_amd.backend.sharded import RcclGather
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rccl = {rccl!r}
    dev = torch.device("cuda", rank if rccl else 0)
    torch.cuda.set_device(dev)
    if rccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = RcclGather(world, rank, dev) if rccl else None   # a second communicator beside torch's
    freq = bench_data.t256_table()
    model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)   # on cuda:rank -- the handle must record THAT device
    assert model.info().device == dev.index
    n_chunks, chunk_len = 1000, 384
    sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=31, device=dev)   # same on both ranks
    a, b = shard_range(n_chunks, world, rank)
    # (1) plain gather of the rank's whole shard
    enc = model.encode_batch(sym[a:b])
    dense, offs = models.compact(enc)
    total, out, goffs = gather_streams_to_root(dense, offs, world, rank, dev, return_data=True, comm=comm)
    # (2) the overlapped pipeline, 3 sub-batches per rank
    timings, parts = encode_gather_overlapped(model, sym[a:b], world, rank, n_sub=3, comm=comm)
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
    if comm is not None:
        comm.close()
    dist.destroy_process_group()
""")


def _run_ranks(tmp_path, world, rccl):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, rccl=rccl))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p_ in procs:
            outs.append(p_.communicate(timeout=900)[0])
    finally:
        for p_ in procs:  # a rank that failed leaves its peers waiting in a collective: never leave them behind
            if p_.poll() is None:
                p_.kill()
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(o[-1500:] for o in outs)
    assert "SHARDED_GPU_OK" in outs[0]


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_share_the_gpu_gloo(tmp_path, world):
    """the N-rank code path (block-contiguous shards, per-rank encode + compaction on the HIP kernels, the root's global
    offset table, the overlapped pipeline) at world sizes 2 and 8 -- configs[4]'s rank count -- on ONE device over gloo"""
    _run_ranks(tmp_path, world, rccl=False)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_on_their_own_gpus_rccl(tmp_path, world):
    """the real thing, at every world size the box has devices for: ranks on cuda:0 .. cuda:world-1, torch's RCCL process
    group plus the C ABI's own communicator; ncclAllGather / grouped ncclSend / ncclRecv meet peers; the plain gather and
    the overlapped pipeline are compared chunk by chunk with a one-process run.  Also covers model handles created on a
    device other than 0.  Skips cleanly on the one-GPU boxes of the gpurun pool, so that the first multi-GPU lease
    validates the exchange by itself."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} HIP devices, this box has {torch.cuda.device_count()}")
    _run_ranks(tmp_path, world, rccl=True)


def test_model_records_its_device():
    """the rANS handle used to record device 0 whatever was current at create"""
    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend import lib, models

    lib.require_device()
    last = torch.cuda.device_count() - 1
    freq = bench_data.t256_table()
    with torch.cuda.device(last):
        model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
        tmodel = models.TansModel(freq.tolist(), 1, 32)
        assert model.info().device == last and tmodel.info().device == last
        dev = torch.device("cuda", last)
        sym = bench_data.iid_chunks_device(freq, 64, 256, seed=3, device=dev)
        enc = model.encode_batch(sym)
        dec = model.decode_batch(enc.data, enc.bit_offset, enc.nbits, 256)
        assert torch.equal(dec[0], sym)
    if last > 0:  # a foreign current device is refused, not dereferenced
        with torch.cuda.device(0), pytest.raises(lib.SclHipError):
            L = lib.load()
            st = torch.cuda.current_stream(0).cuda_stream
            sym0 = sym.to("cuda:0")
            out = model.alloc_encoded(64, 256, torch.device("cuda:0"))
            lib.check(L.scl_rans_encode_batch(model._h, sym0.data_ptr(), 256, None, 256, 64, out.data.data_ptr(),
                                              out.stride, out.bit_offset.data_ptr(), out.nbits.data_ptr(),
                                              out.status.data_ptr(), st), "scl_rans_encode_batch")


def test_rccl_gatherv_refuses_bad_layout_without_leaving_a_group_open():
    """error paths of scl_streams_gatherv_rccl: arguments are validated before anything is posted, and a later
    collective on the same communicator still works"""
    from stanford_compression_library_amd.backend import lib
    from stanford_compression_library_amd.backend.sharded import RcclGather

    lib.require_device()
    dev = torch.device("cuda:0")
    comm = RcclGather(1, 0, dev)
    try:
        payload = torch.arange(100, dtype=torch.uint8, device=dev)
        out = torch.zeros(100, dtype=torch.uint8, device=dev)
        with pytest.raises(lib.SclHipError, match="agreed layout"):
            comm.gatherv([(payload, 100, np.array([0, 99], np.uint64), out)])
        with pytest.raises(lib.SclHipError, match="receive buffer"):
            comm.gatherv([(payload, 100, np.array([0, 100], np.uint64), None)])
        comm.gatherv([(payload, 100, np.array([0, 100], np.uint64), out),
                      (payload[:7], 7, np.array([0, 7], np.uint64), out[50:])])
        torch.cuda.synchronize()
        assert torch.equal(out[:50], payload[:50]) and torch.equal(out[50:57], payload[:7])
        assert list(comm.counts(12345)) == [12345]
        a = torch.tensor([5, 6], dtype=torch.int64, device=dev)
        b = torch.zeros((1, 2), dtype=torch.int64, device=dev)
        comm.allgather_async(a, b)
        torch.cuda.synchronize()
        assert b.tolist() == [[5, 6]]
    finally:
        comm.close()
