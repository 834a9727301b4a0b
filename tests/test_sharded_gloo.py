"""Multi-rank layer on CPU: world_size-2 `gloo` processes shard a batch of chunks block-contiguously,
produce their streams (here with the CPU oracle standing in for the per-rank kernels -- the gather does not
care who made the bytes), and gather them to rank 0.  The gathered buffer + offsets must equal what one
process produces for the whole batch."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT
from stanford_compression_library_amd.backend.sharded import shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1000, 8192):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import numpy as np, torch, torch.distributed as dist
    import scl_oracle as orc
    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend.sharded import shard_range, gather_streams_to_root
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    freq = bench_data.t256_table()
    n_chunks, chunk_len = 37, 257
    sym = bench_data.iid_chunks_host(freq, n_chunks, chunk_len, seed=99)
    def dense_of(rows):
        parts, offs = [], [0]
        for row in rows:
            b, nb = orc.rans_encode(row, freq)
            parts.append(b); offs.append(offs[-1] + b.size)
        data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        return torch.from_numpy(data.copy()), torch.tensor(offs, dtype=torch.int64)
    a, b = shard_range(n_chunks, world, rank)
    dense, offs = dense_of(sym[a:b])
    total, out, goffs = gather_streams_to_root(dense, offs, world, rank, device=torch.device("cpu"), return_data=True)
    if rank == 0:
        ref, ref_offs = dense_of(sym)
        assert total == ref.numel(), (total, ref.numel())
        assert torch.equal(out, ref) and torch.equal(goffs, ref_offs)
        print("GATHER_OK", total)
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gather_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "GATHER_OK" in outs[0]
