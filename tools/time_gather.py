"""host-side cost breakdown of backend/sharded.py::encode_gather_overlapped on one GPU (one-rank RCCL communicator)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models, sharded

dev = torch.device("cuda:0")
freq = bench_data.t256_table()
model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
sym = bench_data.iid_chunks_device(freq, 262144, 4096, seed=5000, device=dev)
comm = sharded.RcclGather(1, 0, dev)
for n_sub in (8, 4, 2, 1):
    ws = sharded.GatherWorkspace(model, 262144, 4096, 1, dev, n_sub)
    for rep in range(4):
        t, _ = sharded.encode_gather_overlapped(model, sym, 1, 0, n_sub=n_sub, comm=comm, workspace=ws)
    print(n_sub, t)
    del ws
# sequential reference
enc = model.alloc_encoded(262144, 4096, dev)
stride = enc.stride
dense = torch.empty(models.compact_capacity(262144, stride), dtype=torch.uint8, device=dev)
offs = torch.empty(262145, dtype=torch.int64, device=dev)
scratch = torch.empty(models.compact_scratch_bytes(262144), dtype=torch.uint8, device=dev)
out = torch.empty_like(dense)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.encode_batch(sym, out=enc); torch.cuda.synchronize(); t1 = time.perf_counter()
    models.compact_into(enc, dense, offs, scratch); torch.cuda.synchronize(); t2 = time.perf_counter()
    n = int(offs[-1]); out[:n].copy_(dense[:n]); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("seq ms", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3)
if os.environ.get("SCL_TRACE"):
    sharded._TRACE = []
    ws = sharded.GatherWorkspace(model, 262144, 4096, 1, dev, 4)
    sharded.encode_gather_overlapped(model, sym, 1, 0, n_sub=4, comm=comm, workspace=ws)
    sharded._TRACE = []
    sharded.encode_gather_overlapped(model, sym, 1, 0, n_sub=4, comm=comm, workspace=ws)
    for name, ms in sharded._TRACE: print(f"{name:30s} {ms:8.3f}")
comm.close()
