import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
n_chunks, chunk_len = 65536, 4096
sym = bench_data.markov1_chunks_device(256, n_chunks, chunk_len, 4000, dev)
model = models.AecModel(2, None, 256, 1, 1 << 30, 32, 32)
enc = model.alloc_encoded(n_chunks, chunk_len, dev)
for _ in range(2): model.encode_batch(sym, out=enc)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); model.encode_batch(sym, out=enc); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(f"encode K=256 {np.median(ts):.3f} ms")
