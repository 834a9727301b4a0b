"""Does a kernel's time depend on WHICH allocation its buffers are?  One process, one batch; the rANS decode output and then the
encoder's output go to eight fresh allocations each (the earlier ones stay alive, so every one is new memory)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
n_chunks, chunk_len = 262144, 4096
model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=1, device=dev)
enc = model.encode_batch(sym)
torch.cuda.synchronize()
def timed(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
keep = []
for k in range(8):
    out = model.alloc_decoded(n_chunks, chunk_len, dev); keep.append(out)
    t = timed(lambda: model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=out))
    print(f"decode into allocation {k} at {hex(out[0].data_ptr())}: {t:.4f} ms", flush=True)
for k in range(8):
    eo = model.alloc_encoded(n_chunks, chunk_len, dev); keep.append(eo)
    t = timed(lambda: model.encode_batch(sym, out=eo))
    print(f"encode into allocation {k} at {hex(eo.data.data_ptr())}: {t:.4f} ms", flush=True)
for k in range(4):
    s2 = sym.clone(); keep.append(s2)
    t = timed(lambda: model.encode_batch(s2, out=enc))
    print(f"encode from input copy {k} at {hex(s2.data_ptr())}: {t:.4f} ms", flush=True)
