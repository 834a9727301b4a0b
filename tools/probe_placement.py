"""Does the rANS decoder's time depend on where its buffers lie?  One process, one encoded batch; the decode output (and then
the encoded input) is placed at different offsets inside one big arena and timed (20 launches each)."""
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
arena = torch.empty((3 << 30) + (64 << 20), dtype=torch.uint8, device=dev)
print("arena", hex(arena.data_ptr()), "enc.data", hex(enc.data.data_ptr()), "sym", hex(sym.data_ptr()))
lens = torch.empty(n_chunks, dtype=torch.int32, device=dev); used = torch.empty_like(lens); status = torch.empty_like(lens)
def time_decode(data, out_sym, reps=20):
    out = (out_sym, lens, used, status)
    for _ in range(5): model.decode_batch(data, enc.bit_offset, enc.nbits, chunk_len, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): model.decode_batch(data, enc.bit_offset, enc.nbits, chunk_len, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
base = (-arena.data_ptr()) % (2 << 20)  # 2 MiB-aligned start
for off in [0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 65536, 16 << 20, (32 << 20) + 4096, 48 << 20]:
    o = arena[base + off: base + off + n_chunks * chunk_len].view(n_chunks, chunk_len)
    print(f"out at +{off:>9}: decode {time_decode(enc.data, o):.4f} ms")
nb = enc.data.numel()
for off in [0, 4096, 65536, 1 << 20, (2 << 20) + 65536, 16 << 20]:
    d = arena[base + (1 << 30) + (32 << 20) + off:][:nb]
    d.copy_(enc.data)
    o = arena[base: base + n_chunks * chunk_len].view(n_chunks, chunk_len)
    print(f"in  at +{off:>9}: decode {time_decode(d, o):.4f} ms")
dec, *_ = model.alloc_decoded(n_chunks, chunk_len, dev)
print(f"torch-allocated out {hex(dec.data_ptr())}: decode {time_decode(enc.data, dec):.4f} ms")
