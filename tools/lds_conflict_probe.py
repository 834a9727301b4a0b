"""How much of the headline kernels' time is LDS bank conflicts on the per-symbol table read?  Uniform table (f = 16,
8 bits per symbol whatever the data), three inputs of the same size and output volume: random bytes (lanes read random
entries), one constant byte (all lanes read ONE entry: broadcast, no conflicts), and lane-distinct constants
(symbol = lane id of the chunk mod 256: distinct entries, fixed pattern)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
n_chunks, chunk_len = 262144, 4096
freq = bench_data.uniform256_table()
model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
inputs = {
    "random": bench_data.iid_chunks_device(freq, n_chunks, chunk_len, 7, dev),
    "constant": torch.full((n_chunks, chunk_len), 37, dtype=torch.uint8, device=dev),
    "per_chunk": (torch.arange(n_chunks, device=dev) % 256).to(torch.uint8)[:, None].expand(n_chunks, chunk_len).contiguous(),
    "per_chunk_x8": ((torch.arange(n_chunks, device=dev) % 8) * 2).to(torch.uint8)[:, None].expand(n_chunks, chunk_len).contiguous(),
}
for name, sym in inputs.items():
    enc = model.alloc_encoded(n_chunks, chunk_len, dev)
    dec = model.alloc_decoded(n_chunks, chunk_len, dev)
    for _ in range(5):
        model.encode_batch(sym, out=enc)
        model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec)
    torch.cuda.synchronize()
    te, td = [], []
    for _ in range(20):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); model.encode_batch(sym, out=enc); b.record()
        model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec); c.record()
        torch.cuda.synchronize()
        te.append(a.elapsed_time(b)); td.append(b.elapsed_time(c))
    ok = bool((dec[0][:, :chunk_len] == sym).all())
    print(f"{name:14s} enc {np.median(te):.4f} ms  dec {np.median(td):.4f} ms  bits/chunk {int(enc.nbits[0])}  round trip {ok}")
