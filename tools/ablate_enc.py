"""time the rANS fast encode/decode kernels only (no verification) -- for ablation builds"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
n_chunks, chunk_len = int(os.environ.get('NCHUNKS', 262144)), 4096
sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=5000, device=dev)
PAD = int(os.environ.get('PAD', 0))
if PAD:
    padded = torch.zeros((n_chunks, chunk_len + PAD), dtype=torch.uint8, device=dev)
    padded[:, :chunk_len] = sym
    sym = padded[:, :chunk_len]
enc = model.alloc_encoded(n_chunks, chunk_len, dev, out_stride=int(os.environ.get('SLOT', 0)) or None)
dec = model.alloc_decoded(n_chunks, chunk_len + PAD, dev)
for _ in range(int(os.environ.get('WARM', 20))):
    model.encode_batch(sym, out=enc); model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
te = td = 0
REPS = int(os.environ.get("TREPS", os.environ.get("REPS", 30)))
for _ in range(REPS):
    e[0].record(); model.encode_batch(sym, out=enc); e[1].record()
    model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec); e[2].record()
    torch.cuda.synchronize(); te += e[0].elapsed_time(e[1]); td += e[1].elapsed_time(e[2])
print(f"{os.environ.get('ABL','base')} chunks={n_chunks} pad={PAD} slot={enc.stride}: encode {te/REPS:.3f} ms  decode {td/REPS:.3f} ms")
