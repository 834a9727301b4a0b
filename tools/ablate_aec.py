"""time the LDS-table arithmetic-coder kernels only (no verification) -- for tools/ablate_aec.sh"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
K, n_chunks, chunk_len = 16, int(os.environ.get("NCHUNKS", 65536)), 4096
model = models.AecModel(2, None, K, 1, 1 << 30, 32, 32)
base = np.stack([bench_data.markov1_host(K, chunk_len, seed=40 + c) for c in range(256)])
sym = torch.from_numpy(base).to(dev).repeat(n_chunks // 256, 1)
enc = model.alloc_encoded(n_chunks, chunk_len, dev)
dec = model.alloc_decoded(n_chunks, chunk_len, dev)
decode = os.environ.get("ABL", "0") == "0" or os.environ.get("DECODE")
for _ in range(2):
    model.encode_batch(sym, out=enc)
    if decode: model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
te = td = 0
for _ in range(3):
    e[0].record(); model.encode_batch(sym, out=enc); e[1].record()
    if decode: model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec)
    e[2].record(); torch.cuda.synchronize(); te += e[0].elapsed_time(e[1]); td += e[1].elapsed_time(e[2])
print(f"AF_ABLATE={os.environ.get('ABL','0')} chunks={n_chunks}: encode {te/3:.3f} ms  decode {td/3:.3f} ms", flush=True)
