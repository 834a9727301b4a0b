"""time the range coder kernels on uniform bytes (configs[2]) without verification -- for ablation builds"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
from stanford_compression_library_amd import bench_data
freq = bench_data.t256_table() if os.environ.get("TABLE") == "t256" else np.ones(256, dtype=np.int64)
model = models.RangeModel(freq.tolist(), 32, 32)
n_chunks, chunk_len = 262144, 4096
sym = torch.randint(0, 256, (n_chunks, chunk_len), dtype=torch.uint8, device=dev)
enc = model.alloc_encoded(n_chunks, chunk_len, dev)
dec = model.alloc_decoded(n_chunks, chunk_len, dev)
for _ in range(10):
    model.encode_batch(sym, out=enc); model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
te = td = 0
R = 20
for _ in range(R):
    e[0].record(); model.encode_batch(sym, out=enc); e[1].record()
    model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec); e[2].record()
    torch.cuda.synchronize(); te += e[0].elapsed_time(e[1]); td += e[1].elapsed_time(e[2])
print(f"{os.environ.get('ABL','base')} range uniform: encode {te/R:.3f} ms  decode {td/R:.3f} ms")
