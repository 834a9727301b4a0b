import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
n_chunks, chunk_len = int(os.environ.get('NCHUNKS', 65536)), 4096
sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=7, device=dev)
for name, model in [("iid K=256", models.AecModel(1, [1] * 256, 256, 0, 1 << 30, 32, 32)),
                    ("fixed K=256", models.AecModel(0, freq.tolist(), 256, 0, 1 << 30, 32, 32))]:
    enc = model.alloc_encoded(n_chunks, chunk_len, dev); dec = model.alloc_decoded(n_chunks, chunk_len, dev)
    model.encode_batch(sym, out=enc); model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); model.encode_batch(sym, out=enc); e[1].record()
    model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len, out=dec); e[2].record(); torch.cuda.synchronize()
    te, td = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    ok = torch.equal(dec[0][:, :chunk_len], sym)
    print(f"{name}: encode {te:.2f} ms decode {td:.2f} ms for {n_chunks*chunk_len/2**20:.0f} MiB -> {n_chunks*chunk_len/(te+td)/1e6:.2f} GB/s round trip ok={ok}", flush=True)
