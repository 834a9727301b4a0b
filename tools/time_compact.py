"""time scl_streams_compact (dense and framed) on the 1 GiB headline batch, buffers preallocated"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models, lib as _lib
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
n_chunks, chunk_len = 262144, 4096
sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=5000, device=dev)
model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
enc = model.encode_batch(sym)
L = _lib.load()
total_bits = int(enc.nbits.to(torch.int64).sum().item())
for framed in (False, True):
    cap = total_bits // 8 + n_chunks * 6 + 16
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    offsets = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
    scratch = torch.empty(int(L.scl_streams_compact_scratch_bytes(n_chunks)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    def run():
        rc = L.scl_streams_compact(enc.data.data_ptr(), enc.bit_offset.data_ptr(), enc.nbits.data_ptr(), n_chunks,
                                   _lib.COMPACT_FRAMED if framed else _lib.COMPACT_DENSE, out.data_ptr(), cap,
                                   offsets.data_ptr(), scratch.data_ptr(), st)
        assert rc == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    nbytes = int(offsets[-1].item())
    ms = e0.elapsed_time(e1) / 5
    print(f"compact framed={framed}: {ms:.3f} ms for {nbytes/1e9:.3f} GB of records -> {2*nbytes/ms/1e9:.2f} TB/s read+write")
