"""Stress the arithmetic-coder fast kernels at full occupancy: N full-size encodes compared word for word with the
any-parameter kernel's streams, and every decode compared with the input (rare, timing-dependent faults -- see the
note in AnsFwdWriter::put -- only show up at this scale).  MODEL=fixed|order1, NCHUNKS, REPS."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
n_chunks, chunk_len = int(os.environ.get("NCHUNKS", 262144)), 4096
mode = os.environ.get("MODEL", "fixed")
if mode in ("fixed", "rans", "tans", "range", "range_uniform1", "iid", "rans_k64", "rans_k200", "rans_m3000", "fixed_k64",
            "rans_b8"):
    freq = bench_data.t256_table()
    if mode == "range_uniform1":  # configs[2]: f = 1, M = 256 -- the table-free range kernels (cooperative line stores)
        freq = np.ones(256, dtype=np.int64)
    if mode in ("rans_k64", "rans_k200", "fixed_k64"):  # alphabets below 256: the symbol-checking encoder variants
        K = 64 if mode.endswith("k64") else 200
        freq = freq[:K].copy()
        freq[0] += 4096 - freq.sum()
    if mode == "rans_m3000":  # a total that is not a power of two
        freq = np.maximum(1, (freq.astype(np.int64) * 3000) // 4096)
        freq[np.argmax(freq)] += 3000 - freq.sum()
    sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=5000, device=dev)
    model = {"fixed": lambda: models.AecModel(0, freq.tolist(), int(freq.size), 0, 1 << 30, 32, 32),
             "iid": lambda: models.AecModel(1, [1] * 256, 256, 0, 1 << 30, 32, 32),
             "rans": lambda: models.RansModel(freq.tolist(), 1 << 16, 1, 32),
             "rans_k64": lambda: models.RansModel(freq.tolist(), 1 << 16, 1, 32),
             "rans_k200": lambda: models.RansModel(freq.tolist(), 1 << 16, 1, 32),
             "rans_m3000": lambda: models.RansModel(freq.tolist(), 1 << 16, 1, 32),
             "rans_b8": lambda: models.RansModel(freq.tolist(), 1 << 8, 8, 32),  # NUM_BITS_OUT = 8 kernels
             "fixed_k64": lambda: models.AecModel(0, freq.tolist(), int(freq.size), 0, 1 << 30, 32, 32),
             "tans": lambda: models.TansModel(freq.tolist(), 1, 32),
             "range": lambda: models.RangeModel(freq.tolist(), 32, 32),
             "range_uniform1": lambda: models.RangeModel(freq.tolist(), 32, 32)}[mode]()
elif mode == "order1_k256":  # scl_aec_wide.hip (device-memory rows): NCHUNKS=65536 keeps the tables at 17.8 GB
    sym = bench_data.markov1_chunks_device(256, n_chunks, chunk_len, seed=901, device=dev)
    model = models.AecModel(2, None, 256, 1, 1 << 30, 32, 32)
else:  # order-1, K = 16: scl_aec_split.hip (encode) / scl_aec_fast.hip (decode); every chunk distinct
    sym = bench_data.markov1_chunks_device(16, n_chunks, chunk_len, seed=900, device=dev)
    model = models.AecModel(2, None, 16, 1, 1 << 30, 32, 32)
if isinstance(model, models.AecModel):
    assert model.fast_path(chunk_len)
elif hasattr(model, 'fast_path'):
    assert model.fast_path()
# reference streams from the any-parameter kernels
ref = model.encode_batch(sym, any_parameter_kernels=True)
torch.cuda.synchronize()
stride = ref.stride
nwords = int((ref.nbits.max().item() + 31) // 32)
total_bad = total_dec = 0
for rep in range(int(os.environ.get("REPS", 10))):
    enc = model.encode_batch(sym)
    dec, dlens, used, status = model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len)
    torch.cuda.synchronize()
    assert torch.equal(enc.nbits, ref.nbits)
    if mode.startswith("rans") or mode == "tans":  # streams end at the slot end: compare the last whole words
        a = enc.data[:n_chunks * stride].view(n_chunks, stride)[:, stride - 4 * nwords:].contiguous().view(torch.int32)
        b = ref.data[:n_chunks * stride].view(n_chunks, stride)[:, stride - 4 * nwords:].contiguous().view(torch.int32)
        col = torch.arange(a.shape[1], device=dev)[None, :]
        nbad = int(((a != b) & (col >= nwords - (ref.nbits.to(torch.int64)[:, None] // 32))).sum())
    else:
        a = enc.data[:n_chunks * stride].view(n_chunks, stride)[:, :4 * nwords].contiguous().view(torch.int32)
        b = ref.data[:n_chunks * stride].view(n_chunks, stride)[:, :4 * nwords].contiguous().view(torch.int32)
        col = torch.arange(a.shape[1], device=dev)[None, :]
        nbad = int(((a != b) & (col < (ref.nbits.to(torch.int64)[:, None] // 32))).sum())
    ndec = int((dec[:, :chunk_len] != sym).any(dim=1).sum())
    total_bad += nbad
    total_dec += ndec
    print("rep", rep, "wrong stream words:", nbad, " chunks decoded wrong:", ndec, flush=True)
    del enc, dec, a, b
print(f"TOTAL {mode}: wrong stream words {total_bad}, chunks decoded wrong {total_dec}")
