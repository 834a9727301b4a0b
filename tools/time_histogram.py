"""row f3: scl_histogram_u8 / _u16 on 1 GiB resident in HBM (ms per call, GB/s)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend.modeling import histogram_u8, histogram_u16
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
for name, sym in (("t256 i.i.d.", bench_data.iid_chunks_device(freq, 262144, 4096, seed=1, device=dev)),
                  ("one value", torch.full((262144, 4096), 7, dtype=torch.uint8, device=dev))):
    for _ in range(3):
        h = histogram_u8(sym)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(10):
        h = histogram_u8(sym)
    e[1].record(); torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / 10
    assert int(h.sum()) == sym.numel()
    print(f"histogram_u8 {name}: {ms:.3f} ms per GiB incl. the download of the counts ({sym.numel() / ms / 1e6:.0f} GB/s)")
s16 = torch.randint(0, 1000, (1 << 29,), dtype=torch.int16, device=dev)
for _ in range(3):
    h = histogram_u16(s16, 1000)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
e[0].record()
for _ in range(10):
    h = histogram_u16(s16, 1000)
e[1].record(); torch.cuda.synchronize()
ms = e[0].elapsed_time(e[1]) / 10
print(f"histogram_u16 K=1000: {ms:.3f} ms per GiB ({s16.numel() * 2 / ms / 1e6:.0f} GB/s)")
