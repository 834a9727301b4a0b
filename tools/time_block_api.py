"""latency of the one-block drop-in calls (encode_block / decode_block) and where it goes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
from stanford_compression_library_amd.compressors.range_coder import RangeCoderParams, RangeDecoder, RangeEncoder
from stanford_compression_library_amd.core.data_block import DataBlock
from stanford_compression_library_amd.core.prob_dist import Frequencies
from stanford_compression_library_amd.compressors._common import symbols_to_indices

freq = bench_data.t256_table()
fr = Frequencies({i: int(f) for i, f in enumerate(freq.tolist())})
p = rANSParams(fr)
enc, dec = rANSEncoder(p), rANSDecoder(p)
rng = np.random.default_rng(0)
for n in (256, 4096, 65536, 1 << 20):
    data = rng.choice(256, size=n, p=freq / freq.sum()).tolist()
    blk = DataBlock(data)
    bits = enc.encode_block(blk)
    out, used = dec.decode_block(bits)
    assert out.data_list == data and used == len(bits)
    reps = max(3, min(200, (1 << 22) // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        bits = enc.encode_block(blk)
    t1 = time.perf_counter()
    for _ in range(reps):
        out, used = dec.decode_block(bits)
    t2 = time.perf_counter()
    model = p._device_model()
    idx = symbols_to_indices(blk, p._index_of)
    t3 = time.perf_counter()
    for _ in range(reps):
        idx = symbols_to_indices(blk, p._index_of)
    t4 = time.perf_counter()
    for _ in range(reps):
        packed, nb = model.encode_host(idx)
    t5 = time.perf_counter()
    for _ in range(reps):
        i2, u2 = model.decode_host(packed, nb, 32)
    t6 = time.perf_counter()
    print(f"n={n}: encode_block {(t1-t0)/reps*1e3:.3f} ms  decode_block {(t2-t1)/reps*1e3:.3f} ms | symbols_to_indices {(t4-t3)/reps*1e3:.3f}  "
          f"encode_host {(t5-t4)/reps*1e3:.3f}  decode_host {(t6-t5)/reps*1e3:.3f} ms")
