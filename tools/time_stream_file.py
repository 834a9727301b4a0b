"""End-to-end rate of the streaming driver (row f2): byte file -> framed block file -> byte file, PCIe and file I/O included.
SIZE_MB (default 64), BLOCK (default 4096), CODER=rans|tans|range, DIR (default /dev/shm or /tmp).  Prints one line."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend.modeling import frequencies_from_counts
from stanford_compression_library_amd.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
from stanford_compression_library_amd.compressors.tANS import tANSDecoder, tANSEncoder, tANSParams
from stanford_compression_library_amd.compressors.range_coder import RangeCoderParams, RangeDecoder, RangeEncoder
from stanford_compression_library_amd.core.data_stream import Uint8FileDataStream
from stanford_compression_library_amd.core.encoded_stream import EncodedBlockReader, EncodedBlockWriter
from stanford_compression_library_amd.core.prob_dist import Frequencies

size = int(float(os.environ.get("SIZE_MB", 64)) * (1 << 20))
block = int(os.environ.get("BLOCK", 4096))
coder = os.environ.get("CODER", "rans")
d = os.environ.get("DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
freq = bench_data.t256_table()
rng = np.random.default_rng(11)
data = rng.choice(256, size=size, p=np.asarray(freq, dtype=np.float64) / float(np.sum(freq))).astype(np.uint8)
src, enc_path, out = (os.path.join(d, f"scl_ts_{os.getpid()}_{n}") for n in ("in.bin", "enc.bin", "out.bin"))
data.tofile(src)
fr = Frequencies({i: int(f) for i, f in enumerate(np.asarray(freq).tolist())})
if coder == "rans":
    p = rANSParams(fr); enc, dec = rANSEncoder(p), rANSDecoder(p)
elif coder == "tans":
    p = tANSParams(fr, RANGE_FACTOR=1); enc, dec = tANSEncoder(p), tANSDecoder(p)
else:
    enc, dec = RangeEncoder(RangeCoderParams(), fr), RangeDecoder(RangeCoderParams(), fr)
reps = int(os.environ.get("REPS", 2))
te = td = 1e9
for r in range(reps):   # first repetition warms the library, the allocator and the page cache
    for stale in (enc_path, out):  # (re-opening an existing file for writing truncates it first: not encode / decode time)
        if os.path.exists(stale):
            os.remove(stale)
    t0 = time.perf_counter()
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(enc_path) as w:
        enc.encode(s, block, w)
    t1 = time.perf_counter()
    with EncodedBlockReader(enc_path) as rd, Uint8FileDataStream(out, "wb") as s:
        dec.decode(rd, s)
    t2 = time.perf_counter()
    te, td = min(te, t1 - t0), min(td, t2 - t1)
ok = open(out, "rb").read() == data.tobytes()
esz = os.path.getsize(enc_path)
for pth in (src, enc_path, out):
    os.remove(pth)
print(f"stream_file coder={coder} size={size >> 20} MiB block={block} dir={d}: encode {te:.3f} s ({size / te / 1e6:.1f} MB/s)  "
      f"decode {td:.3f} s ({size / td / 1e6:.1f} MB/s)  framed {esz} B  round_trip_ok={ok}")
