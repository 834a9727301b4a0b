"""where the time of the reference-API file path goes: main-thread profile + time inside the file reads / writes (row f2)"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
from stanford_compression_library_amd.core.data_stream import Uint8FileDataStream
from stanford_compression_library_amd.core.encoded_stream import EncodedBlockReader, EncodedBlockWriter
from stanford_compression_library_amd.core.prob_dist import Frequencies
from stanford_compression_library_amd.compressors import _stream_batch as sb
size = 1 << 30
freq = bench_data.t256_table()
rng = np.random.default_rng(11)
data = rng.choice(256, size=size, p=np.asarray(freq, dtype=np.float64) / float(np.sum(freq))).astype(np.uint8)
src, mid, out = "/dev/shm/pf_in", "/dev/shm/pf_mid", "/dev/shm/pf_out"
data.tofile(src)
fr = Frequencies({i: int(f) for i, f in enumerate(np.asarray(freq).tolist())})
p = rANSParams(fr); enc, dec = rANSEncoder(p), rANSDecoder(p)
T = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t; T[name + "#"] = T.get(name + "#", 0) + 1; return r
    setattr(mod, name, g)
wrap(sb, "_fill"); wrap(sb, "_fill_from")
ow = EncodedBlockWriter.write_framed_bytes
def wfb(self, b):
    t = time.perf_counter(); ow(self, b); T["write_framed"] = T.get("write_framed", 0) + time.perf_counter() - t
EncodedBlockWriter.write_framed_bytes = wfb
owc = Uint8FileDataStream.write_codes
def wc(self, c):
    t = time.perf_counter(); owc(self, c); T["write_codes"] = T.get("write_codes", 0) + time.perf_counter() - t
Uint8FileDataStream.write_codes = wc
for rep in range(2):
    for stale in (mid, out):
        if os.path.exists(stale):
            os.remove(stale)
    T.clear()
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable()
    with Uint8FileDataStream(src, "rb") as s, EncodedBlockWriter(mid) as w:
        enc.encode(s, 4096, w)
    pr.disable(); t1 = time.perf_counter()
    print("encode", round(t1 - t0, 3), {k: round(v, 3) for k, v in T.items()})
    if rep == 1:
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(12); print(st.getvalue()[:2500])
    T.clear(); pr = cProfile.Profile()
    t1 = time.perf_counter(); pr.enable()
    with EncodedBlockReader(mid) as rd, Uint8FileDataStream(out, "wb") as s:
        dec.decode(rd, s)
    pr.disable(); t2 = time.perf_counter()
    print("decode", round(t2 - t1, 3), {k: round(v, 3) for k, v in T.items()})
    if rep == 1:
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(12); print(st.getvalue()[:2500])
for f in (src, mid, out): os.remove(f)
