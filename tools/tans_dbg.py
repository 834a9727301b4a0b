import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import scl_oracle as orc
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
m = models.TansModel(freq.tolist(), 1, 32)
for n_chunks, L in ((64, 4096), (3000, 4096), (3000, 1024), (2048, 128), (2048, 256)):
    sym = bench_data.iid_chunks_device(freq, n_chunks, L, seed=1, device=dev)
    enc = m.encode_batch(sym)
    dec, lens, used, st = m.decode_batch(enc.data, enc.bit_offset, enc.nbits, L)
    torch.cuda.synchronize()
    bad_enc = 0
    h = sym.cpu().numpy(); data = enc.data.cpu().numpy(); offs = enc.bit_offset.cpu().numpy(); nb = enc.nbits.cpu().numpy()
    for c in range(n_chunks):
        rb, rn = orc.tans_encode(h[c], freq, RF=1)
        got = np.unpackbits(data[int(offs[c]) // 8:(int(offs[c]) + int(nb[c]) + 7) // 8 + 1])[int(offs[c]) % 8:][:rn]
        if rn != nb[c] or not np.array_equal(got, np.unpackbits(rb)[:rn]): bad_enc += 1
    print(n_chunks, L, "enc status", int(enc.status.abs().sum()), "bad_enc_sample", bad_enc, "dec status nonzero", int((st != 0).sum()), "dec mismatch rows", int((dec != sym).any(dim=1).sum()))
n_chunks, L = 3000, 4096
sym = bench_data.iid_chunks_device(freq, n_chunks, L, seed=1, device=dev)
enc = m.encode_batch(sym)
dec, lens, used, st = m.decode_batch(enc.data, enc.bit_offset, enc.nbits, L)
torch.cuda.synchronize()
bad = (dec != sym).any(dim=1).nonzero().flatten().cpu().numpy()
offs = enc.bit_offset.cpu().numpy()
print("bad w0:", sorted(set(((offs[bad] >> 5) & 31).tolist())), "pos:", sorted(set((offs[bad] & 31).tolist()))[:10])
print("first mismatch index per bad chunk:", [(int(c), int((dec[c] != sym[c]).nonzero()[-1])) for c in bad[:8]], "status", st[bad[:8]].tolist(), "used-nbits", (used[bad[:8]] - enc.nbits[bad[:8]]).tolist())
mr = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
enc = mr.encode_batch(sym); dec, lens, used, st = mr.decode_batch(enc.data, enc.bit_offset, enc.nbits, L); torch.cuda.synchronize()
print("rans bad rows", int((dec != sym).any(dim=1).sum()))
