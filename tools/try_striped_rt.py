"""Round 6: the headline ROUND TRIP on wave-striped slots against linear slots, same box, alternating; both decodes verified
against the input; the dense compaction of either timed as well.  CODER=rans|tans|range (T256 table)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stanford_compression_library_amd import bench_data  # noqa: E402
from stanford_compression_library_amd.backend import models  # noqa: E402

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
chunk_len = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
CODER = os.environ.get("CODER", "rans")  # rans | tans | range
model = {"rans": lambda: models.RansModel(freq.tolist(), 1 << 16, 1, 32), "tans": lambda: models.TansModel(freq.tolist(), 1, 32),
         "range": lambda: models.RangeModel(freq.tolist(), 32, 32)}[CODER]()
sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=1, device=dev)


enc = {False: model.alloc_encoded(n_chunks, chunk_len, dev), True: model.alloc_encoded(n_chunks, chunk_len, dev, layout='striped')}
dec = model.alloc_decoded(n_chunks, chunk_len, dev)
for striped in (False, True):
    model.encode_batch(sym, out=enc[striped])
    out = model.decode_encoded(enc[striped], chunk_len, out=dec)
    torch.cuda.synchronize()
    assert int(out[3].abs().sum()) == 0, f"striped={striped}: status"
    assert torch.equal(out[0], sym), f"striped={striped}: decoded symbols differ"
    assert torch.equal(out[2], enc[striped].nbits), f"striped={striped}: consumed bits differ"
print("round trips verified (linear and striped)")


def time_rt(striped, reps=20):
    e = enc[striped]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps + 1)]
    ev[0].record()
    for i in range(reps):
        model.encode_batch(sym, out=e)
        ev[2 * i + 1].record()
        model.decode_encoded(e, chunk_len, out=dec)
        ev[2 * i + 2].record()
    torch.cuda.synchronize()
    te = np.mean([ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(reps)])
    td = np.mean([ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(reps)])
    return te, td


cap = models.compact_capacity(n_chunks, enc[False].stride)
c_out = torch.empty(cap, dtype=torch.uint8, device=dev)
c_offs = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
c_scr = torch.empty(models.compact_scratch_bytes(n_chunks), dtype=torch.uint8, device=dev)


def time_compact(striped, reps=10):
    e = enc[striped]
    models.compact_into(e, c_out, c_offs, c_scr)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        models.compact_into(e, c_out, c_offs, c_scr)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


for _ in range(3):
    time_rt(False)
res = {False: [], True: []}
for rnd in range(7):
    for striped in (False, True):
        res[striped].append(time_rt(striped))
alg = n_chunks * chunk_len + int(((enc[False].nbits.to(torch.int64) + 7) // 8).sum())
line = []
for striped in (False, True):
    te = float(np.median([r[0] for r in res[striped]]))
    td = float(np.median([r[1] for r in res[striped]]))
    print(f"{'striped' if striped else 'linear '}: enc {te:.4f} ms ({alg / te / 8e9:.4f})  dec {td:.4f} ms ({alg / td / 8e9:.4f})  "
          f"round trip {te + td:.4f} ms = {n_chunks * chunk_len / (te + td) / 1e3:.0f} MB/s")
    tc = float(np.median([time_compact(striped) for _ in range(3)]))
    print(f"         compact {tc:.4f} ms")
    line.append(f"{'T' if striped else 'L'} {te:.4f} {td:.4f} {tc:.4f}")
print("  ".join(line))
