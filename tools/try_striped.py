"""Round-6 experiment: the rANS headline encoder with wave-striped output slots (AnsBackWriterT, four
workgroups per CU) against the shipped AnsBackWriterL, same box, alternating -- and bit-exactness of the striped output
after de-striping (a permutation of 16-byte pieces, done here with torch).

    python tools/try_striped.py [n_chunks] [chunk_len]
"""
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
model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=1, device=dev)
S = model.slot_bytes(chunk_len)
n_pad = (n_chunks + 63) // 64 * 64


def alloc():
    e = model.alloc_encoded(n_pad, chunk_len, dev)
    e.n_chunks = n_chunks
    e.bit_offset, e.nbits, e.status = e.bit_offset[:n_chunks], e.nbits[:n_chunks], e.status[:n_chunks]
    return e


def destripe(data):
    g = n_pad // 64
    return data[: n_pad * S].view(g, S // 16, 64, 16).permute(0, 2, 1, 3).contiguous().view(-1)


def run(writer, out):
    if writer:
        os.environ["SCL_RANS_ENC_WRITER"] = writer
    else:
        os.environ.pop("SCL_RANS_ENC_WRITER", None)
    model.encode_batch(sym, out=out)


TIME_ONLY = os.environ.get("TIME_ONLY") == "1"  # ablation builds (wrong output): skip the comparison
ref, tst = alloc(), alloc()
tst.data.zero_()
ref.data.zero_()
run(None, ref)
run("T", tst)
torch.cuda.synchronize()
nb = ref.nbits.to(torch.int64)


def compare():
    assert torch.equal(ref.nbits, tst.nbits), "stream lengths differ"
    assert torch.equal(ref.bit_offset, tst.bit_offset), "bit offsets differ"
    assert int(tst.status.abs().sum()) == 0
    flat = destripe(tst.data)
    # every stream's bytes: the whole bytes behind its first one, and the first (partial) byte under its mask; all streams
    # end at their slot end
    off = ref.bit_offset
    first, end = off // 8, (off + nb) // 8
    pos = torch.arange(n_pad * S, device=dev)
    slot = (pos // S).clamp(max=n_chunks - 1)
    inside = (pos >= (first[slot] + 1)) & (pos < end[slot]) & (pos // S < n_chunks)
    a, b = ref.data[: n_pad * S], flat
    bad = int(((a != b) & inside).sum())
    print("bytes compared:", int(inside.sum()), "mismatching:", bad)
    mask = (0xFF >> (off % 8)).to(torch.uint8)
    bad_first = int(((a[first] & mask) != (b[first] & mask)).sum())
    print("first partial bytes mismatching:", bad_first)
    assert bad == 0 and bad_first == 0, "striped output differs from the shipped writer's"


if not TIME_ONLY:
    compare()


def time_one(writer, out, reps=30):
    run(writer, out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        model.encode_batch(sym, out=out)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


for _ in range(10):  # warm the clocks
    run(None, ref)
torch.cuda.synchronize()
res = {"L": [], "T": []}
for rnd in range(7):
    res["L"].append(time_one(None, ref))
    res["T"].append(time_one("T", tst))
alg = n_chunks * chunk_len + int(((nb + 7) // 8).sum())
line = []
for k, v in res.items():
    med = float(np.median(v))
    print(f"writer {k}: median {med:.4f} ms  min {min(v):.4f}  max {max(v):.4f}   frac {alg / (med * 1e-3) / 8e12:.4f}")
    line.append(f"{k} {med:.4f}")
print("  ".join(line))
