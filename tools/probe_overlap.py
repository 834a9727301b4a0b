"""Can the memory-bound compaction run UNDER the issue-bound rANS encoder?  (cheaper probe first.)

Times, on the 1 GiB headline batch: the encoder alone, scl_streams_compact alone (of a batch encoded earlier), and
both at once on two streams -- the encoder filling a second set of slots while the compaction reads the first -- with
and without a high-priority compaction stream, in both launch orders.  If "both" ~ max(enc, cp) the two kinds of work
overlap on this chip and a fused / concurrent dense encoder is worth building; if "both" ~ enc + cp it is not."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import lib as _lib
from stanford_compression_library_amd.backend import models

dev = torch.device("cuda:0")
freq = bench_data.t256_table()
n_chunks, chunk_len = int(os.environ.get("N_CHUNKS", 262144)), 4096
sym = bench_data.iid_chunks_device(freq, n_chunks, chunk_len, seed=5000, device=dev)
model = models.RansModel(freq.tolist(), 1 << 16, 1, 32)
enc_a = model.encode_batch(sym)
enc_b = model.alloc_encoded(n_chunks, chunk_len, dev)
torch.cuda.synchronize()
L = _lib.load()
cap = models.compact_capacity(n_chunks, enc_a.stride)
dense = torch.empty(cap, dtype=torch.uint8, device=dev)
offsets = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
scratch = torch.empty(models.compact_scratch_bytes(n_chunks), dtype=torch.uint8, device=dev)


def run_enc(stream):
    model.encode_batch(sym, out=enc_b, stream=stream.cuda_stream)


def run_cp(stream):
    models.compact_into(enc_a, dense, offsets, scratch, stream=stream)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cur = torch.cuda.current_stream(dev)
# warm the clocks
for _ in range(200):
    run_enc(cur)
torch.cuda.synchronize()
t_enc = timed(lambda: run_enc(cur))
t_cp = timed(lambda: run_cp(cur))
print(f"encoder alone {t_enc:.3f} ms, compaction alone {t_cp:.3f} ms, sum {t_enc + t_cp:.3f} ms")

lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for label, pri_cp in (("default priorities", 0), ("compaction stream high priority", -1)):
    s_enc = torch.cuda.Stream(device=dev)
    s_cp = torch.cuda.Stream(device=dev, priority=pri_cp)
    for order in ("enc first", "cp first"):
        def both():
            s_enc.wait_stream(cur)
            s_cp.wait_stream(cur)
            if order == "enc first":
                run_enc(s_enc)
                run_cp(s_cp)
            else:
                run_cp(s_cp)
                run_enc(s_enc)
            cur.wait_stream(s_enc)
            cur.wait_stream(s_cp)
        t = timed(both)
        print(f"both at once, {label}, {order}: {t:.3f} ms  (max {max(t_enc, t_cp):.3f}, sum {t_enc + t_cp:.3f})")

# the decoder under a compaction, for completeness (decode reads what the compaction reads)
dec_out = model.alloc_decoded(n_chunks, chunk_len, dev)
t_dec = timed(lambda: model.decode_batch(enc_a.data, enc_a.bit_offset, enc_a.nbits, chunk_len, out=dec_out))
print(f"decoder alone {t_dec:.3f} ms")
