"""the striped layout on very long chunks (1 and 5 MiB: slot strides up to the 16 MiB limit), against the linear layout"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import models
dev = torch.device("cuda:0")
freq = bench_data.t256_table()
for name, model in (("rans", models.RansModel(freq.tolist(), 1 << 16, 1, 32)), ("range", models.RangeModel(freq.tolist(), 32, 32)),
                    ("tans", models.TansModel(freq.tolist(), 1, 32))):
    for n, L in ((130, 1 << 20), (70, 5 << 20)):
        sym = bench_data.iid_chunks_device(freq, n, L, seed=3, device=dev)
        lin = model.encode_batch(sym)
        if lin.stride > model.STRIPED_MAX_STRIDE:  # slots of 16 MiB and more: linear only (auto says so, "striped" raises)
            assert model.pick_layout("auto", 1 << 20, stride=lin.stride) == "linear"
            print(name, n, L, "stride", lin.stride, "-> linear only")
            continue
        st = model.encode_batch(sym, layout="striped")
        assert int(st.status.abs().sum()) == 0 and torch.equal(st.nbits, lin.nbits), (name, n, L)
        a, ao = models.compact(lin); b, bo = models.compact(st)
        assert torch.equal(ao, bo) and torch.equal(a[:int(ao[-1])], b[:int(ao[-1])]), (name, n, L)
        dec, dl, used, stt = model.decode_encoded(st, L)
        assert int(stt.abs().sum()) == 0 and torch.equal(dec, sym) and torch.equal(used, st.nbits), (name, n, L)
        print(name, n, L, "stride", st.stride, "ok")
        del sym, lin, st, a, b, dec
