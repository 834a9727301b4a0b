"""GPU tests of round 5's boundary fixes:

* a chunk whose size header exceeds ``out_cap`` decoded into UNALIGNED output rows (the library re-lays those through
  scratch, ``RowRelay``): nothing of the refused row's scratch reaches the caller's buffer;
* the lone-wave arithmetic decoders' unchecked stretches in waves that are not whole (n_chunks % 64 != 0, one damaged
  chunk per wave): bit-exact against the oracle, damaged chunks flagged, neighbours untouched;
* ``decode_block`` refuses a size header above ``max_block_size`` before it allocates;
* the kernel names the library reports (C ABI 6) are the ones the committed kernel-trace summaries hold.
"""
import os
import re

import numpy as np
import pytest

import scl_oracle as orc
from conftest import ROOT
from stanford_compression_library_amd import bench_data
from stanford_compression_library_amd.backend import lib as backend_lib
from stanford_compression_library_amd.backend import models

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

FILL = 0xA5


@pytest.fixture(scope="module")
def dev():
    backend_lib.require_device()
    return torch.device("cuda:0")


def _models(freq):
    fl = freq.tolist()
    return {
        "rans": models.RansModel(fl, 1 << 16, 1, 32),
        "tans": models.TansModel(fl, 1, 32),
        "range": models.RangeModel(fl, 32, 32),
        "aec_fixed": models.AecModel(backend_lib.MODEL_FIXED, fl, len(fl), 0, 1 << 30, 32, 32),
        "aec_iid": models.AecModel(backend_lib.MODEL_IID, [1] * len(fl), len(fl), 0, 1 << 30, 32, 32),
    }


@pytest.mark.parametrize("name", ["rans", "tans", "range", "aec_fixed", "aec_iid"])
def test_refused_chunk_leaves_unaligned_row_untouched(name, dev):
    """out_cap below some chunks' length, output rows at an odd stride: the tuned decoders write through aligned scratch
    and the relay copies back what was decoded -- for a SCL_ST_CAPACITY chunk that is nothing (the scratch behind it is
    uninitialised pool memory; until round 5 min(out_lens, out_cap) bytes of it were copied)."""
    freq = bench_data.t256_table()
    model = _models(freq)[name]
    n_chunks, chunk_len, out_cap = 200, 640, 512
    lens = np.full(n_chunks, 300, dtype=np.int32)
    lens[::7] = 640          # above out_cap: refused
    lens[3::11] = 512        # exactly out_cap: decoded
    sym = bench_data.iid_chunks_host(freq, n_chunks, chunk_len, seed=77)
    d_sym = torch.from_numpy(sym).to(dev)
    enc = model.encode_batch(d_sym, lens=torch.from_numpy(lens).to(dev))
    torch.cuda.synchronize()
    assert int(enc.status.abs().sum()) == 0
    # poison the allocator's pool so that "stale scratch" is recognisable, then decode into rows of stride 517
    junk = torch.full((n_chunks * 1024 + 4096,), 0x3C, dtype=torch.uint8, device=dev)
    del junk
    out_stride = 517
    out = torch.full((n_chunks * out_stride + 64,), FILL, dtype=torch.uint8, device=dev)
    rows = out[1:1 + n_chunks * out_stride].view(n_chunks, out_stride)  # base address odd as well
    d_lens = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    used = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    status = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    model.decode_batch(enc.data, enc.bit_offset, enc.nbits, out_cap, out=(rows, d_lens, used, status))
    torch.cuda.synchronize()
    got, st = rows.cpu().numpy(), status.cpu().numpy()
    for c in range(n_chunks):
        if lens[c] > out_cap:
            assert st[c] & backend_lib.ST_CAPACITY, f"chunk {c}: status {st[c]:#x}"
            assert (got[c] == FILL).all(), f"chunk {c}: a refused chunk's row was written"
        else:
            assert st[c] == 0
            assert np.array_equal(got[c, :lens[c]], sym[c, :lens[c]])
            assert (got[c, lens[c]:] == FILL).all(), f"chunk {c}: bytes behind the decoded symbols were written"
    assert (out[:1].cpu().numpy() == FILL).all() and (out[1 + n_chunks * out_stride:].cpu().numpy() == FILL).all()


@pytest.mark.parametrize("kind,K", [("orderk", 16), ("iid", 200)])
def test_partial_waves_with_damaged_chunks_vs_oracle(kind, K, dev):
    """the unchecked-refill stretches of scl_aec_fast.hip / scl_aec_iid.hip take their length from a minimum over the
    wave's lanes (af_wave_min): waves with lanes that have left -- a batch that is not a multiple of 64, a chunk whose
    stream is too short for its header -- must still decode every other chunk exactly and flag the damaged ones."""
    n_chunks, n = 64 * 3 + 37, 1500
    if kind == "orderk":
        sym = np.stack([bench_data.markov1_host(K, n, seed=900 + c) for c in range(n_chunks)])
        model = models.AecModel(backend_lib.MODEL_ORDERK, None, K, 1, 1 << 30, 32, 32)
        okw = dict(model_kind=orc.MODEL_ORDERK, K=K, k=1)
    else:
        sym = np.random.default_rng(5).integers(0, K, (n_chunks, n)).astype(np.uint8)
        model = models.AecModel(backend_lib.MODEL_IID, [1] * K, K, 0, 1 << 30, 32, 32)
        okw = dict(model_kind=orc.MODEL_IID, K=K, k=0)
    assert model.fast_path(n) or kind == "iid"
    enc = model.encode_batch(torch.from_numpy(sym).to(dev))
    torch.cuda.synchronize()
    assert int(enc.status.abs().sum()) == 0
    streams, nbits = orc.encode_batch("aec", sym, None if kind == "orderk" else [1] * K, **okw)
    assert np.array_equal(enc.nbits.cpu().numpy().astype(np.uint64), nbits)
    # one damaged chunk per wave: the available bits end inside the size header (the lane leaves before the loop), and in
    # another lane in the middle of the stream (zero fill past the end: decodes garbage, must stay inside its row)
    avail = enc.nbits.clone()
    short, mid = list(range(5, n_chunks, 64)), list(range(40, n_chunks, 64))
    for c in short:
        avail[c] = 20
    for c in mid:
        avail[c] = int(avail[c]) // 2
    out = model.alloc_decoded(n_chunks, n, dev)
    for t in out:
        t.view(torch.uint8).fill_(FILL)
    dec, dlens, used, status = model.decode_batch(enc.data, enc.bit_offset, avail, n, out=out)
    torch.cuda.synchronize()
    dec, st, used = dec.cpu().numpy(), status.cpu().numpy(), used.cpu().numpy()
    for c in range(n_chunks):
        if c in short:
            assert st[c] & backend_lib.ST_TRUNCATED and (dec[c] == FILL).all()
        elif c in mid:
            continue  # decodes zero-filled bits: any symbols, no claim -- only that nobody else is disturbed
        else:
            assert st[c] == 0 and used[c] == nbits[c], f"chunk {c}"
            assert np.array_equal(dec[c], sym[c]), f"chunk {c}"


def test_decode_block_refuses_oversized_header(dev):
    from stanford_compression_library_amd.compressors.rANS import rANSDecoder, rANSEncoder, rANSParams
    from stanford_compression_library_amd.core.data_block import DataBlock
    from stanford_compression_library_amd.core.prob_dist import Frequencies
    from stanford_compression_library_amd.utils.bitarray_utils import BitArray, uint_to_bitarray

    params = rANSParams(Frequencies({"A": 1, "B": 3}))
    bits = rANSEncoder(params).encode_block(DataBlock(list("ABBB" * 50)))
    dec = rANSDecoder(params)
    block, used = dec.decode_block(bits)
    assert block.data_list == list("ABBB" * 50) and used == len(bits)
    # the same stream with a header announcing 2^31 symbols, read by a decoder that opted into a cap (untrusted input):
    # refused before any allocation, as an AssertionError.  (The default accepts what a 32-bit header can announce --
    # reference parity)
    lie = uint_to_bitarray(1 << 31, 32) + bits[32:]
    assert dec.max_block_size is None and dec.params._device_model().DEFAULT_MAX_BLOCK_SIZE == (1 << 32) - 1
    dec.max_block_size = 1 << 24
    with pytest.raises(AssertionError, match="max_block_size"):
        dec.decode_block(lie)
    # a caller that really has large blocks raises the cap; a small cap refuses an honest block
    dec.max_block_size = 100
    with pytest.raises(AssertionError, match="max_block_size"):
        dec.decode_block(bits)
    dec.max_block_size = 200
    assert dec.decode_block(bits)[0].data_list == list("ABBB" * 50)
    assert isinstance(lie, BitArray)


def _summary_kernels(path):
    names = []
    for line in open(path):
        m = re.match(r"(?:void )?(\S.*?)\s+calls=", line)
        if m:
            names.append(m.group(1).strip())
    return names


def test_library_names_its_kernels_like_rocprof(dev):
    """scl_rans_kernel_names / scl_tans_kernel_names against the kernel-trace summaries committed for the same workloads"""
    import ctypes as C

    L = backend_lib.load()
    freq = bench_data.t256_table()

    def names(fn, model, n):
        e, d = C.create_string_buffer(160), C.create_string_buffer(160)
        assert getattr(L, fn)(model._h, n, e, d, 160) == 0
        return e.value.decode(), d.value.decode()

    cases = [
        ("rans_headline", "scl_rans_kernel_names", models.RansModel(freq.tolist(), 1 << 16, 1, 32), 262144),
        ("config2_64Ki", "scl_rans_kernel_names", models.RansModel(freq.tolist(), 1 << 16, 1, 32), 65536),
        ("rans_b8", "scl_rans_kernel_names", models.RansModel(freq.tolist(), 1 << 8, 8, 32), 262144),
        ("tans", "scl_tans_kernel_names", models.TansModel(freq.tolist(), 1, 32), 262144),
    ]
    checked = 0
    for tag, fn, model, n in cases:
        enc, dec = names(fn, model, n)
        assert enc.startswith("rans_encode_fast_kernel<AnsBackWriter") and dec.startswith("rans_decode_fast_kernel<")
        # (the LINEAR-slot kernels: round 6's traces name the decoder with its fifth template argument, `false`; its
        # rans_headline family runs on striped slots -- rans_headline_linear is the same batch on these kernels)
        for rnd, t in (("r06", "rans_headline_linear" if tag == "rans_headline" else tag),):
            if tag in ("rans_b8", "tans"):  # round 6 profiled these on striped slots: their names are checked below
                continue
            path = os.path.join(ROOT, "profiles", f"{rnd}_{t}_kernel_trace_summary.txt")
            if os.path.exists(path):
                have = _summary_kernels(path)
                assert enc in have and dec in have, f"{tag}: {enc} / {dec} not in {path}: {have[:4]}"
                checked += 1
                break
    assert checked >= 2
    # the striped-slot kernels (ABI 8) against the round-6 traces of the families profiled on them
    for tag, fn, model, n in [(c[0], c[1] + "_striped", c[2], c[3]) for c in cases if c[0] != "config2_64Ki"]:
        enc, dec = names(fn, model, n)
        assert "AnsBackWriterT<256>" in enc and dec.endswith(", true>")
        have = _summary_kernels(os.path.join(ROOT, "profiles", f"r06_{tag}_kernel_trace_summary.txt"))
        assert enc in have and dec in have, f"{tag}: {enc} / {dec} not in the round-6 trace: {have[:4]}"
    # forced any-parameter kernels: the function names
    prev = L.scl_set_any_parameter_kernels(1)
    try:
        assert names("scl_rans_kernel_names", cases[0][2], 1000) == ("rans_encode_generic", "rans_decode_generic")
    finally:
        L.scl_set_any_parameter_kernels(prev)


@pytest.mark.parametrize("table", ["uniform1", "uniform16"])
def test_range_cooperative_line_store(table, dev):
    """the table-free range encoder stores its lines cooperatively (eight lanes per line after a register transpose) when a
    whole wave of equally long chunks reaches a flush point in step -- streams equal the oracle's for whole waves (random
    bytes keep their lanes within a byte or two of each other: most lines go out cooperatively, the rest lane by lane), for
    a partial last wave, and for slots too small for their stream (every chunk reports CAPACITY, nothing is written behind
    the last slot)"""
    from stanford_compression_library_amd.backend.models import EncodedBatch

    freq = np.ones(256, dtype=np.int64) if table == "uniform1" else np.full(256, 16, dtype=np.int64)
    model = models.RangeModel(freq.tolist(), 32, 32)
    rng = np.random.default_rng(len(table))
    n_chunks, chunk_len = 64 * 3 + 17, 1536
    sym_np = rng.integers(0, 256, (n_chunks, chunk_len), dtype=np.uint8)
    sym = torch.from_numpy(sym_np).to(dev)
    enc = model.encode_batch(sym)
    torch.cuda.synchronize()
    assert not enc.status.any().item()
    data, offs, nbits = enc.data.cpu().numpy(), enc.bit_offset.cpu().numpy(), enc.nbits.cpu().numpy()
    for c in list(range(0, n_chunks, 13)) + [63, 64, 191, 192, n_chunks - 1]:
        packed, nb = orc.range_encode(sym_np[c], freq.tolist())
        assert nb == nbits[c] and offs[c] % 8 == 0
        assert np.array_equal(data[offs[c] // 8: offs[c] // 8 + (nb + 7) // 8], packed[: (nb + 7) // 8]), c
    out, lens, used, status = model.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_len)
    torch.cuda.synchronize()
    assert torch.equal(out, sym) and not status.any().item()
    # slots of 1024 bytes for streams of ~1544: CAPACITY on every chunk, the bytes behind the last slot untouched
    stride = 1024
    buf = torch.full((n_chunks * stride + 16 + 4096,), FILL, dtype=torch.uint8, device=dev)
    small = EncodedBatch(buf[: n_chunks * stride + 16], stride, torch.empty(n_chunks, dtype=torch.int64, device=dev),
                         torch.empty(n_chunks, dtype=torch.int32, device=dev),
                         torch.empty(n_chunks, dtype=torch.int32, device=dev), n_chunks)
    model.encode_batch(sym, out=small)
    torch.cuda.synchronize()
    assert bool((small.status & backend_lib.ST_CAPACITY).bool().all().item())
    assert bool((buf[n_chunks * stride + 16:] == FILL).all().item())
