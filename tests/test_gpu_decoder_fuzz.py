"""GPU: the decoders face untrusted input.  For every decoder family -- rANS (tuned NUM_BITS_OUT = 1 incl. a total that is
not a power of two, tuned NUM_BITS_OUT = 8, any-parameter), tANS (table-free and lookup-table kernels), range coder (tuned
with and without slot table, any-parameter), arithmetic coder (static model, adaptive with LDS tables, adaptive i.i.d.,
order-k on dense and on sparse device-memory rows, any-parameter) and the uint16 twins -- valid batches are damaged in four
ways:

  (i)   random bit flips inside the streams,
  (ii)  streams overwritten with random bytes,
  (iii) ``in_nbits`` cut short (down to less than the header),
  (iv)  a lie in the size header (up to ``out_cap`` and beyond, all ones),

and decoded into buffers carved out of an arena pre-filled with 0xA5 (4 KiB guard bands, as in test_gpu_guard_bands.py).
Required: the call returns, no HIP error, no byte outside any output buffer is written, every damaged chunk either decodes
to *something* with status 0 or reports SCL_ST_TRUNCATED / SCL_ST_STATE / SCL_ST_CAPACITY / SCL_ST_SIZE (the status-word
form of the reference's ``assert state == INITIAL_STATE`` rANS.py:295, ``num_bits_consumed == len(block)``
data_encoder_decoder.py:141, and of reading past the end), ``out_lens <= out_cap`` for every chunk that wrote symbols,
and the undamaged neighbours decode bit-exactly with status 0.

``SCL_FUZZ_SEEDS`` (default 2) rounds per family and damage kind; the soak logged under profiles/ ran 2000+.
"""
import os
import zlib

import numpy as np
import pytest

from conftest import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_guard_bands import FILL, GUARD, Arena  # noqa: E402

ST_CAPACITY, ST_SYMBOL, ST_TRUNCATED, ST_STATE, ST_TOTAL, ST_SIZE = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
ALLOWED = ST_CAPACITY | ST_TRUNCATED | ST_STATE | ST_SIZE
SEEDS = int(os.environ.get("SCL_FUZZ_SEEDS", "2"))


def _families():
    from stanford_compression_library_amd import bench_data
    from stanford_compression_library_amd.backend import lib, models

    t256 = bench_data.t256_table()
    ones = np.ones(256, dtype=np.int64)
    odd_total = np.array([3, 1, 7, 2, 19, 40, 5, 9], dtype=np.int64)  # total 86: the exact-division rANS decoder
    wide = np.random.default_rng(5).integers(1, 20, 700).astype(np.int64)

    def f(make, K, freq, env=None, any_par=False, u16=False, striped=False):
        return dict(make=make, K=K, freq=freq, env=env or {}, any_par=any_par, u16=u16, striped=striped)

    return {
        "rans_b1": f(lambda: models.RansModel(t256.tolist(), 1 << 16, 1, 32), 256, t256),
        "rans_b1_total86": f(lambda: models.RansModel(odd_total.tolist(), 1 << 16, 1, 32), 8, odd_total),
        "rans_b8": f(lambda: models.RansModel(t256.tolist(), 1 << 8, 8, 32), 256, t256),
        "rans_any": f(lambda: models.RansModel(t256.tolist(), 1 << 16, 1, 32), 256, t256, any_par=True),
        # wave-striped slots (ABI 8): the same decoders behind AnsBitReaderT -- piece-granular refill, per-lane row offsets
        # clamped to the lane's slot
        "rans_b1_striped": f(lambda: models.RansModel(t256.tolist(), 1 << 16, 1, 32), 256, t256, striped=True),
        "rans_b1_total86_striped": f(lambda: models.RansModel(odd_total.tolist(), 1 << 16, 1, 32), 8, odd_total, striped=True),
        "rans_b8_striped": f(lambda: models.RansModel(t256.tolist(), 1 << 8, 8, 32), 256, t256, striped=True),
        "tans_tablefree_striped": f(lambda: models.TansModel(t256.tolist(), 1, 32), 256, t256, striped=True),
        "range_t256_striped": f(lambda: models.RangeModel(t256.tolist(), 32, 32), 256, t256, striped=True),
        "range_uniform1_striped": f(lambda: models.RangeModel([1] * 256, 32, 32), 256, ones, striped=True),
        "tans_tablefree": f(lambda: models.TansModel(t256.tolist(), 1, 32), 256, t256),
        "tans_table": f(lambda: models.TansModel(t256.tolist(), 1, 32), 256, t256, env={"SCL_TANS_KERNELS": "table"}),
        "range_t256": f(lambda: models.RangeModel(t256.tolist(), 32, 32), 256, t256),
        "range_uniform1": f(lambda: models.RangeModel([1] * 256, 32, 32), 256, ones),
        "range_any": f(lambda: models.RangeModel(t256.tolist(), 32, 32), 256, t256, any_par=True),
        "aec_static": f(lambda: models.AecModel(lib.MODEL_FIXED, t256.tolist(), 256, 0, 1 << 30, 32, 32), 256, t256),
        "aec_lds_k16": f(lambda: models.AecModel(lib.MODEL_ORDERK, None, 16, 1, 1 << 30, 32, 32), 16, np.ones(16, dtype=np.int64)),
        "aec_iid256": f(lambda: models.AecModel(lib.MODEL_IID, [1] * 256, 256, 0, 1 << 30, 32, 32), 256, t256),
        "aec_sparse_k256": f(lambda: models.AecModel(lib.MODEL_ORDERK, None, 256, 1, 1 << 30, 32, 32), 256, t256),
        "aec_wide_k40": f(lambda: models.AecModel(lib.MODEL_ORDERK, None, 40, 1, 1 << 30, 32, 32), 40, np.ones(40, dtype=np.int64),
                          env={"SCL_AEC_WIDE": "dense"}),
        "aec_any_k16": f(lambda: models.AecModel(lib.MODEL_ORDERK, None, 16, 1, 1 << 30, 32, 32), 16, np.ones(16, dtype=np.int64),
                         any_par=True),
        "rans_u16": f(lambda: models.RansModel(wide.tolist(), 1 << 16, 1, 32), 700, wide, u16=True),
        "range_u16": f(lambda: models.RangeModel(wide.tolist(), 32, 32), 700, wide, u16=True),
        "aec_u16": f(lambda: models.AecModel(lib.MODEL_IID, [1] * 700, 700, 0, 1 << 30, 32, 32), 700, wide, u16=True),
    }


def _damage(kind, rng, data, offs, nbits, victims, stride, chunk_len, size_bits=32):
    """damages the victims' streams in place (numpy views of the encoded batch); returns the in_nbits to pass"""
    in_nbits = nbits.copy()
    for c in victims:
        o, nb = int(offs[c]), int(nbits[c])
        lo, hi = o // 8, (o + nb + 7) // 8
        if kind == "flip":
            for _ in range(int(rng.integers(1, 9))):
                bit = o + int(rng.integers(0, max(nb, 1)))
                data[bit // 8] ^= np.uint8(0x80 >> (bit % 8))
        elif kind == "random":
            data[lo:hi] = rng.integers(0, 256, hi - lo, dtype=np.uint8)
        elif kind == "truncate":
            in_nbits[c] = int(rng.choice([0, 1, size_bits - 1, size_bits, size_bits + 5, nb // 2, max(nb - 1, 0), max(nb - 9, 0)]))
        elif kind == "size":
            lie = int(rng.choice([chunk_len + 1, chunk_len + 16, 2 * chunk_len, (1 << 31) - 1, (1 << 32) - 1, chunk_len,
                                  max(chunk_len - 1, 0), 0]))
            # the size header is the first size_bits bits of the stream, MSB first, at an arbitrary bit offset
            for i in range(size_bits):
                bit = o + i
                v = (lie >> (size_bits - 1 - i)) & 1
                mask = np.uint8(0x80 >> (bit % 8))
                data[bit // 8] = (data[bit // 8] & ~mask) | (mask if v else np.uint8(0))
    return in_nbits


@pytest.mark.parametrize("kind", ["flip", "random", "truncate", "size"])
@pytest.mark.parametrize("name", sorted(_families()) if torch.cuda.is_available() else [])
def test_damaged_streams(name, kind, monkeypatch):
    from stanford_compression_library_amd.backend import lib

    lib.require_device()
    dev = torch.device("cuda:0")
    fam = _families()[name]
    for k, v in fam["env"].items():
        monkeypatch.setenv(k, v)
    model = fam["make"]()
    K, freq = fam["K"], fam["freq"]
    L = lib.load()
    sym_np_t, sym_t = (np.uint16, torch.int16) if fam["u16"] else (np.uint8, torch.uint8)
    for seed in range(SEEDS):
        rng = np.random.default_rng(zlib.crc32(f"{name}/{kind}/{seed}".encode()))
        n_chunks = int(rng.choice([64, 130, 256, 333]))
        chunk_len = int(rng.choice([48, 200, 700, 1024]))
        row = (chunk_len + 15) // 16 * 16
        p = freq / freq.sum()
        host_sym = rng.choice(K, size=(n_chunks, row), p=p).astype(sym_np_t)
        sym = torch.from_numpy(host_sym.view(np.int16) if fam["u16"] else host_sym).to(dev)[:, :chunk_len]
        enc = model.encode_batch(sym, any_parameter_kernels=fam["any_par"], layout="striped" if fam["striped"] else None)
        torch.cuda.synchronize()
        assert int(enc.status.abs().sum()) == 0
        # (striped batches are damaged in their LINEAR view -- what bit_offset indexes -- and striped again below)
        data = enc.linear_data().cpu().numpy().copy()
        offs, nbits = enc.bit_offset.cpu().numpy(), enc.nbits.cpu().numpy()
        victims = rng.choice(n_chunks, size=max(1, n_chunks // 3), replace=False)
        in_nbits = _damage(kind, rng, data, offs, nbits, victims, enc.stride, chunk_len)
        damaged = np.zeros(n_chunks, dtype=bool)
        damaged[victims] = True

        if fam["striped"]:
            n64, S = (n_chunks + 63) // 64, enc.stride
            body = data[: n64 * 64 * S].reshape(n64, 64, S // 16, 16).transpose(0, 2, 1, 3).reshape(-1)
            data = np.concatenate([body, np.zeros(16, dtype=np.uint8)])
        esz = 2 if fam["u16"] else 1
        out_stride = row + (0 if not fam["any_par"] else 8)  # symbols per row
        arena = Arena(data.size + n_chunks * (out_stride * esz + 64) + 40 * GUARD + (1 << 20), dev)
        d_in = arena.take(data.size)
        d_in.copy_(torch.from_numpy(data))
        d_nbits = arena.take(4 * n_chunks, torch.int32)
        d_nbits.copy_(torch.from_numpy(in_nbits.astype(np.int32)))
        sym_out = arena.take(n_chunks * out_stride * esz, sym_t, (n_chunks, out_stride))
        lens_out, used, status = (arena.take(4 * n_chunks, torch.int32) for _ in range(3))
        dargs = [model._h, d_in.data_ptr(), enc.stride if fam["striped"] else d_in.numel(), enc.bit_offset.data_ptr(),
                 d_nbits.data_ptr(), n_chunks,
                 sym_out.data_ptr(), out_stride, chunk_len, lens_out.data_ptr(), used.data_ptr(), status.data_ptr()]
        keep = None
        if model._needs_scratch:
            keep, nb = model._scratch(n_chunks, dev)
            dargs += [keep.data_ptr() if keep is not None else None, nb]
        prev = L.scl_set_any_parameter_kernels(1 if fam["any_par"] else -1)
        try:
            rc = model._sym_fn("decode_batch_striped" if fam["striped"] else "decode_batch")(
                *dargs, torch.cuda.current_stream(dev).cuda_stream)
        finally:
            L.scl_set_any_parameter_kernels(prev)
        lib.check(rc, f"{name} decode_batch")
        torch.cuda.synchronize()  # a fault or a hang would surface here
        arena.check(f"{name} {kind} seed {seed}")

        st = status.cpu().numpy().astype(np.uint32)
        ln = lens_out.cpu().numpy().astype(np.uint32)
        got = sym_out.cpu().numpy()
        if fam["u16"]:
            got = got.view(np.uint16)
        assert not (st & ~np.uint32(ALLOWED)).any(), f"{name} {kind}: unexpected status bits {sorted(set(st.tolist()))}"
        # undamaged neighbours: bit-exact, status 0
        ok = ~damaged
        assert (st[ok] == 0).all() and (ln[ok] == chunk_len).all(), f"{name} {kind}: an undamaged chunk was affected"
        assert np.array_equal(got[ok][:, :chunk_len], host_sym[ok][:, :chunk_len])
        assert np.array_equal(used.cpu().numpy()[ok], nbits[ok])
        # damaged chunks: a clean status means the decoder produced a block that fits
        clean = damaged & (st == 0)
        assert (ln[clean] <= chunk_len).all(), f"{name} {kind}: status 0 with out_lens > out_cap"
        if kind == "truncate" and model._prefix != "aec":
            # (the arithmetic decoder reads zeros past the end of its input, like the reference's -- arithmetic_coding.py:
            # 222-229, 258-261 -- and may well finish; the ANS and range decoders consume every bit the encoder wrote)
            cut = damaged & (in_nbits < nbits)
            assert (st[cut] != 0).all(), f"{name}: a stream cut short decoded without any status bit"


@pytest.mark.parametrize("name", ["aec_lds_k16", "aec_iid256", "aec_static", "aec_sparse_k256"])
@pytest.mark.parametrize("nbytes", [1, 3, 4, 7])
def test_input_buffers_of_a_few_bytes(name, nbytes):
    """The tuned arithmetic decoders load whole 32-bit words: a buffer shorter than one word goes to the any-parameter kernel
    (scl_aec_decode_batch), a buffer of one or two words must not be read past its end.  Either way: a status, no fault."""
    from stanford_compression_library_amd.backend import lib

    lib.require_device()
    dev = torch.device("cuda:0")
    model = _families()[name]["make"]()
    n_chunks = 70
    arena = Arena(1 << 20, dev)
    d_in = arena.take(nbytes)
    d_in.copy_(torch.arange(nbytes, dtype=torch.uint8) * 37 + 11)
    offs = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
    nb = torch.full((n_chunks,), 8 * nbytes, dtype=torch.int32, device=dev)
    nb[1::2] = 200  # a lie: more bits than the buffer holds
    sym, lens, used, status = model.decode_batch(d_in, offs, nb, 64)
    torch.cuda.synchronize()
    arena.check(f"{name} {nbytes}-byte input")
    st = status.cpu().numpy().astype(np.uint32)
    assert not (st & ~np.uint32(ALLOWED)).any()
    assert (lens.cpu().numpy()[st == 0] <= 64).all()
