"""Device model handles + the two call shapes of the C ABI (include/scl_hip.h):

* ``encode_host`` / ``decode_host`` -- one chunk in host memory (what the drop-in ``encode_block`` /
  ``decode_block`` classes use; the library allocates, copies, runs N = 1 and synchronises);
* ``encode_batch`` / ``decode_batch`` -- N chunks already resident in HBM, given as torch tensors
  (torch is only the allocator / stream provider here).

Everything raises ``SclHipError`` when the HIP library or a device is missing: no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import lib as _lib


def _freq_array(freq_list) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray([int(f) for f in freq_list], dtype=np.int64))
    if arr.size and (arr.min() < 0 or arr.max() >= (1 << 32)):
        raise AssertionError("frequencies must fit an unsigned 32-bit integer")
    return arr.astype(np.uint32)


class _any_parameter:
    """context manager: the library's tuned kernels are kept out of THIS THREAD's batch calls while it is active
    (``scl_set_any_parameter_kernels``, a thread-local switch of the C ABI -- the process environment is not touched)"""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            self.prev = _lib.load().scl_set_any_parameter_kernels(1)

    def __exit__(self, *exc):
        if self.on:
            _lib.load().scl_set_any_parameter_kernels(self.prev)
        return False


@dataclass
class EncodedBatch:
    """Device-resident result of a batched encode: ``data`` holds one slot of ``stride`` bytes per chunk;
    stream c is the ``nbits[c]`` bits starting at absolute bit ``bit_offset[c]`` of ``data``.

    ``layout``: ``"linear"`` -- that bit position is a memory position; ``"striped"`` (ABI 8, rANS / tANS tuned kernels) --
    the same LOGICAL position in wave-striped slots: logical byte ``c * stride + b`` lives at ``(c // 64) * 64 * stride +
    (b // 16) * 1024 + 16 * (c % 64) + b % 16`` (``linear_data()`` undoes it; :func:`compact` and ``decode_encoded`` read
    either)."""

    data: "torch.Tensor"        # uint8 [n_chunks * stride + 16]  (striped: round_up(n_chunks, 64) * stride + 16)
    stride: int
    bit_offset: "torch.Tensor"  # uint64 as int64 [n_chunks]
    nbits: "torch.Tensor"       # uint32 as int32 [n_chunks]
    status: "torch.Tensor"      # uint32 as int32 [n_chunks]
    n_chunks: int
    layout: str = "linear"

    def linear_data(self):
        """``data`` in the linear layout (a copy for striped batches: a permutation of 16-byte pieces) -- for code that
        reads slots by ``bit_offset`` (tests, tools); the product path never needs it"""
        if self.layout == "linear":
            return self.data
        n64 = (self.n_chunks + 63) // 64
        body = self.data[: n64 * 64 * self.stride].view(n64, self.stride // 16, 64, 16).permute(0, 2, 1, 3).reshape(-1)
        import torch

        return torch.cat([body, self.data.new_zeros(16)])


# layout="auto": batches from this size on take the striped kernels.  They win by occupancy (four 256-lane workgroups per CU
# against two), so they need the chip full: measured on MI355X (256 CUs) 65 536 chunks -17 %, 131 072 chunks -6 % (two
# workgroups per CU either way), 196 608 chunks +2.7 %, 262 144 chunks +3.5 % on the round trip (profiles/r06_striped_ab.txt)
STRIPED_MIN_CHUNKS = 196608
# the range coder's linear kernels already run four workgroups per CU; what the striped form changes is the shape of the
# stores and of the decoder's loads: measured never slower on encode, 5 % faster on decode from 32 768 chunks of 4 KiB on
# (only the compaction of the result is slower, 0.08 against 0.05 ms per 128 MiB)
STRIPED_MIN_CHUNKS_BY_CODER = {"range": 32768}


class _DeviceModel:
    """Common part of the four model kinds; subclasses set ``_prefix`` and create the handle."""

    _prefix = ""
    _needs_scratch = False
    K = 0  # alphabet size; above 256 the symbols travel as uint16 indices through the *_u16 entry points

    def __init__(self):
        self._h = C.c_void_p()
        self._L = _lib.load()

    @property
    def wide(self) -> bool:
        return self.K > 256

    @property
    def sym_dtype(self):
        """numpy dtype of this model's symbol index arrays"""
        return np.uint16 if self.wide else np.uint8

    def _torch_sym_dtypes(self):
        import torch

        return (torch.uint16, torch.int16) if self.wide else (torch.uint8,)

    def _sym_fn(self, name):
        """entry point that takes / returns symbol arrays: the *_u16 twin for alphabets above 256"""
        return getattr(self._L, f"scl_{self._prefix}_{name}" + ("_u16" if self.wide else ""))

    def _host_ptr(self, a: np.ndarray):
        return C.c_void_p(a.ctypes.data) if self.wide else _lib.u8_ptr(a)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            getattr(self._L, f"scl_{self._prefix}_model_destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fn(self, name):
        return getattr(self._L, f"scl_{self._prefix}_{name}")

    def slot_bytes(self, n_symbols: int) -> int:
        return int(self._fn("slot_bytes")(self._h, int(n_symbols)))

    # -- one chunk, host memory -------------------------------------------------------------------
    def encode_host(self, sym: np.ndarray):
        """index array (``sym_dtype``) -> (packed MSB-first bytes, nbits), left-aligned like BitArray.tobytes()."""
        sym = np.ascontiguousarray(sym, dtype=self.sym_dtype)
        cap = self.slot_bytes(sym.size) + 16
        out = np.zeros(cap, dtype=np.uint8)
        nbits = C.c_uint64(0)
        rc = self._sym_fn("encode_host")(self._h, self._host_ptr(sym), sym.size, _lib.u8_ptr(out), cap, C.byref(nbits))
        _lib.check(rc, f"scl_{self._prefix}_encode_host")
        return out[: (nbits.value + 7) // 8], int(nbits.value)

    # decode_host sizes its host output from the stream's OWN size header: a damaged or hostile block can ask for up to
    # 2^32 symbols (a symbol can cost 0 bits, so the stream length is no bound).  The DEFAULT accepts whatever the
    # reference's default header (DATA_BLOCK_SIZE_BITS = 32) can announce -- reference parity: ``encode_block`` ->
    # ``decode_block`` of any block the reference codes must decode here too (a 2^24 default refused valid
    # blocks and broke the file decoder's one-block path).  A caller reading untrusted streams sets ``max_block_size`` on
    # the decoder (opt-in): above it the call raises the AssertionError the size check in ``encode_block`` raises (same
    # exception type as the reference's ``assert data_block.size < (1 << DATA_BLOCK_SIZE_BITS)``), before anything is
    # allocated.  Raised explicitly, not with ``assert``: ``python -O`` must not remove the guard.
    DEFAULT_MAX_BLOCK_SIZE = (1 << 32) - 1

    def _check_block_size(self, n: int, max_block_size: Optional[int]):
        cap = self.DEFAULT_MAX_BLOCK_SIZE if max_block_size is None else int(max_block_size)
        if int(n) > cap:
            raise AssertionError(f"encoded block announces {n} symbols, more than max_block_size = {cap} "
                                 "(damaged stream? set max_block_size to decode larger blocks)")

    def decode_host(self, packed: np.ndarray, nbits: int, size_bits: int, max_block_size: Optional[int] = None):
        """(packed bytes, available bits) -> (index array of ``sym_dtype``, num_bits_consumed).  ``max_block_size``
        (default ``DEFAULT_MAX_BLOCK_SIZE`` = 2^32 - 1 symbols): largest block size the header may announce."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        n = C.c_uint64(0)
        rc = self._L.scl_stream_block_size_host(_lib.u8_ptr(packed), int(nbits), int(size_bits), C.byref(n))
        _lib.check(rc, "scl_stream_block_size_host")
        self._check_block_size(n.value, max_block_size)
        out = np.zeros(max(int(n.value), 1), dtype=self.sym_dtype)
        n_out, used = C.c_uint64(0), C.c_uint64(0)
        rc = self._sym_fn("decode_host")(self._h, _lib.u8_ptr(packed), int(nbits), self._host_ptr(out), int(n.value),
                                         C.byref(n_out), C.byref(used))
        _lib.check(rc, f"scl_{self._prefix}_decode_host")
        return out[: n_out.value], int(used.value)

    # -- N chunks, device memory (torch tensors) ----------------------------------------------------
    def _scratch(self, n_chunks, device):
        import torch

        if not self._needs_scratch:
            return None, 0
        nbytes = int(self._L.scl_aec_scratch_bytes(self._h, int(n_chunks)))
        if nbytes == 0:
            return None, 0
        return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes

    def _keep_scratch(self, stream_handle, scratch):
        """the kernels read the scratch asynchronously: keep the tensor referenced until the next call on the SAME
        stream replaces it (stream order then guarantees the previous launch is done with the old one)"""
        if not hasattr(self, "_scratch_by_stream"):
            self._scratch_by_stream = {}
        if isinstance(stream_handle, tuple):  # encode_rows_into: one scratch per sub-batch, kept until replaced
            # ("rows", first row, rows of the whole shard): a shard of another shape starts over, so that a changing
            # shard shape cannot pile up one scratch per sub-batch start ever seen
            shard = stream_handle[2]
            side = stream_handle[3] if len(stream_handle) > 3 else 0
            for k in [k for k in self._scratch_by_stream if isinstance(k, tuple) and k[2] != shard]:
                old = self._scratch_by_stream.pop(k)
                # the evicted scratch was used on a side stream (raw handle in k[3] when the caller gave it): tell the
                # caching allocator, so that the memory is not handed out again before that stream is done with it
                old_side = k[3] if len(k) > 3 else 0
                if old is not None and old_side:
                    import torch

                    old.record_stream(torch.cuda.ExternalStream(int(old_side), device=old.device))
            if scratch is not None and side:
                import torch

                scratch.record_stream(torch.cuda.ExternalStream(int(side), device=scratch.device))
            self._scratch_by_stream[stream_handle] = scratch
            return
        self._scratch_by_stream[int(stream_handle or 0)] = scratch
        if scratch is not None and stream_handle:
            import torch

            if int(stream_handle) != torch.cuda.current_stream(scratch.device).cuda_stream:
                # allocated on torch's current stream, used on another: tell the caching allocator
                scratch.record_stream(torch.cuda.ExternalStream(int(stream_handle), device=scratch.device))

    def striped_ok(self) -> bool:
        """do the striped entry points (ABI 8) serve this model?  (rANS / tANS models on the tuned kernels)"""
        fn = getattr(self._L, f"scl_{self._prefix}_striped_ok", None)
        return bool(fn is not None and self._prefix in ("rans", "tans", "range") and not self.wide and fn(self._h))

    STRIPED_MAX_STRIDE = (1 << 24) - 16  # the striped kernels address a workgroup's 256 slots with 32-bit offsets

    def pick_layout(self, layout: Optional[str], n_chunks: int, any_parameter_kernels: bool = False,
                    stride: Optional[int] = None) -> str:
        """``"linear"`` / ``"striped"`` as asked (striped must be served), ``"auto"``: striped for batches that fill the
        chip when the model is served, else linear; ``None`` = linear"""
        if layout in (None, "linear"):
            return "linear"
        can = (self.striped_ok() and not any_parameter_kernels and not _lib.any_parameter_forced()
               and (stride is None or int(stride) <= self.STRIPED_MAX_STRIDE))
        if layout == "auto":
            return "striped" if can and n_chunks >= STRIPED_MIN_CHUNKS_BY_CODER.get(self._prefix, STRIPED_MIN_CHUNKS) else "linear"
        assert layout == "striped", f"unknown layout {layout!r}"
        if not can:
            raise ValueError("layout='striped': this model (or the any-parameter setting, or a slot stride of 16 MiB and "
                             "more) has no striped kernels")
        return "striped"

    def alloc_encoded(self, n_chunks: int, chunk_len: int, device, out_stride: Optional[int] = None,
                      layout: Optional[str] = None) -> EncodedBatch:
        """Output buffers of a batched encode (reusable across calls of the same shape).  ``layout``: see
        :class:`EncodedBatch` and :meth:`pick_layout`."""
        import torch

        stride = int(out_stride or self.slot_bytes(chunk_len))
        layout = self.pick_layout(layout, n_chunks, stride=stride)
        slots = (n_chunks + 63) // 64 * 64 if layout == "striped" else n_chunks
        return EncodedBatch(torch.empty(slots * stride + 16, dtype=torch.uint8, device=device), stride,
                            torch.empty(n_chunks, dtype=torch.int64, device=device),
                            torch.empty(n_chunks, dtype=torch.int32, device=device),
                            torch.empty(n_chunks, dtype=torch.int32, device=device), n_chunks, layout)

    def encode_batch(self, sym, lens=None, out_stride: Optional[int] = None, stream=None,
                     out: Optional[EncodedBatch] = None, any_parameter_kernels: bool = False,
                     layout: Optional[str] = None) -> EncodedBatch:
        """sym: uint8 CUDA tensor [n_chunks, chunk_len] (row-contiguous).  lens: optional int32 [n_chunks].
        ``out`` reuses buffers from :meth:`alloc_encoded` (and fixes the layout).  ``any_parameter_kernels`` (tests,
        stress tools) keeps the tuned kernels out (``SCL_ANY_PARAMETER_KERNELS=1`` for the duration of the call).
        ``layout``: ``None`` / ``"linear"``, ``"striped"``, ``"auto"`` (:meth:`pick_layout`)."""
        import torch

        assert sym.is_cuda and sym.dtype in self._torch_sym_dtypes() and sym.dim() == 2 and sym.stride(1) == 1
        n_chunks, chunk_len = sym.shape
        dev = sym.device
        # rows that do not start on 16-byte boundaries are re-laid INSIDE the library (RowRelay, csrc/scl_core.hip)
        if out is None:
            out = self.alloc_encoded(n_chunks, chunk_len, dev, out_stride,
                                     self.pick_layout(layout, n_chunks, any_parameter_kernels,
                                                      stride=int(out_stride or self.slot_bytes(chunk_len))))
        assert out.n_chunks == n_chunks
        striped = out.layout == "striped"
        assert not (striped and any_parameter_kernels), "striped slots are written by the tuned kernels only"
        stride, data, bit_off, nbits, status = out.stride, out.data, out.bit_offset, out.nbits, out.status
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        args = [self._h, sym.data_ptr(), sym.stride(0), lens.data_ptr() if lens is not None else None,
                chunk_len, n_chunks, data.data_ptr(), stride, bit_off.data_ptr(), nbits.data_ptr(),
                status.data_ptr()]
        if self._needs_scratch:
            scratch, nbytes = self._scratch(n_chunks, dev)
            args += [scratch.data_ptr() if scratch is not None else None, nbytes]
            self._keep_scratch(st, scratch)
        with torch.cuda.device(dev), _any_parameter(any_parameter_kernels):
            # the library launches on the CURRENT device and checks it is the model's
            rc = self._sym_fn("encode_batch_striped" if striped else "encode_batch")(*args, st)
        _lib.check(rc, f"scl_{self._prefix}_encode_batch" + ("_striped" if striped else ""))
        return out

    def encode_rows_into(self, sym, a: int, b: int, out: EncodedBatch, stream_handle: int):
        """rows a..b of ``sym`` -> slots a..b of ``out`` on the raw stream handle, by pointer arithmetic (no tensor
        views: the overlapped pipeline of backend/sharded.py calls this once per sub-batch).  The bit offsets written for
        those chunks count from SLOT a, i.e. from ``out.data.data_ptr() + a * out.stride``."""
        import torch

        n_rows, chunk_len = sym.shape
        assert 0 <= a <= b <= n_rows == out.n_chunks and sym.stride(1) == 1
        striped = out.layout == "striped"
        assert not striped or a % 64 == 0, "striped slots: a sub-batch starts on a multiple of 64 chunks"
        assert sym.stride(0) % 16 == 0 and sym.data_ptr() % 16 == 0, "rows must start on 16-byte boundaries"
        assert not self.wide, "the overlapped pipeline carries uint8 symbols"
        args = [self._h, sym.data_ptr() + a * sym.stride(0), sym.stride(0), None, chunk_len, b - a,
                out.data.data_ptr() + a * out.stride, out.stride, out.bit_offset.data_ptr() + 8 * a,
                out.nbits.data_ptr() + 4 * a, out.status.data_ptr() + 4 * a]
        if self._needs_scratch:
            scratch, nbytes = self._scratch(b - a, sym.device)
            args += [scratch.data_ptr() if scratch is not None else None, nbytes]
            self._keep_scratch(("rows", a, n_rows, int(stream_handle or 0)), scratch)
        with torch.cuda.device(sym.device):
            rc = self._fn("encode_batch_striped" if striped else "encode_batch")(*args, stream_handle)
        _lib.check(rc, f"scl_{self._prefix}_encode_batch" + ("_striped" if striped else ""))

    def alloc_decoded(self, n_chunks: int, chunk_cap: int, device):
        import torch

        out_stride = (int(chunk_cap) + 15) // 16 * 16
        return (torch.empty((n_chunks, out_stride), dtype=self._torch_sym_dtypes()[0], device=device),
                torch.empty(n_chunks, dtype=torch.int32, device=device),
                torch.empty(n_chunks, dtype=torch.int32, device=device),
                torch.empty(n_chunks, dtype=torch.int32, device=device))

    def decode_encoded(self, enc: EncodedBatch, chunk_cap: int, stream=None, out=None, any_parameter_kernels: bool = False):
        """:meth:`decode_batch` of an :class:`EncodedBatch` in whichever layout it has"""
        return self.decode_batch(enc.data, enc.bit_offset, enc.nbits, chunk_cap, stream=stream, out=out,
                                 any_parameter_kernels=any_parameter_kernels,
                                 striped_stride=enc.stride if enc.layout == "striped" else None)

    def decode_batch(self, data, bit_offset, nbits, chunk_cap: int, stream=None, out=None,
                     any_parameter_kernels: bool = False, striped_stride: Optional[int] = None):
        """-> (sym uint8 [n_chunks, chunk_cap], lens int32, consumed int32, status int32) on the device.
        ``out`` reuses buffers from :meth:`alloc_decoded`.  ``any_parameter_kernels`` (tests) keeps the tuned
        kernels out (``SCL_ANY_PARAMETER_KERNELS=1`` for the duration of the call).  ``striped_stride``: ``data`` holds
        wave-striped slots of that stride (stream c inside logical slot c; :class:`EncodedBatch`)."""
        import torch

        assert data.is_cuda and data.dtype == torch.uint8
        n_chunks = int(bit_offset.numel())
        dev = data.device
        sym, lens, used, status = out if out is not None else self.alloc_decoded(n_chunks, chunk_cap, dev)
        out_stride = sym.stride(0)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        striped = striped_stride is not None
        assert not (striped and any_parameter_kernels), "striped slots are read by the tuned kernels only"
        # (a striped reader walks whole wave-slots: the buffer must hold round_up(n_chunks, 64) slots)
        assert not striped or data.numel() >= (n_chunks + 63) // 64 * 64 * int(striped_stride), "striped buffer too short"
        args = [self._h, data.data_ptr(), int(striped_stride) if striped else data.numel(), bit_offset.data_ptr(),
                nbits.data_ptr(), n_chunks,
                sym.data_ptr(), out_stride, int(chunk_cap), lens.data_ptr(), used.data_ptr(), status.data_ptr()]
        if self._needs_scratch:
            scratch, nbytes = self._scratch(n_chunks, dev)
            args += [scratch.data_ptr() if scratch is not None else None, nbytes]
            self._keep_scratch(st, scratch)
        with torch.cuda.device(dev), _any_parameter(any_parameter_kernels):
            rc = self._sym_fn("decode_batch_striped" if striped else "decode_batch")(*args, st)
        _lib.check(rc, f"scl_{self._prefix}_decode_batch" + ("_striped" if striped else ""))
        return sym[:, :chunk_cap], lens, used, status

    def kernel_names(self, n_chunks: int, layout: str = "linear"):
        """(encode kernel, decode kernel) a batch of that size runs, as rocprofv3 prints them (rANS / tANS models)"""
        enc, dec = C.create_string_buffer(160), C.create_string_buffer(160)
        fn = f"scl_{self._prefix}_kernel_names" + ("_striped" if layout == "striped" else "")
        _lib.check(getattr(self._L, fn)(self._h, int(n_chunks), enc, dec, 160), fn)
        return enc.value.decode(), dec.value.decode()


class RansModel(_DeviceModel):
    _prefix = "rans"

    def __init__(self, freq_list, range_factor: int, num_bits_out: int, size_bits: int):
        super().__init__()
        f = _freq_array(freq_list)
        rc = self._L.scl_rans_model_create(_lib.u32_ptr(f), f.size, int(range_factor), int(num_bits_out),
                                           int(size_bits), C.byref(self._h))
        _lib.check(rc, "scl_rans_model_create")
        self.size_bits = int(size_bits)
        self.K = int(f.size)

    def info(self) -> _lib.RansInfo:
        info = _lib.RansInfo()
        _lib.check(self._L.scl_rans_model_info(self._h, C.byref(info)), "scl_rans_model_info")
        return info


class TansModel(_DeviceModel):
    _prefix = "tans"

    def __init__(self, freq_list, range_factor: int, size_bits: int):
        super().__init__()
        f = _freq_array(freq_list)
        rc = self._L.scl_tans_model_create(_lib.u32_ptr(f), f.size, int(range_factor), int(size_bits),
                                           C.byref(self._h))
        _lib.check(rc, "scl_tans_model_create")
        self.size_bits = int(size_bits)
        self.K = int(f.size)

    def info(self) -> _lib.RansInfo:
        info = _lib.RansInfo()
        _lib.check(self._L.scl_tans_model_info(self._h, C.byref(info)), "scl_tans_model_info")
        return info

    def tables(self):
        """The device-built lookup tables as numpy arrays (enc, nbits, thresh, dec_sym, dec_xs)."""
        L = int(self.info().L)
        enc, dsym, dxs = (np.zeros(L, np.uint32) for _ in range(3))
        nb, th = np.zeros(self.K, np.uint32), np.zeros(self.K, np.uint32)
        rc = self._L.scl_tans_model_tables(self._h, _lib.u32_ptr(enc), _lib.u32_ptr(nb), _lib.u32_ptr(th),
                                           _lib.u32_ptr(dsym), _lib.u32_ptr(dxs))
        _lib.check(rc, "scl_tans_model_tables")
        return dict(enc=enc, nbits=nb, thresh=th, dec_sym=dsym, dec_xs=dxs)


class RangeModel(_DeviceModel):
    _prefix = "range"

    def __init__(self, freq_list, precision: int, size_bits: int):
        super().__init__()
        f = _freq_array(freq_list)
        rc = self._L.scl_range_model_create(_lib.u32_ptr(f), f.size, int(precision), int(size_bits),
                                            C.byref(self._h))
        _lib.check(rc, "scl_range_model_create")
        self.size_bits = int(size_bits)
        self.K = int(f.size)

    def fast_path(self) -> bool:
        """True if the tuned kernels (scl_range_fast.hip) serve this model."""
        return bool(self._L.scl_range_fast_path(self._h))


class AecModel(_DeviceModel):
    _prefix = "aec"
    _needs_scratch = True

    def __init__(self, model_kind: int, freq_init, K: int, order_k: int, max_total: int, precision: int,
                 size_bits: int):
        super().__init__()
        f = _freq_array(freq_init if freq_init is not None else [1] * K)
        rc = self._L.scl_aec_model_create(int(model_kind), _lib.u32_ptr(f), int(K), int(order_k), int(max_total),
                                          int(precision), int(size_bits), C.byref(self._h))
        _lib.check(rc, "scl_aec_model_create")
        self.size_bits = int(size_bits)
        self.K = int(K)

    def fast_path(self, max_symbols: int) -> bool:
        """True if chunks of up to max_symbols symbols run on the per-lane-LDS-table kernels (scl_aec_fast.hip)."""
        return bool(self._L.scl_aec_fast_path(self._h, int(max_symbols)))

    # -- coder objects that live across blocks (reference quirk Q4; include/scl_hip.h "*_resume") -------------
    def state_counts(self) -> int:
        return int(self._L.scl_aec_state_counts(self._h))

    def encode_host_resume(self, sym: np.ndarray, counts: np.ndarray, past_k: np.ndarray):
        """one block of a coder whose model state is (counts, past_k); both arrays are updated in place"""
        sym = np.ascontiguousarray(sym, dtype=self.sym_dtype)
        assert counts.dtype == np.uint32 and past_k.dtype == np.uint32 and counts.flags.c_contiguous
        cap = self.slot_bytes(sym.size) + 16
        out = np.zeros(cap, dtype=np.uint8)
        nbits = C.c_uint64(0)
        rc = self._sym_fn("encode_host_resume")(self._h, self._host_ptr(sym), sym.size, _lib.u8_ptr(out), cap,
                                                C.byref(nbits), _lib.u32_ptr(counts), _lib.u32_ptr(past_k))
        _lib.check(rc, "scl_aec_encode_host_resume")
        return out[: (nbits.value + 7) // 8], int(nbits.value)

    def decode_host_resume(self, packed: np.ndarray, nbits: int, size_bits: int, counts: np.ndarray,
                           past_k: np.ndarray, max_block_size: Optional[int] = None):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        assert counts.dtype == np.uint32 and past_k.dtype == np.uint32 and counts.flags.c_contiguous
        n = C.c_uint64(0)
        rc = self._L.scl_stream_block_size_host(_lib.u8_ptr(packed), int(nbits), int(size_bits), C.byref(n))
        _lib.check(rc, "scl_stream_block_size_host")
        self._check_block_size(n.value, max_block_size)
        out = np.zeros(max(int(n.value), 1), dtype=self.sym_dtype)
        n_out, used = C.c_uint64(0), C.c_uint64(0)
        rc = self._sym_fn("decode_host_resume")(self._h, _lib.u8_ptr(packed), int(nbits), self._host_ptr(out),
                                                int(n.value), C.byref(n_out), C.byref(used), _lib.u32_ptr(counts),
                                                _lib.u32_ptr(past_k))
        _lib.check(rc, "scl_aec_decode_host_resume")
        return out[: n_out.value], int(used.value)

    def alloc_state(self, n_coders: int, device):
        """device state of n fresh coder objects (for encode_batch_resume / decode_batch_resume).  The buffer's
        layout depends on ``n_coders``; it is remembered on the tensor (``state.n_coders``) and handed to every
        resume call, which refuses batches longer than that."""
        import torch

        nbytes = int(self._L.scl_aec_state_bytes(self._h, int(n_coders)))
        state = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            rc = self._L.scl_aec_state_reset(self._h, state.data_ptr(), state.numel(), int(n_coders),
                                             torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "scl_aec_state_reset")
        state.n_coders = int(n_coders)
        return state

    @staticmethod
    def _state_coders(state, n_chunks: int) -> int:
        n_coders = getattr(state, "n_coders", None)
        if n_coders is None:
            raise AssertionError("state does not come from AecModel.alloc_state (its layout depends on n_coders)")
        if n_chunks > n_coders:
            raise AssertionError(f"{n_chunks} chunks but the state holds {n_coders} coders")
        return int(n_coders)

    def state_download(self, state, n_coders: int, coder: int):
        import torch

        assert int(n_coders) == self._state_coders(state, 0), "n_coders differs from the one the state was made with"
        counts = np.zeros(max(self.state_counts(), 1), np.uint32)
        past = np.zeros(4, np.uint32)
        with torch.cuda.device(state.device):
            rc = self._L.scl_aec_state_download(self._h, state.data_ptr(), int(n_coders), int(coder),
                                                _lib.u32_ptr(counts), _lib.u32_ptr(past),
                                                torch.cuda.current_stream(state.device).cuda_stream)
        _lib.check(rc, "scl_aec_state_download")
        return counts[: self.state_counts()], past

    def encode_batch_resume(self, sym, state, lens=None, out_stride: Optional[int] = None, stream=None,
                            out: Optional[EncodedBatch] = None) -> EncodedBatch:
        """chunk c CONTINUES coder c of ``state`` (from :meth:`alloc_state`) and leaves the advanced state there"""
        import torch

        assert sym.is_cuda and sym.dtype in self._torch_sym_dtypes() and sym.dim() == 2 and sym.stride(1) == 1
        n_chunks, chunk_len = sym.shape
        n_coders = self._state_coders(state, n_chunks)
        if out is None:
            out = self.alloc_encoded(n_chunks, chunk_len, sym.device, out_stride)
        st = stream if stream is not None else torch.cuda.current_stream(sym.device).cuda_stream
        with torch.cuda.device(sym.device):
            rc = self._sym_fn("encode_batch_resume")(
                self._h, sym.data_ptr(), sym.stride(0), lens.data_ptr() if lens is not None else None, chunk_len,
                n_chunks, out.data.data_ptr(), out.stride, out.bit_offset.data_ptr(), out.nbits.data_ptr(),
                out.status.data_ptr(), state.data_ptr(), state.numel(), n_coders, st)
        _lib.check(rc, "scl_aec_encode_batch_resume")
        return out

    def decode_batch_resume(self, data, bit_offset, nbits, chunk_cap: int, state, stream=None, out=None):
        import torch

        n_chunks = int(bit_offset.numel())
        n_coders = self._state_coders(state, n_chunks)
        dev = data.device
        sym, lens, used, status = out if out is not None else self.alloc_decoded(n_chunks, chunk_cap, dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            rc = self._sym_fn("decode_batch_resume")(
                self._h, data.data_ptr(), data.numel(), bit_offset.data_ptr(), nbits.data_ptr(), n_chunks,
                sym.data_ptr(), sym.stride(0), int(chunk_cap), lens.data_ptr(), used.data_ptr(), status.data_ptr(),
                state.data_ptr(), state.numel(), n_coders, st)
        _lib.check(rc, "scl_aec_decode_batch_resume")
        return sym[:, :chunk_cap], lens, used, status


def compact_into(enc: EncodedBatch, out, offsets, scratch, framed: bool = False, stream=None):
    """Asynchronous form of :func:`compact`: the dense (or framed) concatenation of ``enc`` goes into caller-owned
    device buffers -- ``out`` uint8 (capacity >= every record: ``compact_capacity``), ``offsets`` int64 [n + 1],
    ``scratch`` uint8 [``scl_streams_compact_scratch_bytes(n)``] -- on ``stream`` (a torch stream; default: current).
    Nothing synchronises; ``offsets[-1]`` is the total once the stream has got there."""
    import torch

    L = _lib.load()
    dev = enc.data.device
    st = stream if stream is not None else torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        rc = _compact_call(L, enc, 0, enc.n_chunks, _lib.COMPACT_FRAMED if framed else _lib.COMPACT_DENSE, out.data_ptr(),
                           out.numel(), offsets.data_ptr(), None, scratch.data_ptr(), st.cuda_stream)
    _lib.check(rc, "scl_streams_compact")


def _compact_call(L, enc: EncodedBatch, a: int, b: int, mode: int, out_ptr: int, out_cap: int, offsets_ptr: int, base_ptr,
                  scratch_ptr: int, stream_handle):
    """chunks a..b of ``enc`` through scl_streams_compact_at or, for striped batches, scl_streams_compact_striped (a must
    then be a multiple of 64: a sub-batch starts with a whole wave)"""
    if enc.layout == "striped":
        assert a % 64 == 0
        return L.scl_streams_compact_striped(enc.data.data_ptr() + a * enc.stride, enc.stride,
                                             enc.bit_offset.data_ptr() + 8 * a, enc.nbits.data_ptr() + 4 * a, b - a, mode,
                                             out_ptr, out_cap, offsets_ptr, base_ptr, scratch_ptr, stream_handle)
    return L.scl_streams_compact_at(enc.data.data_ptr() + a * enc.stride, enc.bit_offset.data_ptr() + 8 * a,
                                    enc.nbits.data_ptr() + 4 * a, b - a, mode, out_ptr, out_cap, offsets_ptr, base_ptr,
                                    scratch_ptr, stream_handle)


def compact_capacity(n_chunks: int, stride: int, framed: bool = False) -> int:
    """bytes that always hold the compaction of ``n_chunks`` slots of ``stride`` bytes"""
    return int(n_chunks) * (int(stride) + (6 if framed else 1)) + 16


def compact_scratch_bytes(n_chunks: int) -> int:
    return int(_lib.load().scl_streams_compact_scratch_bytes(int(n_chunks)))


def compact(enc: EncodedBatch, framed: bool = False, stream=None):
    """Dense (or EncodedBlockWriter-framed) concatenation of a batch: -> (bytes tensor, int64 offsets[n+1])."""
    import torch

    L = _lib.load()
    dev = enc.data.device
    n = enc.n_chunks
    tstream = stream if isinstance(stream, torch.cuda.Stream) else None
    if stream is not None and tstream is None:
        tstream = torch.cuda.ExternalStream(int(stream), device=dev)  # a raw hipStream_t
    if tstream is None:
        tstream = torch.cuda.current_stream(dev)
    # capacity: every record is at most ceil(nbits/8) (+5 framed) bytes
    with torch.cuda.stream(tstream):  # nbits were written on that stream
        total_bits = int(enc.nbits.to(torch.int64).sum().item())
    cap = total_bits // 8 + n * (6 if framed else 1) + 16
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    scratch = torch.empty(int(L.scl_streams_compact_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _compact_call(L, enc, 0, n, _lib.COMPACT_FRAMED if framed else _lib.COMPACT_DENSE, out.data_ptr(), cap,
                           offsets.data_ptr(), None, scratch.data_ptr(), tstream.cuda_stream)
    _lib.check(rc, "scl_streams_compact")
    tstream.synchronize()  # the stream the kernels were queued on: results readable, scratch reusable on return
    return out, offsets


_index_cache = threading.local()  # framed_index_host's three index arrays, reused from slab to slab (per thread)


def _index_arrays(cap: int):
    have = getattr(_index_cache, "arrays", None)
    if have is None or have[0].size < cap:
        have = tuple(np.empty(cap, dtype=np.uint64) for _ in range(3))
        _index_cache.arrays = have
    return have


def framed_index_host(buf: np.ndarray, size_bits: int, max_records: Optional[int] = None, partial: bool = False):
    """Index of a framed block file held in host memory (``scl_framed_index_host``: the walk
    ``EncodedBlockReader.get_block`` + ``Padder.remove_byte_padding`` make one record at a time, encoded_stream.py:196-225,
    :48-58) -> (bit_offset uint64[n], nbits uint64[n], block_size uint64[n], consumed_bytes).  ``buf`` is a contiguous
    uint8 array; records that cross its end are left for the caller (``consumed_bytes`` says where they start).  A
    malformed record raises ``AssertionError`` -- the exception the reference's reader asserts with; with ``partial`` the
    call returns a fifth element instead: ``None``, or the message of that error, the four others then describing the
    valid records IN FRONT of the malformed one (the reference decodes and writes those before it raises).

    The index arrays (worst case one record per 5 bytes: 4.8 x the buffer) are kept per thread and reused; what is
    returned are copies of the used part."""
    assert buf.dtype == np.uint8 and buf.ndim == 1 and buf.flags.c_contiguous
    L = _lib.load()
    # a record is at least 5 bytes (4-byte size + one payload byte)
    cap = int(buf.size // 5 + 1 if max_records is None else max_records)
    offs, nbits, sizes = _index_arrays(cap)
    n, used = C.c_uint64(0), C.c_uint64(0)
    rc = L.scl_framed_index_host(buf.ctypes.data if buf.size else None, buf.size, int(size_bits), cap,
                                 offs.ctypes.data, nbits.ctypes.data, sizes.ctypes.data, C.byref(n), C.byref(used))
    err = None
    if rc == _lib.E_PARAM:
        err = _lib.last_error()
        if not partial:
            raise AssertionError(err)
    else:
        _lib.check(rc, "scl_framed_index_host")
    k = int(n.value)
    out = (offs[:k].copy(), nbits[:k].copy(), sizes[:k].copy(), int(used.value))
    return out + (err,) if partial else out


class DensePipeline:
    """Encode a batch straight to DENSE streams (every chunk's ``BitArray.tobytes()`` back to back + int64 offsets[n + 1],
    or the reference's framed file bytes) in sub-batches on two streams: the compaction of sub-batch i runs while sub-batch
    i + 1 is still being encoded.  The encoders fill slots back to front and only know a stream's length at its end, so the
    left-align / compaction pass is a second pass over the output whatever one does; what can be hidden is its time -- the
    encoders are bound by instruction issue, the compaction by memory.  ``scl_streams_compact_at`` keeps the offsets on the
    device (sub-batch i + 1 starts where sub-batch i ended), so nothing here waits for the host.

    Buffers are owned by the object (worst-case sizes, reused across calls); results are valid once the caller's current
    stream has passed the call.  Static-model coders only (rANS / tANS / range / FixedFreqModel: no per-call scratch)."""

    def __init__(self, model, n_chunks: int, chunk_len: int, device, n_sub: int = 2, framed: bool = False,
                 layout: Optional[str] = None):
        import torch

        assert not model._needs_scratch and not model.wide
        self.model, self.n_chunks, self.chunk_len, self.framed = model, int(n_chunks), int(chunk_len), bool(framed)
        n_sub = max(1, min(int(n_sub), self.n_chunks))
        self.bounds = [self.n_chunks * i // n_sub for i in range(n_sub + 1)]
        self.enc = model.alloc_encoded(self.n_chunks, self.chunk_len, device, layout=layout)
        if self.enc.layout == "striped":  # a sub-batch starts with a whole wave of 64 slots
            self.bounds = sorted({min(self.n_chunks, (b + 63) // 64 * 64) for b in self.bounds[:-1]} | {self.n_chunks})
        self.dense = torch.empty(compact_capacity(self.n_chunks, self.enc.stride, framed), dtype=torch.uint8, device=device)
        self.offsets = torch.empty(self.n_chunks + 1, dtype=torch.int64, device=device)
        per = max(b - a for a, b in zip(self.bounds, self.bounds[1:]))
        self.scr = compact_scratch_bytes(per)
        self.scratch = torch.empty(self.scr * n_sub, dtype=torch.uint8, device=device)
        self.s_enc, self.s_cmp = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)

    def run(self, sym):
        """-> (dense uint8 tensor, int64 offsets [n + 1]); ``offsets[-1]`` bytes of ``dense`` are the streams"""
        import torch

        L = _lib.load()
        dev = sym.device
        assert sym.shape == (self.n_chunks, self.chunk_len) and sym.stride(1) == 1
        cur = torch.cuda.current_stream(dev)
        self.s_enc.wait_stream(cur)
        self.s_cmp.wait_stream(cur)
        enc, stride = self.enc, self.enc.stride
        mode = _lib.COMPACT_FRAMED if self.framed else _lib.COMPACT_DENSE
        with torch.cuda.device(dev):
            for i, (a, b) in enumerate(zip(self.bounds, self.bounds[1:])):
                self.model.encode_rows_into(sym, a, b, enc, self.s_enc.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(self.s_enc)
                self.s_cmp.wait_event(ev)
                rc = _compact_call(
                    L, enc, a, b, mode, self.dense.data_ptr(), self.dense.numel(), self.offsets.data_ptr() + 8 * a,
                    (self.offsets.data_ptr() + 8 * a) if i else None,  # starts where the previous sub-batch ended
                    self.scratch.data_ptr() + self.scr * i, self.s_cmp.cuda_stream)
                _lib.check(rc, "scl_streams_compact_at")
        cur.wait_stream(self.s_cmp)
        cur.wait_stream(self.s_enc)
        return self.dense, self.offsets
