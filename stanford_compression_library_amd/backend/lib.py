"""ctypes binding of ``libscl_hip.so`` (the C ABI declared in include/scl_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing, cannot be loaded or reports
no device, the product path raises.  (The CPU oracle lives in oracle/ and is test infrastructure.)
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_DIR, "libscl_hip.so")

OK = 0
E_PARAM, E_ALLOC, E_HIP, E_NODEVICE, E_CHUNK = -2, -7, -8, -9, -10
ST_CAPACITY, ST_SYMBOL, ST_TRUNCATED, ST_STATE, ST_TOTAL, ST_SIZE = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
MODEL_FIXED, MODEL_IID, MODEL_ORDERK = 0, 1, 2
COMPACT_DENSE, COMPACT_FRAMED = 0, 1


class SclHipError(RuntimeError):
    """A call into libscl_hip.so failed (carries the status code and the library's message)."""

    def __init__(self, code: int, what: str, message: str):
        super().__init__(f"{what}: status {code}: {message}")
        self.code = code
        self.message = message


class RansInfo(C.Structure):
    _fields_ = [("M", C.c_uint64), ("L", C.c_uint64), ("H", C.c_uint64), ("K", C.c_uint32),
                ("num_state_bits", C.c_uint32), ("size_bits", C.c_uint32), ("num_bits_out", C.c_uint32),
                ("max_bits_per_symbol", C.c_uint32), ("fast_path", C.c_uint32), ("device", C.c_int32)]


_u8p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_vp, _u32, _u64, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int

# name -> (restype, argtypes); device pointers travel as void* (integers from tensor.data_ptr())
_ENC_BATCH = [_vp, _vp, _u64, _vp, _u32, _u64, _vp, _u64, _vp, _vp, _vp, _vp]
_DEC_BATCH = [_vp, _vp, _u64, _vp, _vp, _u64, _vp, _u64, _u32, _vp, _vp, _vp, _vp]
_ENC_HOST = [_vp, _u8p, _u64, _u8p, _u64, _u64p]
_DEC_HOST = [_vp, _u8p, _u64, _u8p, _u64, _u64p, _u64p]
_SIGNATURES = {
    "scl_last_error": (C.c_char_p, []),
    "scl_device_count": (_int, [C.POINTER(_int)]),
    "scl_abi_version": (_int, []),
    "scl_rans_encoder_kind": (_int, [_vp, _u64]),
    "scl_rans_kernel_names": (_int, [_vp, _u64, C.c_char_p, C.c_char_p, _u64]),
    "scl_tans_kernel_names": (_int, [_vp, _u64, C.c_char_p, C.c_char_p, _u64]),
    "scl_set_any_parameter_kernels": (_int, [_int]),
    "scl_rans_model_create": (_int, [_u32p, _u32, _u64, _u32, _u32, C.POINTER(_vp)]),
    "scl_rans_model_destroy": (None, [_vp]),
    "scl_rans_model_info": (_int, [_vp, C.POINTER(RansInfo)]),
    "scl_rans_slot_bytes": (_u64, [_vp, _u64]),
    "scl_rans_encode_batch": (_int, _ENC_BATCH),
    "scl_rans_decode_batch": (_int, _DEC_BATCH),
    "scl_rans_encode_host": (_int, _ENC_HOST),
    "scl_rans_decode_host": (_int, _DEC_HOST),
    "scl_tans_model_create": (_int, [_u32p, _u32, _u64, _u32, C.POINTER(_vp)]),
    "scl_tans_model_destroy": (None, [_vp]),
    "scl_tans_model_info": (_int, [_vp, C.POINTER(RansInfo)]),
    "scl_tans_slot_bytes": (_u64, [_vp, _u64]),
    "scl_tans_model_tables": (_int, [_vp, _u32p, _u32p, _u32p, _u32p, _u32p]),
    "scl_tans_encode_batch": (_int, _ENC_BATCH),
    "scl_tans_decode_batch": (_int, _DEC_BATCH),
    "scl_tans_encode_host": (_int, _ENC_HOST),
    "scl_tans_decode_host": (_int, _DEC_HOST),
    "scl_range_model_create": (_int, [_u32p, _u32, _u32, _u32, C.POINTER(_vp)]),
    "scl_range_model_destroy": (None, [_vp]),
    "scl_range_slot_bytes": (_u64, [_vp, _u64]),
    "scl_range_fast_path": (_int, [_vp]),
    "scl_range_encode_batch": (_int, _ENC_BATCH),
    "scl_range_decode_batch": (_int, _DEC_BATCH),
    "scl_range_encode_host": (_int, _ENC_HOST),
    "scl_range_decode_host": (_int, _DEC_HOST),
    "scl_aec_model_create": (_int, [_int, _u32p, _u32, _u32, _u64, _u32, _u32, C.POINTER(_vp)]),
    "scl_aec_model_destroy": (None, [_vp]),
    "scl_aec_slot_bytes": (_u64, [_vp, _u64]),
    "scl_aec_scratch_bytes": (_u64, [_vp, _u64]),
    "scl_aec_fast_path": (_int, [_vp, _u64]),
    "scl_aec_encode_batch": (_int, _ENC_BATCH[:-1] + [_vp, _u64, _vp]),
    "scl_aec_decode_batch": (_int, _DEC_BATCH[:-1] + [_vp, _u64, _vp]),
    "scl_aec_encode_host": (_int, _ENC_HOST),
    "scl_aec_decode_host": (_int, _DEC_HOST),
    "scl_aec_state_bytes": (_u64, [_vp, _u64]),
    "scl_aec_state_counts": (_u64, [_vp]),
    "scl_aec_state_reset": (_int, [_vp, _vp, _u64, _u64, _vp]),
    "scl_aec_state_upload": (_int, [_vp, _vp, _u64, _u64, _u32p, _u32p, _vp]),
    "scl_aec_state_download": (_int, [_vp, _vp, _u64, _u64, _u32p, _u32p, _vp]),
    "scl_aec_encode_batch_resume": (_int, _ENC_BATCH[:-1] + [_vp, _u64, _u64, _vp]),
    "scl_aec_decode_batch_resume": (_int, _DEC_BATCH[:-1] + [_vp, _u64, _u64, _vp]),
    "scl_aec_encode_host_resume": (_int, _ENC_HOST + [_u32p, _u32p]),
    "scl_aec_decode_host_resume": (_int, _DEC_HOST + [_u32p, _u32p]),
    "scl_streams_compact_scratch_bytes": (_u64, [_u64]),
    "scl_streams_compact": (_int, [_vp, _vp, _vp, _u64, _int, _vp, _u64, _vp, _vp, _vp]),
    "scl_streams_compact_at": (_int, [_vp, _vp, _vp, _u64, _int, _vp, _u64, _vp, _vp, _vp, _vp]),
    # wave-striped slots (ABI 8): the tuned rANS kernels (and the tANS models they serve) on the interleaved layout
    "scl_rans_striped_ok": (_int, [_vp]),
    "scl_tans_striped_ok": (_int, [_vp]),
    "scl_range_striped_ok": (_int, [_vp]),
    "scl_range_encode_batch_striped": (_int, _ENC_BATCH),
    "scl_range_decode_batch_striped": (_int, _DEC_BATCH),
    "scl_rans_kernel_names_striped": (_int, [_vp, _u64, C.c_char_p, C.c_char_p, _u64]),
    "scl_tans_kernel_names_striped": (_int, [_vp, _u64, C.c_char_p, C.c_char_p, _u64]),
    "scl_rans_encode_batch_striped": (_int, _ENC_BATCH),
    "scl_rans_decode_batch_striped": (_int, _DEC_BATCH),
    "scl_tans_encode_batch_striped": (_int, _ENC_BATCH),
    "scl_tans_decode_batch_striped": (_int, _DEC_BATCH),
    "scl_streams_compact_striped": (_int, [_vp, _u64, _vp, _vp, _u64, _int, _vp, _u64, _vp, _vp, _vp, _vp]),
    "scl_stream_block_size_host": (_int, [_u8p, _u64, _u32, _u64p]),
    "scl_framed_index_host": (_int, [_vp, _u64, _u32, _u64, _vp, _vp, _vp, _u64p, _u64p]),
    "scl_histogram_u8": (_int, [_vp, _u64, _vp, _vp]),
    "scl_histogram_u16": (_int, [_vp, _u64, _u32, _vp, _vp, _vp]),
    "scl_rccl_inject_api": (_int, [_vp]),
    "scl_rccl_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "scl_rccl_unique_id": (_int, [_u8p]),
    "scl_rccl_comm_create": (_int, [_u8p, _int, _int, C.POINTER(_vp)]),
    "scl_rccl_comm_destroy": (None, [_vp]),
    "scl_rccl_allgather_u64": (_int, [_vp, _u64, _u64p, _vp]),
    "scl_rccl_allgather_async": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "scl_streams_gather_rccl": (_int, [_vp, _int, _vp, _u64, _vp, _u64p, _vp]),
    "scl_streams_gather_blocks_rccl": (_int, [_vp, _int, _vp, _u64, _vp, _u64, _vp, _vp, _u64p, _u64p, _vp]),
    "scl_streams_gatherv_rccl": (_int, [_vp, _int, _u32, C.POINTER(_vp), _u64p, C.POINTER(_vp), _u64p, _vp]),
}

# the *_u16 twins (alphabets up to 65536 symbols): same arguments, symbol arrays are uint16; host arrays travel as void*
_ENC_HOST16 = [_vp, _vp, _u64, _u8p, _u64, _u64p]
_DEC_HOST16 = [_vp, _u8p, _u64, _vp, _u64, _u64p, _u64p]
for _coder in ("rans", "tans", "range", "aec"):
    for _op, _host in (("encode", _ENC_HOST16), ("decode", _DEC_HOST16)):
        _SIGNATURES[f"scl_{_coder}_{_op}_batch_u16"] = _SIGNATURES[f"scl_{_coder}_{_op}_batch"]
        _SIGNATURES[f"scl_{_coder}_{_op}_host_u16"] = (_int, _host)
for _op, _host in (("encode", _ENC_HOST16), ("decode", _DEC_HOST16)):
    _SIGNATURES[f"scl_aec_{_op}_batch_resume_u16"] = _SIGNATURES[f"scl_aec_{_op}_batch_resume"]
    _SIGNATURES[f"scl_aec_{_op}_host_resume_u16"] = (_int, _host + [_u32p, _u32p])

def any_parameter_forced() -> bool:
    """are the tuned kernels kept out of the calling thread's batch calls right now?  (``scl_set_any_parameter_kernels``
    has no getter: set-and-restore, which is thread-local and therefore race-free)"""
    L = load()
    prev = L.scl_set_any_parameter_kernels(-1)
    L.scl_set_any_parameter_kernels(prev)
    if prev >= 0:
        return prev == 1
    return os.environ.get("SCL_ANY_PARAMETER_KERNELS", "")[:1] == "1"


_lib = None
_lock = threading.Lock()


def load(path: str = None) -> C.CDLL:
    """Load libscl_hip.so and declare every entry point of include/scl_hip.h.  Raises
    ``SclHipError`` when the library was not built -- there is no fallback."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise SclHipError(E_NODEVICE, "load", f"{p} not found: build it with __graft_entry__.build() "
                              "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
        if "torch" not in sys.modules:
            # One HIP runtime per process: torch bundles its own libamdhip64, libscl_hip.so resolves to the system's.  Whichever
            # is loaded first serves both -- but if the system's comes first, torch later initialises ITS copy beside it and
            # finds no devices ("No HIP GPUs are available": __graft_entry__.build() followed by smoke() in one process,
            # round 6).  The batch entry points are fed by torch tensors anyway: let torch's runtime be the first.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(p)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the binding disagree
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def last_error() -> str:
    msg = load().scl_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise SclHipError(rc, what, last_error())


def device_count() -> int:
    n = _int(0)
    rc = load().scl_device_count(C.byref(n))
    return n.value if rc == OK else 0


def require_device() -> None:
    if device_count() < 1:
        raise SclHipError(E_NODEVICE, "require_device", "no HIP device visible; the entropy-coding "
                          "path runs on MI355X only (no CPU fallback): " + last_error())


def u8_ptr(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def u32_ptr(a: np.ndarray):
    return a.ctypes.data_as(_u32p)
