"""one-off bring-up: rANS host entry points vs goldens, straight through ctypes (no package code)"""
import ctypes as C, json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL("stanford_compression_library_amd/libscl_hip.so")
L.scl_last_error.restype = C.c_char_p
u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
L.scl_rans_model_create.argtypes = [u32p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
L.scl_rans_encode_host.argtypes = [C.c_void_p, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
L.scl_rans_decode_host.argtypes = [C.c_void_p, u8p, C.c_uint64, u8p, C.c_uint64, u64p, u64p]
z = np.load("tests/golden/golden_rans.npz")
cases = json.loads(str(z["manifest"]))
bad = 0
for m in cases:
    f = np.asarray(m["freq"], np.uint32)
    h = C.c_void_p()
    rc = L.scl_rans_model_create(f.ctypes.data_as(u32p), f.size, m["RF"], m["b"], m["size_bits"], C.byref(h))
    assert rc == 0, L.scl_last_error()
    sym = np.ascontiguousarray(z[f"c{m['id']}_sym"], np.uint8)
    out = np.zeros(sym.size * 8 + 64, np.uint8)
    nb = C.c_uint64()
    rc = L.scl_rans_encode_host(h, sym.ctypes.data_as(u8p), sym.size, out.ctypes.data_as(u8p), out.size, C.byref(nb))
    exp = z[f"c{m['id']}_out"]
    ok = rc == 0 and nb.value == m["nbits"] and np.array_equal(out[: exp.size], exp)
    dec = np.zeros(sym.size + 16, np.uint8); n = C.c_uint64(); used = C.c_uint64()
    g = z[f"c{m['id']}_garbage61"]
    bits = np.concatenate([np.unpackbits(exp)[: m["nbits"]], g]); packed = np.packbits(bits)
    rc2 = L.scl_rans_decode_host(h, packed.ctypes.data_as(u8p), bits.size, dec.ctypes.data_as(u8p), dec.size, C.byref(n), C.byref(used))
    ok2 = rc2 == 0 and n.value == sym.size and used.value == m["nbits"] and np.array_equal(dec[: sym.size], sym)
    if not (ok and ok2):
        bad += 1
        print("FAIL", m["id"], m["group"], rc, nb.value, m["nbits"], rc2, n.value, used.value, L.scl_last_error())
print("cases", len(cases), "bad", bad)
