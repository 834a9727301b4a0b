"""BitArray surface of the hot path.

The reference aliases ``BitArray`` to the third-party ``bitarray.bitarray`` C
extension (reference scl/utils/bitarray_utils.py:25) and uses a small subset of
it on the rANS / tANS / range / arithmetic path (SURVEY.md section 8b lists the
operations).  That package does not exist on the MI355X image, so this module
carries its own implementation with the same observable behaviour:

* big-endian bit order (bit 0 of the array is the MSB of byte 0),
* ``tobytes`` zero-pads the tail of the last byte,
* value semantics for ``+`` / slicing, in-place ``+=`` / ``extend`` / ``frombytes``.

Storage is one ``numpy.uint8`` per bit (0/1): the host side only wraps what the
HIP kernels produce, so simplicity beats packing here; the device side works on
packed MSB-first bytes ("stream layout").
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "BitArray",
    "get_bit_width",
    "uint_to_bitarray",
    "bitarray_to_uint",
    "get_random_bitarray",
]


class BitArray:
    """Minimal big-endian bit vector (drop-in for the subset of
    ``bitarray.bitarray`` the hot path touches, reference
    scl/utils/bitarray_utils.py:25)."""

    __slots__ = ("_b",)

    def __init__(self, init=None):
        if init is None:
            self._b = np.zeros(0, dtype=np.uint8)
        elif isinstance(init, BitArray):
            self._b = init._b.copy()
        elif isinstance(init, str):
            raw = np.frombuffer(init.encode("ascii"), dtype=np.uint8)
            bits = raw - ord("0")
            if bits.size and int(bits.max(initial=0)) > 1:
                raise ValueError(f"expected only '0'/'1' characters, got {init!r}")
            self._b = bits.astype(np.uint8)
        elif isinstance(init, (int, np.integer)) and not isinstance(init, bool):
            # bitarray(n) -> n uninitialised bits; zero them (deterministic)
            self._b = np.zeros(int(init), dtype=np.uint8)
        else:
            arr = np.asarray(list(init) if not isinstance(init, np.ndarray) else init)
            if arr.size and (arr.min() < 0 or arr.max() > 1):
                raise ValueError("bit values must be 0 or 1")
            self._b = arr.astype(np.uint8).reshape(-1)

    # -- constructors used by the device shim -------------------------------------------
    @classmethod
    def from_packed(cls, packed, nbits: int, bit_offset: int = 0) -> "BitArray":
        """Build from MSB-first packed bytes: bits [bit_offset, bit_offset+nbits)."""
        buf = np.frombuffer(bytes(packed), dtype=np.uint8) if not isinstance(packed, np.ndarray) else packed
        out = cls.__new__(cls)
        first = bit_offset >> 3
        last = (bit_offset + nbits + 7) >> 3
        bits = np.unpackbits(np.ascontiguousarray(buf[first:last], dtype=np.uint8))
        lo = bit_offset - (first << 3)
        out._b = bits[lo : lo + nbits].copy()
        if out._b.size != nbits:
            raise ValueError("packed buffer shorter than requested bit range")
        return out

    @classmethod
    def _wrap(cls, bits: np.ndarray) -> "BitArray":
        out = cls.__new__(cls)
        out._b = bits
        return out

    # -- container protocol --------------------------------------------------------------
    def __len__(self):
        return int(self._b.size)

    def __iter__(self):
        return iter(self._b.tolist())

    def __getitem__(self, key):
        if isinstance(key, slice):
            return BitArray._wrap(self._b[key].copy())
        return int(self._b[key])

    def __setitem__(self, key, value):
        if isinstance(value, BitArray):
            value = value._b
        self._b[key] = value

    def __eq__(self, other):
        if not isinstance(other, BitArray):
            return NotImplemented
        return self._b.size == other._b.size and bool(np.array_equal(self._b, other._b))

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __add__(self, other):
        return BitArray._wrap(np.concatenate([self._b, _coerce(other)]))

    def __radd__(self, other):
        return BitArray._wrap(np.concatenate([_coerce(other), self._b]))

    def __iadd__(self, other):
        self._b = np.concatenate([self._b, _coerce(other)])
        return self

    def __repr__(self):
        return f"BitArray('{self.to01()}')"

    def __copy__(self):
        return BitArray(self)

    def __deepcopy__(self, memo):
        return BitArray(self)

    # -- bitarray API subset -------------------------------------------------------------
    def to01(self) -> str:
        return (self._b + ord("0")).astype(np.uint8).tobytes().decode("ascii")

    def tolist(self):
        return self._b.tolist()

    def copy(self):
        return BitArray(self)

    def append(self, bit):
        self._b = np.append(self._b, np.uint8(1 if bit else 0))

    def extend(self, other):
        self._b = np.concatenate([self._b, _coerce(other)])

    def frombytes(self, data: bytes):
        self._b = np.concatenate([self._b, np.unpackbits(np.frombuffer(bytes(data), dtype=np.uint8))])

    def tobytes(self) -> bytes:
        return np.packbits(self._b).tobytes()

    def count(self, value=1) -> int:
        ones = int(self._b.sum())
        return ones if value else int(self._b.size) - ones

    def any(self) -> bool:
        return bool(self._b.any())

    def packed(self) -> np.ndarray:
        """MSB-first packed bytes as a numpy array (zero-padded tail)."""
        return np.packbits(self._b)


def _coerce(x) -> np.ndarray:
    if isinstance(x, BitArray):
        return x._b
    return BitArray(x)._b


def get_bit_width(x) -> int:
    """Minimum number of bits that represent the unsigned int ``x`` (1 for 0).

    Reference scl/utils/bitarray_utils.py:8-20 computes ceil(log2(x+1)) in float64, which
    equals ``int.bit_length`` for every x + 1 <= 2**53; all parameter sets this package accepts
    keep H < 2**63 and are validated against that."""
    x = int(x)
    assert x >= 0
    return 1 if x == 0 else x.bit_length()


def uint_to_bitarray(x, bit_width=None) -> BitArray:
    """Unsigned int -> MSB-first bits (reference scl/utils/bitarray_utils.py:28-34).

    ``bit_width=None`` gives the minimal representation (one bit for 0); a value that does not
    fit in ``bit_width`` raises OverflowError like ``bitarray.util.int2ba``."""
    assert isinstance(x, (int, np.integer))
    x = int(x)
    if x < 0:
        raise OverflowError("unsigned integer expected")
    if bit_width is None:
        bit_width = max(1, x.bit_length())
    elif bit_width <= 0:
        raise ValueError("bit_width must be > 0")
    if x.bit_length() > bit_width:
        raise OverflowError(f"{x} does not fit in {bit_width} bits")
    nbytes = (bit_width + 7) // 8
    bits = np.unpackbits(np.frombuffer(x.to_bytes(nbytes, "big"), dtype=np.uint8))
    return BitArray._wrap(bits[nbytes * 8 - bit_width :].copy())


def bitarray_to_uint(bit_array: BitArray) -> int:
    """MSB-first bits -> unsigned int (reference scl/utils/bitarray_utils.py:37-38)."""
    if len(bit_array) == 0:
        raise ValueError("non-empty bitarray expected")
    return int.from_bytes(np.packbits(np.concatenate(
        [np.zeros((-len(bit_array)) % 8, dtype=np.uint8), bit_array._b])).tobytes(), "big")


def get_random_bitarray(size) -> BitArray:
    """``size`` random bits (reference scl/utils/bitarray_utils.py:41-42; unseeded there too)."""
    return BitArray._wrap(np.random.randint(0, 2, int(size)).astype(np.uint8))
