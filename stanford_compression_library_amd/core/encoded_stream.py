"""On-disk framing of encoded blocks (row f1 of the scope table).

File = repeat[ 4-byte big-endian payload size in bytes | payload ], payload = 3-bit pad count, that many zero
bits, then the block's bits -- byte-identical to reference scl/core/encoded_stream.py (``Padder`` :17-58,
``HeaderHandler`` :81-112, ``EncodedBlockWriter`` :137-175, ``EncodedBlockReader`` :178-225), so files written
here are readable by the reference and vice versa.  The device produces the same bytes for a whole batch in one
pass (``scl_streams_compact`` with ``SCL_COMPACT_FRAMED``); these classes are the host-side, one-block-at-a-time
form of it.
"""
from __future__ import annotations

from ..utils.bitarray_utils import BitArray, bitarray_to_uint, uint_to_bitarray

__all__ = ["Padder", "HeaderHandler", "EncodedBlockWriter", "EncodedBlockReader"]


class Padder:
    NUM_PAD_BITS = 3

    @classmethod
    def add_byte_padding(cls, payload_bitarray: BitArray) -> BitArray:
        assert isinstance(payload_bitarray, BitArray)
        num_pad = (-(len(payload_bitarray) + cls.NUM_PAD_BITS)) % 8
        return uint_to_bitarray(num_pad, bit_width=cls.NUM_PAD_BITS) + BitArray("0" * num_pad) + payload_bitarray

    @classmethod
    def remove_byte_padding(cls, payload_pad_bitarray: BitArray) -> BitArray:
        assert isinstance(payload_pad_bitarray, BitArray)
        num_pad = bitarray_to_uint(payload_pad_bitarray[: cls.NUM_PAD_BITS])
        return payload_pad_bitarray[cls.NUM_PAD_BITS + num_pad:]


class HeaderHandler:
    NUM_HEADER_BYTES = 4
    NUM_HEADER_BITS = NUM_HEADER_BYTES * 8
    MAX_PAYLOAD_SIZE = 1 << NUM_HEADER_BITS

    @classmethod
    def add_header(cls, payload_bitarray: BitArray) -> BitArray:
        assert len(payload_bitarray) % 8 == 0
        arr_size = len(payload_bitarray) // 8
        assert arr_size < cls.MAX_PAYLOAD_SIZE
        return uint_to_bitarray(arr_size, bit_width=cls.NUM_HEADER_BITS) + payload_bitarray

    @classmethod
    def get_payload_size(cls, header_bytes: bytes) -> int:
        assert isinstance(header_bytes, bytes) and len(header_bytes) == cls.NUM_HEADER_BYTES
        return int.from_bytes(header_bytes, "big")


class EncodedBlockWriter:
    def __init__(self, file_path: str):
        self.file_path = file_path

    def __enter__(self):
        self.file_writer = open(self.file_path, "wb")
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        self.file_writer.close()

    def write_block(self, encoded_block: BitArray):
        assert isinstance(encoded_block, BitArray)
        framed = HeaderHandler.add_header(Padder.add_byte_padding(encoded_block))
        self.file_writer.write(framed.tobytes())

    def write_framed_bytes(self, framed: bytes):
        """append bytes that are already framed (the output of the device-side batch framing)"""
        self.file_writer.write(framed)


class EncodedBlockReader:
    def __init__(self, file_path: str):
        self.file_path = file_path

    def __enter__(self):
        self.file_reader = open(self.file_path, "rb")
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        self.file_reader.close()

    def get_block(self):
        header_bytes = self.file_reader.read(HeaderHandler.NUM_HEADER_BYTES)
        if len(header_bytes) == 0:
            return None
        assert len(header_bytes) == HeaderHandler.NUM_HEADER_BYTES
        payload_size = HeaderHandler.get_payload_size(header_bytes)
        payload_bytes = self.file_reader.read(payload_size)
        assert len(payload_bytes) == payload_size
        padded = BitArray()
        padded.frombytes(payload_bytes)
        return Padder.remove_byte_padding(padded)
