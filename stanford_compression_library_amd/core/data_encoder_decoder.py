"""``DataEncoder`` / ``DataDecoder``: the drop-in boundary of the hot path.

Same method names and contracts as reference scl/core/data_encoder_decoder.py (``DataEncoder``
:15-86, ``DataDecoder`` :89-160).  Subclasses implement ``encode_block`` / ``decode_block``; the
block loop (``encode`` / ``decode``) is kept so stream drivers written against the reference keep
working with any writer/reader/stream object that offers ``write_block`` / ``get_block``.
"""
from __future__ import annotations

import abc

from ..utils.bitarray_utils import BitArray
from .data_block import DataBlock

__all__ = ["DataEncoder", "DataDecoder"]


class DataEncoder(abc.ABC):
    def reset(self):
        """Reset coder state, if any (no-op hook, reference :23-27)."""

    def encode_block(self, data_block: DataBlock) -> BitArray:
        raise NotImplementedError

    def encode(self, data_stream, block_size: int, encode_writer):
        """Chop ``data_stream`` into blocks, encode each, hand the bits to ``encode_writer``
        (reference :43-69)."""
        while True:
            data_block = data_stream.get_block(block_size)
            if data_block is None:
                break
            output = self.encode_block(data_block)
            assert isinstance(output, BitArray)
            encode_writer.write_block(output)

    def encode_file(self, input_file_path: str, encoded_file_path: str, block_size: int = 10000):
        """Text file -> framed block file (reference :71-86)."""
        from .data_stream import TextFileDataStream
        from .encoded_stream import EncodedBlockWriter

        with TextFileDataStream(input_file_path, "r") as fds:
            with EncodedBlockWriter(encoded_file_path) as writer:
                self.encode(fds, block_size=block_size, encode_writer=writer)


class DataDecoder(abc.ABC):
    # Largest block size a size header may announce to ``decode_block`` of the HIP-backed decoders before they allocate
    # the output (None: the backend's default, 2^32 - 1 = whatever the reference's 32-bit header can announce; set it
    # when reading untrusted streams).  A symbol can cost 0 bits, so the stream's length is no
    # bound; a damaged header above this raises AssertionError -- the exception the encoder's own size check raises.
    # No reference counterpart (the reference decodes symbol by symbol into a Python list).
    max_block_size = None

    def reset(self):
        """Reset coder state, if any (no-op hook, reference :96-100)."""

    def decode_block(self, bitarray: BitArray):
        """-> (DataBlock, num_bits_consumed); must tolerate trailing bits after the block."""
        raise NotImplementedError

    def decode(self, encode_reader, output_stream):
        """Decode every framed block of ``encode_reader`` into ``output_stream`` (reference :118-144)."""
        while True:
            encoded_block = encode_reader.get_block()
            if encoded_block is None:
                break
            output_block, num_bits_consumed = self.decode_block(encoded_block)
            assert num_bits_consumed == len(encoded_block)
            output_stream.write_block(output_block)

    def decode_file(self, encoded_file_path: str, output_file_path: str):
        """Framed block file -> text file (reference :146-160)."""
        from .data_stream import TextFileDataStream
        from .encoded_stream import EncodedBlockReader

        with EncodedBlockReader(encoded_file_path) as reader:
            with TextFileDataStream(output_file_path, "w") as fds:
                self.decode(reader, fds)
