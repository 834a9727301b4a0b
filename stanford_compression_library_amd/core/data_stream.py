"""Block-wise data streams feeding the coders (row f2 of the scope table).

Same classes and methods as reference scl/core/data_stream.py (``DataStream`` :10-101, ``ListDataStream``
:104-160, ``FileDataStream`` :163-213, ``TextFileDataStream`` :216-235, ``Uint8FileDataStream`` :238-258).  The
reference pulls one symbol per ``file.read(1)``; here ``get_block`` / ``write_block`` move whole blocks in one
read / write, which is what lets a stream feed a batched encode.

Bulk extension (no reference counterpart; used by compressors/_stream_batch.py when a stream offers it): the file streams
also move symbols as numpy arrays of integer CODES -- ``read_codes(n)`` -> up to n symbols (``None`` at the end),
``write_codes(array)``, ``symbol_of(code)`` (what ``get_symbol`` returns for it) / ``code_of(symbol)`` (what ``write_symbol``
writes for it) -- a byte is its own code, a
character its code point.  A stream of the batched coders then never builds a Python list of symbols.
"""
from __future__ import annotations

import abc

import numpy as np

from .data_block import DataBlock

__all__ = ["DataStream", "ListDataStream", "FileDataStream", "TextFileDataStream", "Uint8FileDataStream"]


class DataStream(abc.ABC):
    """get_block(block_size) -> DataBlock | None at the end;  write_block(DataBlock)."""

    @abc.abstractmethod
    def seek(self, pos: int):
        ...

    @abc.abstractmethod
    def get_symbol(self):
        """next symbol, or None when the stream is exhausted"""

    @abc.abstractmethod
    def write_symbol(self, s):
        ...

    def get_block(self, block_size: int):
        data = []
        for _ in range(block_size):
            s = self.get_symbol()
            if s is None:
                break
            data.append(s)
        return DataBlock(data) if data else None

    def write_block(self, data_block: DataBlock):
        for s in data_block.data_list:
            self.write_symbol(s)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        pass


class ListDataStream(DataStream):
    """A list as a stream: reads advance a cursor, writes append."""

    def __init__(self, input_list):
        assert isinstance(input_list, list)
        self.input_list = input_list
        self.current_ind = 0

    def seek(self, pos: int):
        assert pos <= len(self.input_list)
        self.current_ind = pos

    def get_symbol(self):
        if self.current_ind >= len(self.input_list):
            return None
        s = self.input_list[self.current_ind]
        self.current_ind += 1
        return s

    def get_block(self, block_size: int):
        chunk = self.input_list[self.current_ind:self.current_ind + block_size]
        self.current_ind += len(chunk)
        return DataBlock(chunk) if chunk else None

    def write_symbol(self, s):
        self.input_list.append(s)

    def write_block(self, data_block: DataBlock):
        self.input_list.extend(data_block.data_list)


class FileDataStream(DataStream):
    """Opens the file on ``__enter__`` with the given permissions, closes it on ``__exit__``."""

    def __init__(self, file_path: str, permissions="r"):
        self.file_path = file_path
        self.permissions = permissions

    def __enter__(self):
        self.file_obj = open(self.file_path, self.permissions)
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        self.file_obj.close()

    def seek(self, pos: int):
        self.file_obj.seek(pos)


class TextFileDataStream(FileDataStream):
    """characters of a text file"""

    def get_symbol(self):
        s = self.file_obj.read(1)
        return s if s else None

    def get_block(self, block_size: int):
        text = self.file_obj.read(block_size)
        return DataBlock(list(text)) if text else None

    def write_symbol(self, s):
        self.file_obj.write(s)

    def write_block(self, data_block: DataBlock):
        self.file_obj.write("".join(data_block.data_list))

    # -- bulk extension: characters as code points ------------------------------------------------------
    @staticmethod
    def code_of(symbol):
        """the code point ``write_symbol`` writes for an alphabet symbol, or None when it is not one character"""
        return ord(symbol) if isinstance(symbol, str) and len(symbol) == 1 else None

    @staticmethod
    def symbol_of(code: int):
        return chr(code)

    def read_codes(self, n: int):
        """the next up to n characters as code points: uint8 when they all fit a byte, else uint32"""
        text = self.file_obj.read(n)
        if not text:
            return None
        try:
            return np.frombuffer(text.encode("latin-1"), dtype=np.uint8)
        except UnicodeEncodeError:
            return np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)

    def write_codes(self, codes: np.ndarray):
        if codes.dtype == np.uint8:
            self.file_obj.write(codes.tobytes().decode("latin-1"))
        else:
            self.file_obj.write(codes.astype("<u4").tobytes().decode("utf-32-le", "surrogatepass"))


class Uint8FileDataStream(FileDataStream):
    """bytes of a binary file as ints 0..255 (open with "rb" / "wb")"""

    def get_symbol(self):
        s = self.file_obj.read(1)
        return s[0] if s else None

    def get_block(self, block_size: int):
        raw = self.file_obj.read(block_size)
        return DataBlock(list(raw)) if raw else None

    def write_symbol(self, s):
        assert 0 <= s <= 255
        self.file_obj.write(bytes([s]))

    def write_block(self, data_block: DataBlock):
        self.file_obj.write(bytes(data_block.data_list))

    # -- bulk extension: a byte is its own code -----------------------------------------------------------
    @staticmethod
    def code_of(symbol):
        """the byte ``write_symbol`` writes for an alphabet symbol, or None when it cannot write it"""
        if isinstance(symbol, (int, np.integer)) and 0 <= int(symbol) <= 255:
            return int(symbol)
        return None

    @staticmethod
    def symbol_of(code: int):
        return int(code)

    def read_codes(self, n: int, out: np.ndarray = None):
        """the next up to n bytes as a uint8 array (read straight into ``out`` -- e.g. a pinned staging buffer -- when
        given); None at the end of the file"""
        if out is not None:
            got = self.file_obj.readinto(memoryview(out)[:n])
            return out[:got] if got else None
        raw = self.file_obj.read(n)
        return np.frombuffer(raw, dtype=np.uint8) if raw else None

    def write_codes(self, codes: np.ndarray):
        assert codes.dtype == np.uint8
        self.file_obj.write(memoryview(np.ascontiguousarray(codes)))
