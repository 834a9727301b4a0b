"""CPU: utils/fileio.py -- positional reads on several threads must be indistinguishable from the file object's own
``readinto`` (same bytes, same position afterwards), and must decline what they are not made for."""
import gzip
import io
import os

import numpy as np
import pytest

from stanford_compression_library_amd.utils import fileio


@pytest.fixture()
def small_threshold(monkeypatch):
    monkeypatch.setattr(fileio, "MIN_BYTES", 1000)
    monkeypatch.setattr(fileio, "_ALIGN", 4096)


@pytest.mark.parametrize("size,start,n", [(100_000, 0, 100_000), (100_000, 777, 50_000), (100_000, 90_000, 50_000),
                                          (5_000_000, 123, 4_000_000), (4096, 0, 4096), (50_000, 50_000, 1000)])
def test_read_into_equals_readinto(size, start, n, tmp_path, small_threshold):
    data = np.random.default_rng(size + start).integers(0, 256, size, dtype=np.uint8)
    path = os.path.join(tmp_path, "f.bin")
    data.tofile(path)
    buf = np.zeros(n, dtype=np.uint8)
    with open(path, "rb") as f:
        f.read(3)  # the buffered reader has read ahead: positions must still be the logical ones
        f.seek(start)
        got = fileio.read_into(f, memoryview(buf), n)
        want = min(n, size - start)
        assert got == want and f.tell() == start + want
        assert np.array_equal(buf[:want], data[start:start + want])
        rest = f.read()
        assert rest == data[start + want:].tobytes()  # the file object carries on where the positional reads ended


def test_declines_what_it_is_not_made_for(tmp_path, small_threshold):
    buf = np.zeros(5000, dtype=np.uint8)
    assert fileio.read_into(io.BytesIO(b"x" * 5000), memoryview(buf), 5000) is None       # no file descriptor
    gz = os.path.join(tmp_path, "f.gz")
    with gzip.open(gz, "wb") as f:
        f.write(b"y" * 5000)
    with gzip.GzipFile(gz, "rb") as f:
        assert fileio.read_into(f, memoryview(buf), 5000) is None                          # mode is an int, data is not at fd offsets
    txt = os.path.join(tmp_path, "t.txt")
    open(txt, "w").write("z" * 5000)
    with open(txt, "r") as f:
        assert fileio.read_into(f, memoryview(buf), 5000) is None                          # text mode
    with open(txt, "rb") as f:
        assert fileio.read_into(f, memoryview(buf), 10) is None                            # below the threshold
    r, w = os.pipe()
    with os.fdopen(w, "wb") as fw, os.fdopen(r, "rb") as fr:
        assert fileio.read_into(fr, memoryview(buf), 5000) is None                         # not a regular file
