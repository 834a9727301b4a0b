"""Positional file READS on several threads for the bulk shape of the streaming driver (row f2; no reference counterpart).

One thread's ``readinto`` moves 8-12 GB/s out of the page cache (5-8 ms per 64 MiB slab, more than the device needs for the
slab); four ``os.preadv`` calls at explicit offsets (they release the GIL) move 28-38 GB/s on the MI355X boxes
(tools/ubench_fileio.py).  For REGULAR, seekable, binary files the slab is read that way and the file object's own position
is moved behind it; anything else (pipes, sockets, ``gzip.GzipFile``, text streams, small transfers) is left to the caller's
ordinary path: ``read_into`` says so instead of guessing.

(Writes were measured too and are NOT split: a new file in /dev/shm takes its pages one thread at a time -- 5.9-7.3 GB/s with
one ``write``, 3.5-7.1 with two to eight ``pwritev`` -- so the writer side of ``encode()`` / ``decode()``, which is what bounds
them now, keeps the file object's own ``write``.)
"""
from __future__ import annotations

import os
import stat
import threading
from concurrent.futures import ThreadPoolExecutor

THREADS = 4
MIN_BYTES = 8 << 20      # below this one call is as fast
_ALIGN = 1 << 20
_pool = None
_pool_lock = threading.Lock()


def _get_pool() -> ThreadPoolExecutor:
    global _pool
    with _pool_lock:
        if _pool is None:
            _pool = ThreadPoolExecutor(THREADS, thread_name_prefix="scl-fileio")
        return _pool


def _regular(fobj):
    """(fd, size) of a regular, seekable, binary file object, else None"""
    try:
        mode = getattr(fobj, "mode", "")
        if not isinstance(mode, str) or "b" not in mode or not fobj.seekable():
            return None
        fd = fobj.fileno()
        st = os.fstat(fd)
        return (fd, st.st_size) if stat.S_ISREG(st.st_mode) else None
    except (OSError, AttributeError, ValueError):
        return None


def _parts(n: int):
    step = max(_ALIGN, (n // THREADS + _ALIGN - 1) // _ALIGN * _ALIGN)
    return [(a, min(a + step, n)) for a in range(0, n, step)]


def read_into(fobj, view: memoryview, n: int):
    """Fill ``view[:n]`` from the file's current position; -> bytes read (short only at the end of the file), the position
    moved behind them -- or ``None`` when this file object is not one for positional reads (the caller reads it the
    ordinary way)."""
    if n < MIN_BYTES:
        return None
    info = _regular(fobj)
    if info is None:
        return None
    fd, size = info
    pos = fobj.tell()
    n = min(n, size - pos)
    if n <= 0:
        return 0

    def part(ab):
        a, b = ab
        done = a
        while done < b:
            k = os.preadv(fd, [view[done:b]], pos + done)
            if k <= 0:
                break
            done += k
        return done - a

    total = 0
    for (a, b), got in zip(_parts(n), _get_pool().map(part, _parts(n))):
        total += got
        if got < b - a:  # the file shrank under us: what lies behind the gap does not count
            break
    fobj.seek(pos + total)
    return total
