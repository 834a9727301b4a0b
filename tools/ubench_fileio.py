"""one thread's readinto / write against several positional reads / writes, 64 MiB at a time in /dev/shm (pageable and pinned buffers)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from stanford_compression_library_amd.utils import fileio
n = 64 << 20
src = "/dev/shm/io_ub_src"; dst = "/dev/shm/io_ub_dst"
np.random.default_rng(0).integers(0, 256, 16 * n, dtype=np.uint8).tofile(src)
for pinned in (False, True):
    t = torch.empty(n, dtype=torch.uint8, pin_memory=pinned); h = t.numpy(); view = memoryview(h)
    for threads in (1, 2, 4, 8):
        fileio.THREADS = threads; fileio._pool = None
        with open(src, "rb") as f:
            t0 = time.perf_counter()
            for _ in range(16):
                if threads == 1:
                    got = 0
                    while got < n:
                        got += f.readinto(view[got:n])
                else:
                    assert fileio.read_into(f, view, n) == n
            tr = time.perf_counter() - t0
        if os.path.exists(dst): os.remove(dst)
        with open(dst, "wb") as f:
            t0 = time.perf_counter()
            for _ in range(16):
                if threads == 1:
                    f.write(view)
                else:  # (what utils/fileio.py does NOT do: measured here to say why)
                    f.flush(); pos = f.tell(); fd = f.fileno()
                    def part(ab):
                        a, b = ab
                        while a < b:
                            a += os.pwritev(fd, [view[a:b]], pos + a)
                    list(fileio._get_pool().map(part, fileio._parts(n))); f.seek(pos + n)
            tw = time.perf_counter() - t0
        print(f"pinned={pinned} threads={threads}: read {16 * n / tr / 1e9:.1f} GB/s  write(new file) {16 * n / tw / 1e9:.1f} GB/s")
os.remove(src); os.remove(dst)
