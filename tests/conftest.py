import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the `gpu` tests instead of erroring 300 times.  On a box
    WITH a GPU nothing is skipped: a missing libscl_hip.so then fails loudly (no CPU fallback exists)."""
    try:
        import torch

        have_gpu = torch.cuda.is_available() and torch.cuda.device_count() > 0
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """The native pieces are built in tree (`__graft_entry__.build()`); `make` is a no-op when they are up to date
    and rebuilds them when a snapshot arrived without them or with older ones.  A failing toolchain is only fatal if
    there is nothing to fall back on -- the tests that need the libraries say so themselves."""
    import subprocess

    for sub, args in (("stanford_compression_library_amd/csrc", ["-j8", "ARCH=gfx950"]), ("oracle", [])):
        try:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, sub)] + args, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        except Exception:
            pass


class GoldenCase:
    """One reference-generated vector (see oracle/gen_goldens.py)."""

    def __init__(self, meta, arrays):
        self.meta = meta
        self._arrays = arrays

    def __getattr__(self, k):
        try:
            return self.meta[k]
        except KeyError:
            raise AttributeError(k)

    def arr(self, name):
        return self._arrays[f"c{self.meta['id']}_{name}"]

    @property
    def bits_with_garbage(self):
        """[(packed bytes, total bits, expected consumed)] for each stored garbage length."""
        out = []
        base = np.unpackbits(self.arr("out"))[: self.nbits]
        for g, used in zip(self.meta.get("garbage_lens", []), self.meta.get("consumed", [])):
            bits = np.concatenate([base, self.arr(f"garbage{g}").astype(np.uint8)])
            out.append((np.packbits(bits), int(bits.size), used))
        return out

    def __repr__(self):
        m = self.meta
        return f"{m['kind']}[{m['id']}:{m.get('group')},n={m.get('n')}]"


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"golden_{name}.npz"))
    manifest = json.loads(str(z["manifest"]))
    arrays = {k: z[k] for k in z.files if k != "manifest"}
    return [GoldenCase(m, arrays) for m in manifest]


def golden_ids(cases):
    return [repr(c) for c in cases]


def stream_blocks(case):
    """[(symbol indices, packed bits, nbits)] of one golden_stream case, block by block"""
    sym, nbits, packed = case.arr("sym"), case.arr("block_nbits"), case.arr("block_out")
    out, pos = [], 0
    for i, nb in enumerate(nbits.tolist()):
        nbytes = (nb + 7) // 8
        out.append((sym[i * case.block_size:(i + 1) * case.block_size], packed[pos:pos + nbytes], nb))
        pos += nbytes
    return out


def frame_blocks(blocks):
    """EncodedBlockWriter framing of [(packed bits, nbits)] restated on bytes (encoded_stream.py:23-46,94-103):
    [u32 BE payload bytes][3-bit pad count][pad zeros][bits]"""
    out = []
    for packed, nb in blocks:
        pad = (-(nb + 3)) % 8
        bits = np.concatenate([np.unpackbits(np.array([pad << 5], np.uint8))[:3], np.zeros(pad, np.uint8),
                               np.unpackbits(np.asarray(packed, np.uint8))[:nb]])
        payload = np.packbits(bits)
        out.append(np.frombuffer(len(payload).to_bytes(4, "big"), np.uint8))
        out.append(payload)
    return np.concatenate(out) if out else np.zeros(0, np.uint8)
