"""Synthetic inputs of the measurement plan: frequency tables and seeded sources.

Shared by bench.py, smoke() and the tests so that "the headline input" means one thing everywhere.
Host generators use ``numpy.random.default_rng`` (identical streams under numpy 1.26 and 2.2); the
1 GiB headline input is sampled on the device from the same table through the slot -> symbol map
(u ~ U{0..M-1} -> symbol whose cumulative interval contains u), which is exactly i.i.d. with p = f/M.
"""
from __future__ import annotations

import numpy as np


def t256_table() -> np.ndarray:
    """S2 table: 256 symbols, M = 4096, every f >= 1 (Dirichlet(1) weights, seed 1); ~7.3 bit/symbol."""
    w = np.random.default_rng(1).dirichlet(np.ones(256))
    f = np.maximum(1, np.floor(4096 * w)).astype(np.int64)
    f[int(np.argmax(f))] += 4096 - int(f.sum())
    assert f.sum() == 4096 and f.min() >= 1
    return f


def uniform256_table() -> np.ndarray:
    """256 symbols with f = 16 (M = 4096): 8 bit/symbol, output ~ input."""
    return np.full(256, 16, dtype=np.int64)


def slot_to_symbol(freq: np.ndarray) -> np.ndarray:
    return np.repeat(np.arange(freq.size, dtype=np.uint8), freq)


def iid_chunks_host(freq: np.ndarray, n_chunks: int, chunk_len: int, seed: int) -> np.ndarray:
    p = freq / freq.sum()
    return np.random.default_rng(seed).choice(freq.size, size=(n_chunks, chunk_len), p=p).astype(np.uint8)


def iid_chunks_device(freq: np.ndarray, n_chunks: int, chunk_len: int, seed: int, device):
    """uint8 CUDA tensor [n_chunks, chunk_len], i.i.d. with p = f/M, generated on the device piecewise."""
    import torch

    M = int(freq.sum())
    lut = torch.from_numpy(slot_to_symbol(freq)).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    out = torch.empty((n_chunks, chunk_len), dtype=torch.uint8, device=device)
    flat = out.view(-1)
    piece = 1 << 26
    for start in range(0, flat.numel(), piece):
        n = min(piece, flat.numel() - start)
        u = torch.randint(0, M, (n,), dtype=torch.int32, device=device, generator=gen)
        flat[start:start + n] = lut[u.long()]
    return out


def markov1_host(K: int, n: int, seed: int = 4) -> np.ndarray:
    """S4 source: order-1 Markov chain, Dirichlet(0.3) rows, previous symbol starts at index 0."""
    rng = np.random.default_rng(seed)
    P = rng.dirichlet(0.3 * np.ones(K), size=K)
    cdf = np.cumsum(P, axis=1)
    u = rng.random(n)
    x = np.zeros(n, dtype=np.uint8)
    prev = 0
    for t in range(n):
        prev = min(int(np.searchsorted(cdf[prev], u[t], side="right")), K - 1)
        x[t] = prev
    return x


def markov1_matrix(K: int, seed: int = 4) -> np.ndarray:
    """S4 transition matrix: K rows ~ Dirichlet(0.3) (``default_rng(4)``)."""
    return np.random.default_rng(seed).dirichlet(0.3 * np.ones(K), size=K)


def markov1_chunks_device(K: int, n_chunks: int, chunk_len: int, seed: int, device, matrix_seed: int = 4):
    """uint8 CUDA tensor [n_chunks, chunk_len]: every row is its own order-1 Markov chain over K symbols (one shared
    transition matrix, ``x_-1 = 0``), generated on the device so that a 1 GiB batch is 1 GiB of DISTINCT
    data.  The cumulative rows are quantised to 24 bits: a symbol is the number of thresholds its uniform draw passes."""
    import torch

    P = markov1_matrix(K, matrix_seed)
    thresh = np.minimum(np.floor(np.cumsum(P, axis=1) * (1 << 24)), (1 << 24)).astype(np.int32)
    thresh[:, -1] = 1 << 24  # the last symbol takes whatever mass rounding left
    cdf = torch.from_numpy(thresh).to(device)  # [K, K]
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    out = torch.empty((n_chunks, chunk_len), dtype=torch.uint8, device=device)
    rows = max(1, min(n_chunks, (1 << 26) // max(K, 1)))  # bound the [rows, K] temporaries to 256 MiB
    for a in range(0, n_chunks, rows):
        b = min(n_chunks, a + rows)
        prev = torch.zeros(b - a, dtype=torch.int64, device=device)
        for t in range(chunk_len):
            u = torch.randint(0, 1 << 24, (b - a, 1), dtype=torch.int32, device=device, generator=gen)
            prev = (u >= cdf[prev]).sum(dim=1).clamp_(max=K - 1)
            out[a:b, t] = prev.to(torch.uint8)
    return out
