"""Seeded synthetic inputs: CA traces, embedding databases, queries.

The reference snapshot ships neither trained weights nor database payloads
(.MISSING_LARGE_BLOBS), so tests, smoke and bench run on inputs generated here
(SURVEY.md 8d "Synthetic inputs").  Everything is a pure function of its seed.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

DIM = 128
CHUNK_ROWS = 1 << 20     # device-side DB generation granule (rows)

# Empirical length distribution of the shipped TED slice (examples/database/ted100_9606_small: min 25, median 97,
# mean 119.75, p95 267, max 683): the histogram of its domain lengths, one count per length 0..683
# (ted_length_hist.npy, 2.7 KB; the same array as tests/golden/ted_length_hist.npz).  E[N] = 119.75, E[N^2] = 20,506:
# 1,000 domains drawn from it cost the encoder 10.9 TFLOP.
_TED_LEN_QUANTILES = np.array([25, 43, 56, 68, 81, 97, 113, 135, 166, 222, 683], dtype=np.float64)      # (its deciles)
_TED_HIST_FILE = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "ted_length_hist.npy")
_ted_cdf = None


def random_walk(n: int, seed: int, step: float = 3.8) -> np.ndarray:
    """A CA-like trace: n points, consecutive points `step` Angstrom apart. float32 [n,3]."""
    rng = np.random.default_rng(seed)
    steps = rng.standard_normal((n, 3))
    steps = step * steps / np.linalg.norm(steps, axis=1, keepdims=True)
    return np.cumsum(steps, axis=0).astype(np.float32)


def ted_lengths(count: int, seed: int) -> np.ndarray:
    """`count` domain lengths drawn from the length histogram of the shipped TED slice (inverse CDF)."""
    global _ted_cdf
    if _ted_cdf is None:
        hist = np.load(_TED_HIST_FILE).astype(np.float64)
        _ted_cdf = np.cumsum(hist) / hist.sum()
    rng = np.random.default_rng(seed)
    return np.maximum(1, np.searchsorted(_ted_cdf, rng.random(count), side="right")).astype(np.int32)


def decile_lengths(count: int, seed: int) -> np.ndarray:
    """`count` lengths drawn piecewise-uniformly from the slice's DECILES (top decile uniform on 222..683: E[N^2] = 33,205, a
    heavier tail than the slice has).  Kept because the golden vectors of tests/golden were generated from inputs built with it
    (raw_database / raw_queries); workloads that claim the TED distribution use ted_lengths."""
    rng = np.random.default_rng(seed)
    u = rng.random(count) * 10.0
    b = np.minimum(u.astype(np.int64), 9)
    lo, hi = _TED_LEN_QUANTILES[b], _TED_LEN_QUANTILES[b + 1]
    return np.maximum(1, np.floor(lo + (u - b) * (hi - lo))).astype(np.int32)


_AA = np.array(list("ACDEFGHIKLMNPQRSTVWY"))


def synthetic_structures(count: int, seed: int, min_len: int = 20, max_len: int = 90
                         ) -> Tuple[List[str], List[np.ndarray], List[str]]:
    """(names, coords, seqs) of `count` random-walk structures with lengths in [min_len,max_len]."""
    rng = np.random.default_rng(seed)
    names, coords, seqs = [], [], []
    for i in range(count):
        n = int(rng.integers(min_len, max_len + 1))
        names.append(f"/db/syn{i:05d}.pdb")
        coords.append(random_walk(n, seed * 100003 + i))
        seqs.append("".join(_AA[rng.integers(0, 20, size=n)]))
    return names, coords, seqs


def raw_database(n: int, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """Un-normalised `.pt`-style database: (float32 [n,128] ~ N(0,1), lengths float32 [n])."""
    rng = np.random.default_rng(seed)
    db = rng.standard_normal((n, DIM)).astype(np.float32)
    lengths = decile_lengths(n, seed + 7).astype(np.float32)
    return db, lengths


def raw_queries(nq: int, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, DIM)).astype(np.float32)
    qlen = decile_lengths(nq, seed + 7).astype(np.float32)
    return q, qlen


def normalized_database(n: int, seed: int) -> np.ndarray:
    """Pre-normalised faiss-layout style matrix (rows unit L2 norm), float32 [n,128]."""
    db, _ = raw_database(n, seed)
    nrm = np.sqrt((db.astype(np.float64) ** 2).sum(1, keepdims=True))
    return (db / nrm).astype(np.float32)


def plant_neighbours(db: np.ndarray, q: np.ndarray, per_query: int, seed: int, sigma: float = 0.05,
                     normalize: bool = True) -> np.ndarray:
    """Overwrite `per_query` rows per query with q + sigma*noise (known, tie-free neighbours).

    Returns int64 [nq, per_query] of the planted row numbers (recall checks).
    """
    rng = np.random.default_rng(seed)
    nq = q.shape[0]
    rows = rng.choice(db.shape[0], size=nq * per_query, replace=False).reshape(nq, per_query)
    for i in range(nq):
        noise = rng.standard_normal((per_query, DIM)).astype(np.float32)
        v = q[i][None, :] / np.linalg.norm(q[i]) * np.sqrt(DIM) + sigma * np.sqrt(DIM) * noise * \
            rng.uniform(0.5, 1.5, size=(per_query, 1)).astype(np.float32)
        if normalize:
            v = v / np.linalg.norm(v, axis=1, keepdims=True)
        db[rows[i]] = v.astype(np.float32)
    return rows.astype(np.int64)


def device_database(n_rows: int, row_offset: int, seed: int, device, normalize: bool = True):
    """Rows [row_offset, row_offset+n_rows) of the infinite synthetic database, generated on `device`.

    Row r depends only on (seed, r // CHUNK_ROWS, r % CHUNK_ROWS): a shard of any size or
    offset sees the same rows as the unsharded database.  float32 [n_rows,128].
    """
    import torch

    out = torch.empty((n_rows, DIM), dtype=torch.float32, device=device)
    r = row_offset
    end = row_offset + n_rows
    while r < end:
        chunk = r // CHUNK_ROWS
        c0 = chunk * CHUNK_ROWS
        take_to = min(end, c0 + CHUNK_ROWS)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1000003 + chunk)
        block = torch.randn((CHUNK_ROWS, DIM), generator=g, dtype=torch.float32, device=device)
        part = block[r - c0:take_to - c0]
        if normalize:
            part = part / part.norm(dim=1, keepdim=True).clamp_min(1e-12)
        out[r - row_offset:take_to - row_offset] = part
        del block
        r = take_to
    return out
