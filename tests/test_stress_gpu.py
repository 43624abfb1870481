"""Randomised GPU-vs-oracle sweeps (seeded): search in both modes incl. duplicates, multi-pass k,
row offsets; encoder on ragged batches."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_search_random_sweep():
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "stress_search.py"), "7", "40"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "40 cases, 0 mismatches" in out.stdout


def test_encoder_random_ragged_batches(synthetic_weights):
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    weights, pe = synthetic_weights
    enc = ops.EgnnEncoder(weights, pe, "cuda:0")
    rng = np.random.default_rng(3)
    for case in range(6):
        nb = int(rng.integers(1, 12))
        lens = [int(x) for x in rng.integers(1, 180, size=nb)]
        coords = [syn.random_walk(n, seed=int(rng.integers(0, 1 << 30)), step=float(rng.uniform(2.0, 6.0))) for n in lens]
        e = enc.embed(coords).cpu().numpy()
        ref = orc.egnn_embed(weights, pe, coords)
        for a, b in zip(e, ref):
            scale = np.abs(b).max()
            assert np.abs(a - b).max() <= 1e-5 * scale, (case, lens, np.abs(a - b).max() / scale)


def test_encoder_long_structures(synthetic_weights):
    """N = 1000 (oracle on the host cores) and the positional-table limit N = 3000 (finite, repeatable)."""
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    weights, pe = synthetic_weights
    enc = ops.EgnnEncoder(weights, pe, "cuda:0")
    c1000 = syn.random_walk(1000, seed=77)
    e = enc.embed([c1000]).cpu().numpy()[0]
    if (os.cpu_count() or 1) >= 32:               # ~2.1 TFLOP of literal arithmetic: only on a many-core host
        ref = orc.egnn_embed(weights, pe, [c1000])[0]
        assert np.abs(e - ref).max() <= 1e-5 * np.abs(ref).max()
    c3000 = syn.random_walk(3000, seed=78)
    a = enc.embed([c3000, c1000]).cpu().numpy()
    b = enc.embed([c3000, c1000]).cpu().numpy()
    assert np.isfinite(a).all() and np.array_equal(a, b) and np.array_equal(a[1], e)
