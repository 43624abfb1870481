"""GPU parity of the HIP EGNN encoder (through the C ABI) against the reference goldens
(FoldClassNet run on CPU with the same synthetic weights) and the CPU oracle.

Floating point: the kernel restructures the first edge Linear and accumulates in a different
order than torch, so parity is relative: against the REFERENCE's goldens max|delta| <= EGNN_REL * max|e|
with EGNN_REL = 1e-6, the criterion SURVEY.md 8c states (it measured 2e-8..1e-6 between two fp32
evaluations of this network), and cosine within 1e-6 of 1 (north_star: cosine scores within 1e-5).
Against the C oracle -- itself an fp32 evaluation 1.4e-7 .. 2.0e-7 away from the reference -- the bar is
the sum of two such errors, ORACLE_REL = 2e-6."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EGNN_REL = 1e-6          # vs the reference's goldens
ORACLE_REL = 2e-6        # vs the C oracle: two fp32 evaluations, each within EGNN_REL of the truth
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def encoder(synthetic_weights):
    from merizo_search_amd import ops
    weights, pe = synthetic_weights
    return ops.EgnnEncoder(weights, pe, "cuda:0")


def _check(e, ref, rel=EGNN_REL):
    scale = np.abs(ref).max()
    err = np.abs(e - ref).max()
    assert err <= rel * scale, (err, scale, err / scale)
    cos = float(np.dot(e.astype(np.float64), ref.astype(np.float64)) / (np.linalg.norm(e.astype(np.float64)) * np.linalg.norm(ref.astype(np.float64))))
    assert abs(cos - 1.0) < 1e-6, cos


CASES = ["M0", "3w5h", "AF-Q96HM7-F1-model_v4", "AF-Q96PD2-F1-model_v4", "walk1", "walk2", "walk64", "walk257"]


@pytest.mark.parametrize("case", CASES)
def test_embedding_matches_reference_golden(case, encoder, golden_dir):
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    e = encoder.embed([g[f"coords_{case}"]]).cpu().numpy()[0]
    _check(e, g[f"emb_{case}"])


def test_ragged_batch_equals_one_by_one_and_goldens(encoder, golden_dir):
    """The reference embeds with batch = 1 (dbsearch.py:288-301); one ragged launch must give
    the same vectors bit for bit, in any batch composition."""
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    coords = [g[f"coords_{c}"] for c in CASES]
    batch = encoder.embed(coords).cpu().numpy()
    for idx, c in enumerate(CASES):
        _check(batch[idx], g[f"emb_{c}"])
    singles = np.stack([encoder.embed([x]).cpu().numpy()[0] for x in coords])
    assert np.array_equal(batch, singles)
    rev = encoder.embed(coords[::-1]).cpu().numpy()[::-1]
    assert np.array_equal(batch, rev)
    again = encoder.embed(coords).cpu().numpy()
    assert np.array_equal(batch, again)            # no atomics: run-to-run bit reproducible


def test_matches_oracle_on_ted_like_batch(encoder, synthetic_weights):
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    weights, pe = synthetic_weights
    lens = [1, 3, 31, 32, 33, 45, 64, 97, 100, 127, 128, 129, 160]
    coords = [syn.random_walk(n, seed=700 + n) for n in lens]
    e = encoder.embed(coords).cpu().numpy()
    ref = orc.egnn_embed(weights, pe, coords)
    for a, b in zip(e, ref):
        _check(a, b, ORACLE_REL)


def test_rejects_structures_longer_than_positional_table(encoder):
    from merizo_search_amd._lib import MerizoHipError
    from merizo_search_amd.foldclass import synthetic as syn
    with pytest.raises(MerizoHipError):
        encoder.embed([syn.random_walk(3001, seed=1)])


def test_thousand_domain_batch_properties(encoder):
    """BASELINE config C3 encoder side (1k TED-length domains) through properties: finite,
    permutation-equivariant over the batch, equal to one-by-one on a sample."""
    from merizo_search_amd.foldclass import synthetic as syn
    lens = syn.ted_lengths(1000, seed=5)
    coords = [syn.random_walk(int(n), seed=9000 + i) for i, n in enumerate(lens)]
    e = encoder.embed(coords).cpu().numpy()
    assert np.isfinite(e).all()
    perm = np.random.default_rng(0).permutation(1000)
    e2 = encoder.embed([coords[i] for i in perm]).cpu().numpy()
    assert np.array_equal(e[perm], e2)
    for i in (0, 17, 511, 999):
        assert np.array_equal(encoder.embed([coords[i]]).cpu().numpy()[0], e[i])


@pytest.mark.parametrize("case", ["M0", "walk97", "walk292"])
def test_embedding_with_a_large_distance_weight_saturated_silu(case, golden_dir):
    """d2_scale = 1.0: the squared-distance column of edge_mlp.0.weight at full scale drives the first SiLU to pre-activations
    of +-1e2..1e3: v_exp_f32 overflows to +inf on one side and underflows on the other, v_rcp_f32 must turn that
    into exactly 0 / 1.  Against the reference's goldens (oracle/gen_golden_d2.py) and the oracle."""
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import weights as W
    from oracle import oracle as orc
    g = np.load(os.path.join(golden_dir, "egnn_d2.npz"))
    weights, pe = W.pack_state_dict(W.synthetic_state_dict(0, d2_scale=float(g["d2_scale"])))
    enc = ops.EgnnEncoder(weights, pe, "cuda:0")
    coords = g[f"coords_{case}"]
    e = enc.embed([coords]).cpu().numpy()[0]
    assert np.isfinite(e).all()
    _check(e, g[f"emb_{case}"])
    _check(e, orc.egnn_embed(weights, pe, [coords])[0], ORACLE_REL)


@pytest.mark.parametrize("n", [1000, 2000])
@pytest.mark.parametrize("which", ["std", "d2"])
def test_long_chains_match_the_reference_goldens(which, n, golden_dir):
    """Chains of 1000 and 2000 residues (2000 = createdb's truncation length, makedb.py:68-69): 1M / 4M edges per layer, sums over
    up to 2000 neighbours per residue -- with the default fixture weights and with the distance column at full scale (d2_scale = 1).
    Reference goldens from oracle/gen_golden_d2.py long; the bar of SURVEY.md 8c, 1e-6 of max |e|, in the default (split-bf16) form."""
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn, weights as W
    g = np.load(os.path.join(golden_dir, "egnn_long.npz"))
    sd = W.synthetic_state_dict(0, d2_scale=1.0) if which == "d2" else W.synthetic_state_dict(0)
    weights, pe = W.pack_state_dict(sd)
    enc = ops.EgnnEncoder(weights, pe, "cuda:0")
    coords = g[f"coords_walk{n}"]
    assert np.array_equal(coords, syn.random_walk(n, seed={1000: 1197, 2000: 2197}[n]))
    e = enc.embed([coords]).cpu().numpy()[0]
    assert np.isfinite(e).all()
    _check(e, g[f"emb_{which}_walk{n}"])


def test_hundred_runs_of_a_ragged_batch_return_identical_bits(synthetic_weights):
    """The encoder keeps loads in flight across compiler-scheduled matrix instructions (buffer loads the compiler can see, since
    round 4; LDS-DMA pieces it cannot): 100 runs of one ragged batch -- 40 structures, 25..400 residues, two of them long enough
    for several edge tiles per residue row -- must return the same bits, and two different batch compositions the same embeddings.
    (The scan's lost-row bug of round 3 showed in one run of twenty and in no single-run parity test.)"""
    import torch
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    weights, pe = synthetic_weights
    enc = ops.EgnnEncoder(weights, pe, "cuda:0")
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.integers(25, 160, size=38)] + [333, 400]
    coords = [syn.random_walk(n, seed=500 + i) for i, n in enumerate(lens)]
    first = enc.embed(coords).clone()
    for run in range(100):
        again = enc.embed(coords)
        assert torch.equal(again.view(torch.int32), first.view(torch.int32)), "run %d differs" % run
    perm = list(rng.permutation(len(coords)))
    shuffled = enc.embed([coords[i] for i in perm])
    assert torch.equal(shuffled.view(torch.int32), first[perm].view(torch.int32))


def test_fp32_edge_gemm_form_still_matches_the_goldens_and_the_split_form(encoder, golden_dir, tmp_path):
    """MS_EGNN_SPLIT=0 selects the fp32 matrix instructions for the edge GEMM (the form of rounds 1-3; the split-bf16 form is the
    default since round 4).  A fresh process (the switch is read once) embeds the golden structures with it: same parity bar against
    the reference goldens, and within 2e-6 (relative to max |e|) of the split form of this process."""
    import subprocess
    import sys
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    out = tmp_path / "fp32_form.npy"
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from merizo_search_amd import ops\n"
        "from merizo_search_amd.foldclass import weights as W\n"
        "g = np.load(%r)\n"
        "w, pe = W.pack_state_dict(W.synthetic_state_dict(0))\n"
        "enc = ops.EgnnEncoder(w, pe, 'cuda:0')\n"
        "np.save(%r, enc.embed([g['coords_' + c] for c in %r]).cpu().numpy())\n"
    ) % (REPO, os.path.join(REPO, "tests"), os.path.join(golden_dir, "egnn.npz"), str(out), list(CASES))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MS_EGNN_SPLIT="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    fp32 = np.load(out)
    split = encoder.embed([g[f"coords_{c}"] for c in CASES]).cpu().numpy()
    for idx, c in enumerate(CASES):
        _check(fp32[idx], g[f"emb_{c}"])
        assert np.abs(fp32[idx] - split[idx]).max() <= 2e-6 * np.abs(split[idx]).max()

