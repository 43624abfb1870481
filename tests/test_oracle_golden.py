"""Pin the CPU oracle (oracle/oracle.c) to the golden vectors produced by the imported
reference (oracle/gen_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from conftest import assert_topk_equivalent
from merizo_search_amd.foldclass import synthetic as syn
from oracle import oracle as orc


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_generators_have_not_drifted(golden_meta, synthetic_weights):
    weights, pe = synthetic_weights
    assert _sha(weights) == golden_meta["g1"]["weights_sha"]
    assert _sha(pe) == golden_meta["g1"]["pe_sha"]
    db, lengths = syn.raw_database(5000, seed=11)
    q, qlen = syn.raw_queries(8, seed=12)
    assert _sha(db, lengths) == golden_meta["g2"]["db_sha"]
    assert _sha(q, qlen) == golden_meta["g2"]["q_sha"]
    assert weights.size == 2 * orc.layer_floats() == 792330


def test_positional_table_is_reference_data(golden_dir, synthetic_weights):
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    _, pe = synthetic_weights
    assert pe.shape == (3000, 128)
    assert np.array_equal(pe[:4], g["pe_head"]) and np.array_equal(pe[-4:], g["pe_tail"])


# EGNN parity criterion: max_abs(delta) <= 1e-6 * max_abs(e) is what two fp32
# evaluations of the same network reach; we allow 5e-6 relative and require cosine ~ 1.
def _check_emb(e, ref, rel=5e-6):
    scale = np.abs(ref).max()
    assert np.abs(e - ref).max() <= rel * scale, (np.abs(e - ref).max(), scale)
    cos = float(np.dot(e, ref) / (np.linalg.norm(e) * np.linalg.norm(ref)))
    assert abs(cos - 1.0) < 1e-6


@pytest.mark.parametrize("case", ["M0", "walk1", "walk2", "walk64"])
def test_egnn_oracle_matches_reference(case, golden_dir, synthetic_weights):
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    weights, pe = synthetic_weights
    e = orc.egnn_embed(weights, pe, [g[f"coords_{case}"]])[0]
    _check_emb(e, g[f"emb_{case}"])


def test_egnn_oracle_per_layer_and_ragged_batch(golden_dir, synthetic_weights):
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    weights, pe = synthetic_weights
    coords = [g["coords_walk2"], g["coords_M0"], g["coords_walk1"]]
    out, layers = orc.egnn_embed(weights, pe, coords, return_layers=True)
    _check_emb(out[0], g["emb_walk2"]); _check_emb(out[1], g["emb_M0"]); _check_emb(out[2], g["emb_walk1"])
    for l, key in enumerate(("layer1_M0", "layer2_M0")):
        got = layers[l, 2:33]
        assert np.abs(got - g[key]).max() <= 5e-6 * np.abs(g[key]).max()


@pytest.mark.slow
@pytest.mark.parametrize("case", ["walk257", "3w5h"])
def test_egnn_oracle_matches_reference_large(case, golden_dir, synthetic_weights):
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    weights, pe = synthetic_weights
    e = orc.egnn_embed(weights, pe, [g[f"coords_{case}"]])[0]
    _check_emb(e, g[f"emb_{case}"])


def test_normalize_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "normalize.npz"))
    y = orc.l2_normalize_rows(g["x"], 1e-12)
    np.testing.assert_allclose(y, g["y12"], rtol=5e-7, atol=0)   # norm summation order differs by <= 2 ulp
    assert np.all(y[3] == 0.0)                      # zero row stays zero (0 / eps)
    # cosine eps (1e-8) observed through unit probes: cos(x, e_c) = x_c / max(|x|, 1e-8)
    yc = orc.l2_normalize_rows(g["x"], 1e-8)
    np.testing.assert_allclose(yc[:, :4], g["cos_probe"], rtol=5e-7, atol=1e-30)


@pytest.mark.parametrize("mincov", [0.0, 0.7])
@pytest.mark.parametrize("k", [1, 10, 100])
def test_cosine_topk_matches_reference(mincov, k, golden_dir):
    g = np.load(os.path.join(golden_dir, "search.npz"))
    db, lengths = syn.raw_database(5000, seed=11)
    q, qlen = syn.raw_queries(8, seed=12)
    s, i = orc.cosine_topk(db, q, k, lengths, qlen, mincov)
    assert_topk_equivalent(s, i, g[f"s_cov{mincov}_k{k}"], g[f"i_cov{mincov}_k{k}"], tol=3e-7)


def test_cosine_topk_all_masked_and_k_full(golden_dir):
    g = np.load(os.path.join(golden_dir, "search.npz"))
    db, lengths = syn.raw_database(5000, seed=11)
    q, _ = syn.raw_queries(8, seed=12)
    s, i = orc.cosine_topk(db, q[:1], 10, lengths, np.array([10.0], np.float32), 0.7)
    assert np.all(s == 0.0) and np.all(g["s_allmasked"] == 0.0)      # masked rows score +-0, not -inf
    assert i[0].tolist() == list(range(10))                            # our tie policy: lowest index first
    db2, len2 = syn.raw_database(50, seed=13)
    s, i = orc.cosine_topk(db2, q[1:2], 50, len2, np.array([200.0], np.float32), 0.7)
    assert_topk_equivalent(s[0], i[0], g["s_kfull"], g["i_kfull"], tol=3e-7)
    with pytest.raises(RuntimeError):
        orc.cosine_topk(db2, q[1:2], 51, len2, np.array([200.0], np.float32), 0.7)   # torch.topk raises


def test_faiss_path_restatement_agrees_with_pinned_torch_path(golden_dir):
    """knn_exact_faiss (unpinned: faiss absent) vs the pinned `.pt` path on a pre-normalised
    DB with mincov 0: same indices, scores within 3e-7; also blockwise ==
    one-shot and both dot orders agree to rounding."""
    g = np.load(os.path.join(golden_dir, "search.npz"))
    db, _ = syn.raw_database(5000, seed=11)
    q, _ = syn.raw_queries(8, seed=12)
    dbn = orc.l2_normalize_rows(db, 1e-12)
    qn = orc.l2_normalize_rows(q, 1e-12)
    for order in (0, 1):
        s, i = orc.knn_exact_blockwise(dbn, qn, 10, block=777, order=order)
        assert_topk_equivalent(s, i, g["s_prenorm_k10"], g["i_prenorm_k10"], tol=3e-7)
        s1, i1 = orc.ip_topk(dbn, qn, 10, order=order)
        assert np.array_equal(s, s1) and np.array_equal(i, i1)


def test_topk_merge_and_padding():
    rng = np.random.default_rng(5)
    db = rng.standard_normal((300, 128)).astype(np.float32)
    q = rng.standard_normal((3, 128)).astype(np.float32)
    parts_s, parts_i = [], []
    for a, b in ((0, 7), (7, 150), (150, 300)):       # first shard has fewer rows than k
        s, i = orc.ip_topk(db[a:b], q, 10, row_offset=a)
        parts_s.append(s); parts_i.append(i)
    assert parts_i[0][0, 7:].tolist() == [-1, -1, -1] and np.all(np.isneginf(parts_s[0][:, 7:]))
    s, i = orc.topk_merge(np.stack(parts_s), np.stack(parts_i))
    s_full, i_full = orc.ip_topk(db, q, 10)
    assert np.array_equal(s, s_full) and np.array_equal(i, i_full)


def test_torch_cpu_baseline_legs_are_the_reference_arithmetic(golden_dir):
    """oracle/torch_baseline.py (the torch-CPU legs bench.py times next to the GPU) against the
    reference-derived goldens: EGNN embedding of M0 (G1), cosine+mask top-k (G2), blockwise IP top-k."""
    import torch
    from oracle import torch_baseline as tb
    from merizo_search_amd.foldclass import weights as W
    g = np.load(os.path.join(golden_dir, "egnn.npz"))
    sd = tb.state_dict_tensors(W.synthetic_state_dict(0))
    p = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    e = tb.egnn_forward(sd, torch.from_numpy(p["coords"])[None]).numpy()[0]
    ref = g["emb_M0"]
    assert np.abs(e - ref).max() <= 1e-6 * np.abs(ref).max()
    s = np.load(os.path.join(golden_dir, "search.npz"))
    db, lengths = syn.raw_database(5000, seed=11)
    q, qlen = syn.raw_queries(8, seed=12)
    ts, ti = tb.pt_path_search(torch.from_numpy(db), torch.from_numpy(lengths), torch.from_numpy(q), qlen, 0.7, 10)
    assert_topk_equivalent(ts.numpy(), ti.numpy(), s["s_cov0.7_k10"], s["i_cov0.7_k10"], tol=3e-7)
    dbn = orc.l2_normalize_rows(db, 1e-12)
    fs, fi = tb.faiss_path_search(torch.from_numpy(dbn), torch.from_numpy(q), 10, block=777)
    assert_topk_equivalent(fs.numpy(), fi.numpy(), s["s_prenorm_k10"], s["i_prenorm_k10"], tol=3e-7)


@pytest.mark.parametrize("case", ["M0", "walk97", "walk292"])
def test_oracle_egnn_with_a_large_distance_weight(case, golden_dir):
    """The C restatement against the reference with the distance column of edge_mlp.0.weight at full scale (saturated SiLU:
    oracle/gen_golden_d2.py)."""
    from merizo_search_amd.foldclass import weights as W
    from oracle import oracle as orc
    g = np.load(os.path.join(golden_dir, "egnn_d2.npz"))
    weights, pe = W.pack_state_dict(W.synthetic_state_dict(0, d2_scale=float(g["d2_scale"])))
    e = orc.egnn_embed(weights, pe, [g[f"coords_{case}"]])[0]
    ref = g[f"emb_{case}"]
    assert np.abs(e - ref).max() <= 1e-5 * np.abs(ref).max()
