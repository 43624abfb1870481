"""GPU parity of the HIP search path (through the C ABI) against the CPU oracle and the
reference goldens.  Integer work (indices) is bit-exact; scores are bit-exact in
IP mode (the oracle restates the kernel's k order).  In cosine mode the kernel scales the raw
dot product by 1/||row|| where the reference normalises the row first, so scores agree to
rounding only: COS_TOL = 2e-6 (north_star allows 1e-5)."""
import os

import numpy as np
import pytest

from conftest import assert_topk_equivalent

pytestmark = pytest.mark.gpu
COS_TOL = 2e-6
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    from merizo_search_amd import _lib
    _lib.require_gpu()
    return torch


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _norm_db(n, seed):
    from merizo_search_amd.foldclass import synthetic as syn
    return syn.normalized_database(n, seed)


@pytest.mark.parametrize("n,nq,k", [
    (1, 1, 1), (31, 1, 5), (32, 8, 10), (33, 33, 10), (1000, 64, 10), (5000, 100, 1), (4097, 256, 10),
    (70000, 7, 64), (70000, 256, 10), (20000, 129, 33), (300000, 32, 10),
])
def test_ip_topk_bit_exact_vs_oracle(n, nq, k, torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    db = _norm_db(n, seed=40 + n % 7)
    q, _ = syn.raw_queries(nq, seed=50 + nq)
    q = orc.l2_normalize_rows(q, 1e-12)
    kk = min(k, n) if n >= 1 else k
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), kk, row_offset=1000)
    s_ref, i_ref = orc.ip_topk(db, q, kk, row_offset=1000, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


def test_ip_topk_pads_when_shard_smaller_than_k(torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    db = _norm_db(7, seed=3)
    q = _norm_db(5, seed=4)
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), 10)
    s_ref, i_ref = orc.ip_topk(db, q, 10, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref) and np.array_equal(s.cpu().numpy(), s_ref)
    assert (i.cpu().numpy()[:, 7:] == -1).all() and np.isneginf(s.cpu().numpy()[:, 7:]).all()


@pytest.mark.parametrize("k", [65, 100, 130, 200])
def test_ip_topk_multipass_large_k(k, torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    db = _norm_db(3000, seed=9)
    q = _norm_db(37, seed=10)
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), k)
    s_ref, i_ref = orc.ip_topk(db, q, k, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


def test_ip_topk_ties_resolve_to_lowest_row(torch_gpu):
    """Duplicate rows (exact score ties) anywhere in the tile order: lowest row index wins."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    db = _norm_db(2048, seed=21)
    q = _norm_db(40, seed=22)
    rng = np.random.default_rng(0)
    for src in rng.choice(2048, 64, replace=False):       # copies of strong rows at scattered places
        for dst in rng.choice(2048, 6, replace=False):
            db[dst] = db[src]
    db[5] = q[3]; db[1] = q[3]; db[4] = q[3]; db[36] = q[3]; db[37] = q[3]   # same-tile ties at the top
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), 10)
    s_ref, i_ref = orc.ip_topk(db, q, 10, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    assert i.cpu().numpy()[3, :5].tolist() == [1, 4, 5, 36, 37]


@pytest.mark.parametrize("mincov", [0.0, 0.7])
@pytest.mark.parametrize("k", [1, 10, 100])
def test_cosine_mask_topk_matches_reference_goldens(mincov, k, torch_gpu, golden_dir):
    """search_query_against_db goldens (reference dbsearch.py:75-81), batched on the GPU."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    g = np.load(os.path.join(golden_dir, "search.npz"))
    db, lengths = syn.raw_database(5000, seed=11)
    q, qlen = syn.raw_queries(8, seed=12)
    d_db = _dev(torch, db)
    inv = ops.row_inv_norms(d_db)
    s, i = ops.ip_topk(d_db, _dev(torch, q), k, mode=ops.MODE_COSINE_RAW, inv_norm=inv,
                       lengths=_dev(torch, lengths), qlen=_dev(torch, qlen), mincov=mincov)
    assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), g[f"s_cov{mincov}_k{k}"], g[f"i_cov{mincov}_k{k}"], tol=COS_TOL)
    # inverse norms computed inside the call (as the reference does per query) give the same answer
    s2, i2 = ops.ip_topk(d_db, _dev(torch, q), k, mode=ops.MODE_COSINE_RAW, lengths=_dev(torch, lengths),
                         qlen=_dev(torch, qlen), mincov=mincov)
    assert torch.equal(s, s2) and torch.equal(i, i2)


def test_cosine_all_masked_and_k_equals_ndb(torch_gpu, golden_dir):
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    g = np.load(os.path.join(golden_dir, "search.npz"))
    db, lengths = syn.raw_database(5000, seed=11)
    q, _ = syn.raw_queries(8, seed=12)
    ql = np.array([10.0], np.float32)
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q[:1]), 10, mode=ops.MODE_COSINE_RAW,
                       lengths=_dev(torch, lengths), qlen=_dev(torch, ql), mincov=0.7)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    s_ref, i_ref = orc.cosine_topk(db, q[:1], 10, lengths, ql, 0.7)
    assert (s == 0.0).all() and (g["s_allmasked"] == 0.0).all()           # masked rows score +-0.0, not -inf
    assert np.array_equal(i, i_ref) and i[0].tolist() == list(range(10))   # ties -> ascending row
    assert np.array_equal(np.signbit(s), np.signbit(s_ref))                # cos * 0 keeps its sign, as in torch
    db2, len2 = syn.raw_database(50, seed=13)
    ql = np.array([200.0], np.float32)
    s, i = ops.ip_topk(_dev(torch, db2), _dev(torch, q[1:2]), 50, mode=ops.MODE_COSINE_RAW,
                       lengths=_dev(torch, len2), qlen=_dev(torch, ql), mincov=0.7)
    assert_topk_equivalent(s.cpu().numpy()[0], i.cpu().numpy()[0], g["s_kfull"], g["i_kfull"], tol=COS_TOL)


def test_cosine_vs_oracle_larger(torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    db, lengths = syn.raw_database(60000, seed=31)
    q, qlen = syn.raw_queries(70, seed=32)
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), 10, mode=ops.MODE_COSINE_RAW,
                       lengths=_dev(torch, lengths), qlen=_dev(torch, qlen), mincov=0.7)
    s_ref, i_ref = orc.cosine_topk(db, q, 10, lengths, qlen, 0.7)
    assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), s_ref, i_ref, tol=COS_TOL)


def test_normalize_rows_matches_reference_golden(torch_gpu, golden_dir):
    torch = torch_gpu
    from merizo_search_amd import ops
    g = np.load(os.path.join(golden_dir, "normalize.npz"))
    y = ops.l2_normalize_rows_(_dev(torch, g["x"].copy()), 1e-12).cpu().numpy()
    np.testing.assert_allclose(y, g["y12"], rtol=5e-7, atol=0)
    assert (y[3] == 0).all()
    x_dev = _dev(torch, g["x"].copy())
    y2 = ops.l2_normalize_rows(x_dev, 1e-12)                    # out of place: same bits, input untouched
    assert np.array_equal(y2.cpu().numpy().view(np.uint32), y.view(np.uint32)) and np.array_equal(x_dev.cpu().numpy(), g["x"])
    inv = ops.row_inv_norms(_dev(torch, g["x"]), 1e-8).cpu().numpy()
    nrm = np.maximum(np.sqrt((g["x"].astype(np.float64) ** 2).sum(1)), 1e-8)
    np.testing.assert_allclose(inv, 1.0 / nrm, rtol=5e-7)


def test_topk_merge_shards_equals_unsharded(torch_gpu):
    """sharded(S) == unsharded for S in {2,3,8} (SURVEY.md 8e invariant), uneven shards."""
    torch = torch_gpu
    from merizo_search_amd import ops
    db = _norm_db(10000, seed=61)
    q = _norm_db(50, seed=62)
    d_db, d_q = _dev(torch, db), _dev(torch, q)
    s_full, i_full = ops.ip_topk(d_db, d_q, 10)
    for S in (2, 3, 8):
        bounds = np.linspace(0, 10000, S + 1).astype(int)
        bounds[1] = 5                                        # first shard smaller than k
        parts = [ops.ip_topk(d_db[a:b].contiguous(), d_q, 10, row_offset=int(a)) for a, b in zip(bounds[:-1], bounds[1:])]
        s, i = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        assert torch.equal(i, i_full) and torch.equal(s, s_full)


def test_topk_merge_packed_reads_allgather_blocks_in_place(torch_gpu):
    """ms_topk_merge_strided on S packed blocks [scores | pad | rows] == ms_topk_merge on dense arrays."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass.sharded import PackedExchange
    for nq, k, S in ((9, 3, 3), (64, 10, 8)):
        g = torch.Generator(device="cpu"); g.manual_seed(nq)
        sc = torch.sort(torch.randn((S, nq, k), generator=g), dim=2, descending=True).values
        sc[1, :, k - 1] = sc[0, :, k - 1]                                        # ties across shards
        ix = torch.stack([torch.arange(nq * k, dtype=torch.int64).reshape(nq, k) * S + s for s in range(S)])
        ix[S - 1, 0, k - 1] = -1; sc[S - 1, 0, k - 1] = float("-inf")             # padding entry
        ex = PackedExchange(nq, k, "cuda:0")
        ex.world = S
        ex.gathered = torch.zeros((S, ex.block_bytes), dtype=torch.uint8, device="cuda:0")
        for s in range(S):
            ex.gathered[s, : 4 * nq * k] = sc[s].contiguous().view(torch.uint8).reshape(-1).cuda()
            ex.gathered[s, ex.idx_offset:] = ix[s].contiguous().view(torch.uint8).reshape(-1).cuda()
        ms, mi = ex.merge()
        rs, ri = ops.topk_merge(sc.cuda(), ix.cuda())
        assert torch.equal(ms, rs) and torch.equal(mi, ri)


def test_full_size_properties_c2(torch_gpu):
    """BASELINE config C2 (1M x 128, nq 256, k 10) through size-independent properties:
    planted neighbours are recalled, lists are sorted, scores reproduce from the returned
    rows, and the result is invariant to splitting the database."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    n, nq, k = 1_000_000, 256, 10
    db = syn.device_database(n, 0, seed=0, device="cuda:0")
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    q = torch.randn((nq, 128), generator=g, device="cuda:0")
    q = q / q.norm(dim=1, keepdim=True)
    rows = torch.randperm(n, generator=g, device="cuda:0")[: nq * 3].reshape(nq, 3)
    noise = torch.randn((nq, 3, 128), generator=g, device="cuda:0") * 0.02
    planted = q[:, None, :] + noise
    db[rows.reshape(-1)] = (planted / planted.norm(dim=2, keepdim=True)).reshape(-1, 128)
    s, i = ops.ip_topk(db, q, k)
    assert (s[:, :-1] >= s[:, 1:]).all()
    assert all(set(rows[j].tolist()) <= set(i[j].tolist()) for j in range(nq))       # recall of planted rows
    rescored = (db[i.reshape(-1)].reshape(nq, k, 128) * q[:, None, :]).sum(-1)
    assert (rescored - s).abs().max() < 2e-6
    # brute-force check of the k-th score with torch on a sample of queries
    ref = (q[:8] @ db.T).topk(k, dim=1)
    assert torch.equal(ref.indices, i[:8]) and (ref.values - s[:8]).abs().max() < 2e-6
    half = n // 2 + 12345
    p0 = ops.ip_topk(db[:half], q, k)
    p1 = ops.ip_topk(db[half:], q, k, row_offset=half)
    s2, i2 = ops.topk_merge(torch.stack([p0[0], p1[0]]), torch.stack([p0[1], p1[1]]))
    assert torch.equal(i2, i) and torch.equal(s2, s)


def test_sample_prepass_is_exact_on_adversarial_order(torch_gpu):
    """Sizes where the sample pre-pass is active (>= 32 tiles per row stream).  The sampled rows
    (first tiles of each stream) are made UNREPRESENTATIVE: the database is sorted so that every
    stream starts with its worst rows for query 0, plus exact duplicates straddling streams;
    the bound from the sample must still never drop a true top-k row."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, nq, k = 160_000, 130, 10
    db = _norm_db(n, seed=71)
    q = _norm_db(nq, seed=72)
    order = np.argsort(db @ q[0])                      # ascending score for query 0
    db = np.ascontiguousarray(db[order])
    db[150_000] = db[10]; db[77_777] = db[159_999]; db[5] = db[159_999]        # ties across streams
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), k)
    s_ref, i_ref = orc.ip_topk(db, q, k, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


@pytest.mark.parametrize("n,nq,k", [
    (131_073, 96, 10), (131_105, 96, 10), (200_000, 130, 5), (262_113, 97, 1), (400_000, 256, 10), (300_000, 100, 64),
    (1_000_003, 256, 10), (400_000, 130, 21), (400_000, 256, 32), (131_105, 97, 25),
    (1_000_003, 256, 64), (400_000, 130, 33), (131_105, 97, 50), (20_000, 129, 33), (262_113, 512, 64),
])
def test_loader_wave_scan_with_sample_bound(n, nq, k, torch_gpu):
    """Batches of >= 3 query tiles run the loader-wave kernel, thresholded from its first tile by
    the sample pass's lower bound.  Sizes around stream boundaries (last stream shorter than the
    sample, partial last tile), duplicates between sampled and unsampled rows."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    db = _norm_db(n, seed=81 + n % 5)
    q = _norm_db(nq, seed=82)
    db[n - 1] = db[3]; db[n // 2] = db[3]; db[40] = db[n - 2]          # duplicates: sampled vs unsampled rows
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), k, row_offset=7)
    s_ref, i_ref = orc.ip_topk(db, q, k, row_offset=7, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


@pytest.mark.parametrize("k", [33, 48, 64])
def test_long_lists_append_path_with_ties_in_every_tile(k, torch_gpu):
    """k > 32 (lists of 32 entries per lane): candidates are buffered in LDS and the lists take them at flushes, a query's two
    buffers merged in row order -- a database of 64 distinct rows repeated all over makes every tile tie with every other one
    (the lowest rows must win, whatever the buffering did), then every score equal, then the cosine mode on unit rows with a
    length mask (the mask is applied in the rare path), and a database too short for a sample pass (no threshold: every row
    is a candidate and the buffers are emptied inside every tile)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    rng = np.random.default_rng(k)
    n, nq = 200_000, 130
    base = _norm_db(64, seed=7)
    db = np.ascontiguousarray(base[rng.integers(0, 64, size=n)])
    q = _norm_db(nq, seed=8)
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), k, row_offset=11)
    s_ref, i_ref = orc.ip_topk(db, q, k, row_offset=11, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    same = np.tile(base[:1], (n, 1))
    s, i = ops.ip_topk(_dev(torch, same), _dev(torch, q), k)
    assert (i.cpu().numpy() == np.arange(k)[None, :]).all()
    for n2 in (150_000, 3_000):
        raw, lengths = syn.raw_database(n2, seed=92)
        raw[rng.integers(0, n2, size=n2 // 2)] = raw[rng.integers(0, 50, size=n2 // 2)]       # half the rows repeat 50 of them
        rq, qlen = syn.raw_queries(nq, seed=93)
        unit = ops.l2_normalize_rows_(_dev(torch, raw), 1e-8)
        s, i = ops.ip_topk(unit, _dev(torch, rq), k, mode=ops.MODE_COSINE_UNIT, lengths=_dev(torch, lengths), qlen=_dev(torch, qlen), mincov=0.7)
        s_ref, i_ref = orc.cosine_topk(raw, rq, k, lengths, qlen, 0.7)
        assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), s_ref, i_ref, tol=COS_TOL)


def test_loader_wave_all_equal_scores_and_cosine_mask(torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    n, nq, k = 300_000, 96, 7
    db = np.tile(_norm_db(1, seed=5), (n, 1))          # every score ties: the lowest rows win
    s, i = ops.ip_topk(_dev(torch, db), _dev(torch, _norm_db(nq, seed=6)), k)
    assert (i.cpu().numpy() == np.arange(k)[None, :]).all()
    # cosine mode (row scale + length mask staged by the loader wave) against the oracle
    n, nq, k = 250_000, 100, 10
    raw, lengths = syn.raw_database(n, seed=90)
    rq, qlen = syn.raw_queries(nq, seed=91)
    d = _dev(torch, raw)
    s, i = ops.ip_topk(d, _dev(torch, rq), k, mode=ops.MODE_COSINE_RAW, inv_norm=ops.row_inv_norms(d),
                       lengths=_dev(torch, lengths), qlen=_dev(torch, qlen), mincov=0.7)
    s_ref, i_ref = orc.cosine_topk(raw, rq, k, lengths, qlen, 0.7)
    assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), s_ref, i_ref, tol=COS_TOL)


def test_staged_api_equals_one_shot_and_all_equal_scores(torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    n, nq, k = 150_000, 64, 7
    db = np.tile(_norm_db(1, seed=5), (n, 1))          # every row identical: every score ties
    q = _norm_db(nq, seed=6)
    d_db, d_q = _dev(torch, db), _dev(torch, q)
    s, i = ops.ip_topk(d_db, d_q, k)
    assert (i.cpu().numpy() == np.arange(k)[None, :]).all()      # ties -> lowest rows
    ws = ops.TopKWorkspace(d_db.device).get(n, nq, k)
    out_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    out_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ops.ip_topk_prepare(d_db, d_q, k, ws); ops.ip_topk_scan(d_db, d_q, k, ws); ops.ip_topk_finish(n, nq, k, ws, out_s, out_i, row_offset=42)
    assert torch.equal(out_s, s) and torch.equal(out_i, i + 42)


@pytest.mark.parametrize("n", [10, 33, 42, 70, 1000])
def test_cosine_tiny_databases_partial_tiles(n, torch_gpu):
    """Streams shorter than one tile / with a partial tail in cosine mode (regression: scores of a
    non-existent pipeline tile scaled by uninitialised LDS)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    db, lengths = syn.raw_database(n, seed=90 + n)
    q, qlen = syn.raw_queries(5, seed=91)
    junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk       # dirty the allocator's memory
    for _ in range(3):
        s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), min(5, n), mode=ops.MODE_COSINE_RAW,
                           lengths=_dev(torch, lengths), qlen=_dev(torch, qlen), mincov=0.7)
        s_ref, i_ref = orc.cosine_topk(db, q, min(5, n), lengths, qlen, 0.7)
        assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), s_ref, i_ref, tol=COS_TOL)


@pytest.mark.parametrize("S,nq,k", [(65, 9, 10), (200, 3, 1), (130, 40, 7)])
def test_topk_merge_beyond_64_lists(S, nq, k, torch_gpu):
    """ms_topk_merge with more than 64 lists (no per-thread head array) == the 64-list kernel applied in two levels
    == the oracle's merge; padding entries and cross-list score ties included."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    rng = np.random.default_rng(S)
    scores = np.sort(rng.standard_normal((S, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    scores[3] = scores[70 % S]                                   # equal scores in two lists: the lower index wins
    idx = rng.permutation(S * nq * k).astype(np.int64).reshape(S, nq, k)
    idx.sort(axis=2)                                             # (ties inside a list stay in index order)
    scores[5, :, k // 2:] = -np.inf; idx[5, :, k // 2:] = -1     # a short list
    s, i = ops.topk_merge(_dev(torch, scores), _dev(torch, idx))
    s_ref, i_ref = orc.topk_merge(scores, idx)
    assert np.array_equal(i.cpu().numpy(), i_ref) and np.array_equal(s.cpu().numpy(), s_ref)


def test_nan_and_inf_rows_are_never_returned(torch_gpu):
    """NaN policy: a database row whose score is NaN or -inf is never returned, whatever the kernel form;
    +inf scores sort first.  (torch.topk would rank NaN first; the reference never produces one from finite embeddings.)"""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    for n, nq in ((5000, 3), (200_000, 100)):
        db = _norm_db(n, seed=31)
        q = _norm_db(nq, seed=32)
        bad = np.array([0, 17, 33, n // 2, n - 1])
        db[bad[:3], 5] = np.nan
        db[bad[3]] = -np.inf                                     # -inf * (mixed signs) = NaN / -inf
        db[bad[4], :] = np.nan
        s, i = ops.ip_topk(_dev(torch, db), _dev(torch, q), 10)
        s, i = s.cpu().numpy(), i.cpu().numpy()
        assert np.isfinite(s).all() and not np.isin(i, bad).any()
        keep = np.setdiff1d(np.arange(n), bad)
        s_ref, i_ref = orc.ip_topk(db[keep], q, 10, order=1)
        assert np.array_equal(i, keep[i_ref]) and np.array_equal(s.view(np.uint32), s_ref.view(np.uint32))


def test_empty_shard_is_searchable(torch_gpu):
    """A rank of a sharded search may hold no rows at all (more ranks than rows): both modes return (-inf, -1) lists that
    the merge then ignores."""
    torch = torch_gpu
    from merizo_search_amd import ops
    q = _dev(torch, _norm_db(5, seed=2))
    empty = torch.empty((0, 128), dtype=torch.float32, device="cuda")
    for nq in (1, 5):
        s, i = ops.ip_topk(empty, q[:nq], 3, row_offset=77)
        assert np.isneginf(s.cpu().numpy()).all() and (i.cpu().numpy() == -1).all()
        s, i = ops.ip_topk(empty, q[:nq], 3, mode=ops.MODE_COSINE_RAW, inv_norm=ops.row_inv_norms(empty),
                           lengths=torch.empty(0, device="cuda"), qlen=torch.ones(nq, device="cuda"), mincov=0.7)
        assert np.isneginf(s.cpu().numpy()).all() and (i.cpu().numpy() == -1).all()
    db = _dev(torch, _norm_db(100, seed=3))
    full = ops.ip_topk(db, q, 3)
    e = ops.ip_topk(empty, q, 3, row_offset=100)
    ms, mi = ops.topk_merge(torch.stack([full[0], e[0]]), torch.stack([full[1], e[1]]))
    assert torch.equal(ms, full[0]) and torch.equal(mi, full[1])


@pytest.mark.parametrize("n,nq,k", [(5000, 4, 100), (5000, 70, 130), (40_000, 3, 64), (300, 33, 300)])
def test_cosine_mask_large_k_vs_reference_arithmetic(n, nq, k, torch_gpu):
    """Cosine + length mask with k beyond one pass (k > 64: ceil(k / 64) scans with an exclusive upper bound) and k = n,
    against the oracle's reference arithmetic."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    raw, lengths = syn.raw_database(n, seed=95 + n % 5)
    rq, qlen = syn.raw_queries(nq, seed=96)
    d = _dev(torch, raw)
    s, i = ops.ip_topk(d, _dev(torch, rq), k, mode=ops.MODE_COSINE_RAW, inv_norm=ops.row_inv_norms(d),
                       lengths=_dev(torch, lengths), qlen=_dev(torch, qlen), mincov=0.7)
    s_ref, i_ref = orc.cosine_topk(raw, rq, k, lengths, qlen, 0.7)
    assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), s_ref, i_ref, tol=COS_TOL)


@pytest.mark.parametrize("n", [100_000, 400_000])
@pytest.mark.parametrize("nq", [8, 96])
def test_cosine_mask_zero_ties_under_the_sample_bound(n, nq, torch_gpu):
    """The sample pass is active here (nq >= 8, >= 12 tiles per stream).  Queries so short that EVERY row is masked have
    all scores +-0.0 (reference dbsearch.py:76-79: the mask multiplies, it does not exclude), so their sample bound is
    lb = +-0.0 and the full pass runs on floor = nextbelow(0) = -1.4e-45, a denormal threshold; queries with fewer than k
    unmasked rows end their lists in zeros.  Indices and sign bits as the oracle's: all-masked lists are rows 0..k-1 in
    ascending order (easy-search of short domains against CATH)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    k, mincov = 10, 0.7
    db, lengths = syn.raw_database(n, seed=61 + nq)
    lengths = lengths.copy()
    short = np.random.default_rng(5).choice(n, 7, replace=False)         # a handful of very short targets
    lengths[short] = np.array([20, 20, 21, 22, 24, 25, 26], np.float32)[: len(short)]
    q, qlen = syn.raw_queries(nq, seed=62)
    qlen = qlen.copy()
    qlen[0::4] = 3.0                  # every row masked: 3 >= len * 0.7 never holds (len >= 20)
    qlen[1::4] = 15.0                 # unmasked rows: only targets with len <= 21.4 -> 3 rows (< k)
    qlen[2::4] = 17.0                 # unmasked rows: len <= 24.3 -> 5 rows (< k)
    d_db = _dev(torch, db)
    inv = ops.row_inv_norms(d_db)
    s, i = ops.ip_topk(d_db, _dev(torch, q), k, mode=ops.MODE_COSINE_RAW, inv_norm=inv, lengths=_dev(torch, lengths),
                       qlen=_dev(torch, qlen), mincov=mincov)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    s_ref, i_ref = orc.cosine_topk(db, q, k, lengths, qlen, mincov)
    assert_topk_equivalent(s, i, s_ref, i_ref, tol=COS_TOL)
    allm = np.arange(0, nq, 4)
    assert (s[allm] == 0.0).all() and np.array_equal(i[allm], np.tile(np.arange(k), (len(allm), 1)))
    assert np.array_equal(np.signbit(s[allm]), np.signbit(s_ref[allm]))
    for j in range(1, nq, 4):         # 3 unmasked rows: positive cosines first, then zeros of the masked rows by ascending row
        pos = int((s_ref[j] > 0).sum())
        assert np.array_equal(i[j, :pos], i_ref[j, :pos])
        zero = s_ref[j] == 0.0
        assert np.array_equal(i[j, zero], i_ref[j, zero]) and np.array_equal(np.signbit(s[j, zero]), np.signbit(s_ref[j, zero]))


def _cosine_unit(torch, ops, db, q, k, lengths=None, qlen=None, mincov=0.0, **kw):
    """MS_MODE_COSINE_UNIT as the engine uses it: rows normalised once on the device (eps 1e-8), raw queries."""
    rows = ops.l2_normalize_rows_(_dev(torch, db).clone(), 1e-8)
    return ops.ip_topk(rows, _dev(torch, q), k, mode=ops.MODE_COSINE_UNIT, lengths=None if lengths is None else _dev(torch, lengths),
                       qlen=None if qlen is None else _dev(torch, qlen), mincov=mincov, **kw)


@pytest.mark.parametrize("mincov", [0.0, 0.7])
@pytest.mark.parametrize("k", [1, 10, 100])
def test_cosine_unit_mode_matches_reference_goldens(mincov, k, torch_gpu, golden_dir):
    """The `.pt` search on rows normalised ahead of time (MS_MODE_COSINE_UNIT, what the engine keeps resident) against
    the reference's search_query_against_db goldens (dbsearch.py:75-81) and against the raw-row mode."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    g = np.load(os.path.join(golden_dir, "search.npz"))
    db, lengths = syn.raw_database(5000, seed=11)
    q, qlen = syn.raw_queries(8, seed=12)
    s, i = _cosine_unit(torch, ops, db, q, k, lengths, qlen, mincov)
    assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), g[f"s_cov{mincov}_k{k}"], g[f"i_cov{mincov}_k{k}"], tol=COS_TOL)


@pytest.mark.parametrize("n,nq,k", [(70_000, 7, 10), (70_000, 100, 10), (300_000, 256, 20), (150_000, 96, 32), (60_000, 130, 64), (3000, 70, 100)])
def test_cosine_unit_mode_vs_oracle_all_kernel_forms(n, nq, k, torch_gpu):
    """Few queries (row streams per wave), >= 3 query tiles (loader-wave form: mask applied in the rare path), long lists,
    k > 64 (bounded passes): masked cosine top-k of the oracle (reference arithmetic on the raw rows)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    db, lengths = syn.raw_database(n, seed=71)
    q, qlen = syn.raw_queries(nq, seed=72)
    s, i = _cosine_unit(torch, ops, db, q, k, lengths, qlen, 0.7, row_offset=17)
    s_ref, i_ref = orc.cosine_topk(db, q, k, lengths, qlen, 0.7, row_offset=17)
    assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), s_ref, i_ref, tol=COS_TOL)
    s2, i2 = _cosine_unit(torch, ops, db, q, k)                       # no mask: plain cosine
    s_ref, i_ref = orc.cosine_topk(db, q, k, None, None, 0.0)
    assert_topk_equivalent(s2.cpu().numpy(), i2.cpu().numpy(), s_ref, i_ref, tol=COS_TOL)


@pytest.mark.parametrize("n", [100_000, 400_000])
@pytest.mark.parametrize("nq", [8, 96])
def test_cosine_unit_mode_zero_ties_and_negative_thresholds(n, nq, torch_gpu):
    """The unit-row form filters on unmasked scores and applies the mask in the rare path, which is only sound while a
    query's threshold is >= 0: queries whose lists hold masked zeros or negative cosines (all rows masked, fewer than k
    unmasked rows, or mostly negative scores) keep a negative threshold and must take the exact path.  Same construction
    as the raw-row zero-tie test, plus queries pointing AWAY from every row (all cosines negative)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    k, mincov = 10, 0.7
    db, lengths = syn.raw_database(n, seed=61 + nq)
    db = np.abs(db)                                   # every row in the positive orthant ...
    lengths = lengths.copy()
    short = np.random.default_rng(5).choice(n, 7, replace=False)
    lengths[short] = np.array([20, 20, 21, 22, 24, 25, 26], np.float32)
    q, qlen = syn.raw_queries(nq, seed=62)
    q = q.copy(); qlen = qlen.copy()
    q[3::4] = -np.abs(q[3::4])                        # ... and these queries in the negative one: all cosines < 0
    qlen[3::4] = 18.0                                 # 6 unmasked rows (negative scores), everything else masked to -0.0
    qlen[0::4] = 3.0; qlen[1::4] = 15.0; qlen[2::4] = 17.0
    s, i = _cosine_unit(torch, ops, db, q, k, lengths, qlen, mincov)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    s_ref, i_ref = orc.cosine_topk(db, q, k, lengths, qlen, mincov)
    assert_topk_equivalent(s, i, s_ref, i_ref, tol=COS_TOL)
    zero = s_ref == 0.0
    assert np.array_equal(i[zero], i_ref[zero]) and np.array_equal(np.signbit(s[zero]), np.signbit(s_ref[zero]))
    assert (s_ref[3::4] <= 0).all() and (s_ref[3::4, 0] == 0).all()          # masked zeros outrank the negative cosines


def test_staged_scan_twice_after_one_prepare_is_still_exact(torch_gpu):
    """ms_ip_topk_prepare leaves the shared-bound counters zeroed for ONE scan; a second ms_ip_topk_scan on the same workspace
    without a new prepare must not count rows twice (the library zeroes the counters itself then)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, nq, k = 300_000, 128, 10
    db, q = _norm_db(n, seed=5), _norm_db(nq, seed=6)
    d, dq = _dev(torch, db), _dev(torch, q)
    ws = ops.TopKWorkspace(d.device).get(n, nq, k)
    out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
    s_ref, i_ref = orc.ip_topk(db, q, k, order=1)
    ops.ip_topk_prepare(d, dq, k, ws)
    for _ in range(3):
        ops.ip_topk_scan(d, dq, k, ws)
        ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
        assert np.array_equal(out_i.cpu().numpy(), i_ref) and np.array_equal(out_s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


@pytest.mark.parametrize("n,nq,k", [(1, 1, 1), (5000, 1, 10), (200_000, 3, 10), (1_000_000, 1, 10), (1_000_000, 8, 16), (300_000, 5, 1),
                                    (40, 2, 10), (777, 8, 10)])
def test_few_queries_merge_inside_the_scan_launch(n, nq, k, torch_gpu):
    """Few queries (<= MS_FUSED_MERGE_MAX_NQ: 2 by default, 8 under tests/conftest.py): the last workgroup of the scan launch
    merges the per-stream lists itself (no merge launch).  Bit-exact
    against the oracle, also when called again and again on the same workspace (the arrival counters reset themselves) and
    for shards shorter than k (padding)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    db, q = _norm_db(n, seed=81 + nq), _norm_db(nq, seed=82 + k)
    d, dq = _dev(torch, db), _dev(torch, q)
    ws = ops.TopKWorkspace(d.device)
    s_ref, i_ref = orc.ip_topk(db, q, k, row_offset=5, order=1)
    for _ in range(3):
        s, i = ops.ip_topk(d, dq, k, row_offset=5, workspace=ws)
        assert np.array_equal(i.cpu().numpy(), i_ref)
        assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


@pytest.mark.parametrize("n,nq,k", [(50_000, 1, 10), (50_000, 5, 10), (50_000, 16, 10), (50_000, 17, 10), (400_000, 32, 10), (400_000, 33, 10), (400_000, 64, 20), (400_000, 100, 10),
                                    (1_000_000, 256, 10), (3000, 70, 100)])
def test_ip_normq_mode_equals_normalize_then_search(n, nq, k, torch_gpu):
    """MS_MODE_IP_NORMQ (raw queries, F.normalize fused into the call: inside the scan launch for a handful of queries
    (MS_INKERNEL_NORM_MAX_NQ), in the query preparation kernel above) == ms_l2_normalize_rows_to followed by MS_MODE_IP_PRENORM, bit for bit."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    db = _norm_db(n, seed=91)
    q_raw, _ = syn.raw_queries(nq, seed=92)
    q_raw = (q_raw * 3.0).astype(np.float32)
    q_raw[0] = 0.0 if nq > 1 else q_raw[0]                     # a zero query: x / max(0, 1e-12) = 0
    d, dq = _dev(torch, db), _dev(torch, q_raw)
    s1, i1 = ops.ip_topk(d, ops.l2_normalize_rows(dq, 1e-12), k)
    s2, i2 = ops.ip_topk(d, dq, k, mode=ops.MODE_IP_NORMQ)
    assert torch.equal(i1, i2) and torch.equal(s1.view(torch.int32), s2.view(torch.int32))


@pytest.mark.parametrize("n,nq,k", [(300_000, 1, 10), (300_000, 40, 10), (1_000_000, 8, 16), (200_000, 300, 10), (300_000, 20, 64),
                                    (900, 3, 10), (4000, 33, 20)])
def test_merge_by_threshold_handles_ties_by_the_hundred_and_short_shards(n, nq, k, torch_gpu):
    """The workgroup merge (threshold = k-th best list head, survivors rank themselves) and the shapes it hands to the
    head-advance merge: every score equal (hundreds of entries tie with the threshold), blocks of duplicated rows, shards with
    fewer lists than k, and ordinary data -- all bit-exact against the oracle, lowest row first among ties."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    q = _norm_db(nq, seed=300 + nq)
    dq = _dev(torch, q)
    # (a) every row the same: all scores of a query tie
    db = np.tile(_norm_db(1, seed=301), (n, 1))
    s, i = ops.ip_topk(_dev(torch, db), dq, k)
    assert (i.cpu().numpy() == np.arange(min(k, n))[None, :]).all()
    # (b) ordinary rows with the best row of query 0 copied into every 1000th row: as many ties at the top as there are lists
    db = _norm_db(n, seed=302)
    db[::1000] = q[0]
    s, i = ops.ip_topk(_dev(torch, db), dq, k)
    s_ref, i_ref = orc.ip_topk(db, q, k, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    assert i.cpu().numpy()[0].tolist() == list(range(0, 1000 * k, 1000))[:k] or n < 1000 * k


@pytest.mark.parametrize("k", [10, 48, 64])
def test_repeated_searches_return_identical_results(k, torch_gpu):
    """Run-to-run determinism (no atomics decide a result; thresholds only decide how fast).  With 32-entry lists in the
    loader-wave form, fragment registers still in flight when the compiler-scheduled insertion path started were moved by
    hipcc: about one run in twenty lost a row.  150 runs of one search, and the first against the oracle."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, nq = 202_000, 738
    db, q = _norm_db(n, seed=46), _norm_db(nq, seed=47)
    d, dq = _dev(torch, db), _dev(torch, q)
    s0, i0 = ops.ip_topk(d, dq, k)
    s_ref, i_ref = orc.ip_topk(db, q, k, order=1)
    assert np.array_equal(i0.cpu().numpy(), i_ref) and np.array_equal(s0.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    for _ in range(150):
        s, i = ops.ip_topk(d, dq, k)
        assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))


def test_few_query_shapes_with_the_production_thresholds(tmp_path):
    """tests/conftest.py widens the few-query shortcuts (MS_FUSED_MERGE_MAX_NQ = 8, MS_INKERNEL_NORM_MAX_NQ = 16) so that the parity
    cases run through them; THIS test runs 1..20 raw queries with the library's own defaults (2 and 4: what a user gets) in a fresh
    process -- the switches are read once -- against the oracle: indices and score bits identical."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
from oracle import oracle as orc
assert ops.small_batch_thresholds() == (2, 4), ops.small_batch_thresholds()
db = syn.normalized_database(150_003, seed=881)
d = torch.from_numpy(db).cuda()
for nq in (1, 2, 3, 4, 5, 8, 9, 16, 17, 20):
    q, _ = syn.raw_queries(nq, seed=882 + nq)
    dq = torch.from_numpy(q).cuda()
    s, i = ops.ip_topk(d, dq, 10, mode=ops.MODE_IP_NORMQ, row_offset=5)
    qn = ops.l2_normalize_rows(dq, 1e-12).cpu().numpy()
    s_ref, i_ref = orc.ip_topk(db, qn, 10, row_offset=5, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref) and np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32)), nq
print("production thresholds ok")
'''
    env = {k: v for k, v in os.environ.items() if k not in ("MS_FUSED_MERGE_MAX_NQ", "MS_INKERNEL_NORM_MAX_NQ")}
    r = subprocess.run([sys.executable, "-c", code, REPO], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "production thresholds ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
