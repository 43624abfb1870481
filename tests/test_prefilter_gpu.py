"""GPU parity of the prefiltered search (ms_ip_topk_prefiltered): the rows are scanned with bf16 matrix instructions on split
operands, the survivors re-scored with the exact fp32 chain, and every query's answer is proved complete -- or the exact
pipeline runs after all.  The bar is the one of ms_ip_topk: indices AND score bits identical to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _check(torch, ops, orc, db, q, k, bound, row_offset=0, raw=False, expect_fallback=None):
    d, dq = _dev(torch, db), _dev(torch, q)
    ws = ops.PrefilterWorkspace(d.device).get(db.shape[0], q.shape[0], k)
    mode = ops.MODE_IP_NORMQ if raw else ops.MODE_IP_PRENORM
    s, i = ops.ip_topk_prefiltered(d, dq, k, bound, mode=mode, row_offset=row_offset, workspace=ws)
    fell_back = ops.prefilter_fell_back(ws)
    qn = ops.l2_normalize_rows(dq, 1e-12).cpu().numpy() if raw else q      # (the library's own F.normalize: ms_l2_normalize_rows_to)
    s_ref, i_ref = orc.ip_topk(db, qn, k, row_offset=row_offset, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    if expect_fallback is not None:
        assert fell_back == expect_fallback
    return fell_back


@pytest.mark.parametrize("n,nq,k", [(70_000, 65, 1), (70_000, 100, 5), (131_105, 97, 10), (300_000, 256, 10), (262_113, 130, 16),
                                    (400_000, 200, 20), (200_000, 129, 32), (300_000, 100, 40), (1_000_003, 256, 10), (100_000, 1000, 3)])
def test_prefiltered_is_bit_identical_and_needs_no_exact_pass_on_ordinary_data(n, nq, k, torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    db, q = _norm_db(n, seed=401 + k), _norm_db(nq, seed=402)
    db[n - 1] = db[3]; db[n // 2] = db[3]                     # a few exact duplicates (ties resolve to the lowest row)
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, row_offset=11, expect_fallback=False)


def test_prefiltered_raw_queries_and_rows_that_are_not_unit_vectors(torch_gpu):
    """MS_MODE_IP_NORMQ (F.normalize inside the call) and a database whose rows have norms 0.2 .. 3: the error bound scales with
    the row-norm bound the caller measured."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    n, nq, k = 200_000, 150, 10
    rng = np.random.default_rng(5)
    db = (_norm_db(n, seed=411) * rng.uniform(0.2, 3.0, size=(n, 1))).astype(np.float32)
    q_raw, _ = syn.raw_queries(nq, seed=412)
    bound = float(np.linalg.norm(db.astype(np.float64), axis=1).max()) * (1 + 1e-6)
    _check(torch, ops, orc, db, (q_raw * 2.5).astype(np.float32), k, bound, raw=True, expect_fallback=False)
    q = (_norm_db(nq, seed=413) * 4.0).astype(np.float32)    # MS_MODE_IP_PRENORM with queries that are not unit vectors either
    _check(torch, ops, orc, db, q, k, bound)


def test_prefiltered_near_ties_by_the_hundred_fall_back_to_the_exact_pipeline(torch_gpu):
    """300 rows within 1e-6 of each other around every query's best score (copies of the query's own direction with tiny
    perturbations) and blocks of exact duplicates: the proof cannot succeed, the gate opens, the exact pipeline answers."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, nq, k = 150_000, 96, 10
    db, q = _norm_db(n, seed=421), _norm_db(nq, seed=422)
    rng = np.random.default_rng(6)
    rows = rng.choice(n, size=(nq, 300), replace=False)
    for j in range(nq):
        v = q[j][None, :] + rng.normal(0, 2e-7, size=(300, 128)).astype(np.float32)
        db[rows[j]] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    db[rows[0, :50]] = db[rows[0, 0]]                         # exact ties too
    fell_back = _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, expect_fallback=True)
    assert fell_back


@pytest.mark.parametrize("n,nq,k", [(300_000, 40, 10), (300_000, 100, 49), (20_000, 100, 10), (300_000, 100, 100)])
def test_prefiltered_shapes_it_does_not_serve_take_the_plain_path(n, nq, k, torch_gpu):
    """<= 64 queries, k > 48, small databases: the call is ms_ip_topk (same results, of course)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    _check(torch, ops, orc, _norm_db(n, seed=431), _norm_db(nq, seed=432), k, 1.0 + 1e-6)


def test_prefiltered_stages_equal_the_one_shot_call_and_repeat(torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    n, nq, k = 500_000, 256, 10
    db, q = _norm_db(n, seed=441), _norm_db(nq, seed=442)
    d, dq = _dev(torch, db), _dev(torch, q)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    s0, i0 = ops.ip_topk(d, dq, k)
    out = (torch.empty_like(s0), torch.empty_like(i0))
    for _ in range(3):
        ops.ip_topk_prefiltered_stage("prepare", d, dq, k, ws)
        ops.ip_topk_prefiltered_stage("scan", d, dq, k, ws)
        ops.ip_topk_prefiltered_stage("finish", d, dq, k, ws, out=out)
        assert torch.equal(out[1], i0) and torch.equal(out[0].view(torch.int32), s0.view(torch.int32))
        out[0].zero_(); out[1].zero_()
    ops.ip_topk_prefiltered_stage("prepare", d, dq, k, ws)
    for _ in range(2):                                         # the same workspace scanned twice after one prepare
        ops.ip_topk_prefiltered_stage("scan", d, dq, k, ws)
        ops.ip_topk_prefiltered_stage("finish", d, dq, k, ws, out=out)
        assert torch.equal(out[1], i0) and torch.equal(out[0].view(torch.int32), s0.view(torch.int32))


def test_engine_uses_the_prefilter_for_large_batches_on_a_resident_database(torch_gpu):
    """foldclass/engine.py: ip_topk with a row-norm bound (what dbsearch_faiss passes for a resident shard) == without."""
    torch = torch_gpu
    from merizo_search_amd.foldclass import engine as eng
    e = eng.HipEngine("cuda:0")
    db, q = _norm_db(300_000, seed=451), _norm_db(200, seed=452)
    d, dq = _dev(torch, db), _dev(torch, (q * 3).astype(np.float32))
    bound = e.row_norm_bound(d)
    assert 1.0 <= bound < 1.0001
    s1, i1 = e.ip_topk(d, dq, 10, row_offset=5, normalize_queries=True, row_norm_bound=bound)
    s0, i0 = e.ip_topk(d, dq, 10, row_offset=5, normalize_queries=True)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))


def test_prefiltered_shards_merge_to_the_unsharded_answer(torch_gpu):
    """Two row shards searched through the prefilter (row offsets, the shard's own row-norm bound) + ms_topk_merge == one
    search over all rows: what dbsearch_faiss does on two ranks."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import engine as eng
    e = eng.HipEngine("cuda:0")
    n, nq, k = 400_000, 150, 10
    db, q = _norm_db(n, seed=461), _norm_db(nq, seed=462)
    d, dq = _dev(torch, db), _dev(torch, q)
    s_all, i_all = ops.ip_topk(d, dq, k)
    parts = []
    for lo, hi in ((0, 170_000), (170_000, n)):
        shard = d[lo:hi].contiguous()
        parts.append(e.ip_topk(shard, dq, k, row_offset=lo, row_norm_bound=e.row_norm_bound(shard)))
    s, i = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(i, i_all) and torch.equal(s.view(torch.int32), s_all.view(torch.int32))


def test_prefiltered_declines_databases_with_non_finite_rows(torch_gpu):
    """A row with an infinite element has no norm bound: engine.row_norm_bound is not finite and the call is ms_ip_topk."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import engine as eng
    e = eng.HipEngine("cuda:0")
    db, q = _norm_db(100_000, seed=471), _norm_db(100, seed=472)
    db[777, 5] = np.inf
    d, dq = _dev(torch, db), _dev(torch, q)
    bound = e.row_norm_bound(d)
    assert not np.isfinite(bound)
    s1, i1 = e.ip_topk(d, dq, 10, row_norm_bound=bound)
    s0, i0 = ops.ip_topk(d, dq, 10)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))


def test_engine_switches_the_prefilter_off_after_two_batches_that_needed_the_exact_pass(torch_gpu):
    """foldclass/engine.py: near-duplicate families make every batch run both scans; after two in a row the engine uses the fp32
    scan for this database.  Results are exact before and after."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import engine as eng
    e = eng.HipEngine("cuda:0")
    n, nq, k = 150_000, 96, 10
    db, q = _norm_db(n, seed=481), _norm_db(nq, seed=482)
    rng = np.random.default_rng(7)
    rows = rng.choice(n, size=(nq, 100), replace=False)
    for j in range(nq):
        v = q[j][None, :] + rng.normal(0, 2e-7, size=(100, 128)).astype(np.float32)
        db[rows[j]] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    d, dq = _dev(torch, db), _dev(torch, q)
    bound = e.row_norm_bound(d)
    s0, i0 = ops.ip_topk(d, dq, k)
    for call in range(4):
        s, i = e.ip_topk(d, dq, k, row_norm_bound=bound)
        assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))
        s.cpu()
        e.prefilter_feedback()
        assert e._pf_enabled == (call < 1)


@pytest.mark.parametrize("k", [10, 40])
def test_prefiltered_repeated_searches_return_identical_results(k, torch_gpu):
    """Run-to-run determinism of the prefiltered search (100 runs of one search; the first against the fp32 scan)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    n, nq = 202_000, 738
    d, dq = _dev(torch, _norm_db(n, seed=46)), _dev(torch, _norm_db(nq, seed=47))
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    s_ref, i_ref = ops.ip_topk(d, dq, k)
    for _ in range(100):
        s, i = ops.ip_topk_prefiltered(d, dq, k, 1.0 + 1e-6, workspace=ws)
        assert torch.equal(i, i_ref) and torch.equal(s.view(torch.int32), s_ref.view(torch.int32))
    assert not ops.prefilter_fell_back(ws)
