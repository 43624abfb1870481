"""GPU parity of the prefiltered search (ms_ip_topk_prefiltered): the rows are scanned with 16-bit matrix instructions -- over an
image built once per database (ms_pf_build_image: fp16 rows against split or unsplit fp16 queries, MS_PF_F16X2 / MS_PF_F16X1, or
bf16 hi + lo halves of both, MS_PF_BF16X3), or splitting in registers without one -- the survivors re-scored with the exact fp32
chain, and every query's answer is proved complete -- or that query gets an exact pass of its own.  The bar is the one of
ms_ip_topk, for every arithmetic: indices AND score bits identical to the oracle."""
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


IMAGES = ["f16x2", "f16x1", "bf16x3", None]           # the three image arithmetics, and no image (rows split in registers)
FORMATS = ["f16x2", "f16x1", "bf16x3"]


def _img(ops, d, image, bound=None):
    """image: a format name, True (the driver's default format) or None / False (no image)."""
    if not image:
        return None
    fmt = None if image is True else {"f16x2": ops.PF_F16X2, "f16x1": ops.PF_F16X1, "bf16x3": ops.PF_BF16X3}[image]
    return ops.pf_build_image(d, fmt=fmt, row_norm_bound=bound)


def _check(torch, ops, orc, db, q, k, bound, row_offset=0, raw=False, expect_fallback=None, image=True, expect_flagged=None):
    d, dq = _dev(torch, db), _dev(torch, q)
    ws = ops.PrefilterWorkspace(d.device).get(db.shape[0], q.shape[0], k)
    mode = ops.MODE_IP_NORMQ if raw else ops.MODE_IP_PRENORM
    img = _img(ops, d, image, bound)
    s, i = ops.ip_topk_prefiltered(d, dq, k, bound, mode=mode, row_offset=row_offset, workspace=ws, image=img)
    flagged = ops.prefilter_flagged(ws)
    fell_back = flagged > 0
    if expect_flagged is not None:       # (a family may also sit inside the candidate range of a query that does not own it: a few more)
        assert expect_flagged <= flagged <= expect_flagged + max(4, expect_flagged // 2), (flagged, expect_flagged)
    qn = ops.l2_normalize_rows(dq, 1e-12).cpu().numpy() if raw else q      # (the library's own F.normalize: ms_l2_normalize_rows_to)
    s_ref, i_ref = orc.ip_topk(db, qn, k, row_offset=row_offset, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    if expect_fallback is not None:
        assert fell_back == expect_fallback
    return fell_back


@pytest.mark.parametrize("image", IMAGES)
@pytest.mark.parametrize("n,nq,k", [(70_000, 65, 1), (70_000, 100, 5), (131_105, 97, 10), (300_000, 256, 10), (262_113, 130, 16),
                                    (400_000, 200, 20), (200_000, 129, 32), (300_000, 100, 40), (1_000_003, 256, 10), (100_000, 1000, 3),
                                    (65_536, 160, 10), (99_999, 161, 7), (250_000, 300, 48)])
def test_prefiltered_is_bit_identical_and_needs_no_exact_pass_on_ordinary_data(n, nq, k, image, torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    db, q = _norm_db(n, seed=401 + k), _norm_db(nq, seed=402)
    db[n - 1] = db[3]; db[n // 2] = db[3]                     # a few exact duplicates (ties resolve to the lowest row)
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, row_offset=11, expect_fallback=False, image=image)


@pytest.mark.parametrize("image", IMAGES)
def test_prefiltered_raw_queries_and_rows_that_are_not_unit_vectors(image, torch_gpu):
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
    _check(torch, ops, orc, db, (q_raw * 2.5).astype(np.float32), k, bound, raw=True, expect_fallback=False, image=image)
    q = (_norm_db(nq, seed=413) * 4.0).astype(np.float32)    # MS_MODE_IP_PRENORM with queries that are not unit vectors either
    _check(torch, ops, orc, db, q, k, bound, image=image)


@pytest.mark.parametrize("image", IMAGES)
def test_prefiltered_near_ties_by_the_hundred_fall_back_to_the_exact_pipeline(image, torch_gpu):
    """300 rows within 1e-6 of each other around every query's best score (copies of the query's own direction with tiny
    perturbations) and blocks of exact duplicates: no proof can succeed, every query gets the exact pass."""
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
    fell_back = _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, expect_fallback=True, image=image, expect_flagged=nq)
    assert fell_back


def _family(rng, qvec, count, sigma=2e-7):
    v = qvec[None, :] + rng.normal(0, sigma, size=(count, 128)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("image", FORMATS)
@pytest.mark.parametrize("which", ["one", "seventeen", "all", "tile_edges", "last_only"])
def test_queries_whose_proof_fails_get_an_exact_pass_of_their_own(which, image, torch_gpu):
    """Per-query exact fallback (the reference's semantics are per query: dbsearch.py:234-242): only the queries that own a family
    of 200 near-duplicates fail their proof; they are compacted on the device, scanned exactly and scattered back; every other
    query keeps the prefilter's (proved) answer.  All 256 answers == the oracle's, and the flagged count is exactly the planted one."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, nq, k = 200_000, 256, 10
    db, q = _norm_db(n, seed=491), _norm_db(nq, seed=492)
    owners = {"one": [77], "seventeen": list(range(5, 256, 15))[:17], "all": list(range(256)),
              "tile_edges": [0, 31, 32, 63, 64, 95, 96, 127, 128, 159, 160, 191, 192, 223, 224, 255], "last_only": [255]}[which]
    rng = np.random.default_rng(8)
    rows = rng.choice(n, size=(len(owners), 200), replace=False)
    for j, qi in enumerate(owners):
        db[rows[j]] = _family(rng, q[qi], 200)
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, row_offset=3, expect_flagged=len(owners), image=image)


def test_flagged_queries_in_a_large_batch_and_long_lists(torch_gpu):
    """1000 queries, k = 32 (64 candidates per query), 40 owners of 300-row families: compaction across several query groups."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, nq, k = 150_000, 1000, 32
    db, q = _norm_db(n, seed=493), _norm_db(nq, seed=494)
    owners = list(range(3, 1000, 25))
    rng = np.random.default_rng(9)
    rows = rng.choice(n, size=(len(owners), 300), replace=False)
    for j, qi in enumerate(owners):
        db[rows[j]] = _family(rng, q[qi], 300)
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, expect_flagged=len(owners))


def _cosine_case(n, nq, seed, masked_fraction=0.5):
    from merizo_search_amd.foldclass import synthetic as syn
    rng = np.random.default_rng(seed)
    db = (_norm_db(n, seed=seed) * rng.uniform(0.5, 2.0, size=(n, 1))).astype(np.float32)
    q, _ = syn.raw_queries(nq, seed=seed + 1)
    lengths = rng.integers(40, 400, size=n).astype(np.float32)
    qlen = rng.integers(40, 400, size=nq).astype(np.float32)
    return db, q.astype(np.float32), lengths, qlen


@pytest.mark.parametrize("n,nq,k,mincov", [(200_000, 256, 10, 0.7), (131_071, 100, 5, 0.0), (300_000, 1000, 10, 0.7), (100_000, 130, 20, 1.5),
                                           (70_000, 97, 1, 0.7)])
@pytest.mark.parametrize("image", FORMATS)
def test_prefiltered_cosine_on_unit_rows_equals_the_fp32_scan(n, nq, k, mincov, image, torch_gpu):
    """MS_MODE_COSINE_UNIT through the prefilter (search_query_against_db on rows normalised once, dbsearch.py:75-81): the length
    mask multiplies the approximate and the exact score by the same 0 / 1, so the proof holds; results == ms_ip_topk's bit for bit
    and == the oracle's cosine_topk (near-tie aware)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    from conftest import assert_topk_equivalent
    db, q, lengths, qlen = _cosine_case(n, nq, seed=500 + k)
    d, dq, dl, dql = _dev(torch, db), _dev(torch, q), _dev(torch, lengths), _dev(torch, qlen)
    unit = ops.l2_normalize_rows_(d.clone(), 1e-8)
    img = _img(ops, unit, image, 1.0 + 1e-5)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=dl, qlen=dql, mincov=mincov)
    s0, i0 = ops.ip_topk(unit, dq, k, row_offset=7, **kw)
    s1, i1 = ops.ip_topk_prefiltered(unit, dq, k, 1.0 + 1e-5, row_offset=7, workspace=ws, image=img, **kw)
    if mincov <= 1.0:           # (mincov 1.5 masks every row for the shortest queries: all-zero scores, no proof, exact pass)
        assert ops.prefilter_flagged(ws) == 0
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    sel = np.arange(0, nq, max(1, nq // 40))
    s_ref, i_ref = orc.cosine_topk(db, q[sel], k, lengths, qlen[sel], mincov, row_offset=7)
    assert_topk_equivalent(s1.cpu().numpy()[sel], i1.cpu().numpy()[sel], s_ref, i_ref, tol=2e-6)


@pytest.mark.parametrize("image", FORMATS)
def test_prefiltered_cosine_with_every_row_masked_and_with_families(image, torch_gpu):
    """Everything masked for some queries (all scores +-0: ties by the thousand, resolved by row; those queries fail their proof and
    get the exact pass) and near-duplicate families for others."""
    torch = torch_gpu
    from merizo_search_amd import ops
    n, nq, k = 120_000, 200, 10
    db, q, lengths, qlen = _cosine_case(n, nq, seed=520)
    qlen[:5] = 1.0                                             # shorter than every row * mincov: everything masked
    rng = np.random.default_rng(10)
    rows = rng.choice(n, size=(6, 150), replace=False)
    for j in range(6):
        db[rows[j]] = _family(rng, q[50 + j] / np.linalg.norm(q[50 + j]), 150) * 1.3
        lengths[rows[j]] = 50.0
        qlen[50 + j] = 300.0
    d, dq, dl, dql = _dev(torch, db), _dev(torch, q), _dev(torch, lengths), _dev(torch, qlen)
    unit = ops.l2_normalize_rows_(d.clone(), 1e-8)
    img = _img(ops, unit, image, 1.0 + 1e-5)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=dl, qlen=dql, mincov=0.7)
    s0, i0 = ops.ip_topk(unit, dq, k, **kw)
    s1, i1 = ops.ip_topk_prefiltered(unit, dq, k, 1.0 + 1e-5, workspace=ws, image=img, **kw)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    assert ops.prefilter_flagged(ws) >= 11


@pytest.mark.parametrize("n,nq,k", [(300_000, 40, 10), (300_000, 100, 49), (20_000, 100, 10), (300_000, 100, 100)])
def test_prefiltered_shapes_it_does_not_serve_take_the_plain_path(n, nq, k, torch_gpu):
    """<= 64 queries, k > 48, small databases: the call is ms_ip_topk (same results, of course)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    _check(torch, ops, orc, _norm_db(n, seed=431), _norm_db(nq, seed=432), k, 1.0 + 1e-6)


@pytest.mark.parametrize("image", IMAGES)
def test_prefiltered_stages_equal_the_one_shot_call_and_repeat(image, torch_gpu):
    torch = torch_gpu
    from merizo_search_amd import ops
    n, nq, k = 500_000, 256, 10
    db, q = _norm_db(n, seed=441), _norm_db(nq, seed=442)
    d, dq = _dev(torch, db), _dev(torch, q)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    img = _img(ops, d, image)
    s0, i0 = ops.ip_topk(d, dq, k)
    out = (torch.empty_like(s0), torch.empty_like(i0))
    for _ in range(3):
        ops.ip_topk_prefiltered_stage("prepare", d, dq, k, ws, image=img)
        ops.ip_topk_prefiltered_stage("scan", d, dq, k, ws, image=img)
        ops.ip_topk_prefiltered_stage("finish", d, dq, k, ws, out=out, image=img)
        assert torch.equal(out[1], i0) and torch.equal(out[0].view(torch.int32), s0.view(torch.int32))
        out[0].zero_(); out[1].zero_()
    ops.ip_topk_prefiltered_stage("prepare", d, dq, k, ws, image=img)
    for _ in range(2):                                         # the same workspace scanned twice after one prepare
        ops.ip_topk_prefiltered_stage("scan", d, dq, k, ws, image=img)
        ops.ip_topk_prefiltered_stage("finish", d, dq, k, ws, out=out, image=img)
        assert torch.equal(out[1], i0) and torch.equal(out[0].view(torch.int32), s0.view(torch.int32))


def test_engine_uses_the_prefilter_for_large_batches_on_a_resident_database(torch_gpu):
    """foldclass/engine.py: ip_topk with a row-norm bound (what dbsearch_faiss passes for a resident shard) == without."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import engine as eng
    e = eng.HipEngine("cuda:0")
    db, q = _norm_db(300_000, seed=451), _norm_db(200, seed=452)
    d, dq = _dev(torch, db), _dev(torch, (q * 3).astype(np.float32))
    bound = e.row_norm_bound(d)
    assert 1.0 <= bound < 1.0001
    img = e.pf_image(d)
    assert img is not None and img.format == ops.PF_F16X1 and 256 * d.shape[0] <= img.numel() <= 256 * (d.shape[0] + 64) + 256
    lazy = e.lazy_pf_image(d, bound)
    assert not lazy.built
    s1, i1 = e.ip_topk(d, dq[:40], 10, row_offset=5, normalize_queries=True, row_norm_bound=bound, pf_image=lazy)     # 40 queries: no image built
    assert not lazy.built
    s1, i1 = e.ip_topk(d, dq, 10, row_offset=5, normalize_queries=True, row_norm_bound=bound, pf_image=lazy)
    assert lazy.built and lazy.get() is not None
    s0, i0 = e.ip_topk(d, dq, 10, row_offset=5, normalize_queries=True)
    for image in (img, None):
        s1, i1 = e.ip_topk(d, dq, 10, row_offset=5, normalize_queries=True, row_norm_bound=bound, pf_image=image)
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
        parts.append(e.ip_topk(shard, dq, k, row_offset=lo, row_norm_bound=e.row_norm_bound(shard), pf_image=e.pf_image(shard)))
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


def test_engine_keeps_the_prefilter_on_clustered_data_batch_after_batch(torch_gpu):
    """foldclass/engine.py: near-duplicate families for a quarter of the queries; every batch flags exactly those and stays on the
    prefiltered path (no two-strikes switch any more: a clustered query costs only itself).  Results exact every time."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import engine as eng
    e = eng.HipEngine("cuda:0")
    n, nq, k = 150_000, 96, 10
    db, q = _norm_db(n, seed=481), _norm_db(nq, seed=482)
    rng = np.random.default_rng(7)
    owners = list(range(0, nq, 4))
    rows = rng.choice(n, size=(len(owners), 100), replace=False)
    for j, qi in enumerate(owners):
        db[rows[j]] = _family(rng, q[qi], 100)
    d, dq = _dev(torch, db), _dev(torch, q)
    bound, img = e.row_norm_bound(d), e.pf_image(d)
    s0, i0 = ops.ip_topk(d, dq, k)
    for call in range(4):
        s, i = e.ip_topk(d, dq, k, row_norm_bound=bound, pf_image=img)
        assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))
        assert len(owners) <= ops.prefilter_flagged(e._pws.buf) <= len(owners) + 4


@pytest.mark.parametrize("image", IMAGES)
@pytest.mark.parametrize("k", [10, 40])
def test_prefiltered_repeated_searches_return_identical_results(k, image, torch_gpu):
    """Run-to-run determinism of the prefiltered search (100 runs of one search; the first against the fp32 scan)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    n, nq = 202_000, 738
    d, dq = _dev(torch, _norm_db(n, seed=46)), _dev(torch, _norm_db(nq, seed=47))
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    img = _img(ops, d, image)
    s_ref, i_ref = ops.ip_topk(d, dq, k)
    for _ in range(100):
        s, i = ops.ip_topk_prefiltered(d, dq, k, 1.0 + 1e-6, workspace=ws, image=img)
        assert torch.equal(i, i_ref) and torch.equal(s.view(torch.int32), s_ref.view(torch.int32))
    assert not ops.prefilter_fell_back(ws)


def test_fp16_image_builder_declines_row_norms_outside_its_range_and_the_engine_falls_back_to_split_bf16(torch_gpu):
    """The fp16 image stores row * 2^sr: a row-norm bound outside [2^-40, 2^40] is refused (MS_ERR_RANGE); the engine then builds
    the split-bf16 image, whose arithmetic has the range of fp32.  Tiny rows (norm 1e-14) searched through it: exact."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd._lib import MerizoHipError
    from merizo_search_amd.foldclass import engine as eng
    from oracle import oracle as orc
    n, nq, k = 100_000, 100, 10
    db = (_norm_db(n, seed=531) * np.float32(1e-14)).astype(np.float32)
    q = _norm_db(nq, seed=532)
    d, dq = _dev(torch, db), _dev(torch, q)
    with pytest.raises(MerizoHipError, match="row-norm bound"):
        ops.pf_build_image(d, fmt=ops.PF_F16X2, row_norm_bound=1e-14)
    e = eng.HipEngine("cuda:0")
    bound = e.row_norm_bound(d)
    img = e.pf_image(d, bound)
    assert img is not None and img.format == ops.PF_BF16X3
    s, i = e.ip_topk(d, dq, k, row_norm_bound=bound, pf_image=img)
    s_ref, i_ref = orc.ip_topk(db, q, k, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref) and np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


@pytest.mark.parametrize("scale", [2.0 ** -30, 3.0e4, 7.5e8])
@pytest.mark.parametrize("image", ["f16x2", "f16x1"])
def test_fp16_image_scales_rows_and_queries_of_any_magnitude_into_range(image, scale, torch_gpu):
    """Rows of norm ~scale (and queries 1000x smaller / larger than that) go through the fp16 image: the image stores row * 2^sr, every
    query is scaled by its own power of two, the proof's bound scales with the row-norm bound.  Exact; nothing flagged."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, nq, k = 150_000, 130, 10
    db = (_norm_db(n, seed=541) * np.float32(scale)).astype(np.float32)
    q = _norm_db(nq, seed=542)
    q[: nq // 2] *= np.float32(1e-3 / scale)
    q[nq // 2:] *= np.float32(1e3 * scale)
    bound = float(np.linalg.norm(db.astype(np.float64), axis=1).max()) * (1 + 1e-6)
    _check(torch, ops, orc, db, q.astype(np.float32), k, bound, expect_fallback=False, image=image)


def test_fp16_image_of_the_wrong_database_traps_instead_of_answering(torch_gpu, tmp_path):
    """The fp16 image carries a trailer (magic, scale exponent, rows) the scan checks: a split-bf16 image passed as MS_PF_F16X2 makes
    the launch trap -- in a child process, because a trap poisons the HIP context."""
    import subprocess
    import sys
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys; sys.path.insert(0, %r)
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
d = syn.device_database(100000, 0, seed=0, device="cuda:0"); q = syn.device_database(100, 0, seed=1, device="cuda:0")
img = ops.pf_build_image(d, fmt=ops.PF_BF16X3)
bad = ops.PfImage(img.data, ops.PF_F16X2, img.n)         # (a 512 B/row buffer: large enough, wrong content)
try:
    s, i = ops.ip_topk_prefiltered(d, q, 10, 1.0 + 1e-6, image=bad)
    torch.cuda.synchronize()
    print("ANSWERED")
except Exception as exc:
    print("FAILED LOUDLY", type(exc).__name__)
""" % repo
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "ANSWERED" not in r.stdout, r.stdout + r.stderr[-2000:]
    assert "FAILED LOUDLY" in r.stdout or r.returncode != 0


def test_a_poisoned_compaction_ticket_fails_loudly_instead_of_leaving_unproven_answers(torch_gpu, tmp_path):
    """the re-scoring launch recognises its last workgroup by a ticket; a ticket counter that is not where the
    host expects it (an aborted launch, a second stream on the workspace -- here: ms_debug_prefilter_poison) used to leave gate[0] / the
    device plan unwritten, the gated exact pass returned at once and the flagged queries kept their UNPROVEN prefilter answer with rc 0.
    Round 6: the ticket counts on from call to call against a host-side total, and the gated launch traps when the re-scoring's verdict
    (gate[1] == this call's epoch) is missing.  Clustered data (every query owns a family of near-duplicates: all flagged), three
    poisonings -- ticket behind, ticket ahead, slot counter + ticket -- each in a child process (a trap poisons the HIP context); the
    un-poisoned control in the same child answers exactly."""
    import subprocess
    import sys
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys; sys.path.insert(0, %r)
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
n, nq, k = 100000, 96, 10
d = syn.device_database(n, 0, seed=0, device="cuda:0"); q = syn.device_database(nq, 0, seed=1, device="cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(5)
fam = q.cpu()[:, None, :] + torch.randn((nq, 40, 128), generator=g) * 2e-7          # 40 rows within ~1e-6 of every query: no proof possible
fam = fam / fam.norm(dim=2, keepdim=True)
d[torch.arange(nq * 40, device="cuda:0") * 17 + 3] = fam.reshape(-1, 128).cuda()
img = ops.pf_build_image(d, fmt=ops.PF_F16X1, row_norm_bound=1.0 + 1e-6)
ws = torch.empty_like(ops.PrefilterWorkspace("cuda:0").get(n, nq, k))
s0, i0 = ops.ip_topk(d, q, k)
s1, i1 = ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-6, image=img, workspace=ws)
torch.cuda.synchronize()
assert ops.prefilter_flagged(ws) == nq and torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
s1, i1 = ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-6, image=img, workspace=ws)          # (the ticket counts on: a second call is as good)
torch.cuda.synchronize()
assert torch.equal(i0, i1)
print("CONTROL OK", flush=True)
mode = sys.argv[1]
ops.prefilter_poison(ws, *{"behind": (0, 5), "ahead": (0, 2 * nq + 7), "both": (3, 1)}[mode])
try:
    s2, i2 = ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-6, image=img, workspace=ws)
    torch.cuda.synchronize()
    print("ANSWERED", bool(torch.equal(i0, i2)))
except Exception as exc:
    print("FAILED LOUDLY", type(exc).__name__)
""" % repo
    for mode in ("behind", "ahead", "both"):
        r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=600)
        assert "CONTROL OK" in r.stdout, r.stdout + r.stderr[-3000:]
        assert "ANSWERED" not in r.stdout, mode + ": " + r.stdout + r.stderr[-2000:]
        assert "FAILED LOUDLY" in r.stdout or r.returncode != 0, mode


def test_fp16_image_built_for_another_row_count_traps(torch_gpu):
    """the trailer's row count is checked by the scan too (an image of the same tile count built for other rows used to
    answer from the wrong rows): the C ABI called with the image of a 100,000-row database for a 99,990-row one traps.  Child process."""
    import subprocess
    import sys
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys; sys.path.insert(0, %r)
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
d = syn.device_database(100000, 0, seed=0, device="cuda:0"); q = syn.device_database(100, 0, seed=1, device="cuda:0")
img = ops.pf_build_image(d, fmt=ops.PF_F16X2, row_norm_bound=1.0 + 1e-6)
e = ops.pf_build_image(d[:0], fmt=ops.PF_F16X2, row_norm_bound=1.0)        # (ADVICE r05: an empty database builds a trailer-only image)
torch.cuda.synchronize()
assert e.data[:4].cpu().numpy().tobytes() == b"MF16"
small = d[:99990]
bad = ops.PfImage(img.data, ops.PF_F16X2, 99990)          # (the Python wrapper's own check passes: same tile count, same size)
try:
    s, i = ops.ip_topk_prefiltered(small, q, 10, 1.0 + 1e-6, image=bad)
    torch.cuda.synchronize()
    print("ANSWERED")
except Exception as exc:
    print("FAILED LOUDLY", type(exc).__name__)
""" % repo
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "ANSWERED" not in r.stdout, r.stdout + r.stderr[-2000:]
    assert "FAILED LOUDLY" in r.stdout or r.returncode != 0


def test_the_arithmetic_is_chosen_per_database_by_searching_its_own_rows(torch_gpu):
    """ops.pf_choose_format (what engine.pf_image runs once per resident database): i.i.d. rows -> one matrix instruction per 16
    dimensions (MS_PF_F16X1); a database made of families of rows within ~4e-4 of each other (cosine) -> two (MS_PF_F16X2), over the same
    image.  Either way the answers are the fp32 scan's."""
    torch = torch_gpu
    from merizo_search_amd import ops
    n, nq, k = 200_000, 128, 10
    plain = _dev(torch, _norm_db(n, seed=551))
    img = ops.pf_build_image(plain, fmt=ops.PF_F16X2, row_norm_bound=1.0 + 1e-6)
    assert ops.pf_choose_format(plain, img, 1.0 + 1e-6).format == ops.PF_F16X1
    rng = np.random.default_rng(11)
    centres = _norm_db(4000, seed=552)
    fam = centres[rng.integers(0, 4000, size=n)] + rng.normal(0, 2.5e-3, size=(n, 128)).astype(np.float32)      # 1 - cos ~ 4e-4 inside a family
    fam = (fam / np.linalg.norm(fam, axis=1, keepdims=True)).astype(np.float32)
    d = _dev(torch, fam)
    img = ops.pf_build_image(d, fmt=ops.PF_F16X2, row_norm_bound=1.0 + 1e-6)
    chosen = ops.pf_choose_format(d, img, 1.0 + 1e-6)
    assert chosen.format == ops.PF_F16X2 and chosen.data is img.data
    dq = _dev(torch, fam[rng.integers(0, n, size=nq)])
    s0, i0 = ops.ip_topk(d, dq, k)
    for image in (chosen, img.as_format(ops.PF_F16X1)):
        s1, i1 = ops.ip_topk_prefiltered(d, dq, k, 1.0 + 1e-6, image=image)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))


@pytest.mark.parametrize("image", ["f16x2", "f16x1"])
@pytest.mark.parametrize("nq,k", [(1, 10), (3, 1), (17, 10), (32, 20), (33, 5), (64, 10), (5, 40), (2, 48)])
def test_few_queries_take_the_fp16_image_scan_on_large_databases(nq, k, image, torch_gpu):
    """The reference's own CLI regime (one to a few query domains per call, dbsearch.py:531-546) is HBM-bound: from ms_pf_few_min_rows()
    rows on, ms_ip_topk_prefiltered serves ANY number of queries over an fp16 image -- 256 B per row instead of 512.  1.05M rows (the
    threshold is 1M): indices and score bits == the oracle's; nothing flagged on ordinary data; below the threshold, or over a
    split-bf16 image, the same call is ms_ip_topk (same answers)."""
    torch = torch_gpu
    from merizo_search_amd import ops, _lib
    from oracle import oracle as orc
    assert int(_lib.load().ms_pf_few_min_rows(1)) == 1_000_000 and int(_lib.load().ms_pf_few_min_rows(32)) == 1_000_000
    assert int(_lib.load().ms_pf_few_min_rows(33)) == 200_000 and int(_lib.load().ms_pf_few_min_rows(64)) == 200_000
    n = 1_050_000
    db, q = _norm_db(n, seed=561), _norm_db(nq, seed=562 + nq)
    assert ops.prefilter_serves(n, nq, k, {"f16x2": ops.PF_F16X2, "f16x1": ops.PF_F16X1}[image]) and not ops.prefilter_serves(n, nq, k)
    assert not ops.prefilter_serves(n, nq, k, ops.PF_BF16X3) and not ops.prefilter_serves(990_000 if nq <= 32 else 190_000, nq, k, ops.PF_F16X2)
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, row_offset=9, raw=(nq % 2 == 1), expect_fallback=False, image=image)


@pytest.mark.parametrize("image", ["f16x2", "f16x1"])
@pytest.mark.parametrize("nq,k,flag", [(1, 10, 1), (3, 10, 3), (8, 5, 4), (8, 16, 8), (9, 10, 9)])
def test_few_flagged_queries_merge_inside_the_gated_exact_scan(nq, k, flag, image, torch_gpu):
    """Round 6: for a handful of queries (<= 8: MS_PF_FUSE_EXACT_MAX_NQ) over the fp16 image of a large
    database the exact pass behind the prefilter is ONE gated launch -- its last workgroup merges the flagged queries' lists and scatters
    them into their output rows -- instead of a scan and a merge launch.  `flag` of the nq queries own a family of 60 rows within ~1e-6 of
    each other (no proof possible: exactly they take the exact pass); 9 queries: the two-launch form.  Indices and score bits == the oracle,
    raw and prepared queries."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n = 1_050_000
    rng = np.random.default_rng(900 + nq + k)
    db, q = _norm_db(n, seed=581), _norm_db(nq, seed=582 + nq)
    owners = list(range(0, nq, max(1, nq // flag)))[:flag]
    for j, o in enumerate(owners):
        fam = q[o][None, :] + rng.standard_normal((60, 128)).astype(np.float32) * 2e-7
        fam /= np.linalg.norm(fam, axis=1, keepdims=True)
        db[(np.arange(60) * 15_013 + 1_000 * j + 7) % n] = fam.astype(np.float32)
    for raw in (False, True):
        _check(torch, ops, orc, db, (q * np.float32(3.0)) if raw else q, k, 1.0 + 1e-6, row_offset=5, raw=raw, expect_flagged=len(owners), image=image)


def test_engine_builds_the_image_for_few_query_searches_from_the_third_call(torch_gpu):
    """foldclass/engine.py: a resident database of >= ms_pf_few_min_rows() rows searched with a handful of queries per call (the CLI's
    loop over query structures) gets its fp16 image at the third such search -- one pass over the rows -- and every later search
    reads half the bytes; results identical to the fp32 scan before and after."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import engine as eng, synthetic as syn
    e = eng.HipEngine("cuda:0")
    d = syn.device_database(1_200_000, 0, seed=571, device="cuda:0")
    dq = syn.device_database(4, 0, seed=572, device="cuda:0") * 2.0
    bound = e.row_norm_bound(d)
    lazy = e.lazy_pf_image(d, bound)
    s0, i0 = ops.ip_topk(d, dq, 10, mode=ops.MODE_IP_NORMQ)
    for call in range(5):
        s1, i1 = e.ip_topk(d, dq, 10, normalize_queries=True, row_norm_bound=bound, pf_image=lazy)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
        assert lazy.built == (call >= 2)
    assert lazy.get().format in (ops.PF_F16X1, ops.PF_F16X2)


@pytest.mark.parametrize("nq", [1, 6, 40])
def test_few_queries_cosine_with_length_mask_over_the_fp16_image(nq, torch_gpu):
    """search_query_against_db's shape -- ONE query (or a few) against a `.pt` database normalised once, cosine + length mask
    (dbsearch.py:75-81) -- over the fp16 image of 1.05M unit rows: == the fp32 scan bit for bit, both arithmetics."""
    torch = torch_gpu
    from merizo_search_amd import ops
    n, k = 1_050_000, 10
    db, q, lengths, qlen = _cosine_case(n, nq, seed=580 + nq)
    qlen[nq - 1] = 1.0                      # shorter than every row * mincov: everything masked for the last query (all scores +-0: no proof, exact pass)
    d, dq, dl, dql = _dev(torch, db), _dev(torch, q), _dev(torch, lengths), _dev(torch, qlen)
    unit = ops.l2_normalize_rows_(d.clone(), 1e-8)
    kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=dl, qlen=dql, mincov=0.7)
    s0, i0 = ops.ip_topk(unit, dq, k, row_offset=3, **kw)
    img = _img(ops, unit, "f16x2", 1.0 + 1e-5)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    for image in (img, img.as_format(ops.PF_F16X1)):
        s1, i1 = ops.ip_topk_prefiltered(unit, dq, k, 1.0 + 1e-5, row_offset=3, workspace=ws, image=image, **kw)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
        assert ops.prefilter_flagged(ws) >= 1


@pytest.mark.parametrize("image", ["f16x2", "f16x1"])
@pytest.mark.parametrize("nq,owners", [(1, [0]), (3, [1]), (20, [0, 7, 19]), (64, [5, 63])])
def test_few_queries_whose_proof_fails_get_the_exact_pass_too(nq, owners, image, torch_gpu):
    """The few-query plan of the image scan (one workgroup row per CU, <= 2 query tiles) with queries that own a family of 200
    near-duplicates: those are flagged, compacted and scanned exactly (ms_scan_kernel on the device plan for 1 .. 3 queries), the
    others keep the proved answer; all == the oracle."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, k = 1_050_000, 10
    db, q = _norm_db(n, seed=591), _norm_db(nq, seed=592 + nq)
    rng = np.random.default_rng(12)
    rows = rng.choice(n, size=(len(owners), 200), replace=False)
    for j, qi in enumerate(owners):
        db[rows[j]] = _family(rng, q[qi], 200)
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, row_offset=17, expect_flagged=len(owners), image=image)


@pytest.mark.parametrize("nq", [33, 48, 64])
def test_two_query_tiles_take_the_fp16_image_scan_from_200k_rows(nq, torch_gpu):
    """33..64 queries are two query tiles of fp32 matrix work in ms_ip_topk; over the fp16 image they are one pass at the inner-product
    rate: served from 200k rows (ms_pf_few_min_rows(nq)); 32 queries on the same database are not (HBM-bound: 1M rows).  == the oracle."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from oracle import oracle as orc
    n, k = 210_000, 10
    db, q = _norm_db(n, seed=601), _norm_db(nq, seed=602 + nq)
    assert ops.prefilter_serves(n, nq, k, ops.PF_F16X1) and not ops.prefilter_serves(n, 32, k, ops.PF_F16X1)
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, row_offset=5, expect_fallback=False, image="f16x1")
    _check(torch, ops, orc, db, q, k, 1.0 + 1e-6, row_offset=5, expect_fallback=False, image="f16x2", raw=True)
