"""BASELINE configs at their full sizes on the GPU (C1 real-size CLI run, C3 search half, C4 per-GPU
shard shape), plus the checkpoint branch and the library's environment switches.

At these sizes the CPU oracle checks a query sample; everything else is checked through
size-independent properties: planted neighbours recalled, returned scores re-computed from the
returned rows, lists sorted, split-invariance (two half shards merged == one scan, bit for bit),
and an independent brute force on the GPU (torch matmul + topk, near-tie aware comparison)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_topk_equivalent

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COS_TOL = 2e-6            # cosine scores vs reference arithmetic (north_star: 1e-5)


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    from merizo_search_amd import _lib
    _lib.require_gpu()
    return torch


def _torch_bruteforce(torch, db, q, k, chunk=1 << 19, scale=None, mask_fn=None):
    """Independent exact top-k: scores = q @ db^T in row chunks, running torch.topk.  fp32, library GEMM
    (its own summation order: compare near-tie aware)."""
    best_s = best_i = None
    for r0 in range(0, db.shape[0], chunk):
        blk = db[r0:r0 + chunk]
        s = q @ blk.T
        if scale is not None:
            s = s * scale[r0:r0 + chunk][None, :]
        if mask_fn is not None:
            s = s * mask_fn(r0, r0 + blk.shape[0])
        ts, ti = torch.topk(s, min(k, blk.shape[0]), dim=1)
        ti = ti + r0
        if best_s is None:
            best_s, best_i = ts, ti
        else:
            cs, ci = torch.cat([best_s, ts], 1), torch.cat([best_i, ti], 1)
            o = torch.topk(cs, k, dim=1)
            best_s, best_i = o.values, torch.gather(ci, 1, o.indices)
        del s
    return best_s, best_i


def test_c3_search_half_500k_raw_rows_1000_queries_cosine_mask(torch_gpu):
    """C3's search half: `.pt`-style database of 500,000 RAW rows, 1000 queries, cosine + length mask
    (dbsearch.py:75-81), k = 10."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    n, nq, k, mincov = 500_000, 1000, 10, 0.7
    raw, lengths = syn.raw_database(n, seed=300)
    rq, qlen = syn.raw_queries(nq, seed=301)
    planted = syn.plant_neighbours(raw, rq, 2, seed=302, normalize=False)
    lengths[planted.reshape(-1)] = 30.0                                   # never masked: qlen >= 25 >= 30 * 0.7
    d, dq = torch.from_numpy(raw).cuda(), torch.from_numpy(rq).cuda()
    dl, dql = torch.from_numpy(lengths).cuda(), torch.from_numpy(qlen).cuda()
    inv = ops.row_inv_norms(d)
    s, i = ops.ip_topk(d, dq, k, mode=ops.MODE_COSINE_RAW, inv_norm=inv, lengths=dl, qlen=dql, mincov=mincov)
    s_h, i_h = s.cpu().numpy(), i.cpu().numpy()
    # oracle (reference arithmetic) on a query sample
    sample = np.r_[0:24, nq - 8:nq]
    s_ref, i_ref = orc.cosine_topk(raw, rq[sample], k, lengths, qlen[sample], mincov)
    assert_topk_equivalent(s_h[sample], i_h[sample], s_ref, i_ref, tol=COS_TOL)
    # all queries: planted rows on top, sorted, scores re-computed from the returned rows in float64
    assert all(set(planted[j]) <= set(i_h[j, :4]) for j in range(nq))
    assert (np.diff(s_h, axis=1) <= 0).all()
    rows = raw[i_h].astype(np.float64)
    qq = rq.astype(np.float64)
    cos = np.einsum("qkd,qd->qk", rows, qq) / np.linalg.norm(rows, axis=2) / np.linalg.norm(qq, axis=1)[:, None]
    cos *= (qlen[:, None] >= lengths[i_h] * np.float32(mincov))
    assert np.abs(cos - s_h).max() <= COS_TOL
    # independent GPU brute force over all 1000 queries
    qn = dq / dq.norm(dim=1, keepdim=True)
    bs, bi = _torch_bruteforce(torch, d, qn, k, scale=inv,
                               mask_fn=lambda a, b: (dql[:, None] >= dl[a:b][None, :] * mincov).float())
    assert_topk_equivalent(s_h, i_h, bs.cpu().numpy(), bi.cpu().numpy(), tol=COS_TOL)
    # split invariance: two shards + merge == one scan
    h = n // 2 + 13
    parts = [ops.ip_topk(d[a:b], dq, k, mode=ops.MODE_COSINE_RAW, inv_norm=inv[a:b], lengths=dl[a:b], qlen=dql, mincov=mincov,
                         row_offset=a) for a, b in ((0, h), (h, n))]
    ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, i) and torch.equal(ms, s)
    # the product's path for this batch (foldclass/engine.py): rows normalised once, then the prefiltered search over their split
    # image in MS_MODE_COSINE_UNIT == the fp32 scan on the same unit rows bit for bit, and == the reference arithmetic (oracle sample)
    unit = ops.l2_normalize_rows_(d.clone(), 1e-8)
    kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=dl, qlen=dql, mincov=mincov)
    su, iu = ops.ip_topk(unit, dq, k, **kw)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    sp, ip_ = ops.ip_topk_prefiltered(unit, dq, k, 1.0 + 1e-5, workspace=ws, image=ops.pf_build_image(unit), **kw)
    assert torch.equal(ip_, iu) and torch.equal(sp.view(torch.int32), su.view(torch.int32))
    assert_topk_equivalent(sp.cpu().numpy()[sample], ip_.cpu().numpy()[sample], s_ref, i_ref, tol=COS_TOL)


def test_c4_per_gpu_shard_shape_45_6M_rows_4096_queries(torch_gpu):
    """C4's per-GPU shard: 45,625,000 x 128 unit rows (23.4 GB, generated on the device) x 4096 queries,
    k = 10 -- the shape every rank scans when the 365M-row TED database is sharded over 8 GPUs."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    from oracle import oracle as orc
    n, nq, k, lo = 45_625_000, 4096, 10, 3 * 45_625_000                  # rank 3's rows of the 365M-row matrix
    dev = torch.device("cuda", 0)
    db = syn.device_database(n, lo, seed=0, device=dev)
    q = syn.device_database(nq, 0, seed=1, device=dev)
    gen = torch.Generator(device="cpu"); gen.manual_seed(2)
    rows = torch.randperm(n, generator=gen)[: nq * 3].reshape(nq, 3)
    near = q.cpu()[:, None, :] + torch.randn((nq, 3, 128), generator=gen) * 0.02
    near = near / near.norm(dim=2, keepdim=True)
    db[rows.reshape(-1).to(dev)] = near.reshape(-1, 128).to(dev)
    s, i = ops.ip_topk(db, q, k, row_offset=lo)
    torch.cuda.synchronize()
    i_h, s_h = i.cpu().numpy(), s.cpu().numpy()
    # planted neighbours (3 per query) lead every list; lists sorted; rows inside the shard
    assert all(set((rows[j] + lo).tolist()) == set(i_h[j, :3].tolist()) for j in range(nq))
    assert (np.diff(s_h, axis=1) <= 0).all() and i_h.min() >= lo and i_h.max() < lo + n
    # scores re-computed from the returned rows: bit-exact against the oracle's k-order dot product
    got_rows = db[(i - lo).reshape(-1)].cpu().numpy().reshape(nq, k, 128)
    sample = np.r_[0:16, nq - 16:nq]
    for j in sample:
        s_ref, i_ref = orc.ip_topk(got_rows[j], q[j:j + 1].cpu().numpy(), k, order=1)
        assert np.array_equal(np.sort(s_ref[0].view(np.uint32)), np.sort(s_h[j].view(np.uint32)))
    # independent brute force over the whole shard, all 4096 queries (recall@k = 1 up to near-ties)
    bs, bi = _torch_bruteforce(torch, db, q, k)
    assert_topk_equivalent(s_h, i_h, bs.cpu().numpy(), (bi + lo).cpu().numpy(), tol=2e-6)
    del bs, bi
    # split invariance: two half shards + ms_topk_merge == one scan, bit for bit
    h = n // 2 + 7
    parts = [ops.ip_topk(db[a:b], q, k, row_offset=lo + a) for a, b in ((0, h), (h, n))]
    ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, i) and torch.equal(ms, s)
    # HBM-bound regime on the same shard (one query tile): same rows as the batch's first queries
    s1, i1 = ops.ip_topk(db, q[:1], k, row_offset=lo)
    s32, i32 = ops.ip_topk(db, q[:32], k, row_offset=lo)
    assert torch.equal(i1, i[:1]) and torch.equal(s1, s[:1]) and torch.equal(i32, i[:32]) and torch.equal(s32, s[:32])
    # the DEFAULT path of this shape in the driver: the prefiltered search over the split image of the shard (built once, +23.4 GB),
    # and without an image (rows split in registers) -- indices and score bits of all 4096 lists == the fp32 scan's
    del parts, ms, mi
    ws = ops.PrefilterWorkspace(dev).get(n, nq, k)
    img = ops.pf_build_image(db, fmt=ops.PF_F16X2, row_norm_bound=1.0 + 1e-6)        # fp16 image: +11.7 GB; F16X1 runs over the same image
    assert img.numel() == (n + 63) // 64 * 16384 + 256                                  # 256 B per row (+ the trailer)
    for image in (img, img.as_format(ops.PF_F16X1), None):
        sp, ip_ = ops.ip_topk_prefiltered(db, q, k, 1.0 + 1e-6, row_offset=lo, workspace=ws, image=image)
        assert torch.equal(ip_, i) and torch.equal(sp.view(torch.int32), s.view(torch.int32))
        assert ops.prefilter_flagged(ws) == 0
    del img
    img = ops.pf_build_image(db, fmt=ops.PF_BF16X3)                                   # split-bf16 image: +23.4 GB
    sp, ip_ = ops.ip_topk_prefiltered(db, q, k, 1.0 + 1e-6, row_offset=lo, workspace=ws, image=img)
    assert torch.equal(ip_, i) and torch.equal(sp.view(torch.int32), s.view(torch.int32)) and ops.prefilter_flagged(ws) == 0

def test_c4_at_its_real_row_count_on_one_gpu_unsharded_equals_eight_shards_merged(torch_gpu):
    """C4 at the size BASELINE.json states it: 365,000,000 x 128 fp32 rows (186.9 GB) resident on ONE 288 GB MI355X, 4096 queries,
    k = 10.  The reference walks the whole database block by block and merges (dbsearch.py:233-243); here ONE unsharded scan of all
    365M rows must equal -- indices and score bits -- the eight contiguous shards of sharded.shard_bounds scanned one after the other
    (views of the same tensor, global row = shard offset + local row) and merged by ms_topk_merge_strided, which is exactly what the
    eight ranks of the sharded run compute.  Also: planted rows on top, lists sorted, the oracle on a query sample over the returned
    rows, an independent brute force for 64 queries over every row, 1 / 32 queries (HBM-bound regime), and the prefiltered search
    (rows split in registers: the split image of 365M rows would not fit next to them)."""
    torch = torch_gpu
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import sharded, synthetic as syn
    from oracle import oracle as orc
    n, nq, k, S = 365_000_000, 4096, 10, 8
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(dev)
    if free < 200 << 30:
        pytest.skip("C4 at full size needs 200 GB of free HBM on one GPU (186.9 GB of rows + workspaces); %.0f GB free of %.0f" % (free / 2**30, total / 2**30))
    db = syn.device_database(n, 0, seed=0, device=dev)
    assert db.shape == (n, 128) and db.numel() * 4 == 186_880_000_000
    q = syn.device_database(nq, 0, seed=1, device=dev)
    gen = torch.Generator(device="cpu"); gen.manual_seed(2)
    rows = ((torch.arange(nq * 3, dtype=torch.int64) * 2_147_483_629 + 12_345) % n).reshape(nq, 3)      # distinct (a bijection modulo n)
    assert len(set(rows.reshape(-1).tolist())) == nq * 3
    near = q.cpu()[:, None, :] + torch.randn((nq, 3, 128), generator=gen) * 0.02
    near = near / near.norm(dim=2, keepdim=True)
    db[rows.reshape(-1).to(dev)] = near.reshape(-1, 128).to(dev)
    # ONE scan over all 365M rows
    s, i = ops.ip_topk(db, q, k)
    torch.cuda.synchronize()
    i_h, s_h = i.cpu().numpy(), s.cpu().numpy()
    assert all(set(rows[j].tolist()) == set(i_h[j, :3].tolist()) for j in range(nq))
    assert (np.diff(s_h, axis=1) <= 0).all() and i_h.min() >= 0 and i_h.max() < n
    assert (i_h > 2**31 - 1).sum() == 0 and (i_h > 300_000_000).any()               # rows of the last shards are found too
    # eight sequential shard scans (views: no copy) + the strided merge of the gathered blocks, as the eight ranks do it
    # (each shard's lists are written straight into its packed block [scores | rows], the blocks lie back to back as the all-gather
    #  leaves them, and ms_topk_merge_strided reads them in place: sharded.PackedExchange's layout)
    bounds = [sharded.shard_bounds(n, S, r) for r in range(S)]
    assert bounds[0] == (0, 45_625_000) and bounds[-1][1] == n
    idx_off = (4 * nq * k + 7) // 8 * 8
    gathered = torch.zeros((S, idx_off + 8 * nq * k), dtype=torch.uint8, device=dev)
    for r, (lo, hi) in enumerate(bounds):
        out = (gathered[r, : 4 * nq * k].view(torch.float32).reshape(nq, k), gathered[r, idx_off:].view(torch.int64).reshape(nq, k))
        ops.ip_topk(db[lo:hi], q, k, row_offset=lo, out=out)
    ms, mi = torch.empty_like(s), torch.empty_like(i)
    ops.topk_merge_packed(gathered, S, nq, k, idx_off, ms, mi)
    assert torch.equal(mi, i) and torch.equal(ms.view(torch.int32), s.view(torch.int32))
    del gathered, ms, mi
    # scores re-computed from the returned rows: bit-exact against the oracle's k-order dot product (query sample)
    sample = np.r_[0:8, nq - 8:nq]
    got_rows = db[i[sample].reshape(-1)].cpu().numpy().reshape(len(sample), k, 128)
    qh = q.cpu().numpy()
    for a, j in enumerate(sample):
        s_ref, _ = orc.ip_topk(got_rows[a], qh[j:j + 1], k, order=1)
        assert np.array_equal(np.sort(s_ref[0].view(np.uint32)), np.sort(s_h[j].view(np.uint32)))
    # independent brute force over EVERY row for 64 queries (library GEMM, its own summation order: near-tie aware)
    pick = torch.arange(0, nq, nq // 64, device=dev)[:64]
    bs, bi = _torch_bruteforce(torch, db, q[pick], k, chunk=1 << 22)
    assert_topk_equivalent(s_h[pick.cpu().numpy()], i_h[pick.cpu().numpy()], bs.cpu().numpy(), bi.cpu().numpy(), tol=2e-6)
    del bs, bi
    # the reference's own regime (a few queries per call): one full pass over 186.9 GB per call
    for few in (1, 32):
        sf, jf = ops.ip_topk(db, q[:few], k)
        assert torch.equal(jf, i[:few]) and torch.equal(sf.view(torch.int32), s[:few].view(torch.int32))
    # prefiltered search over the fp32 rows (no image), all 4096 lists == the fp32 scan's
    ws = ops.PrefilterWorkspace(dev).get(n, nq, k)
    sp, ip_ = ops.ip_topk_prefiltered(db, q, k, 1.0 + 1e-6, workspace=ws, image=None)
    assert torch.equal(ip_, i) and torch.equal(sp.view(torch.int32), s.view(torch.int32))
    del db, ws
    torch.cuda.empty_cache()


def test_c1_real_size_cli_search_on_gpu(tmp_path, golden_dir):
    """C1 at its real size through the CLI on the HIP engine: M0 against the shipped TED example layout
    (66,943 entries), three neighbours planted at the first / last / a middle row."""
    import c1_case
    from merizo_search_amd.foldclass.network import network_setup
    net, _ = network_setup(device="cuda", allow_synthetic=True)
    p = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    e = net.embed_many([p["coords"]]).cpu().numpy()[0]
    e = (e / np.linalg.norm(e)).astype(np.float32)
    rng = np.random.default_rng(1)
    planted = {}
    for j, row in enumerate((66942, 0, 31337)):
        v = e + 0.02 * (j + 1) * rng.standard_normal(128).astype(np.float32) / np.sqrt(128)
        planted[row] = (v / np.linalg.norm(v)).astype(np.float32)
    prefix = c1_case.build(str(tmp_path / "db"), plant=planted)
    env = dict(os.environ, MERIZO_ALLOW_SYNTHETIC_WEIGHTS="1", PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-m", "merizo_search_amd.cli", "search", os.path.join(golden_dir, "M0_ca.pdb"), prefix,
                        str(tmp_path / "out"), str(tmp_path / "tmp"), "-d", "cuda", "-k", "5", "-s", "-1", "--output_headers",
                        "--format", "query,emb_rank,target,emb_score,q_len,t_len,metadata"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    c1_case.check_search(str(tmp_path / "out"), [66942, 0, 31337], 5)


def test_blockwise_streaming_equals_single_launch(torch_gpu):
    """knn_exact over host blocks (pinned double-buffered upload, per-block scan, running merge: the
    reference's db_iterator loop) == one scan of the resident matrix, bit for bit, for block sizes that
    do and do not divide the row count."""
    torch = torch_gpu
    from merizo_search_amd.foldclass import dbsearch as ds, dbutil, synthetic as syn
    from merizo_search_amd.foldclass.engine import HipEngine
    eng = HipEngine("cuda:0")
    n, nq, k = 300_007, 130, 10
    db = syn.normalized_database(n, seed=71)
    db[n - 1] = db[5]; db[150_000] = db[149_999]
    q = syn.normalized_database(nq, seed=72)
    whole = eng.upload_rows(db, 0, n)
    assert torch.equal(whole.cpu(), torch.from_numpy(db))
    D0, I0 = ds.knn_exact(q, [whole], k, eng, row_offset=11)
    for bs in (262_144, 100_000, 4_099):
        D, I = ds.knn_exact(q, dbutil.db_iterator(db, bs), k, eng, row_offset=11)
        assert np.array_equal(I, I0) and np.array_equal(D.view(np.uint32), D0.view(np.uint32)), bs


def test_checkpoint_weights_embed_like_in_memory_weights(tmp_path, torch_gpu):
    torch = torch_gpu
    from merizo_search_amd.foldclass import network as nw, weights as W, synthetic as syn
    from merizo_search_amd.foldclass.engine import HipEngine
    sd = W.synthetic_state_dict(5)
    path = str(tmp_path / nw.WEIGHTS_NAME)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, path)
    net, _ = nw.network_setup(device="cuda", weights_path=path)
    direct = nw.FoldClassEncoder(HipEngine("cuda:0", state_dict=sd))
    coords = [syn.random_walk(n, seed=n) for n in (33, 120)]
    assert torch.equal(net.embed_many(coords), direct.embed_many(coords))


_ENV_WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
from oracle import oracle as orc
for n, nq, k in ((300_000, 100, 10), (300_000, 4, 10), (300_000, 20, 10), (40_000, 256, 33), (1_200_000, 130, 10)):
    db = syn.normalized_database(n, seed=5); q = syn.normalized_database(nq, seed=6)
    db[n - 1] = db[3]
    s, i = ops.ip_topk(torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda(), k)
    s_ref, i_ref = orc.ip_topk(db, q, k, order=1)
    assert np.array_equal(i.cpu().numpy(), i_ref) and np.array_equal(s.cpu().numpy().view(np.uint32), s_ref.view(np.uint32)), (n, nq, k)
print("env ok")
'''


@pytest.mark.parametrize("env", [{"MS_LOADER_WAVE": "0"}, {"MS_HEAD_MERGE": "0"}, {"MS_SAMPLE_MIN_NQ": "1"},
                                 {"MS_PREPASS_TILES": "0"}, {"MS_PREPASS_TILES": "5", "MS_SAMPLE_MIN_NQ": "200"}])
def test_library_environment_switches_keep_results_exact(env, tmp_path):
    """Non-default values of the library's diagnostic switches (read once per process, hence child
    processes): alternative kernel forms, same bits."""
    script = tmp_path / "w.py"
    script.write_text(_ENV_WORKER)
    r = subprocess.run([sys.executable, str(script), REPO], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "env ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
