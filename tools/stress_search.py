"""Randomised GPU-vs-oracle parity sweep of the search path (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
from oracle import oracle as orc
from conftest import assert_topk_equivalent

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t0 = time.time()
for c in range(ncases):
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 200000), rng.integers(200000, 600000)]))
    nq = int(rng.choice([1, rng.integers(1, 33), rng.integers(33, 130), rng.integers(130, 300)]))
    k = int(min(n, rng.choice([1, 5, 10, rng.integers(1, 65), rng.integers(65, 140)])))
    cosine = bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1 << 30))
    if n * nq > 60_000_000:
        nq = max(1, 60_000_000 // n)
    try:
        if cosine:
            db, lengths = syn.raw_database(n, seed)
            q, qlen = syn.raw_queries(nq, seed + 1)
            dup = rng.integers(0, n, size=min(n, 20)); db[dup] = db[rng.integers(0, n, size=len(dup))]
            mincov = float(rng.choice([0.0, 0.7]))
            use_mask = bool(rng.integers(0, 2))
            kw = dict(lengths=torch.from_numpy(lengths).cuda(), qlen=torch.from_numpy(qlen).cuda(), mincov=mincov) if use_mask else {}
            s, i = ops.ip_topk(torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda(), k, mode=ops.MODE_COSINE_RAW, **kw)
            sr, ir = orc.cosine_topk(db, q, k, lengths if use_mask else None, qlen if use_mask else None, mincov)
            assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), sr, ir, tol=2e-6)
            # the product's form: rows normalised once, MS_MODE_COSINE_UNIT
            unit = ops.l2_normalize_rows_(torch.from_numpy(db).cuda(), 1e-8)
            s, i = ops.ip_topk(unit, torch.from_numpy(q).cuda(), k, mode=ops.MODE_COSINE_UNIT, **kw)
            assert_topk_equivalent(s.cpu().numpy(), i.cpu().numpy(), sr, ir, tol=2e-6)
        else:
            db = syn.normalized_database(n, seed)
            q = syn.normalized_database(nq, seed + 1)
            dup = rng.integers(0, n, size=min(n, 50)); db[dup] = db[rng.integers(0, n, size=len(dup))]
            off = int(rng.integers(0, 1 << 33))
            s, i = ops.ip_topk(torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda(), k, row_offset=off)
            sr, ir = orc.ip_topk(db, q, k, row_offset=off, order=1)
            assert np.array_equal(i.cpu().numpy(), ir), "indices"
            assert np.array_equal(s.cpu().numpy().view(np.uint32), sr.view(np.uint32)), "score bits"
            # the prefiltered search (takes the plain path itself for the shapes it does not serve): same bits; with a few
            # clusters of near-duplicates of a query planted now and then, so that the gated exact pass runs too
            if rng.integers(0, 3) == 0 and n > 1000:
                rows = rng.integers(0, n, size=min(n, 200))
                v = q[0][None, :] + rng.normal(0, 3e-7, size=(len(rows), 128)).astype(np.float32)
                db[rows] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
                sr, ir = orc.ip_topk(db, q, k, row_offset=off, order=1)
            s, i = ops.ip_topk_prefiltered(torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda(), k, 1.0 + 1e-6, row_offset=off)
            assert np.array_equal(i.cpu().numpy(), ir), "prefiltered: indices"
            assert np.array_equal(s.cpu().numpy().view(np.uint32), sr.view(np.uint32)), "prefiltered: score bits"
    except AssertionError as e:
        bad += 1
        print("MISMATCH", dict(n=n, nq=nq, k=k, cosine=cosine, seed=seed), str(e)[:200])
print(f"{ncases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
