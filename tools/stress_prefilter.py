"""Randomised parity sweep of the prefiltered search against the fp32 scan on the GPU (both through the C ABI; the fp32 scan is
the one pinned to the oracle): larger shapes than the oracle sweep can afford, plain / clustered / rescaled databases, inner
product and cosine-on-unit-rows with a length mask, over an image in each of the three arithmetics (MS_PF_F16X2 / F16X1 / BF16X3) and
without one.  Also measures the approximation itself: max |a - s| / (|q| * row-norm bound) over every candidate the scan kept (a = its
approximate score from the workspace, s = the float64 dot product) against the E the proof uses (ms_pf_err_coef) -- the empirical check
of the error budget in csrc/ms_scan_pf16.h.
usage: python tools/stress_prefilter.py SEED CASES"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
from merizo_search_amd import ops, _lib
FMT = {"f16x2": ops.PF_F16X2, "f16x1": ops.PF_F16X1, "bf16x3": ops.PF_BF16X3}
worst = {name: 0.0 for name in FMT}          # max |a - s| / (E-free scale) per arithmetic
over = 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = fell = flagged_total = queries_total = 0
t0 = time.time()
g = torch.Generator(device="cuda")
for c in range(ncases):
    n = int(rng.choice([rng.integers(65_536, 200_000), rng.integers(200_000, 1_000_000), rng.integers(1_000_000, 3_000_000)]))
    nq = int(rng.choice([rng.integers(65, 129), rng.integers(129, 600), rng.integers(600, 1500)]))
    k = int(rng.choice([1, 5, 10, 16, 24, 32, 48]))
    kind = int(rng.integers(0, 4))
    g.manual_seed(int(rng.integers(0, 1 << 30)))
    db = torch.randn((n, 128), generator=g, device="cuda")
    q = torch.randn((nq, 128), generator=g, device="cuda")
    if kind == 1:        # families: 2,000 centres, every row a centre plus a little noise (many rows within 1e-3 of each other)
        centres = torch.randn((2000, 128), generator=g, device="cuda")
        db = centres[torch.randint(0, 2000, (n,), generator=g, device="cuda")] + 0.002 * db
        q = centres[torch.randint(0, 2000, (nq,), generator=g, device="cuda")] + 0.05 * q
    if kind == 2:        # exact duplicates of a few queries scattered through the database
        idx = torch.randint(0, n, (200,), generator=g, device="cuda")
        db[idx] = q[torch.randint(0, min(nq, 8), (200,), generator=g, device="cuda")]
    db = db / db.norm(dim=1, keepdim=True)
    scale = 1.0
    if kind == 3:        # rows that are not unit vectors
        db = db * (0.1 + 4.0 * torch.rand((n, 1), generator=g, device="cuda"))
    bound = float(1.0 / ops.row_inv_norms(db, 1e-30).min()) * (1 + 1e-6)
    raw = bool(rng.integers(0, 2))
    mode = ops.MODE_IP_NORMQ if raw else ops.MODE_IP_PRENORM
    if not raw:
        q = q / q.norm(dim=1, keepdim=True)
    off = int(rng.integers(0, 1 << 33))
    use_image = bool(rng.integers(0, 4))                       # three in four over an image
    fmt_name = str(rng.choice(list(FMT)))
    kw = {}
    if use_image and kind != 3 and rng.integers(0, 3) == 0:    # cosine on unit rows + length mask (needs the image)
        mode, bound = ops.MODE_COSINE_UNIT, 1.0 + 1e-5
        db = ops.l2_normalize_rows_(db, 1e-8)
        kw = dict(lengths=torch.randint(40, 400, (n,), generator=g, device="cuda").float(),
                  qlen=torch.randint(40, 400, (nq,), generator=g, device="cuda").float(), mincov=float(rng.choice([0.0, 0.7])))
    if kind == 3 and rng.integers(0, 2):                       # (fp16 formats: rows and queries far from 1 in magnitude)
        sc = float(2.0 ** rng.integers(-25, 25))
        db = db * sc; bound *= sc
        if not raw:
            q = q * float(2.0 ** rng.integers(-25, 25))
    if os.environ.get("MS_STRESS_VERBOSE"):
        print("case", c, dict(n=n, nq=nq, k=k, kind=kind, raw=raw, mode=mode, image=fmt_name if use_image else None, kw=sorted(kw)), flush=True)
    img = ops.pf_build_image(db, fmt=FMT[fmt_name], row_norm_bound=bound) if use_image else None
    s0, i0 = ops.ip_topk(db, q, k, mode=mode, row_offset=off, **kw)
    ws = ops.PrefilterWorkspace(db.device).get(n, nq, k)
    s1, i1 = ops.ip_topk_prefiltered(db, q, k, bound, mode=mode, row_offset=off, workspace=ws, image=img, **kw)
    fl = ops.prefilter_flagged(ws)
    if os.environ.get("MS_STRESS_VERBOSE"):
        print("   ok, flagged", fl, flush=True)
    if use_image and mode != ops.MODE_COSINE_UNIT and ops.prefilter_serves(n, nq, k):
        # the approximation itself: the candidate lists the scan left in the workspace against float64
        a_s = np.zeros((nq, 64), np.float32); a_i = np.zeros((nq, 64), np.int64); kp = ctypes.c_int(0)
        rc = _lib.load().ms_debug_prefilter_lists(ws.data_ptr(), n, nq, k, 2 if fmt_name != "bf16x3" else 1, a_s.ctypes.data, a_i.ctypes.data, ctypes.byref(kp))
        if rc == 0:
            kp = kp.value
            a_s = a_s.reshape(-1)[:nq * kp].reshape(nq, kp); a_i = a_i.reshape(-1)[:nq * kp].reshape(nq, kp)
            qn = (ops.l2_normalize_rows(q, 1e-12) if raw else q).double()
            rows_ = db[torch.from_numpy(a_i.clip(0)).cuda().reshape(-1)].reshape(nq, kp, 128).double()
            ex = (rows_ * qn[:, None, :]).sum(2).cpu().numpy()
            qnorm = qn.norm(dim=1).cpu().numpy()[:, None]
            ok = (a_i >= 0) & (qnorm > 0)
            ratio = float((np.abs(ex - a_s) / np.maximum(qnorm * bound, 1e-300))[ok].max()) if ok.any() else 0.0
            worst[fmt_name] = max(worst[fmt_name], ratio)
            if ratio > ops.pf_err_coef(FMT[fmt_name]):
                over += 1
                print("ERROR BOUND EXCEEDED", fmt_name, ratio, dict(n=n, nq=nq, k=k, kind=kind, raw=raw))
    fell += int(fl > 0); flagged_total += fl; queries_total += nq
    if not (torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))):
        bad += 1
        print("MISMATCH", dict(n=n, nq=nq, k=k, kind=kind, raw=raw, mode=mode, image=fmt_name if use_image else None, flagged=fl))
    del db, q, ws, img
print(f"{ncases} cases, {bad} mismatches, exact pass needed in {fell} cases for {flagged_total} of {queries_total} queries, {time.time() - t0:.1f} s")
for name in FMT:
    print(f"  {name}: max |a - s| / (|q| * bound) over all candidates = {worst[name]:.3e}  (E = {ops.pf_err_coef(FMT[name]):.3e})")
sys.exit(1 if (bad or over) else 0)
