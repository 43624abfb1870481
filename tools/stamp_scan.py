"""Diagnostic (stamp builds only, -DMS_STAMP): cycles per tile and shader clock of the scan's compute waves."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn

lib = _lib.load()
lib.ms_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
PF = os.environ.get("STAMP_PF") == "1"       # the prefiltered search's scan instead of the fp32 one
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1_000_000, 256, 10), (4_000_000, 256, 10)]
for n, nq, k in cases:
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
    qq = torch.randn(nq, 128, device="cuda"); qq = qq / qq.norm(dim=1, keepdim=True)
    ws = (ops.PrefilterWorkspace if PF else ops.TopKWorkspace)(d.device).get(n, nq, k)
    out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
    if PF:
        prep = lambda: ops.ip_topk_prefiltered_stage("prepare", d, qq, k, ws)
        scan = lambda: ops.ip_topk_prefiltered_stage("scan", d, qq, k, ws)
        fin = lambda: ops.ip_topk_prefiltered_stage("finish", d, qq, k, ws, out=(out_s, out_i))
    else:
        prep = lambda: ops.ip_topk_prepare(d, qq, k, ws)
        scan = lambda: ops.ip_topk_scan(d, qq, k, ws)
        fin = lambda: ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
    for _ in range(30):      # warm clocks
        prep(); scan(); fin()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    prep(); e0.record(); scan(); e1.record(); torch.cuda.synchronize()
    words = 8 * 8 * 4096
    buf = np.zeros(words, dtype=np.uint64)
    assert lib.ms_debug_stamps(buf.ctypes.data, words) == 0
    allw = buf.reshape(-1, 8, 8)
    ld = allw[:, 4, :]; ld = ld[ld[:, 3] > 0].astype(np.float64)
    if len(ld):
        print(f"   loader cycles/tile: poll {np.median(ld[:,0]/ld[:,3]):.0f} issue {np.median(ld[:,1]/ld[:,3]):.0f} vmcnt+publish {np.median(ld[:,2]/ld[:,3]):.0f}")
    act = allw[:, 0, 2] > 0
    for w in range(4):
        ww = allw[act, w, :]
        cw = (ww[:, 3] & np.uint64((1 << 40) - 1)).astype(np.float64) / np.maximum(ww[:, 2].astype(np.float64), 1)
        nw = (ww[:, 3] >> np.uint64(40)).astype(np.float64)
        print(f"   wave {w}: wait cycles/tile median {np.median(cw):.0f}, misses median {np.median(nw):.0f}")
    st = allw[:, :4, :].reshape(-1, 8); st = st[st[:, 2] > 0]
    ni = st[:, 5].astype(np.float64); ci = st[:, 4].astype(np.float64)
    print(f"   insert path: taken in {100*ni.sum()/st[:,2].astype(np.float64).sum():.1f}% of tiles, {ci.sum()/max(ni.sum(),1):.0f} cycles per visit")
    if os.environ.get("STAMP_FLUSH") == "1":      # -DMS_STAMP_FLUSH builds: words 6, 7 = cycles in flushes, (flushes << 32) | rounds
        fl = st[:, 6].astype(np.float64); nf = (st[:, 7] >> np.uint64(32)).astype(np.float64); nr = (st[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.float64)
        print(f"   flushes per wave median {np.median(nf):.0f} max {nf.max():.0f}; rounds per flush {nr.sum()/max(nf.sum(),1):.1f}; cycles per round {fl.sum()/max(nr.sum(),1):.0f}; "
              f"flush cycles per wave median {np.median(fl):.0f} of {np.median(ci):.0f} in the rare path")
        st[:, 6] = 0; st[:, 7] = 0
    cyc, rt, nt = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64), st[:, 2].astype(np.float64)
    ghz = cyc / rt * 0.1
    c1, r1 = st[:, 6].astype(np.float64), st[:, 7].astype(np.float64)
    ok = (r1 > 0) & (rt > r1)
    if ok.any():
        print(f"   clock first half {np.median(c1[ok] / r1[ok]) * 0.1:.3f} GHz, second half {np.median((cyc[ok] - c1[ok]) / (rt[ok] - r1[ok])) * 0.1:.3f} GHz; "
              f"first half takes {np.median(r1[ok]) / 100:.1f} us, second {np.median(rt[ok] - r1[ok]) / 100:.1f} us")
    nwait = (st[:, 3] >> np.uint64(40)).astype(np.float64); cwait = (st[:, 3] & np.uint64((1 << 40) - 1)).astype(np.float64)
    print(f"   flag misses/wave median {np.median(nwait):.0f} max {nwait.max():.0f}; wait cycles/tile median {np.median(cwait/nt):.0f} max {np.max(cwait/nt):.0f}")
    print(f"n={n} nq={nq}: scan kernel {e0.elapsed_time(e1)*1e3:.1f} us | waves {len(st)} tiles/wave {nt.mean():.0f} | "
          f"cycles/tile median {np.median(cyc/nt):.0f} max {np.max(cyc/nt):.0f} min {np.min(cyc/nt):.0f} | clock {np.median(ghz):.3f} GHz | "
          f"wave time max {np.max(rt)/100:.1f} us median {np.median(rt)/100:.1f} us")
    del d
