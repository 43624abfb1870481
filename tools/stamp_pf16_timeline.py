"""Diagnostic (-DMS_STAMP build): the timeline of the fp16-image prefilter scan -- when each wave enters the kernel, finishes its
set-up, sees its first tile, leaves the loop over its stream and exits, against the launch duration the host sees.
usage: MS_LIB_OVERRIDE=.../build/stamp/libmerizo_search_amd.so [MS_PF_FORMAT=f16x1] python3 tools/stamp_pf16_timeline.py ROWS,NQ,K [...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn
lib = _lib.load()
lib.ms_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for a in sys.argv[1:] or ["1000000,256,10"]:
    n, nq, k = (int(x) for x in a.split(","))
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
    qq = torch.randn(nq, 128, device="cuda"); qq = qq / qq.norm(dim=1, keepdim=True)
    img = ops.pf_build_image(d, row_norm_bound=1.0 + 1e-6)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    prep = lambda: ops.ip_topk_prefiltered_stage("prepare", d, qq, k, ws, image=img)
    scan = lambda: ops.ip_topk_prefiltered_stage("scan", d, qq, k, ws, image=img)
    for _ in range(20):
        prep(); scan()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    prep(); e0.record(); scan(); e1.record(); torch.cuda.synchronize()
    words = 8 * 8 * 4096 * 2
    buf = np.zeros(words, dtype=np.uint64)
    assert lib.ms_debug_stamps(buf.ctypes.data, words) == 0
    tl = buf[8 * 8 * 4096:].reshape(-1, 8)
    tl = tl[tl[:, 4] > 0].astype(np.float64) / 100.0            # us
    t0 = tl[:, 0].min()
    tl -= t0
    f = lambda x: "min %.1f med %.1f max %.1f" % (x.min(), np.median(x), x.max())
    print(f"n={n} nq={nq} k={k} format={os.environ.get('MS_PF_FORMAT', 'auto->f16x2')}: launch (events) {e0.elapsed_time(e1)*1e3:.1f} us, {len(tl)} waves")
    print(f"   entry        {f(tl[:, 0])}\n   set-up done  {f(tl[:, 1])}\n   first tile   {f(tl[:, 2])}\n   loop end     {f(tl[:, 3])}\n   exit         {f(tl[:, 4])}")
    print(f"   per wave: set-up {f(tl[:, 1] - tl[:, 0])} | first-tile wait {f(tl[:, 2] - tl[:, 1])} | loop {f(tl[:, 3] - tl[:, 2])} | tail {f(tl[:, 4] - tl[:, 3])}")
    del d, img
