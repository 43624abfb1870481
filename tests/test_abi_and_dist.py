"""CPU checks of the C-ABI surface and of the multi-rank search path (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from merizo_search_amd import _lib
    _lib.build()
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "merizo_search_amd.h")).read()
    declared = set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (ms_[a-z0-9_]+)", out))
    assert declared <= exported
    assert lib.ms_version() >= 100
    assert int(lib.ms_egnn_weight_floats()) == 792330
    assert int(lib.ms_ip_topk_workspace_bytes(1000, 8, 10)) > 0
    assert int(lib.ms_ip_topk_workspace_bytes(-1, 8, 10)) == 0


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device every product entry point raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from merizo_search_amd import _lib, ops
    from merizo_search_amd.foldclass.engine import HipEngine
    with pytest.raises(_lib.MerizoHipError):
        ops.ip_topk(torch.zeros(4, 128), torch.zeros(1, 128), 1)
    with pytest.raises(_lib.MerizoHipError):
        HipEngine("cuda:0")
    src = open(os.path.join(REPO, "merizo_search_amd", "ops.py")).read() + open(os.path.join(REPO, "merizo_search_amd", "_lib.py")).read()
    assert "oracle" not in src.replace("no CPU fallback", "")


def test_product_never_imports_the_oracle():
    for root, _dirs, files in os.walk(os.path.join(REPO, "merizo_search_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "from oracle" not in text and "import oracle" not in text and "liboracle" not in text, f


def test_shard_bounds_cover_rows_exactly():
    from merizo_search_amd.foldclass.sharded import shard_bounds
    for n in (0, 1, 7, 8, 9, 1000, 365_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert all(0 <= lo <= hi for lo, hi in spans)


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from merizo_search_amd.foldclass import synthetic as syn
from merizo_search_amd.foldclass.sharded import ShardedIndex, shard_bounds
from oracle_engine import OracleEngine
from oracle import oracle as orc
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
rank, world = dist.get_rank(), dist.get_world_size()
eng = OracleEngine()
n, nq, k = 5003, 17, 10
db = syn.normalized_database(n, seed=1); q = torch.from_numpy(syn.normalized_database(nq, seed=2))
db[5] = db[4000]; db[4999] = db[17]                      # cross-shard exact ties
lo, hi = shard_bounds(n, world, rank)
idx = ShardedIndex(torch.from_numpy(db[lo:hi].copy()), lo, search_fn=lambda d, qq, kk, row_offset=0: eng.ip_topk(d, qq, kk, row_offset),
                   merge_fn=eng.topk_merge)
s, i = idx.search(q, k)
s_ref, i_ref = orc.ip_topk(db, q.numpy(), k)
assert np.array_equal(i.numpy(), i_ref), (rank, "indices")
assert np.array_equal(s.numpy(), s_ref), (rank, "scores")
# identical on every rank
mine = torch.cat([s.reshape(-1), i.reshape(-1).to(torch.float32)])
other = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(other, mine)
assert all(torch.equal(o, mine) for o in other)
# a rank with fewer rows than k still takes part (padding entries are ignored by the merge)
lo2, hi2 = (0, 3) if rank == 0 else (3, n)
idx2 = ShardedIndex(torch.from_numpy(db[lo2:hi2].copy()), lo2, search_fn=lambda d, qq, kk, row_offset=0: eng.ip_topk(d, qq, kk, row_offset),
                    merge_fn=eng.topk_merge)
s2, i2 = idx2.search(q, k)
assert np.array_equal(i2.numpy(), i_ref) and np.array_equal(s2.numpy(), s_ref)
# the serving-loop form: results written straight into this rank's packed block, one all-gather, blocks read in place
from merizo_search_amd.foldclass.sharded import PackedExchange
for nq3, k3 in ((17, 10), (5, 3)):                       # nq*k odd too: the int64 half stays 8-byte aligned
    ex = PackedExchange(nq3, k3, "cpu")
    s3, i3 = eng.ip_topk(torch.from_numpy(db[lo:hi].copy()), q[:nq3], k3, lo)
    ex.out_s.copy_(s3); ex.out_i.copy_(i3)
    ex.exchange()
    s4, i4 = ex.merge(merge_fn=eng.topk_merge)
    s_ref3, i_ref3 = orc.ip_topk(db, q[:nq3].numpy(), k3)
    assert np.array_equal(i4.numpy(), i_ref3) and np.array_equal(s4.numpy(), s_ref3), (rank, "packed exchange")
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_search_two_ranks_gloo(tmp_path):
    """world_size-2 run of the real collective path (pack -> all_gather -> unpack -> merge) with
    the oracle engine standing in for the GPU kernels: sharded == unsharded, bit for bit."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    from conftest import free_port
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), REPO, port, str(r), "2"], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_c_abi_argument_errors_are_reported_without_a_gpu():
    """Bad arguments are rejected by the host-side checks (no kernel launch): error code + message."""
    import ctypes
    from merizo_search_amd import _lib
    lib = _lib.load()
    rc = lib.ms_ip_topk(None, -1, 0, None, 1, 1, 0, None, None, None, 0.0, None, None, None, 0, None)
    assert rc == -1 and b"n >= 0" in lib.ms_last_error()
    rc = lib.ms_ip_topk(None, 0, 0, ctypes.c_void_p(16), 1, 1, 7, None, None, None, 0.0, None, None, None, 0, None)
    assert rc == -1 and b"unknown mode" in lib.ms_last_error()
    rc = lib.ms_ip_topk(ctypes.c_void_p(16), 2 ** 31, 0, ctypes.c_void_p(16), 1, 1, 0, None, None, None, 0.0, None, None, None, 0, None)
    assert rc == -4 and b"shard the database" in lib.ms_last_error()
    rc = lib.ms_ip_topk(ctypes.c_void_p(16), 10, 0, ctypes.c_void_p(16), 1, 1, 0, ctypes.c_void_p(16), None, None, 0.0,
                        ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, None)
    assert rc == -1 and b"only valid in the cosine modes" in lib.ms_last_error()
    rc = lib.ms_ip_topk(ctypes.c_void_p(16), 10, 0, ctypes.c_void_p(16), 1, 1, 2, ctypes.c_void_p(16), None, None, 0.0,
                        ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, None)
    assert rc == -1 and b"MS_MODE_COSINE_UNIT takes rows that are normalised already" in lib.ms_last_error()
    rc = lib.ms_ip_topk(ctypes.c_void_p(16), 10, 0, ctypes.c_void_p(16), 1, 1, 0, None, None, None, 0.0,
                        ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, None)
    assert rc == -2 and b"workspace" in lib.ms_last_error()
    rc = lib.ms_l2_normalize_rows(ctypes.c_void_p(16), 4, 64, 1e-12, None)
    assert rc == -1 and b"d must be 128" in lib.ms_last_error()
    rc = lib.ms_topk_merge(None, None, 0, 1, 1, None, None, None)
    assert rc == -1
    import numpy as np
    offs = np.array([0, 5, 5], dtype=np.int32)
    rc = lib.ms_egnn_embed(ctypes.c_void_p(16), ctypes.c_void_p(16), 3000, ctypes.c_void_p(16), ctypes.c_void_p(16),
                           offs.ctypes.data, 2, ctypes.c_void_p(16), None, 0, None)
    assert rc == -1 and b"is empty" in lib.ms_last_error()
    offs = np.array([0, 3001], dtype=np.int32)
    rc = lib.ms_egnn_embed(ctypes.c_void_p(16), ctypes.c_void_p(16), 3000, ctypes.c_void_p(16), ctypes.c_void_p(16),
                           offs.ctypes.data, 1, ctypes.c_void_p(16), None, 0, None)
    assert rc == -4 and b"positional table" in lib.ms_last_error()


def test_bench_self_launches_its_ranks_without_a_gpu_call_in_the_parent(tmp_path):
    """`python bench.py --gpus 2` with no outer launcher starts torch.distributed.run as a CHILD before importing torch
    (never exec), passes the ranks' output through and exits with their code.  Here (no GPU) the ranks must fail loudly in
    _lib.require_gpu(), and the parent must report that failure."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert "launching 2 ranks" in r.stderr
    assert "--nproc-per-node=2" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0                      # children ran and refused to run without a GPU: no silent fallback
        assert "GPU" in r.stderr or "HIP" in r.stderr or "hip" in r.stderr
