"""The product drivers on several ranks: run_dbsearch / the CLI with an initialised process group
(row-sharded database, one all-gather, merge, rank 0 reports) must reproduce the one-rank run --
TSV text and score bits.  CPU: gloo + the oracle engine.  GPU: gloo between two ranks that share
cuda:0 (one-GPU box), the HIP engine and ms_topk_merge_strided behind a real collective; plus RCCL
itself at world size 1."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, "tests", "dist_worker.py")


def _make_case(work, n=3001, nq=7, k=10, batch=500):
    """faiss-layout + `.pt` databases of n rows with exact duplicates on both sides of the 2-rank shard
    boundary (ties must resolve to the lower row), queries as CA traces."""
    from merizo_search_amd.foldclass import dbutil, synthetic as syn
    os.makedirs(work, exist_ok=True)
    raw, lengths = syn.raw_database(n, seed=11)
    if n > 20:
        raw[5] = raw[n - 3]; raw[n // 2 - 1] = raw[n // 2 + 4]; raw[7] = raw[8]
    else:
        lengths[:] = 30.0                                      # tiny databases: nothing masked, k <= n stays valid
    names = ["d%06d" % i for i in range(n)]
    seqs = ["A" * int(l) for l in lengths]
    coords = [np.zeros((int(l), 3), np.float32) for l in lengths]
    norm = raw / np.linalg.norm(raw, axis=1, keepdims=True)
    dbutil.write_faiss_db(os.path.join(work, "fa"), norm.astype(np.float32), names, seqs, coords,
                          metadata=['{ "row": %d }' % i for i in range(n)])
    dbutil.write_pt_db(os.path.join(work, "pt"), raw, ["/x/" + nm + ".pdb" for nm in names], coords, seqs)
    qn, qc, qs = syn.synthetic_structures(nq, seed=3, min_len=30, max_len=120)
    np.savez(os.path.join(work, "queries.npz"), coords=np.asarray(qc, dtype=object), seqs=np.asarray(qs), names=np.asarray(qn),
             k=k, batch=batch)


def _run_ranks(work, world, kind):
    from conftest import free_port
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS="2", MERIZO_ALLOW_SYNTHETIC_WEIGHTS="1")
    procs = [subprocess.Popen([sys.executable, WORKER, REPO, port, str(r), str(world), kind, work], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]


def _compare(work, world):
    for tag in ("fa", "fa_stream", "pt"):
        one, many = np.load(os.path.join(work, f"{tag}_w1.npz")), np.load(os.path.join(work, f"{tag}_w{world}.npz"))
        assert np.array_equal(one["rows"], many["rows"]), tag
        assert np.array_equal(one["scores"].view(np.uint32), many["scores"].view(np.uint32)), tag
        assert open(os.path.join(work, f"{tag}_w1.tsv")).read() == open(os.path.join(work, f"{tag}_w{world}.tsv")).read(), tag
    a, b = np.load(os.path.join(work, "fa_w1.npz")), np.load(os.path.join(work, "fa_stream_w1.npz"))
    assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["scores"].view(np.uint32), b["scores"].view(np.uint32))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_run_dbsearch_sharded_equals_one_rank_gloo_oracle(world, tmp_path):
    """(8 ranks: the node the north star names -- PackedExchange's strides and the merge of 8 gathered blocks through the drivers)"""
    work = str(tmp_path / "case")
    _make_case(work)
    _run_ranks(work, 1, "oracle")
    _run_ranks(work, world, "oracle")
    _compare(work, world)


_CLI_SHIM = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from oracle_engine import oracle_network
from merizo_search_amd import cli
from merizo_search_amd.foldclass import dbsearch as ds, makedb
net = oracle_network()
ds.network_setup = lambda **kw: (net, "cpu")
makedb.network_setup = lambda **kw: (net, "cpu")
cli.main(sys.argv[2:])
'''


def _torchrun(nproc, argv, env, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + argv
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)


def test_cli_under_torchrun_two_ranks_gloo_oracle(tmp_path):
    """`torchrun --nproc-per-node 2 ... cli easy-search --multi_domain_search` (engine swapped for the oracle
    by a shim): createdb embeds data-parallel, search shards the rows, rank 0 writes every file once."""
    import md_case
    shim = tmp_path / "shim.py"
    shim.write_text(_CLI_SHIM)
    qpdb, dbdir = md_case.write_inputs(tmp_path)
    env = dict(os.environ, MERIZO_DIST_BACKEND="gloo", MERIZO_TMALIGN=md_case.fake_tmalign(tmp_path), OMP_NUM_THREADS="2")
    from conftest import free_port
    port = free_port(2)
    r = _torchrun(2, [str(shim), REPO, "createdb", dbdir, str(tmp_path / "db"), "--layout", "both"], env, port)
    assert r.returncode == 0, r.stderr[-3000:]
    search = ["easy-search", qpdb, str(tmp_path / "db"), None, str(tmp_path / "tmp"), "-k", "3", "-s", "0.5", "--chopping",
              md_case.CHOPPING, "--multi_domain_search", "--output_headers"]
    os.remove(tmp_path / "db.pt")                                       # search the faiss layout
    for nproc, out in ((2, "two"), (1, "one")):
        argv = [str(shim), REPO] + [a if a is not None else str(tmp_path / out) for a in search]
        r = _torchrun(nproc, argv, env, port + 1)
        assert r.returncode == 0, r.stderr[-3000:]
        md_case.check_outputs(str(tmp_path / out))
    for suffix in ("_search.tsv", "_segment.tsv", "_search_multi_dom.tsv"):
        a, b = open(str(tmp_path / "one") + suffix).read(), open(str(tmp_path / "two") + suffix).read()
        if suffix == "_segment.tsv":                                    # the runtime column differs
            a, b = [l.split("\t")[:6] + l.split("\t")[7:] for l in a.splitlines()], [l.split("\t")[:6] + l.split("\t")[7:] for l in b.splitlines()]
        assert a == b, suffix



def _segment_rows_without_runtime(path):
    """`_segment.tsv` rows with the wall-clock `time_sec` column (index 6, programs/utils.py:161-176) removed: it is the only field
    of any output file that differs between two runs of the same command."""
    return [l.split("\t")[:6] + l.split("\t")[7:] for l in open(path).read().splitlines()]


C5_CHOPPING = "71-189,190-290,291-453"          # /root/reference README.md:128 (AF-Q96PD2)


def _c5_database_pdbs(tmp_path, count, seed=21):
    """A directory of `count` synthetic single-domain PDB files (coordinates on the 0.001 grid of the PDB text format) -- the
    input of createdb -- plus copies of the three AF-Q96PD2 domains of the README chopping, so that the C5 query has real hits."""
    from merizo_search_amd.foldclass import chopping as chop, pdbio, synthetic as syn
    dbdir = tmp_path / "c5_pdbs"
    dbdir.mkdir()
    names, coords, seqs = syn.synthetic_structures(count, seed=seed, min_len=40, max_len=150)
    for n, c, s in zip(names, coords, seqs):
        pdbio.write_pdb(str(dbdir), np.round(c.astype(np.float64), 3).astype(np.float32), s, name=os.path.basename(n).replace(".pdb", ""))
    pd2 = os.path.join(REPO, "tests", "golden", "AF-Q96PD2-F1-model_v4_ca.pdb")
    for j, d in enumerate(chop.domains_from_chopping(pd2, C5_CHOPPING, "A")):
        pdbio.write_pdb(str(dbdir), d["coords"], d["seq"], name="AF-Q96PD2-F1-model_v4_TED%02d" % (j + 1))
    return pd2, str(dbdir)


def _c5_on_eight_ranks(tmp_path, launcher_argv, env, count):
    """C5 as BASELINE.json words it (easy-search of AF-Q96PD2, README chopping, -k 10) on EIGHT ranks under torchrun against the same
    command in one process, fed by createdb on eight ranks against createdb in one process: database files and TSVs identical."""
    from conftest import free_port
    pd2, dbdir = _c5_database_pdbs(tmp_path, count)
    port = free_port(4)
    for nproc, tag in ((8, "db8"), (1, "db1")):
        r = _torchrun(nproc, launcher_argv + ["createdb", dbdir, str(tmp_path / tag), "--layout", "faiss"], env, port + (0 if nproc == 8 else 1))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    import json
    info8, info1 = json.load(open(tmp_path / "db8.json")), json.load(open(tmp_path / "db1.json"))
    assert info8["DB_SIZE"] == info1["DB_SIZE"] == count + 3
    for key in ("dbfname_IP", "db_names_f", "sif", "sdf", "cif", "cdf"):          # every file of the faiss layout, byte for byte
        a, b = open(tmp_path / info8[key], "rb").read(), open(tmp_path / info1[key], "rb").read()
        assert a == b, key
    search = ["easy-search", pd2, str(tmp_path / "db8"), None, str(tmp_path / "tmp"), "-k", "10", "-s", "-1", "--chopping", C5_CHOPPING, "--output_headers"]
    for nproc, out in ((8, "eight"), (1, "one")):
        r = _torchrun(nproc, launcher_argv + [a if a is not None else str(tmp_path / out) for a in search], env, port + (2 if nproc == 8 else 3))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    one, eight = open(str(tmp_path / "one") + "_search.tsv").read(), open(str(tmp_path / "eight") + "_search.tsv").read()
    assert one == eight                                                            # byte-identical
    rows = [l.split("\t") for l in one.splitlines()[1:]]
    assert len(rows) == 30 and {r_[0] for r_ in rows} == {"AF-Q96PD2-F1-model_v4_ca_merizo_0%d" % i for i in (1, 2, 3)}
    for j, dom in enumerate(("71-189", "190-290", "291-453")):                     # every domain finds its own copy first
        top = rows[10 * j]
        assert top[1] == dom and top[4] == "0" and top[5] == "AF-Q96PD2-F1-model_v4_TED%02d" % (j + 1), top
    seg1, seg8 = _segment_rows_without_runtime(str(tmp_path / "one") + "_segment.tsv"), _segment_rows_without_runtime(str(tmp_path / "eight") + "_segment.tsv")
    assert seg1 == seg8 and seg1[1][:5] == ["AF-Q96PD2-F1-model_v4_ca", "775", "383", "392", "3"]


def test_c5_easy_search_af_q96pd2_on_eight_ranks_gloo_oracle(tmp_path):
    """C5 verbatim (merizo.py:367-383, README.md:128) on 8 ranks, CPU: gloo + the oracle engine behind the product CLI."""
    shim = tmp_path / "shim.py"
    shim.write_text(_CLI_SHIM)
    env = dict(os.environ, MERIZO_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    _c5_on_eight_ranks(tmp_path, [str(shim), REPO], env, count=21)


def test_ranks_wait_out_rank0_postprocessing_beyond_the_group_timeout(tmp_path):
    """Rank 0's serial post-processing (TM-align per hit, multi-domain step, TSV files) may outlast the process group's
    collective timeout: the other ranks wait on the store, not in a collective, so nobody is aborted.
    Group timeout 5 s, rank 0 'post-processes' for 9 s."""
    script = tmp_path / "slow_rank0.py"
    script.write_text(r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from merizo_search_amd.foldclass import sharded
rank, world, _dev = sharded.init_distributed()
assert world == 2
t = torch.ones(4) * (rank + 1)
dist.all_reduce(t)                               # the exchange every rank takes part in
assert float(t[0]) == 3.0
if rank == 0:
    time.sleep(9.0)                              # rank 0 alone reports
    open(sys.argv[2], "w").write("written by rank 0\n")
sharded.finalize_distributed()
sys.stdout.write("rank %d finalized\n" % rank)          # (ONE write: two ranks share the pipe)
sys.stdout.flush()
""")
    env = dict(os.environ, MERIZO_DIST_BACKEND="gloo", MERIZO_DIST_TIMEOUT_S="5", OMP_NUM_THREADS="1")
    out = tmp_path / "report.txt"
    from conftest import free_port
    r = _torchrun(2, [str(script), REPO, str(out)], env, free_port())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert out.read_text() == "written by rank 0\n"
    assert "rank 0 finalized" in r.stdout and "rank 1 finalized" in r.stdout


def test_same_device_with_rccl_backend_is_refused(monkeypatch):
    """MERIZO_SAME_DEVICE=1 (every rank on cuda:0) only works over gloo; with the default nccl backend RCCL would fail or
    hang, so init_distributed says so up front."""
    from merizo_search_amd.foldclass import sharded
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("MERIZO_SAME_DEVICE", "1"); monkeypatch.delenv("MERIZO_DIST_BACKEND", raising=False)
    with pytest.raises(RuntimeError, match="gloo"):
        sharded.init_distributed()


def test_balance_by_cost_is_a_partition_and_balanced():
    from merizo_search_amd.foldclass.sharded import balance_by_cost
    rng = np.random.default_rng(0)
    costs = (rng.integers(25, 700, size=101) ** 2).tolist()
    for world in (1, 2, 3, 8):
        shares = balance_by_cost(costs, world)
        assert sorted(i for s in shares for i in s) == list(range(101))
        loads = [sum(costs[i] for i in s) for s in shares]
        assert max(loads) - min(loads) <= max(costs)
    assert balance_by_cost([], 4) == [[], [], [], []]


# ---------------------------------------------------------------------------- GPU ------
@pytest.mark.gpu
def test_run_dbsearch_sharded_equals_one_rank_hip_engine_two_ranks(tmp_path):
    """Two fresh child ranks on cuda:0 over gloo with the HIP engine: sharded scan -> all-gather ->
    ms_topk_merge_strided == the one-rank run, bit for bit (resident, streamed, and the `.pt` cosine path)."""
    work = str(tmp_path / "case")
    _make_case(work, n=200_003, nq=70, k=10, batch=30_000)
    _run_ranks(work, 1, "hip")
    _run_ranks(work, 2, "hip")
    _compare(work, 2)


@pytest.mark.gpu
def test_run_dbsearch_sharded_equals_one_rank_hip_engine_eight_ranks(tmp_path):
    """The C4 / C5 rank count on the hardware there is: EIGHT child ranks sharing cuda:0 over gloo (MERIZO_SAME_DEVICE), the HIP
    engine on every shard, one all-gather of 8 packed blocks, ms_topk_merge_strided at S = 8 -- == the one-rank run bit for bit
    (resident, streamed, and the `.pt` cosine path; the prefiltered search across 8 ranks: test_bench_gpus_8_...)."""
    work = str(tmp_path / "case")
    _make_case(work, n=200_003, nq=70, k=10, batch=30_000)
    _run_ranks(work, 1, "hip")
    _run_ranks(work, 8, "hip")
    _compare(work, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("n,nq,k,world", [(3, 1, 2, 2), (1, 2, 1, 2), (5, 3, 5, 3)])
def test_run_dbsearch_sharded_tiny_databases_hip_engine(n, nq, k, world, tmp_path):
    """Shards smaller than k, and ranks with no rows at all (more ranks than rows): padded / empty per-shard lists go
    through the all-gather and the merge; same TSV and score bits as one rank."""
    work = str(tmp_path / "case")
    _make_case(work, n=n, nq=nq, k=k, batch=2)
    _run_ranks(work, 1, "hip")
    _run_ranks(work, world, "hip")
    _compare(work, world)


@pytest.mark.gpu
def test_cli_under_torchrun_two_ranks_on_one_gpu(tmp_path):
    """The user-facing form: torchrun -m merizo_search_amd.cli createdb / easy-search --multi_domain_search with
    2 ranks (both on cuda:0, gloo: MERIZO_SAME_DEVICE / MERIZO_DIST_BACKEND) against the 1-process CLI."""
    import md_case
    qpdb, dbdir = md_case.write_inputs(tmp_path)
    env = dict(os.environ, MERIZO_DIST_BACKEND="gloo", MERIZO_SAME_DEVICE="1", MERIZO_ALLOW_SYNTHETIC_WEIGHTS="1",
               MERIZO_TMALIGN=md_case.fake_tmalign(tmp_path), PYTHONPATH=REPO)
    from conftest import free_port
    port = free_port(2)
    r = _torchrun(2, ["-m", "merizo_search_amd.cli", "createdb", dbdir, str(tmp_path / "db"), "--layout", "faiss"], env, port)
    assert r.returncode == 0, r.stderr[-3000:]
    search = ["easy-search", qpdb, str(tmp_path / "db"), None, str(tmp_path / "tmp"), "-k", "3", "-s", "0.5", "--chopping",
              md_case.CHOPPING, "--multi_domain_search", "--multi_domain_mode", "exhaustive_tmalign", "--output_headers"]
    r = _torchrun(2, ["-m", "merizo_search_amd.cli"] + [a if a is not None else str(tmp_path / "two") for a in search], env, port + 1)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "merizo_search_amd.cli"] + [a if a is not None else str(tmp_path / "one") for a in search],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    for out in ("one", "two"):
        md_case.check_outputs(str(tmp_path / out))
    for suffix in ("_search.tsv", "_search_multi_dom.tsv"):
        assert open(str(tmp_path / "one") + suffix).read() == open(str(tmp_path / "two") + suffix).read(), suffix



@pytest.mark.gpu
def test_c5_easy_search_af_q96pd2_on_eight_ranks_one_gpu(tmp_path):
    """C5 verbatim on the hardware there is: `torchrun --nproc-per-node 8 -m merizo_search_amd.cli easy-search AF-Q96PD2 ... --chopping
    71-189,190-290,291-453 -k 10` (eight ranks sharing cuda:0 over gloo: MERIZO_SAME_DEVICE / MERIZO_DIST_BACKEND; the HIP engine on every
    shard, 16 rows per shard -- fewer than k + the padded lists through the all-gather) against the one-process run: `_search.tsv`
    byte-identical, `_segment.tsv` identical but for its wall-clock column; the database comes from createdb on 8 ranks and is byte-identical
    to the one createdb writes in one process."""
    env = dict(os.environ, MERIZO_DIST_BACKEND="gloo", MERIZO_SAME_DEVICE="1", MERIZO_ALLOW_SYNTHETIC_WEIGHTS="1", PYTHONPATH=REPO, GLOO_SOCKET_IFNAME="lo")
    _c5_on_eight_ranks(tmp_path, ["-m", "merizo_search_amd.cli"], env, count=125)


@pytest.mark.gpu
def test_rccl_exchange_runs_at_world_size_one(tmp_path):
    """RCCL itself on the hardware there is: a 1-rank "nccl" group, PackedExchange's all_gather_into_tensor
    branch and the in-place merge of the gathered block."""
    script = tmp_path / "rccl1.py"
    script.write_text(r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
from merizo_search_amd.foldclass.sharded import PackedExchange
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
db = syn.device_database(50_000, 0, seed=0, device="cuda:0")
q = syn.device_database(33, 0, seed=1, device="cuda:0")
s, i = ops.ip_topk(db, q, 10, row_offset=1000)
ex = PackedExchange(33, 10, "cuda:0")
ex.out_s.copy_(s); ex.out_i.copy_(i)
g = ex.exchange()                       # dist.all_gather_into_tensor over RCCL
ms, mi = ex.merge()                     # ms_topk_merge_strided on the gathered block
torch.cuda.synchronize()
assert torch.equal(ms, s) and torch.equal(mi, i)
dist.destroy_process_group()
print("rccl ok")
''')
    from conftest import free_port
    r = subprocess.run([sys.executable, str(script), REPO, str(free_port())], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _run_bench(args, env, timeout):
    """`python bench.py ...` as a child in its own process group, ONCE.  A run that exceeds `timeout` is killed as a group (its ranks
    would otherwise keep the GPU) and the test fails with what the ranks printed: MS_BENCH_FAULT_DUMP makes every rank dump its Python
    stacks shortly before the limit, so a stuck rendezvous names the line it sits in (round 4 retried here instead; one suite run in
    twenty had needed it and 60 dedicated repeats never did -- see DESIGN.md section 6).  GLOO_SOCKET_IFNAME=lo: the ranks of these
    self-tests talk over loopback, whatever the box's hostname resolves to."""
    import signal
    env = dict(env, MS_BENCH_FAULT_DUMP=str(max(30, timeout - 40)), GLOO_SOCKET_IFNAME="lo")
    p = subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=env, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)          # (the group this test started: p is its leader)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        pytest.fail("bench.py %s exceeded %d s; stacks of the ranks:\n%s" % (" ".join(args), timeout, (err or "")[-12000:]))
    return subprocess.CompletedProcess(p.args, p.returncode, out, err)


def _bench_line(stdout):
    """The LAST stdout line is the one the driver parses: one JSON object, under 4 KB."""
    import json
    last = stdout.strip().splitlines()[-1]
    assert last.startswith("{") and len(last) < 4096, len(last)
    line = json.loads(last)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    assert "workload" in line["config"] and line["dtype"] == "f32"
    if line["n_gpus"] > 1:          # an N > 1 line proves what ran: backend, ranks, DISTINCT devices, RCCL version, the exchange's own time
        co = line["collective"]
        for key in ("backend", "world", "distinct_devices", "rccl_version", "exchange_us", "same_device_selftest"):
            assert key in co, key
        assert co["world"] == line["n_gpus"] and co["exchange_us"] > 0 and co["distinct_devices"] >= 1
    else:
        assert "collective" not in line
    return line


@pytest.mark.gpu
def test_bench_gpus_2_self_launches_and_is_exact(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts its two ranks itself (a fresh child, never
    an exec), here both on cuda:0 over gloo; the JSON line reports 2 GPUs, recall 1.0 and lists identical to a brute force."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MS_BENCH_SAME_DEVICE="1", MS_BENCH_BACKEND="gloo")
    r = _run_bench(["--gpus", "2", "--rows", "600000", "--nq", "96", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], env, 400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["rows_per_gpu"] == 300000
    assert line["collective"]["backend"] == "gloo" and line["collective"]["distinct_devices"] == 1 and line["collective"]["same_device_selftest"] is True
    assert line["weak_scaling_ref_q_per_s"] > 0 and 0.2 < line["vs_ref"] <= 1.05       # the exchange + merge cost something, not much
    assert line["recall_at_k"] == 1.0 and line["planted_recall"] == 1.0
    assert line["topk_identical_to_torch_bruteforce"].startswith("96 of 96")
    assert line["prefiltered"]["identical_to_fp32"] is True


@pytest.mark.gpu
def test_bench_gpus_8_self_launches_on_tiny_shards(tmp_path):
    """`python bench.py --gpus 8` (the driver's scaling run) end to end on one GPU: eight ranks on cuda:0 over gloo, 100,000 rows
    each; exact, and the prefiltered block of the same step identical to the fp32 scan across the 8-way exchange."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MS_BENCH_SAME_DEVICE="1", MS_BENCH_BACKEND="gloo")
    r = _run_bench(["--gpus", "8", "--rows", "800000", "--nq", "96", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], env, 400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 8 and line["config"]["rows_per_gpu"] == 100000
    assert line["collective"]["backend"] == "gloo" and line["collective"]["world"] == 8 and line["collective"]["distinct_devices"] == 1
    assert line["weak_scaling_ref_q_per_s"] > 0 and line["vs_ref"] > 0
    assert line["recall_at_k"] == 1.0 and line["planted_recall"] == 1.0
    assert line["topk_identical_to_torch_bruteforce"].startswith("96 of 96")
    assert line["prefiltered"]["identical_to_fp32"] is True


@pytest.mark.gpu
def test_bench_default_line_is_parseable_and_carries_roofline_and_cpu_baseline(tmp_path):
    """`python bench.py` as the driver runs it at N = 1 (C2; extras skipped here for time): the last stdout line is under 4 KB and
    carries the contract's keys, `roofline` and `cpu_baseline`; the full document went to bench_full.json."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = _run_bench(["--steps", "20", "--warmup", "5", "--no-extras"], env, 600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1            # ONE stdout line
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["workload"].startswith("C2:") and line["config"]["db_rows"] == 1_000_000
    assert line["recall_at_k"] == 1.0 and line["roofline"]["bound"] == "mfma" and 0.3 < line["roofline"]["frac"] < 1.0
    assert line["roofline"]["kernel_ms"] < line["ms_per_step"]
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["unit"] == "queries/s"
    assert line["prefiltered"]["identical_to_fp32"] is True
    with open(os.path.join(REPO, line["full"])) as fh:
        full = json.load(fh)
    assert full["value"] == pytest.approx(line["value"], rel=1e-5) and "notes" in full and "torch_cpu" in full["cpu_baseline"]


@pytest.mark.gpu
def test_bench_shape_c4_is_the_one_rank_point_of_the_weak_scaling_curve(tmp_path):
    """`python bench.py --gpus 1 --shape c4` (rows per GPU cut down here): the top-level step is one rank's share of C4."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MS_BENCH_ROWS_PER_GPU"] = "2000000"
    r = _run_bench(["--shape", "c4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env, 600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["config"]["workload"].startswith("C4 shape, ONE rank") and line["config"]["queries_per_step"] == 4096
    assert line["config"]["db_rows"] == 2_000_000 and line["recall_at_k"] == 1.0 and line["prefiltered"]["identical_to_fp32"] is True


@pytest.mark.gpu
def test_bench_measures_the_top_level_traffic_itself(tmp_path):
    """`roofline.traffic` of the top-level line comes from rocprofv3 --pmc CHILD passes started by bench.py itself (FETCH_SIZE;
    WRITE_SIZE; gfx950 corrections), not from a committed file: about the algorithmic 512 MB at C2, and the line says where it came from."""
    import shutil
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("no rocprofv3 on this box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MS_BENCH_LIVE_TRAFFIC"] = "1"
    r = _run_bench(["--steps", "10", "--warmup", "3", "--no-extras", "--no-cpu-baseline"], env, 600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["roofline"]["traffic_from"].startswith("measured in this run"), line["roofline"]
    assert 0.98 * 512e6 <= line["roofline"]["traffic"] <= 1.15 * 512e6
    pf = line["prefiltered"]["roofline"]                       # the image scan of the prefiltered step: 256 B per row
    assert pf["traffic_from"].startswith("measured in this run") and 0.98 * 256e6 <= pf["traffic"] <= 1.2 * 256e6, pf
