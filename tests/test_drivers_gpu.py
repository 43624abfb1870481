"""The host drivers and the CLI on the real HIP engine (GPU), against the reference goldens."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMB_FMT = "query,emb_rank,target,emb_score,q_len,t_len,metadata".split(",")


@pytest.fixture(scope="module")
def hip_network():
    from merizo_search_amd.foldclass.network import network_setup
    net, dev = network_setup(device="cuda", allow_synthetic=True)
    return net


def _pt_db(tmp, golden_dir):
    from merizo_search_amd.foldclass import dbutil, synthetic as syn
    g = np.load(os.path.join(golden_dir, "dbsearch.npz"))
    names, coords, seqs = syn.synthetic_structures(40, seed=31, min_len=20, max_len=90)
    for name in ("M0", "3w5h"):
        p = np.load(os.path.join(golden_dir, f"pdb_{name}.npz"))
        names.append(f"/db/{name}.pdb"); coords.append(p["coords"]); seqs.append(str(p["seq"]))
    dbutil.write_pt_db(os.path.join(tmp, "syn"), g["db_emb"], names, coords, seqs)
    return os.path.join(tmp, "syn"), g, (names, coords, seqs)


@pytest.mark.parametrize("mincov", [0.0, 0.7])
def test_run_dbsearch_pt_on_gpu_matches_reference_tsv(mincov, hip_network, golden_dir, tmp_path):
    from merizo_search_amd.foldclass import dbsearch as ds, results
    db_prefix, g, _ = _pt_db(str(tmp_path), golden_dir)
    queries = []
    for q in ("M0", "3w5h"):
        p = np.load(os.path.join(golden_dir, f"pdb_{q}.npz"))
        queries.append(dict(coords=p["coords"], seq=str(p["seq"]), name=f"/q/{q}.pdb"))
    res, all_res = ds.run_dbsearch(queries, db_prefix, str(tmp_path / "tmp"), "cuda", topk=5, fastmode=False, threads=-1,
                                   mincos=-1.0, mintm=0.5, mincov=mincov, inputs_are_ca=True, skip_tmalign=True,
                                   network=hip_network)
    for qi, q in enumerate(("M0", "3w5h")):
        assert [int(h["dbindex"]) for h in res[qi].values()] == g[f"dbindex_{q}_cov{mincov}"].tolist()
        np.testing.assert_allclose([float(h["score"]) for h in res[qi].values()], g[f"scores_{q}_cov{mincov}"], atol=1e-5)
    out = str(tmp_path / "o.tsv")
    results.write_search_results(res, out, EMB_FMT, header=True)
    assert open(out).read() == open(os.path.join(golden_dir, f"dbsearch_cov{mincov}.tsv")).read()


def test_createdb_and_faiss_layout_search_on_gpu(hip_network, golden_dir, tmp_path):
    """createdb (ragged GPU launches) -> both layouts -> search both; embeddings match the
    reference's (golden db_emb was produced by the reference network on the same structures)."""
    from merizo_search_amd.foldclass import dbsearch as ds, pdbio
    from merizo_search_amd.foldclass.makedb import run_createdb
    import torch
    _, g, (names, coords, seqs) = _pt_db(str(tmp_path), golden_dir)
    pdbdir = tmp_path / "pdbs"
    pdbdir.mkdir()
    order = sorted(range(len(names)), key=lambda i: os.path.basename(names[i]))
    for n, c, s in zip(names, coords, seqs):
        pdbio.write_pdb(str(pdbdir), c, s, name=os.path.basename(n).replace(".pdb", ""))
    assert run_createdb(str(pdbdir), str(tmp_path / "made"), network=hip_network, layout="both") == 42
    emb = torch.load(str(tmp_path / "made.pt")).numpy()
    ref = g["db_emb"][order]                                   # createdb sorts by file name
    # coordinates went through the 8.3f PDB text format: compare loosely here (exact parity is
    # covered by test_egnn_gpu.py); cosine must still be ~1
    cos = (emb * ref).sum(1) / np.linalg.norm(emb, axis=1) / np.linalg.norm(ref, axis=1)
    assert cos.min() > 0.9999
    p = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    q = [dict(coords=p["coords"], seq=str(p["seq"]), name="M0.pdb")]
    res_pt, _ = ds.run_dbsearch(q, str(tmp_path / "made"), str(tmp_path / "t"), "cuda", 5, False, -1, mincos=0.0, mintm=0.5,
                                mincov=0.0, inputs_are_ca=True, skip_tmalign=True, network=hip_network)
    os.remove(tmp_path / "made.pt")
    res_fa, _ = ds.run_dbsearch(q, str(tmp_path / "made"), str(tmp_path / "t"), "cuda", 5, False, -1, mincos=0.0, mintm=0.5,
                                mincov=0.0, inputs_are_ca=True, skip_tmalign=True, network=hip_network, search_batchsize=16)
    assert [h["target"] for h in res_pt[0].values()] == [h["target"] for h in res_fa[0].values()]
    assert list(res_fa[0].keys()) == [0, 1, 2, 3, 4] and res_fa[0][0]["target"] == "M0"
    np.testing.assert_allclose([float(h["score"]) for h in res_pt[0].values()], [float(h["score"]) for h in res_fa[0].values()], atol=2e-6)


def test_cli_search_and_easy_search_end_to_end(golden_dir, tmp_path):
    """`python -m merizo_search_amd.cli createdb / search / easy-search` on the GPU (config C5
    shape: AF-Q96PD2 with the README's chopping), synthetic weights via the env switch."""
    env = dict(os.environ, MERIZO_ALLOW_SYNTHETIC_WEIGHTS="1", PYTHONPATH=REPO)
    from merizo_search_amd.foldclass import pdbio, synthetic as syn
    pdbdir = tmp_path / "pdbs"
    pdbdir.mkdir()
    names, coords, seqs = syn.synthetic_structures(30, seed=8, min_len=60, max_len=170)
    for n, c, s in zip(names, coords, seqs):
        pdbio.write_pdb(str(pdbdir), c, s, name=os.path.basename(n).replace(".pdb", ""))
    run = lambda *a: subprocess.run([sys.executable, "-m", "merizo_search_amd.cli", *a], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    r = run("createdb", str(pdbdir), str(tmp_path / "db"), "-d", "cuda", "--layout", "faiss")
    assert r.returncode == 0, r.stderr
    m0 = os.path.join(golden_dir, "M0_ca.pdb")
    r = run("search", m0, str(tmp_path / "db"), str(tmp_path / "out"), str(tmp_path / "tmp"), "-d", "cuda", "-k", "3",
            "-s", "-1", "--output_headers")
    assert r.returncode == 0, r.stderr
    lines = open(tmp_path / "out_search.tsv").read().splitlines()
    assert lines[0].split("\t") == ["query", "emb_rank", "target", "emb_score", "q_len", "t_len", "metadata"]
    assert len(lines) == 4 and lines[1].split("\t")[0] == "M0_ca" and lines[1].split("\t")[1] == "0"
    pd2 = os.path.join(golden_dir, "AF-Q96PD2-F1-model_v4_ca.pdb")
    r = run("easy-search", pd2, str(tmp_path / "db"), str(tmp_path / "easy"), str(tmp_path / "tmp"), "-d", "cuda", "-k", "2",
            "-s", "-1", "--chopping", "71-189,190-290,291-453", "--output_headers")
    assert r.returncode == 0, r.stderr
    seg = open(tmp_path / "easy_segment.tsv").read().splitlines()
    assert seg[1].split("\t")[:5] == ["AF-Q96PD2-F1-model_v4_ca", "775", "383", "392", "3"]
    rows = [l.split("\t") for l in open(tmp_path / "easy_search.tsv").read().splitlines()[1:]]
    assert len(rows) == 6 and {r_[0] for r_ in rows} == {f"AF-Q96PD2-F1-model_v4_ca_merizo_0{i}" for i in (1, 2, 3)}
    assert rows[0][1] == "71-189" and rows[0][4] == "0"
    r = run("search", m0, str(tmp_path / "db"), str(tmp_path / "out"), str(tmp_path / "tmp"), "-d", "cpu")
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)          # no CPU fallback


def test_streamed_bench_block_on_a_small_memmap():
    """bench.py's `streamed` block (the reference's db_iterator loop over a host memmap, dbsearch.py:233-243) on 300,000 rows in blocks of
    65,536: the streamed search returns the bits of the same blocks scanned from HBM, and the entry carries rows/s, H2D GB/s against the
    63 GB/s host link, the copy-only and scan-only legs and the overlap."""
    import torch
    sys.path.insert(0, REPO)
    import bench
    from merizo_search_amd import ops
    from merizo_search_amd.foldclass import synthetic as syn
    out = bench.streamed_bench(torch, ops, syn, torch.device("cuda", 0), 10, lambda m: None, sizes=(300_000,), nqs=(1, 96), block_rows=65_536)
    assert out["bound"] == "pcie" and out["peak_GBps"] == 63.0
    ents = [e for e in out["entries"] if "h2d_GBps" in e]
    assert len(ents) == 2, out["entries"]
    for e in ents:
        assert e["identical_to_resident_blocks"] is True
        assert 0.0 < e["frac_of_pcie_peak"] < 1.0 and e["rows_per_s"] > 1e6 and 0.0 <= e["overlap_hidden_frac"] <= 1.0
        assert e["copy_only_seconds"] > 0 and e["scan_only_seconds"] > 0
    assert len(out["block_sweep"]) == 3 and out["roofline"]["bound"] == "pcie" and out["roofline"]["frac"] == max(e["frac_of_pcie_peak"] for e in ents)


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a C ABI: a plain-C program (tests/c_abi/abi_smoke.c, no Python or torch in the
    process) normalises, searches two shards, merges and checks against its own brute force."""
    import shutil
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the ROCm headers")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(repo, "merizo_search_amd")
    build = subprocess.run(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(repo, "include"),
                            os.path.join(repo, "tests", "c_abi", "abi_smoke.c"), "-L" + libdir, "-lmerizo_search_amd",
                            "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "abi_smoke ok" in run.stdout
