"""Host-side logic on CPU: PDB parsing, database layouts, drivers, TSV writers, CLI checks.

The drivers are run with the oracle-backed engine from tests/oracle_engine.py (test
infrastructure) and compared with outputs of the REFERENCE captured in tests/golden/."""
import json
import mmap
import os
import pickle

import numpy as np
import pytest
import torch

from merizo_search_amd.foldclass import chopping as chop
from merizo_search_amd.foldclass import dbsearch as ds
from merizo_search_amd.foldclass import dbutil, pdbio, results
from merizo_search_amd.foldclass import synthetic as syn
from merizo_search_amd.foldclass.makedb import run_createdb
from oracle_engine import OracleEngine, oracle_network

EXAMPLES = ["M0", "3w5h", "AF-Q96HM7-F1-model_v4", "AF-Q96PD2-F1-model_v4"]
EMB_FMT = "query,emb_rank,target,emb_score,q_len,t_len,metadata".split(",")


# ------------------------------------------------------------------ G5 read_pdb --------
@pytest.mark.parametrize("name", EXAMPLES)
def test_read_pdb_matches_reference(name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"pdb_{name}.npz"))
    d = pdbio.read_pdb(os.path.join(golden_dir, f"{name}_ca.pdb"), "A")
    assert np.array_equal(d["coords"], g["coords"]) and d["coords"].dtype == np.float32
    assert d["seq"] == str(g["seq"])
    if name == "M0":
        full = pdbio.read_pdb(os.path.join(golden_dir, "M0.pdb"), "A")      # whole file incl. non-CA atoms
        assert np.array_equal(full["coords"], g["coords"]) and full["seq"] == "GTLPCGESCVWIPCISSVVGCSCKSKVCYKN"


def test_read_pdb_error_paths(golden_dir, tmp_path):
    with pytest.raises(SystemExit) as e:
        pdbio.read_pdb(os.path.join(golden_dir, "M0_ca.pdb"), "AB")
    assert e.value.code == 127
    with pytest.raises(SystemExit) as e:
        pdbio.read_pdb(os.path.join(golden_dir, "M0_ca.pdb"), "Z")
    assert e.value.code == 128
    short = tmp_path / "short.pdb"
    short.write_text("REMARK\n")
    with pytest.raises(IndexError):                       # the reference indexes column 22 unguarded
        pdbio.read_pdb(str(short), "A")
    coords, seq = pdbio.read_pdb_all_chains(os.path.join(golden_dir, "AF-Q96PD2-F1-model_v4_ca.pdb"), max_len=100)
    assert coords.shape == (100, 3) and len(seq) == 100  # createdb truncation (makedb.py:68-69)


def test_write_pdb_roundtrip(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    fn = pdbio.write_pdb(str(tmp_path), g["coords"], str(g["seq"]), name="x")
    lines = open(fn).read().splitlines()
    assert lines[0] == "ATOM      1  CA  GLY A   1      -0.613   3.564   3.258  1.00  0.00"[:len(lines[0])] or lines[0].startswith("ATOM      1  CA  GLY A   1")
    assert lines[-1] == "END" and len(lines) == 32
    with pytest.raises(IndexError):        # reference quirk kept: its reader cannot parse its writer's "END" line
        pdbio.read_pdb(fn, "A")
    coords, seq = pdbio.read_pdb_all_chains(fn)
    np.testing.assert_allclose(coords, g["coords"], atol=5e-4)
    assert seq == str(g["seq"])


# ------------------------------------------------------------------ G7 dbutil ----------
def test_dbutil_retrieval_matches_reference(golden_dir, golden_meta):
    sl = os.path.join(golden_dir, "ted100_slice")
    meta = golden_meta["g7"]
    with open(os.path.join(sl, "names.index_names"), "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        assert dbutil.retrieve_names_by_idx(meta["idx"], mm).tolist() == meta["names"]
    with open(os.path.join(sl, "seq.index"), "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        assert dbutil.retrieve_start_end_by_idx(meta["idx"], mm).tolist() == meta["seq_startend"]
    se = dbutil.startend_memmap(os.path.join(sl, "ca.index"), meta["n"])
    sq = dbutil.startend_memmap(os.path.join(sl, "seq.index"), meta["n"])
    assert np.array_equal(se[:, 1] - se[:, 0], 12 * (sq[:, 1] - sq[:, 0]))     # 12 bytes of CA per residue
    assert set(meta["ted100_json"]) == {"dbfname_IP", "DB_SIZE", "DB_DIM", "db_names_f", "sif", "sdf", "cif", "cdf", "mif", "mdf"}
    blob = ds._Blob(os.path.join(sl, "cath10.metadata.index"), os.path.join(sl, "cath10.metadata"))
    assert blob.fetch([0], dbutil.ascii_conv)[0] == meta["cath_first"]
    blob.close()


def test_faiss_layout_writer_roundtrip(tmp_path):
    names, coords, seqs = syn.synthetic_structures(25, seed=3)
    short = [os.path.basename(n).replace(".pdb", "") for n in names]
    emb = syn.normalized_database(25, seed=4)
    md = ['{"id": %d}' % i for i in range(25)]
    path = dbutil.write_faiss_db(str(tmp_path / "mini"), emb, short, seqs, coords, md)
    info = dbutil.read_dbinfo(path)
    assert info["DB_SIZE"] == 25 and info["DB_DIM"] == 128
    mmx = dbutil.db_memmap(str(tmp_path / info["dbfname_IP"]), (25, 128))
    assert np.array_equal(np.asarray(mmx), emb)
    assert os.path.getsize(tmp_path / info["db_names_f"]) == 33 * 25
    blocks = list(dbutil.db_iterator(mmx, 10))
    assert [b.shape[0] for b in blocks] == [10, 10, 5]
    with open(tmp_path / info["db_names_f"], "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        assert dbutil.retrieve_names_by_idx([24, 0], mm).tolist() == [short[24], short[0]]
    ca = ds._Blob(str(tmp_path / info["cif"]), str(tmp_path / info["cdf"]))
    got = ca.fetch([7, 3], dbutil.coord_conv)
    assert np.array_equal(got[0], coords[7]) and np.array_equal(got[1], coords[3])
    sq = ds._Blob(str(tmp_path / info["sif"]), str(tmp_path / info["sdf"]))
    assert sq.fetch([5], dbutil.ascii_conv) == [seqs[5]]
    mdb = ds._Blob(str(tmp_path / info["mif"]), str(tmp_path / info["mdf"]))
    assert mdb.fetch([9], dbutil.ascii_conv) == [md[9]]
    for b in (ca, sq, mdb):
        b.close()


# ------------------------------------------------------------------ G4 writers ---------
def _g4_results():
    return [
        {0: dict(query="q1", target="t1", score=np.float32(0.82041), q_len=31, t_len=45, tmalign_output=None,
                 dom_str=None, dom_conf=None, dom_plddt=None, dbindex=7, metadata='{"cath": "3.30.470.20"}'),
         3: dict(query="q1", target="t2", score=np.float32(-0.0), q_len=31, t_len=100, tmalign_output=None,
                 dom_str=None, dom_conf=None, dom_plddt=None, dbindex=9, metadata="{ }")},
        {},
        {0: dict(query="q3", target="AF-X", score=np.float32(0.5), q_len=120, t_len=118,
                 tmalign_output=dict(len_ali=110, rmsd=2.345, seq_id=0.1234, qtm=0.71234, ttm=0.69),
                 dom_str="1-120", dom_conf=0.91234, dom_plddt=88.12345, dbindex=3, metadata="{ }")},
    ]


def test_tsv_writers_match_reference_text(golden_dir, tmp_path):
    res = _g4_results()
    out = str(tmp_path / "a.tsv")
    results.write_search_results(res[:2], out, EMB_FMT, header=True)
    assert open(out).read() == open(os.path.join(golden_dir, "tsv_emb_only.tsv")).read()
    results.write_search_results(res[:2], out, EMB_FMT, header=False)
    assert open(out).read() == open(os.path.join(golden_dir, "tsv_noheader.tsv")).read()
    full = results.EASY_SEARCH_FIELDS.split(",")
    results.write_search_results(res[2:], out, full, header=True)
    assert open(out).read() == open(os.path.join(golden_dir, "tsv_full.tsv")).read()
    seg = [dict(name="/x/AF-Q96PD2-F1-model_v4.pdb", length=775, nres_domain=383, nres_non_domain=392,
                num_domains=3, conf=0.81234, time=0.7174, dom_str="71-189,190-290,291-453")]
    results.write_segment_results(seg, out, header=True)
    assert open(out).read() == open(os.path.join(golden_dir, "tsv_segment.tsv")).read()
    results.write_search_results(res[:1], out, EMB_FMT, header=False, metadata_json=True)
    assert json.load(open(out + ".hit_metadata.json")) == [{"cath": "3.30.470.20"}]


def test_format_and_database_checks(tmp_path):
    assert results.parse_output_format("query,target", results.SEARCH_FIELDS) == ["query", "target"]
    with pytest.raises(SystemExit):
        results.parse_output_format("query,bogus", results.SEARCH_FIELDS)
    with pytest.raises(SystemExit):
        results.check_for_database(str(tmp_path / "nothing"))
    (tmp_path / "x.json").write_text("{}")
    results.check_for_database(str(tmp_path / "x"))
    with pytest.raises(SystemExit):
        ds.read_database(str(tmp_path / "nothing"))


# ------------------------------------------------------------------ G6 `.pt` driver ----
@pytest.fixture(scope="module")
def pt_database(tmp_path_factory, golden_dir):
    """The synthetic `.pt` database of gen_golden.g6_dbsearch, rebuilt through OUR createdb-side
    writer from the reference's own embeddings (so the drivers are tested in isolation)."""
    tmp = tmp_path_factory.mktemp("ptdb")
    g = np.load(os.path.join(golden_dir, "dbsearch.npz"))
    names, coords, seqs = syn.synthetic_structures(40, seed=31, min_len=20, max_len=90)
    for name in ("M0", "3w5h"):
        p = np.load(os.path.join(golden_dir, f"pdb_{name}.npz"))
        names.append(f"/db/{name}.pdb"); coords.append(p["coords"]); seqs.append(str(p["seq"]))
    dbutil.write_pt_db(str(tmp / "syn"), g["db_emb"], names, coords, seqs)
    return str(tmp / "syn"), g


@pytest.mark.parametrize("mincov", [0.0, 0.7])
def test_pt_driver_end_to_end_matches_reference_tsv(mincov, pt_database, golden_dir, tmp_path):
    """run_dbsearch (embed -> cosine+mask top-k -> hit assembly -> TSV) == the reference's
    dbsearch(..., skip_tmalign=True) + write_search_results text."""
    db_prefix, g = pt_database
    net = oracle_network(0)
    queries = []
    for q in ("M0", "3w5h"):
        p = np.load(os.path.join(golden_dir, f"pdb_{q}.npz"))
        queries.append(dict(coords=p["coords"], seq=str(p["seq"]), name=f"/q/{q}.pdb"))
    res, all_res = ds.run_dbsearch(queries, db_prefix, str(tmp_path / "tmp"), "cpu", topk=5, fastmode=False, threads=-1,
                                   mincos=-1.0, mintm=0.5, mincov=mincov, inputs_are_ca=True, skip_tmalign=True, network=net)
    assert len(res) == 2 and all_res == [{}, {}]
    for qi, q in enumerate(("M0", "3w5h")):
        assert [int(h["dbindex"]) for h in res[qi].values()] == g[f"dbindex_{q}_cov{mincov}"].tolist()
        np.testing.assert_allclose([float(h["score"]) for h in res[qi].values()], g[f"scores_{q}_cov{mincov}"], atol=2e-6)
    out = str(tmp_path / "o.tsv")
    results.write_search_results(res, out, EMB_FMT, header=True)
    assert open(out).read() == open(os.path.join(golden_dir, f"dbsearch_cov{mincov}.tsv")).read()


def test_pt_driver_single_query_shapes_and_k_too_large(pt_database, golden_dir):
    db_prefix, g = pt_database
    net = oracle_network(0)
    td = ds.read_database(db_prefix, engine=net.engine)
    assert td["faiss"] is False and td["database"].shape == (42, 128) and td["lengths"].shape == (42,)
    p = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    qd = {"seq": str(p["seq"]), "embedding": net(torch.from_numpy(p["coords"]).unsqueeze(0))}
    r = ds.search_query_against_db(qd, td, 0.7, 5)
    assert r["scores"].shape == (5,) and r["indices"].dtype == torch.int64
    with pytest.raises(RuntimeError):
        ds.search_query_against_db(qd, td, 0.7, 43)


def test_pt_driver_mincos_filter_and_metadata(pt_database, golden_dir, tmp_path):
    db_prefix, g = pt_database
    # add metadata side files: the driver must pick them up (dbsearch.py:59-62)
    md = ['{"n": %d}' % i for i in range(42)]
    names, coords, seqs = [], [], []
    with open(db_prefix + ".index", "rb") as f:
        for n, c, s in pickle.load(f):
            names.append(n); coords.append(c); seqs.append(s)
    pre = str(tmp_path / "withmd")
    dbutil.write_pt_db(pre, g["db_emb"], names, coords, seqs, metadata=md)
    net = oracle_network(0)
    p = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    q = [dict(coords=p["coords"], seq=str(p["seq"]), name="M0.pdb")]
    res, _ = ds.run_dbsearch(q, pre, str(tmp_path / "t"), "cpu", 5, False, -1, mincos=0.9975, mintm=0.5, mincov=0.7,
                             inputs_are_ca=True, skip_tmalign=True, network=net)
    ranks = list(res[0].keys())
    assert ranks == [0, 1, 2, 3]                       # 0.9964 at rank 4 is below mincos; keys = top-k positions
    assert res[0][1]["metadata"] == md[int(res[0][1]["dbindex"])]


# ------------------------------------------------------------------ faiss-layout driver -
def test_faiss_driver_c1_plumbing(golden_dir, tmp_path):
    """BASELINE config C1 (plumbing): search M0 against a faiss-layout database built by our
    writer; blockwise == one-shot; ranks are a dense counter; zero hits do not crash."""
    net = oracle_network(0)
    names, coords, seqs = syn.synthetic_structures(60, seed=77, min_len=20, max_len=60)
    p = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    names.append("/db/M0.pdb"); coords.append(p["coords"]); seqs.append(str(p["seq"]))
    pdbdir = tmp_path / "pdbs"
    pdbdir.mkdir()
    for n, c, s in zip(names, coords, seqs):
        os.replace(pdbio.write_pdb(str(pdbdir), c, s, name=os.path.basename(n).replace(".pdb", "")),
                   str(pdbdir / os.path.basename(n)))
    count = run_createdb(str(pdbdir), str(tmp_path / "mini"), network=net, layout="both")
    assert count == 61 and os.path.exists(tmp_path / "mini.pt") and os.path.exists(tmp_path / "mini.index")
    os.remove(tmp_path / "mini.pt")                    # force the faiss-layout path (`.pt` wins otherwise)
    query = [dict(coords=p["coords"], seq=str(p["seq"]), name="M0.pdb")]
    out = {}
    for bs in (262144, 7):
        res, all_res = ds.run_dbsearch(query, str(tmp_path / "mini"), str(tmp_path / "t"), "cpu", 5, False, -1,
                                       mincos=0.0, mintm=0.5, mincov=0.7, inputs_are_ca=True, search_batchsize=bs,
                                       skip_tmalign=True, network=net)
        out[bs] = [(h["target"], round(float(h["score"]), 6), int(h["dbindex"]), h["t_len"]) for h in res[0].values()]
        assert list(res[0].keys()) == [0, 1, 2, 3, 4]
    assert out[262144] == out[7]
    assert out[7][0][0] == "M0" and abs(out[7][0][1] - 1.0) < 1e-5 and out[7][0][3] == 31
    res, all_res = ds.run_dbsearch(query, str(tmp_path / "mini"), str(tmp_path / "t"), "cpu", 5, False, -1,
                                   mincos=1.5, mintm=0.5, mincov=0.7, inputs_are_ca=True, skip_tmalign=True, network=net)
    assert res == [{}] and all_res == [{}]


def test_knn_exact_matches_oracle_blockwise():
    eng = OracleEngine()
    from oracle import oracle as orc
    db = syn.normalized_database(3000, seed=5)
    q = syn.normalized_database(9, seed=6)
    D, I = ds.knn_exact(q, dbutil.db_iterator(db, 1000), 10, eng)
    Dr, Ir = orc.knn_exact_blockwise(db, q, 10, block=1000)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    D, I = ds.knn_exact(q, iter(()), 10, eng)
    assert (I == -1).all() and np.isneginf(D).all()


# ------------------------------------------------------------------ easy-search hand-off
def test_chopping_handoff_matches_readme_numbers(golden_dir):
    pdb = os.path.join(golden_dir, "AF-Q96PD2-F1-model_v4_ca.pdb")
    doms = chop.domains_from_chopping(pdb, "71-189,190-290,291-453")
    assert [d["name"] for d in doms] == [f"AF-Q96PD2-F1-model_v4_ca_merizo_0{i}" for i in (1, 2, 3)]
    assert [len(d["seq"]) for d in doms] == [119, 101, 163] and doms[0]["coords"].dtype == np.float32
    row = chop.segment_row(pdb, "71-189,190-290,291-453")
    assert (row["length"], row["nres_domain"], row["nres_non_domain"], row["num_domains"]) == (775, 383, 392, 3)  # README.md:128
    d2 = chop.domains_from_chopping(pdb, "10-20_40-45,600")
    assert len(d2[0]["seq"]) == 17 and len(d2[1]["seq"]) == 1 and d2[0]["dom_str"] == "10-20_40-45"
    assert chop.parse_chopping("-5-3,7") == [[range(-5, 4)], [range(7, 8)]]


def test_device_cpu_is_refused():
    from merizo_search_amd.foldclass.engine import resolve_device
    assert resolve_device("cuda") == "cuda:0" and resolve_device("cuda:3") == "cuda:3"
    with pytest.raises(SystemExit):
        resolve_device("cpu")


def test_c1_real_size_search_through_the_cli_on_the_oracle_engine(tmp_path, monkeypatch, golden_dir):
    """BASELINE config C1 at its real size: `search examples/M0.pdb` against the shipped TED example
    database layout (66,943 entries; name / offset files byte-identical to the shipped ones, payloads
    synthesised to the shipped byte counts), CPU plumbing run with the oracle engine."""
    import c1_case
    from oracle_engine import oracle_network
    from merizo_search_amd import cli
    from merizo_search_amd.foldclass import dbsearch as ds
    net = oracle_network()
    m0 = os.path.join(golden_dir, "M0_ca.pdb")
    p = np.load(os.path.join(golden_dir, "pdb_M0.npz"))
    e = net.embed_many([p["coords"]]).numpy()[0]
    e = (e / np.linalg.norm(e)).astype(np.float32)
    rng = np.random.default_rng(1)
    planted = {}
    for j, row in enumerate((66942, 0, 31337)):                # best, second, third: first / last rows included
        v = e + 0.02 * (j + 1) * rng.standard_normal(128).astype(np.float32) / np.sqrt(128)
        planted[row] = (v / np.linalg.norm(v)).astype(np.float32)
    prefix = c1_case.build(str(tmp_path / "db"), plant=planted)
    monkeypatch.setattr(ds, "network_setup", lambda **kw: (net, "cpu"))
    cli.main(["search", m0, prefix, str(tmp_path / "out"), str(tmp_path / "tmp"), "-k", "5", "-s", "-1", "--output_headers",
              "--format", "query,emb_rank,target,emb_score,q_len,t_len,metadata"])
    c1_case.check_search(str(tmp_path / "out"), [66942, 0, 31337], 5)


def test_checkpoint_round_trip_through_network_setup(tmp_path):
    """The real-checkpoint branch (reference dbsearch.py:43: torch.load of a state_dict, strict=False):
    torch.save -> find_checkpoint -> load_checkpoint -> pack_state_dict reproduces the weights, ignores
    extra keys, and network_setup embeds with them."""
    import torch
    from oracle_engine import OracleEngine
    from merizo_search_amd.foldclass import network as nw, weights as W, synthetic as syn
    sd = W.synthetic_state_dict(3)
    ckpt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    ckpt["not_a_model_key"] = torch.zeros(3)                            # strict=False: ignored
    path = str(tmp_path / nw.WEIGHTS_NAME)
    torch.save(ckpt, path)
    w0, pe0 = W.pack_state_dict(sd)
    w1, pe1 = W.pack_state_dict(W.load_checkpoint(path))
    assert np.array_equal(w0, w1) and np.array_equal(pe0, pe1)
    assert nw.find_checkpoint(path) == path
    net, _dev = nw.network_setup(device="cuda", weights_path=path, engine=OracleEngine())
    coords = syn.random_walk(40, seed=4)
    want = nw.FoldClassEncoder(OracleEngine(sd)).embed_many([coords]).numpy()
    assert np.array_equal(net(coords[None]).numpy(), want)
    del ckpt["encode_ca_egnn.0.edge_mlp.0.weight"]
    torch.save(ckpt, path)
    with pytest.raises(KeyError):
        W.pack_state_dict(W.load_checkpoint(path))
