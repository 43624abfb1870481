#!/usr/bin/env python
"""Generate tests/golden/* by running the REFERENCE (imported from /root/reference) on CPU.

Run in the build container only (the reference never travels):

    python oracle/gen_golden.py

What is written is data only: inputs and the reference's outputs (npz / json / tsv / CA-only
PDB records).  No reference source is copied.  Seeds and generators for the synthetic inputs
live in merizo_search_amd.foldclass.synthetic so tests can rebuild them; each fixture stores
a sha256 of its regenerated inputs so drift in a generator is detected.

Goldens:
  G1 egnn.npz        FoldClassNet embeddings (+ per-layer node features for M0) with the
                     synthetic weights (seed 0) for the 4 example PDBs and synthetic walks.
  G2 search.npz      search_query_against_db scores/indices (mask on/off, k in {1,10,100},
                     all-masked query, k == Ndb).
  G3 normalize.npz   F.normalize / cosine eps behaviour incl. a zero row.
  G4 tsv_*.tsv       write_search_results text.
  G5 pdb_*.npz       read_pdb outputs; *_ca.pdb hold the examples' CA records (M0.pdb whole).
  G6 dbsearch_*.tsv  dbsearch(..., skip_tmalign=True) end-to-end on a synthetic .pt DB.
  G7 dbutil.json     retrieve_* results on a slice of the shipped ted100 index files
                     (slice copied to tests/golden/ted100_slice/).
"""
import hashlib
import json
import os
import pickle
import shutil
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "merizo_search"))
sys.path.insert(0, REPO)

from programs.Foldclass.nndef_fold_egnn_embed import FoldClassNet  # noqa: E402
from programs.Foldclass import dbsearch as ref_ds  # noqa: E402
from programs.Foldclass import dbutil as ref_dbutil  # noqa: E402
from programs.Foldclass.utils import read_pdb as ref_read_pdb  # noqa: E402
from programs import utils as ref_fmt  # noqa: E402

from merizo_search_amd.foldclass import weights as W  # noqa: E402
from merizo_search_amd.foldclass import synthetic as syn  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
EXAMPLES = ["M0", "3w5h", "AF-Q96HM7-F1-model_v4", "AF-Q96PD2-F1-model_v4"]


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def ref_network(seed=0):
    sd = W.synthetic_state_dict(seed)
    net = FoldClassNet(128).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return net, sd


def g5_pdb():
    """read_pdb outputs + CA-record fixtures."""
    out = {}
    for name in EXAMPLES:
        src = os.path.join(REF, "examples", name + ".pdb")
        d = ref_read_pdb(src, "A")
        np.savez_compressed(os.path.join(OUT, f"pdb_{name}.npz"), coords=d["coords"], seq=np.array(d["seq"]))
        dst = os.path.join(OUT, f"{name}_ca.pdb")
        if name == "M0":
            shutil.copyfile(src, os.path.join(OUT, "M0.pdb"))   # small data file, kept whole
        with open(src) as f, open(dst, "w") as g:
            for line in f:
                # keep every CA ATOM record (all chains) plus a few non-CA / non-ATOM records so
                # the parser's filters are exercised
                if line.startswith("ATOM") and line[12:16] == " CA ":
                    g.write(line)
                elif line.startswith(("HEADER", "TER", "END")):
                    g.write(line)
        d2 = ref_read_pdb(dst, "A")
        assert np.array_equal(d2["coords"], d["coords"]) and d2["seq"] == d["seq"]
        out[name] = d
    return out


def g1_egnn(pdbs):
    net, sd = ref_network(0)
    weights, pe = W.pack_state_dict(sd)
    res = {"weights_sha": sha(weights), "pe_sha": sha(pe)}
    arrays = {}
    cases = []
    for name in EXAMPLES:
        cases.append((name, pdbs[name]["coords"]))
    for n in (1, 2, 64, 257):
        cases.append((f"walk{n}", syn.random_walk(n, seed=100 + n)))
    with torch.no_grad():
        for name, coords in cases:
            x = torch.from_numpy(coords).unsqueeze(0)
            emb = net(x)[0].numpy()
            arrays[f"emb_{name}"] = emb
            arrays[f"coords_{name}"] = coords
            print("G1", name, coords.shape, float(np.abs(emb).max()))
        # per-layer node features for M0
        x = torch.from_numpy(pdbs["M0"]["coords"]).unsqueeze(0)
        feats = net.posenc_as(x)
        h1 = net.encode_ca_egnn[0]((feats, x, None))[0]
        h2 = net.encode_ca_egnn[1]((h1, x, None))[0]
        arrays["layer1_M0"] = h1[0].numpy()
        arrays["layer2_M0"] = h2[0].numpy()
    arrays["pe_head"] = pe[:4].copy()
    arrays["pe_tail"] = pe[-4:].copy()
    np.savez_compressed(os.path.join(OUT, "egnn.npz"), **arrays)
    res["cases"] = [c[0] for c in cases]
    return res


def g2_search():
    meta = {}
    arrays = {}
    db, lengths = syn.raw_database(5000, seed=11)
    q, qlen = syn.raw_queries(8, seed=12)
    meta["db_sha"] = sha(db, lengths)
    meta["q_sha"] = sha(q, qlen)
    tdb = torch.from_numpy(db)
    td = {"database": tdb, "lengths": torch.from_numpy(lengths)}
    for mincov in (0.0, 0.7):
        for k in (1, 10, 100):
            S = np.zeros((len(q), k), np.float32)
            I = np.zeros((len(q), k), np.int64)
            for i in range(len(q)):
                qd = {"seq": "A" * int(qlen[i]), "embedding": torch.from_numpy(q[i:i + 1])}
                r = ref_ds.search_query_against_db(qd, td, mincov, k)
                S[i], I[i] = r["scores"].numpy(), r["indices"].numpy()
            arrays[f"s_cov{mincov}_k{k}"] = S
            arrays[f"i_cov{mincov}_k{k}"] = I
    # all-masked: a 10-residue query against targets of length >= 25 at mincov 0.7
    qd = {"seq": "A" * 10, "embedding": torch.from_numpy(q[:1])}
    r = ref_ds.search_query_against_db(qd, td, 0.7, 10)
    arrays["s_allmasked"] = r["scores"].numpy()
    arrays["i_allmasked"] = r["indices"].numpy()
    # k == Ndb on a small DB
    db2, len2 = syn.raw_database(50, seed=13)
    td2 = {"database": torch.from_numpy(db2), "lengths": torch.from_numpy(len2)}
    qd = {"seq": "A" * 200, "embedding": torch.from_numpy(q[1:2])}
    r = ref_ds.search_query_against_db(qd, td2, 0.7, 50)
    arrays["s_kfull"] = r["scores"].numpy()
    arrays["i_kfull"] = r["indices"].numpy()
    meta["db2_sha"] = sha(db2, len2)
    # pre-normalised DB, mincov 0: the importable torch path must agree with the faiss-path
    # restatement ("cross-checked against the importable torch path")
    dbn = F.normalize(tdb).numpy()
    tdn = {"database": torch.from_numpy(dbn), "lengths": torch.from_numpy(lengths)}
    S = np.zeros((len(q), 10), np.float32)
    I = np.zeros((len(q), 10), np.int64)
    qn = F.normalize(torch.from_numpy(q))
    for i in range(len(q)):
        qd = {"seq": "A" * int(qlen[i]), "embedding": qn[i:i + 1]}
        r = ref_ds.search_query_against_db(qd, tdn, 0.0, 10)
        S[i], I[i] = r["scores"].numpy(), r["indices"].numpy()
    arrays["s_prenorm_k10"] = S
    arrays["i_prenorm_k10"] = I
    np.savez_compressed(os.path.join(OUT, "search.npz"), **arrays)
    return meta


def g3_normalize():
    rng = np.random.default_rng(21)
    x = rng.standard_normal((16, 128)).astype(np.float32)
    x[3] = 0.0
    x[5] *= 1e-20
    x[7] *= 1e6
    y12 = F.normalize(torch.from_numpy(x)).numpy()                      # eps 1e-12 (dbsearch.py:304)
    # cosine_similarity's internal normalisation (eps 1e-8), observed through unit probes
    probes = torch.eye(128)[:4]
    cos = torch.stack([F.cosine_similarity(torch.from_numpy(x), p[None], dim=-1) for p in probes], 1).numpy()
    np.savez_compressed(os.path.join(OUT, "normalize.npz"), x=x, y12=y12, cos_probe=cos)


def g4_tsv():
    results = [
        {0: dict(query="q1", target="t1", score=torch.tensor(0.82041), q_len=31, t_len=45, tmalign_output=None,
                 dom_str=None, dom_conf=None, dom_plddt=None, dbindex=torch.tensor(7), metadata='{"cath": "3.30.470.20"}'),
         3: dict(query="q1", target="t2", score=torch.tensor(-0.0), q_len=31, t_len=100, tmalign_output=None,
                 dom_str=None, dom_conf=None, dom_plddt=None, dbindex=torch.tensor(9), metadata="{ }")},
        {},
        {0: dict(query="q3", target="AF-X", score=np.float32(0.5), q_len=120, t_len=118,
                 tmalign_output=dict(len_ali=110, rmsd=2.345, seq_id=0.1234, qtm=0.71234, ttm=0.69),
                 dom_str="1-120", dom_conf=0.91234, dom_plddt=88.12345, dbindex=3, metadata="{ }")},
    ]
    fmt_emb = "query,emb_rank,target,emb_score,q_len,t_len,metadata".split(",")
    p = os.path.join(OUT, "tsv_emb_only.tsv")
    ref_fmt.write_search_results(results[:2], p, fmt_emb, header=True)
    fmt_full = "query,chopping,conf,plddt,emb_rank,target,emb_score,q_len,t_len,ali_len,seq_id,q_tm,t_tm,max_tm,rmsd,metadata".split(",")
    p = os.path.join(OUT, "tsv_full.tsv")
    ref_fmt.write_search_results(results[2:], p, fmt_full, header=True)
    p = os.path.join(OUT, "tsv_noheader.tsv")
    ref_fmt.write_search_results(results[:2], p, fmt_emb, header=False)
    seg = [dict(name="/x/AF-Q96PD2-F1-model_v4.pdb", length=775, nres_domain=383, nres_non_domain=392,
                num_domains=3, conf=0.81234, time=0.7174, dom_str="71-189,190-290,291-453")]
    ref_fmt.write_segment_results(seg, os.path.join(OUT, "tsv_segment.tsv"), header=True)


def g6_dbsearch(pdbs):
    """End-to-end `.pt` path: createdb-style DB from the reference net, dbsearch with skip_tmalign."""
    net, _ = ref_network(0)
    names, coords, seqs = syn.synthetic_structures(40, seed=31, min_len=20, max_len=90)
    for name in ("M0", "3w5h"):
        names.append(f"/db/{name}.pdb"); coords.append(pdbs[name]["coords"]); seqs.append(pdbs[name]["seq"])
    with torch.no_grad():
        embs = torch.cat([net(torch.from_numpy(c).unsqueeze(0)) for c in coords], 0)
    tmp = tempfile.mkdtemp()
    try:
        torch.save(embs, os.path.join(tmp, "syn.pt"))
        with open(os.path.join(tmp, "syn.index"), "wb") as f:
            pickle.dump([(n, c, s) for n, c, s in zip(names, coords, seqs)], f)
        td = ref_ds.read_database(os.path.join(tmp, "syn"), "cpu")
        arrays = {"db_emb": embs.numpy()}
        for mincov in (0.0, 0.7):
            results = []
            for q in ("M0", "3w5h"):
                qd = dict(coords=pdbs[q]["coords"], seq=pdbs[q]["seq"], name=f"/q/{q}.pdb")
                res, _ = ref_ds.dbsearch(qd, td, tmp, net, topk=5, mincov=mincov, mincos=-1.0, mintm=0.5,
                                         fastmode=False, device=torch.device("cpu"), inputs_are_ca=True,
                                         skip_tmalign=True)
                results.append(res)
                arrays[f"scores_{q}_cov{mincov}"] = np.array([float(r["score"]) for r in res.values()], np.float32)
                arrays[f"dbindex_{q}_cov{mincov}"] = np.array([int(r["dbindex"]) for r in res.values()], np.int64)
            ref_fmt.write_search_results(results, os.path.join(OUT, f"dbsearch_cov{mincov}.tsv"),
                                         "query,emb_rank,target,emb_score,q_len,t_len,metadata".split(","), header=True)
        np.savez_compressed(os.path.join(OUT, "dbsearch.npz"), **arrays)
    finally:
        shutil.rmtree(tmp)


def g7_dbutil():
    """retrieve_* on a slice of the shipped ted100 name/offset files (data files of the reference)."""
    import mmap
    src = os.path.join(REF, "examples", "database", "ted100_9606_small")
    dst = os.path.join(OUT, "ted100_slice")
    os.makedirs(dst, exist_ok=True)
    nkeep = 2000
    base = "ted100_9606_small"
    with open(os.path.join(src, base + "_raw_128d.index_names"), "rb") as f:
        names = f.read(33 * nkeep)
    with open(os.path.join(dst, "names.index_names"), "wb") as f:
        f.write(names)
    meta = {"n": nkeep}
    for kind in ("seq", "ca", "metadata"):
        a = np.fromfile(os.path.join(src, f"{base}_{kind}.index"), dtype=np.int64).reshape(-1, 2)
        a[:nkeep].tofile(os.path.join(dst, f"{kind}.index"))
        meta[f"{kind}_total"] = int(a[-1, 1])
        meta[f"{kind}_n"] = int(a.shape[0])
    idx = [0, 1, 17, 1999, 5]
    with open(os.path.join(dst, "names.index_names"), "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        meta["names"] = [str(x) for x in ref_dbutil.retrieve_names_by_idx(idx, mm)]
    with open(os.path.join(dst, "seq.index"), "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        meta["seq_startend"] = [[int(v) for v in se] for se in ref_dbutil.retrieve_start_end_by_idx(idx, mm)]
    meta["idx"] = idx
    # lengths of the whole shipped slice (empirical TED length distribution used by bench/synthetic)
    a = np.fromfile(os.path.join(src, f"{base}_seq.index"), dtype=np.int64).reshape(-1, 2)
    lens = (a[:, 1] - a[:, 0]).astype(np.int32)
    hist = np.bincount(lens)
    np.savez_compressed(os.path.join(OUT, "ted_length_hist.npz"), hist=hist)
    meta["len_stats"] = dict(min=int(lens.min()), max=int(lens.max()), mean=float(lens.mean()),
                             median=float(np.median(lens)), n=int(lens.size))
    # example json (key names of the faiss layout)
    with open(os.path.join(REF, "examples", "database", "ted100.json")) as f:
        meta["ted100_json"] = json.load(f)
    # CATH metadata sample (pt-layout side files)
    mi = np.fromfile(os.path.join(REF, "examples", "database", "cath-dataset-nonredundant-S20.metadata.index"),
                     dtype=np.int64).reshape(-1, 2)
    with open(os.path.join(REF, "examples", "database", "cath-dataset-nonredundant-S20.metadata"), "rb") as f:
        blob = f.read(int(mi[9, 1]))
    with open(os.path.join(dst, "cath10.metadata"), "wb") as f:
        f.write(blob)
    mi[:10].tofile(os.path.join(dst, "cath10.metadata.index"))
    meta["cath_first"] = blob[int(mi[0, 0]):int(mi[0, 1])].decode("ascii")
    return meta


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    meta = {"torch": torch.__version__, "numpy": np.__version__}
    pdbs = g5_pdb()
    meta["g1"] = g1_egnn(pdbs)
    meta["g2"] = g2_search()
    g3_normalize()
    g4_tsv()
    g6_dbsearch(pdbs)
    meta["g7"] = g7_dbutil()
    with open(os.path.join(OUT, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
