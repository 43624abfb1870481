"""A small multi-domain search scenario shared by the CPU and GPU CLI tests (TEST INFRASTRUCTURE).

Database chains (TED naming, so that domid2chainid groups them):
    AF-T1-F1-model_v4   TED01 (= query domain 1, 40 residues), TED02 (= query domain 2, 60), TED03 (30)
    AF-T2-F1-model_v4   TED01 (50)                      single-domain chain: can never match two query domains
    AF-T3-F1-model_v4   TED01 (= query domain 2), TED02 (= query domain 1)    same domains, order swapped
Query: one chain `Q.pdb` of 100 residues, chopping "1-40,41-100".
The stand-in aligner scores 0.9 for two structures with the same number of CA atoms, else 0.2.
"""
import os
import stat

import numpy as np

from merizo_search_amd.foldclass import pdbio, synthetic as syn

CHOPPING = "1-40,41-100"
T1 = "Q_merizo_01:AF-T1-F1-model_v4_TED01:0.9,Q_merizo_02:AF-T1-F1-model_v4_TED02:0.9"
T3 = "Q_merizo_01:AF-T3-F1-model_v4_TED02:0.9,Q_merizo_02:AF-T3-F1-model_v4_TED01:0.9"


def fake_tmalign(tmp_path) -> str:
    script = os.path.join(str(tmp_path), "tmalign")
    with open(script, "w") as fh:
        fh.write("""#!/usr/bin/env python3
import sys
n = [sum(1 for l in open(f) if l.startswith("ATOM")) for f in sys.argv[1:3]]
tm = 0.9 if n[0] == n[1] else 0.2
print("Aligned length=  %d, RMSD=   1.00, Seq_ID=n_identical/n_aligned= 0.500" % min(n))
print("TM-score= %.5f (if normalized by length of Chain_1)" % tm)
print("TM-score= %.5f (if normalized by length of Chain_2)" % tm)
""")
    os.chmod(script, os.stat(script).st_mode | stat.S_IEXEC)
    return script


def write_inputs(tmp_path):
    """-> (query pdb path, directory of database pdbs).  Coordinates are multiples of 0.001 so that they
    survive the PDB text format unchanged: query domains and their database copies embed identically."""
    tmp = str(tmp_path)
    walk = lambda n, seed: np.round(syn.random_walk(n, seed).astype(np.float64), 3).astype(np.float32)
    whole = walk(100, 5)
    d1, d2 = whole[:40], whole[40:]
    seq = ("ACDEFGHIKLMNPQRSTVWY" * 5)
    qdir = os.path.join(tmp, "query"); dbdir = os.path.join(tmp, "dbpdbs")
    os.makedirs(qdir); os.makedirs(dbdir)
    qpdb = pdbio.write_pdb(qdir, whole, seq, name="Q")
    entries = {"AF-T1-F1-model_v4_TED01": (d1, seq[:40]), "AF-T1-F1-model_v4_TED02": (d2, seq[40:]),
               "AF-T1-F1-model_v4_TED03": (walk(30, 6), seq[:30]), "AF-T2-F1-model_v4_TED01": (walk(50, 7), seq[:50]),
               "AF-T3-F1-model_v4_TED01": (d2, seq[40:]), "AF-T3-F1-model_v4_TED02": (d1, seq[:40])}
    for name, (c, s) in entries.items():
        pdbio.write_pdb(dbdir, c, s, name=name)
    return qpdb, dbdir


def check_outputs(prefix: str) -> None:
    rows = [l.rstrip("\n").split("\t") for l in open(prefix + "_search_multi_dom.tsv")]
    assert rows[0] == ["query_chain", "nqd", "hit_chain", "nhd", "match_category", "match_info", "hit_metadata"]
    got = {(r[2], r[4]): r for r in rows[1:]}
    # T1: both query domains, in order, contiguous, fewer query than hit domains -> category 2
    assert got[("AF-T1-F1-model_v4", "2")][:4] == ["Q", "2", "AF-T1-F1-model_v4", "3"] and got[("AF-T1-F1-model_v4", "2")][5] == T1
    # T3: both query domains, order swapped -> category 0; the single-domain chain T2 never appears
    assert got[("AF-T3-F1-model_v4", "0")][:4] == ["Q", "2", "AF-T3-F1-model_v4", "2"] and got[("AF-T3-F1-model_v4", "0")][5] == T3
    assert not any(r[2] == "AF-T2-F1-model_v4" for r in rows[1:])
    hits = [l.split("\t") for l in open(prefix + "_search.tsv").read().splitlines()[1:]]
    top = {(h[0], h[4]): h[5] for h in hits}          # (query domain, emb_rank) -> target
    assert {top[("Q_merizo_01", "0")], top[("Q_merizo_01", "1")]} == {"AF-T1-F1-model_v4_TED01", "AF-T3-F1-model_v4_TED02"}
    assert {top[("Q_merizo_02", "0")], top[("Q_merizo_02", "1")]} == {"AF-T1-F1-model_v4_TED02", "AF-T3-F1-model_v4_TED01"}
