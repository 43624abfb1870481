"""Multi-domain ("full-length") search on top of the per-domain top-k results.

Mirror of programs/Foldclass/dbsearch_fulllength.py.  Nothing here is on the GPU path: the
inputs are the hit dictionaries the (GPU) per-domain search produced, the work is
  1. group the query domains by query chain and their hits by target chain
     (dbsearch_fulllength.py:230-272);
  2. for every hit, walk the database index left and right to collect the sibling domains of
     the hit's chain -- database rows of one chain are adjacent (:346-394);
  3. TM-align every query domain of a chain against every collected target domain
     (external binary, a process pool; scores below mintm -> 0) (:55-92, :468-483);
  4. per (query chain, target chain) sub-matrix, enumerate the one-to-one assignments of query
     domains to target domains and classify them 0-3 (:95-180).
Step 3 needs a TM-align executable ($MERIZO_TMALIGN): without one the reference cannot run this
mode either, and `multi_domain_search` raises.  Steps 1, 2 and 4 are plain functions, tested
against outputs of the reference's own functions (tests/golden/multidomain.json).
"""
from __future__ import annotations

import itertools
import logging
import mmap
import os
import re
import shutil
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import dbutil
from .pdbio import read_pdb, write_pdb
from .tmalign import find_tmalign, run_tmalign

logger = logging.getLogger(__name__)

FIELD_SET_SEPARATOR = ","       # between the per-domain entries of one mapping
FIELD_SEPARATOR = ":"           # inside an entry: query domain : hit domain : TM score

_TWO_DIGITS = re.compile(r"[0-9]{2}$")
_MERIZO_SUFFIX = re.compile(r"_merizo_[0-9]*$")


def domid2chainid(domain_id: str) -> str:
    """'cath-dompdb/2pi4A04.pdb' -> '2pi4A'; 'x/AF-Q93009-F1-model_v4_TED02.pdb' -> 'AF-Q93009-F1-model_v4'.

    Same steps as the reference (:36-39), including its use of str.rstrip('.pdb'), which strips
    any trailing run of the characters '.', 'p', 'd', 'b' rather than the literal suffix."""
    stem = os.path.basename(domain_id).rstrip(".pdb")
    stem = _TWO_DIGITS.sub("", stem).rstrip("_")
    return stem[: -len("_TED")] if stem.endswith("_TED") else stem


def chain_mappings(tm: np.ndarray, query_chain: str, hit_chain: str, query_domains: Sequence[str],
                   hit_domains: Sequence[dict]) -> List[tuple]:
    """All one-to-one assignments of a chain's query domains (rows of `tm`) to the domains of one
    target chain (columns; dicts with 'hd' name and 'hm' metadata) whose TM scores are non-zero.

    Returns tuples (query_chain, nqd, hit_chain, nhd, category, 'qd:hd:tm,...', '[metadata,...]')
    in the reference's enumeration order (:95-180).  category: 3 = same number of domains, same
    order; 2 = order kept, contiguous target domains; 1 = order kept with gaps; 0 = any order."""
    tm = np.asarray(tm)
    nqd, nhd = tm.shape
    assert len(query_domains) == nqd and len(hit_domains) == nhd
    nonzero = tm != 0
    if not nonzero.any(axis=1).all():                 # a query domain matches nothing in this chain
        return []
    if int(nonzero.any(axis=0).sum()) < nqd:          # fewer matching target domains than query domains
        return []
    choices = [np.flatnonzero(nonzero[row]).tolist() for row in range(nqd)]
    out = []
    for path in itertools.product(*choices):
        if len(set(path)) != nqd:                     # two query domains on the same target domain
            continue
        steps = np.diff(path)
        if nqd > 1 and not (steps > 0).all():
            category = 0
        elif nqd == nhd:
            category = 3
        elif (steps == 1).all():
            category = 2
        else:
            category = 1
        info = [FIELD_SEPARATOR.join([query_domains[q], hit_domains[c]["hd"], str(tm[q, c])]) for q, c in enumerate(path)]
        meta = [hit_domains[c]["hm"] for c in path]
        out.append((query_chain, nqd, hit_chain, nhd, category, FIELD_SET_SEPARATOR.join(info),
                    "[" + FIELD_SET_SEPARATOR.join(meta) + "]"))
    return out


def sibling_rows(anchor: int, chain: str, n_rows: int, name_of: Callable[[int], str]) -> List[int]:
    """Database rows of the other domains of `chain` around row `anchor`, then `anchor` itself --
    empty if the chain has a single domain (:363-394).  Rows of a chain are adjacent; the walk
    stops at the database ends (the reference indexes past them)."""
    rows = []
    i = anchor - 1
    while i >= 0 and domid2chainid(name_of(i)) == chain:
        rows.append(i)
        i -= 1
    i = anchor + 1
    while i < n_rows and domid2chainid(name_of(i)) == chain:
        rows.append(i)
        i += 1
    if rows:
        rows.append(anchor)
    return rows


def tm_matrix(query_files: Sequence[str], target_files: Sequence[str], threads: int = -1, mintm: float = 0.5,
              options: Optional[str] = None, runner: Callable = run_tmalign) -> np.ndarray:
    """max(TM by query, TM by target) for every (query, target) pair; values below mintm -> 0
    (:55-92).  Pairs run concurrently (each one is a TM-align subprocess)."""
    pairs = [(q, t) for q in query_files for t in target_files]
    if not pairs:
        return np.zeros((len(query_files), len(target_files)))
    workers = threads if threads and threads > 0 else min(len(pairs), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        results = list(pool.map(lambda p: runner(p[0], p[1], options, True), pairs))
    scores = np.asarray([max(r["qtm"], r["ttm"]) for r in results], dtype=np.float64)
    scores = scores.reshape(len(query_files), len(target_files))
    scores[scores < mintm] = 0.0
    return scores


def group_hits(query_names: Sequence[str], query_chains: Sequence[str], search_results) -> Dict[str, Dict[str, list]]:
    """{query chain: {query domain: [{'hc','hd','hi'}, ...]}} from the per-domain search results
    (a list of {rank: hit dict} as run_dbsearch returns them) (:246-272)."""
    index: Dict[str, Dict[str, list]] = {}
    chain_of = {}
    for qc, qd in zip(query_chains, query_names):
        index.setdefault(qc, {}).setdefault(qd, [])
        chain_of[qd] = qc
    for per_query in search_results:
        for hit in per_query.values():
            qd = hit["query"]
            index[chain_of[qd]][qd].append({"hc": domid2chainid(hit["target"]), "hd": hit["target"], "hi": int(hit["dbindex"])})
    return index


class _TargetStore:
    """Names, coordinates, sequences and metadata of database rows, for either on-disk layout."""

    def __init__(self, db_name: str):
        self.maps = []
        self.meta = None
        if os.path.exists(db_name + ".pt"):
            import pickle

            self.faiss = False
            with open(db_name + ".index", "rb") as handle:
                self.index = pickle.load(handle)
            self.n = len(self.index)
            mdfn = db_name + ".metadata"
            if os.path.exists(mdfn) and os.path.exists(mdfn + ".index"):
                self.meta = (self._map(mdfn + ".index"), self._map(mdfn))
        else:
            self.faiss = True
            info = dbutil.read_dbinfo(db_name + ".json")
            folder = os.path.dirname(db_name + ".json")
            path = lambda key: os.path.join(folder, info[key])
            self.names = self._map(path("db_names_f"))
            self.n = len(self.names) // dbutil.NAME_RECORD
            self.seq = (self._map(path("sif")), self._map(path("sdf")))
            self.coords = (self._map(path("cif")), self._map(path("cdf")))
            if "mif" in info and "mdf" in info:
                self.meta = (self._map(path("mif")), self._map(path("mdf")))

    def _map(self, filename):
        handle = open(filename, "rb")
        mm = mmap.mmap(handle.fileno(), 0, access=mmap.ACCESS_READ)
        self.maps.append((mm, handle))
        return mm

    def close(self):
        for mm, handle in self.maps:
            mm.close()
            handle.close()

    def name(self, row: int) -> str:
        if self.faiss:
            return str(dbutil.retrieve_names_by_idx([row], self.names)[0])
        return self.index[row][0]

    def _blob(self, pair, row, conv):
        start, end = dbutil.retrieve_start_end_by_idx([row], pair[0])[0]
        return dbutil.retrieve_bytes(start, end, mm=pair[1], typeconv=conv)

    def entry(self, row: int):
        """(domain name, coords [N,3], sequence, row, metadata json) of a database row (:426-466)."""
        metadata = self._blob(self.meta, row, dbutil.ascii_conv) if self.meta is not None else "{ }"
        if self.faiss:
            return (self.name(row), self._blob(self.coords, row, dbutil.coord_conv), self._blob(self.seq, row, dbutil.ascii_conv),
                    row, metadata)
        name, coords, seq = self.index[row][:3]
        return (os.path.basename(name).replace(".pdb", ""), coords, seq, row, metadata)


def multi_domain_search(queries, search_results, db_name: str, tmp_root: str, device=None, fastmode: bool = False,
                        threads: int = -1, mintm: float = 0.5, inputs_from_easy_search: bool = False,
                        mode: str = "exhaustive_tmalign", pdb_chain: Optional[str] = None):
    """The reference's multi_domain_search (:183-574): same arguments, same result tuples (feed
    them to results.write_all_dom_search_results).  `queries`: PDB file names (search) or domain
    dicts with 'coords', 'seq', 'name' (easy-search)."""
    if mode != "exhaustive_tmalign":
        raise ValueError("Unrecognised multi-domain search mode: " + mode)
    if len(queries) == 1:
        logger.warning("Cannot execute multi-domain search with only one query domain.")
        return None
    if find_tmalign() is None:
        raise FileNotFoundError("multi-domain search aligns every query domain with every candidate target domain: "
                                "it needs a TM-align binary (set $MERIZO_TMALIGN)")
    if not inputs_from_easy_search:
        chains = pdb_chain.rstrip(",").split(",") if pdb_chain else ["A"] * len(queries)
        queries = [read_pdb(pdbfile=q, pdb_chain=c) for q, c in zip(queries, chains)]
    names = [os.path.basename(q["name"]) for q in queries]
    names = [n[: -len(".pdb")] if n.endswith(".pdb") else n for n in names]
    structures = {n: q for n, q in zip(names, queries)}
    query_chains = [_MERIZO_SUFFIX.sub("", n) for n in names] if inputs_from_easy_search else ["A"] * len(names)
    hits = group_hits(names, query_chains, search_results)

    store = _TargetStore(db_name)
    results = []
    try:
        for qc, domains in hits.items():
            nqd = len(domains)
            if nqd < 2:
                logger.info("Query chain %s: only one detected domain, multi-domain hits equal the per-domain hits." % qc)
                continue
            rows = set()
            for per_domain in domains.values():
                for hit in per_domain:
                    chain_rows = sibling_rows(hit["hi"], hit["hc"], store.n, store.name)
                    if len(chain_rows) >= nqd:                 # target chains with fewer domains cannot match
                        rows.update(chain_rows)
            if not rows:
                logger.info("Query chain %s: every hit chain has fewer domains than the query; try a larger -k." % qc)
                continue
            entries = [store.entry(r) for r in sorted(rows)]
            tmp = os.path.join(tmp_root, "MD_search_structures_" + qc)
            os.makedirs(tmp, exist_ok=True)
            try:
                qfiles = [write_pdb(tmp, structures[qd]["coords"], structures[qd]["seq"], name="FSQUERY-" + qd) for qd in domains]
                tfiles = [write_pdb(tmp, e[1], e[2], name="FSTARGET-" + e[0]) for e in entries]
                logger.info("TM-align %d query domains of chain %s against %d target domains" % (nqd, qc, len(tfiles)))
                scores = tm_matrix(qfiles, tfiles, threads=threads, mintm=mintm, options="-fast" if fastmode else None)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
            hit_chain = np.asarray([domid2chainid(e[0]) for e in entries])
            info = [{"hd": e[0], "hc": hc, "hi": e[3], "hm": e[4]} for e, hc in zip(entries, hit_chain)]
            qds = list(domains.keys())
            for hc in np.unique(hit_chain):
                cols = np.flatnonzero(hit_chain == hc)
                results.extend(chain_mappings(scores[:, cols], qc, str(hc), qds, [info[c] for c in cols]))
            logger.info("Finished multi-domain search for query chain %s." % qc)
    finally:
        store.close()
    return results
