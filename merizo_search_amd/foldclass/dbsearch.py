"""Search drivers: the host-side mirror of programs/Foldclass/dbsearch.py on the HIP engine.

Same function names, argument meaning, return shapes and on-disk layouts as the reference:

    network_setup            (network.py)        dbsearch.py:35-45
    read_database                                 dbsearch.py:48-72
    search_query_against_db                       dbsearch.py:75-81
    knn_exact                                     dbsearch.py:213-248 (knn_exact_faiss)
    dbsearch                                      dbsearch.py:84-200   `.pt` database
    dbsearch_faiss                                dbsearch.py:203-472  faiss-layout database
    run_dbsearch                                  dbsearch.py:475-551

What differs, deliberately (DESIGN.md "host drivers"):
  * all queries of a call are embedded in ONE ragged GPU launch and searched in ONE batched
    scan, instead of the reference's per-query Python loop; results are the same lists;
  * database row norms are computed once per database (the reference re-normalises the whole
    database for every query);
  * a faiss-layout database that fits in HBM is uploaded ONCE as one contiguous tensor and scanned
    with one launch per query batch; only a database larger than the resident budget is streamed
    in `search_batchsize` blocks (pinned, double-buffered: engine.device_blocks);
  * with an initialised torch.distributed process group (cli under torchrun) every rank holds and
    scans only its `sharded.shard_bounds` rows, one all-gather + merge gives every rank the global
    top-k, and rank 0 alone assembles hit records (sharded.py; replaces index_cpu_to_all_gpus,
    dbsearch.py:228-230);
  * TM-align is optional (tmalign.py): with no binary the search is embedding-only;
  * reference defects are not reproduced: a single --pdb_chain with several inputs is applied
    to every input (dbsearch.py:523-524 builds a list of lists; :296 raises IndexError); a
    faiss-layout search with zero hits returns empty lists instead of crashing (:390); the
    returned lists always have one entry per query.
"""
from __future__ import annotations

import logging
import mmap
import os
import pickle
import sys
from typing import List, Optional

import numpy as np

from . import sharded
from . import tmalign as tm
from .dbutil import (ascii_conv, coord_conv, db_iterator, db_memmap, read_dbinfo, retrieve_bytes,
                     retrieve_names_by_idx, retrieve_start_end_by_idx)
from .network import network_setup
from .pdbio import read_pdb, write_pdb

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------ database open ------
def read_database(db_name: str, device=None, engine=None) -> dict:
    """Open a database by prefix: `<db>.pt` first, else `<db>.json` (dbsearch.py:50,65).

    pt layout -> {'database': float32 [N,128] on the device (RAW), 'index': list of
    (path, coords, seq), 'lengths': float32 [N], 'faiss': False, 'mdfn', 'mifn'}; once resident on the
    engine's device the rows are kept L2-normalised (engine.cosine_rows: the row half of cosine_similarity,
    done once).  faiss layout -> {'database': json path, 'faiss': True}; the matrix is opened by dbsearch_faiss.
    """
    if os.path.exists(db_name + ".pt"):
        import torch

        try:        # memory-mapped: under N ranks every rank then reads only the pages of ITS rows (_to_engine slices first)
            raw = torch.load(db_name + ".pt", map_location="cpu", weights_only=True, mmap=True)
        except (RuntimeError, TypeError, ValueError):       # (an archive written in the legacy, non-zip format)
            raw = torch.load(db_name + ".pt", map_location="cpu", weights_only=True)
        with open(db_name + ".index", "rb") as handle:
            target_index = pickle.load(handle)
        assert len(target_index) == raw.size(0)
        lengths = np.asarray([len(entry[2]) for entry in target_index], dtype=np.float32)
        mdfn = db_name + ".metadata"
        mifn = mdfn + ".index"
        if not os.path.exists(mdfn) or not os.path.exists(mifn):
            mdfn = mifn = None
        out = {"database": raw, "index": target_index, "lengths": torch.from_numpy(lengths),
               "faiss": False, "mdfn": mdfn, "mifn": mifn}
        if engine is not None:
            _to_engine(out, engine)
        return out
    if os.path.exists(db_name + ".json"):
        return {"database": db_name + ".json", "faiss": True}
    logger.error("%s is not a valid db or the path basename is incorrect; neither %s.pt nor %s.json were found."
                 % (db_name, db_name, db_name))
    sys.exit(1)


def _to_engine(target_dict: dict, engine) -> None:
    """Make this rank's rows of the `.pt` database resident on the engine's device, normalised once for the
    cosine search.  'database' / 'lengths' then hold rows [row_lo, row_hi) only."""
    if target_dict.get("_engine") is engine:
        return
    rank, world = sharded.rank_world()
    n = int(target_dict["database"].shape[0])
    lo, hi = sharded.shard_bounds(n, world, rank)
    target_dict["n_rows"] = n
    target_dict["row_lo"], target_dict["row_hi"] = lo, hi
    target_dict["database"] = engine.cosine_rows(engine.to_device(target_dict["database"][lo:hi].float().contiguous()))
    target_dict["lengths"] = engine.to_device(target_dict["lengths"][lo:hi])
    # (large query batches: the prefiltered search over an image of the normalised rows -- built by the first batch of more than 64
    #  queries, never by a run with a handful of query domains)
    target_dict["pf_image"] = (engine.lazy_pf_image(target_dict["database"], engine.UNIT_ROW_BOUND)
                               if hasattr(engine, "lazy_pf_image") else None)
    target_dict["_engine"] = engine


# ------------------------------------------------------------------ numeric kernels ----
def search_query_against_db(query_dict, target_dict, mincov, topk, score_corrections=None, engine=None):
    """cosine_similarity(db, q) * (len(q_seq) >= lengths * mincov) -> top-k (dbsearch.py:75-81).

    query_dict['embedding'] may hold one query [1,128] (the reference's shape) or a batch
    [nq,128] with query_dict['seq'] a list of sequences; returns {'scores', 'indices'} of shape
    [k] or [nq,k] accordingly.  k > Ndb raises, as torch.topk does.
    """
    engine = engine or target_dict.get("_engine")
    _to_engine(target_dict, engine)
    emb = query_dict["embedding"]
    seqs = query_dict["seq"]
    single = isinstance(seqs, str)
    qlen = np.asarray([len(seqs)] if single else [len(s) for s in seqs], dtype=np.float32)
    if topk > target_dict["n_rows"]:
        raise RuntimeError("selected index k out of range")
    q = engine.to_device(emb).reshape(-1, 128)
    extra = {"pf_image": target_dict["pf_image"]} if target_dict.get("pf_image") is not None else {}
    scores, idx = engine.cosine_topk(target_dict["database"], q, int(topk), lengths=target_dict["lengths"],
                                     qlen=engine.to_device(qlen), mincov=float(mincov), row_offset=target_dict["row_lo"], **extra)
    scores, idx = sharded.exchange_and_merge(scores, idx, engine)        # no-op on one rank
    if single:
        return {"scores": scores[0], "indices": idx[0]}
    return {"scores": scores, "indices": idx}


def knn_exact(xq, db_blocks, k: int, engine, log=logger, row_offset: int = 0, to_host: bool = True, raw_queries: bool = False,
              row_norm_bound=None, pf_image=None):
    """Exact max-inner-product kNN over a database delivered block by block (knn_exact_faiss,
    dbsearch.py:213-248): per block IndexFlat.add/search -> `I += i0` -> ResultHeap merge.

    xq: [nq,d], already normalised -- or raw with raw_queries=True: F.normalize(xq) (eps 1e-12, :303-304) is then applied
    inside every block's search call (bit-identical to normalising first); db_blocks: iterable of float32 [b,d] blocks -- host arrays
    (memmap slices: streamed through engine.device_blocks, the copy of block b+1 overlapping the
    scan of block b) or device tensors (scanned in place; a resident database is ONE such block).
    Rows are numbered from `row_offset`.  Returns (D float32 [nq,k], I int64 [nq,k]), best first,
    as numpy arrays (device tensors with to_host=False); missing entries are (-inf, -1) like faiss.
    """
    import time

    t0 = time.time()
    q = engine.to_device(xq)
    nq = q.shape[0]
    log.info("knn_exact queries size %s k=%d" % (tuple(q.shape), k))
    blocks = list(db_blocks) if isinstance(db_blocks, (list, tuple)) else db_blocks
    on_host = not (isinstance(blocks, list) and all(isinstance(b, engine.torch.Tensor) and b.device == q.device for b in blocks))
    best_s = best_i = None
    i0 = int(row_offset)
    for block in (engine.device_blocks(b for b in blocks if b.shape[0] > 0) if on_host else blocks):
        ni = block.shape[0]
        if ni == 0:
            continue
        # (a resident shard comes with its row-norm bound and, memory permitting, its fp16 image (built on first use): large batches -- and, from 1M rows, any batch -- then take the
        #  prefiltered search -- same results)
        s, i = (engine.ip_topk(block, q, k, row_offset=i0, normalize_queries=raw_queries, row_norm_bound=row_norm_bound, pf_image=pf_image)
                if row_norm_bound is not None else engine.ip_topk(block, q, k, row_offset=i0, normalize_queries=raw_queries))
        if best_s is None:
            best_s, best_i = s, i
        else:
            best_s, best_i = engine.topk_merge(_stack(best_s, s), _stack(best_i, i))
        i0 += ni
        log.info("%d DB elements, %.3f s" % (i0 - int(row_offset), time.time() - t0))
    if best_s is None:
        best_s = engine.to_device(np.full((nq, k), -np.inf, np.float32))
        best_i = engine.to_device(np.full((nq, k), -1, np.int64))
    if not to_host:
        return best_s, best_i
    D, I = best_s.cpu().numpy(), best_i.cpu().numpy()
    log.info("kNN time: %.3f s (%d vectors)" % (time.time() - t0, i0 - int(row_offset)))
    return D, I


def _stack(a, b):
    import torch
    return torch.stack([a, b])


# ------------------------------------------------------------------ helpers ------------
def _query_name(query_dict) -> str:
    return os.path.basename(query_dict["name"]).replace(".pdb", "")


def _hit(query_dict, target_name, score, t_len, tm_output, dbindex, metadata) -> dict:
    """The per-hit record consumed by write_search_results (keys: dbsearch.py:126-138)."""
    return {
        "query": _query_name(query_dict),
        "target": os.path.basename(target_name).replace(".pdb", ""),
        "score": score,
        "q_len": len(query_dict["seq"]),
        "t_len": t_len,
        "tmalign_output": tm_output,
        "dom_str": query_dict.get("dom_str"),
        "dom_conf": query_dict.get("dom_conf"),
        "dom_plddt": query_dict.get("dom_plddt"),
        "dbindex": dbindex,
        "metadata": metadata,
    }


def _load_queries(inputs, inputs_are_ca: bool, pdb_chains: List[str]) -> List[dict]:
    if inputs_are_ca:
        return list(inputs)                       # dicts {coords, seq, name, ...} (merizo.py:367-383)
    return [read_pdb(pdbfile=path, pdb_chain=chain) for path, chain in zip(inputs, pdb_chains)]


def _chain_list(pdb_chain: Optional[str], n_inputs: int) -> List[str]:
    """Chain ids per input: comma separated list, one id broadcast to all inputs, default 'A'."""
    if not pdb_chain:
        return ["A"] * n_inputs
    chains = pdb_chain.rstrip(",").split(",")
    if len(chains) == n_inputs:
        return chains
    if len(chains) == 1:
        return chains * n_inputs
    logger.error("Number of specified chain IDs not equal to number of input PDB files.")
    sys.exit(1)


class _Blob:
    """mmap of an (offsets file, data file) pair of either layout."""

    def __init__(self, index_path: str, data_path: str):
        self._fi = open(index_path, "rb")
        self._fd = open(data_path, "rb")
        self.index = mmap.mmap(self._fi.fileno(), 0, access=mmap.ACCESS_READ)
        self.data = mmap.mmap(self._fd.fileno(), 0, access=mmap.ACCESS_READ) if os.path.getsize(data_path) else b""

    def fetch(self, idx, conv):
        return [retrieve_bytes(s, e, self.data, typeconv=conv) for s, e in retrieve_start_end_by_idx(idx, self.index)]

    def close(self):
        for m in (self.index, self.data):
            if hasattr(m, "close"):
                m.close()
        self._fi.close()
        self._fd.close()


def _tmalign_pair(tmp, query_dict, target_coords, target_seq, fastmode, target_name=None, named=False):
    if named:
        qfn = write_pdb(tmp, query_dict["coords"], query_dict["seq"], name=os.path.basename(query_dict["name"]))
        tfn = write_pdb(tmp, target_coords, target_seq, name=target_name)
    else:
        qfn = write_pdb(tmp, query_dict["coords"], query_dict["seq"])
        tfn = write_pdb(tmp, target_coords, target_seq)
    return tm.run_tmalign(qfn, tfn, options="-fast" if fastmode else None, keep_pdbs=False)


# ------------------------------------------------------------------ `.pt` driver -------
def dbsearch(query, target_dict: dict, tmp: str, network, topk: int, mincov: float, mincos: float, mintm: float,
             fastmode: bool, device=None, inputs_are_ca: bool = False, pdb_chain: str = "A", skip_tmalign: bool = False,
             score_corrections=None, _embedding=None, _topk=None):
    """One query against a `.pt` database -> (results, all_results), dicts keyed by the hit's
    position in the top-k (dbsearch.py:84-200).  `_embedding` / `_topk` let run_dbsearch pass one
    row of a batched embedding / batched scan instead of recomputing per query."""
    query_dict = query if inputs_are_ca else read_pdb(pdbfile=query, pdb_chain=pdb_chain)
    engine = network.engine
    if _topk is None:
        if _embedding is None:
            _embedding = network.embed_many([query_dict["coords"]])
        query_dict["embedding"] = _embedding.reshape(1, -1)
        result = search_query_against_db(query_dict, target_dict, mincov, topk, score_corrections, engine=engine)
        scores = result["scores"].cpu().numpy()
        indices = result["indices"].cpu().numpy()
    else:
        query_dict["embedding"] = _embedding
        scores, indices = _topk

    meta = None
    if target_dict["mdfn"] is not None and target_dict["mifn"] is not None:
        meta = _Blob(target_dict["mifn"], target_dict["mdfn"])
    metadata = "{ }"
    results, all_results = {}, {}
    for rank in range(min(topk, scores.shape[0])):
        score, dbindex = scores[rank], int(indices[rank])
        target_name, target_coords, target_seq = target_dict["index"][dbindex]
        if skip_tmalign:
            if meta is not None:
                metadata = meta.fetch([dbindex], ascii_conv)[0]     # fetched for every rank (dbsearch.py:119-123)
            if score >= mincos:
                results[rank] = _hit(query_dict, target_name, score, len(target_seq), None, dbindex, metadata)
            continue
        if not (score >= mincos):
            continue
        tm_output = _tmalign_pair(tmp, query_dict, target_coords, target_seq, fastmode)
        max_tm = max(tm_output["qtm"], tm_output["ttm"])
        if tm_output["len_ali"] >= len(target_seq) * mincov:       # coverage filter, `.pt` path only (:165)
            if meta is not None:
                metadata = meta.fetch([dbindex], ascii_conv)[0]
            rec = _hit(query_dict, target_name, score, len(target_seq), tm_output, dbindex, metadata)
            if max_tm >= mintm:
                results[rank] = rec
            else:
                all_results[rank] = rec
    if meta is not None:
        meta.close()
    return results, all_results


# ------------------------------------------------------------------ faiss-layout driver -
def dbsearch_faiss(queries, target_dict: dict, tmp: str, network, topk: int, mincov: float, mincos: float,
                   mintm: float, fastmode: bool, device=None, inputs_are_ca: bool = False,
                   search_batchsize: int = 262144, search_type: str = "IP", pdb_chain: str = "A",
                   skip_tmalign: bool = False, score_corrections=None):
    """All queries against a faiss-layout database -> (results, all_results): one dict per query,
    keyed by a dense counter of retained hits (dbsearch.py:203-472).  No mincov length mask on this
    path (acknowledged TODO at dbsearch.py:307-310)."""
    if len(queries) == 0:
        logger.error("No inputs were provided!")
        sys.exit(1)
    if not os.path.exists(tmp):
        os.mkdir(tmp)
    if search_type != "IP":
        logging.error("Invalid/unsupported faiss search type: " + search_type + "\n\tOnly 'IP' is currently supported.")
        sys.exit(1)
    engine = network.engine
    nq = len(queries)
    dbinfofname = target_dict["database"]
    dbinfo = read_dbinfo(dbinfofname)
    db_dir = os.path.dirname(dbinfofname)

    def path(key):
        return os.path.join(db_dir, dbinfo[key])

    dbmm = db_memmap(filename=path("dbfname_IP"), shape=(dbinfo["DB_SIZE"], dbinfo["DB_DIM"]))
    logger.info("DB iterator using batchsize of " + str(search_batchsize))

    query_dicts = _load_queries(queries, inputs_are_ca, _chain_list(pdb_chain, nq))
    emb = sharded.embed_distributed(network, [qd["coords"] for qd in query_dicts])   # ragged launches, data-parallel over ranks

    # this rank's rows of the matrix: [lo, hi) of DB_SIZE (all of them on one rank)
    rank, world = sharded.rank_world()
    lo, hi = sharded.shard_bounds(int(dbinfo["DB_SIZE"]), world, rank)
    shard = _resident_shard(target_dict, engine, dbmm, lo, hi, nq, int(topk))
    if shard is not None:
        # resident shard: F.normalize (:303-304) + knn_exact_faiss (:213-248) as ONE call -- for the few queries of a CLI
        # search that is one launch (normalisation in the scan's prologue, merge by its last workgroup)
        # (more than 64 queries: the prefiltered search, with the shard's row-norm bound measured once when it became resident)
        Ds, Is = knn_exact(emb, [shard], int(topk), engine, row_offset=lo, to_host=False, raw_queries=True,
                           row_norm_bound=target_dict["_resident"].get("row_norm_bound"), pf_image=target_dict["_resident"].get("pf_image"))
    else:
        logger.info("database shard of %d rows exceeds the resident budget: streaming blocks of %d rows"
                    % (hi - lo, int(search_batchsize)))
        emb = engine.normalized(emb, 1e-12)                                 # F.normalize once, out of place; then block by block
        Ds, Is = knn_exact(emb, db_iterator(dbmm[lo:hi], int(search_batchsize)), int(topk), engine, row_offset=lo,
                           to_host=False)
    Ds, Is = sharded.exchange_and_merge(Ds, Is, engine)                   # all-gather + merge; no-op on one rank
    D, I = Ds.cpu().numpy(), Is.cpu().numpy()
    results = [dict() for _ in range(nq)]
    all_results = [dict() for _ in range(nq)]
    if rank != 0:
        return results, all_results                                       # rank 0 assembles the hit records

    keep = np.where((D >= mincos) & (I >= 0))                  # row-major: grouped by query, rank order
    hit_indices, hit_scores, query_indices = I[keep], D[keep], keep[0]
    n_hits = len(hit_indices)
    if n_hits == 0:
        return results, all_results

    logger.info("Retrieve domain hits...")
    with open(path("db_names_f"), "rb") as handle:
        names_mm = mmap.mmap(handle.fileno(), 0, access=mmap.ACCESS_READ)
        hit_ids = retrieve_names_by_idx(hit_indices, names_mm)
    seq_blob = _Blob(path("sif"), path("sdf"))
    hit_seqs = seq_blob.fetch(hit_indices, ascii_conv)
    seq_blob.close()
    hit_coords = None
    if not skip_tmalign:
        ca_blob = _Blob(path("cif"), path("cdf"))
        hit_coords = ca_blob.fetch(hit_indices, coord_conv)
        ca_blob.close()
    if "mif" in dbinfo and "mdf" in dbinfo:
        md_blob = _Blob(path("mif"), path("mdf"))
        hit_metadata = md_blob.fetch(hit_indices, ascii_conv)
        md_blob.close()
    else:
        hit_metadata = ["{ }"] * n_hits

    if not skip_tmalign:
        logger.info("TM-align top hits...")
    counts = [0] * nq
    n_tm_exclude = 0
    for h in range(n_hits):
        qi = int(query_indices[h])
        qd = query_dicts[qi]
        t_len = len(hit_seqs[h])
        if skip_tmalign:
            results[qi][counts[qi]] = _hit(qd, hit_ids[h], hit_scores[h], t_len, None, hit_indices[h], hit_metadata[h])
            counts[qi] += 1
            continue
        tm_output = _tmalign_pair(tmp, qd, hit_coords[h], hit_seqs[h], fastmode, target_name=hit_ids[h], named=True)
        rec = _hit(qd, hit_ids[h], hit_scores[h], t_len, tm_output, hit_indices[h], hit_metadata[h])
        if max(tm_output["qtm"], tm_output["ttm"]) >= mintm:
            results[qi][counts[qi]] = rec
            counts[qi] += 1
        else:
            all_results[qi][n_tm_exclude] = rec
            n_tm_exclude += 1
    if n_tm_exclude > 0:
        logger.info("Excluded " + str(n_tm_exclude) + " hits (across all query domains) by TM-score threshold(>=" + str(mintm) + ")")
    return results, all_results


def _resident_shard(target_dict: dict, engine, dbmm, lo: int, hi: int, nq: int, k: int):
    """Rows [lo,hi) of the matrix as one device tensor, kept in `target_dict` across calls on the same
    database; None when they do not fit the engine's resident budget (-> streaming)."""
    cache = target_dict.setdefault("_resident", {})
    if cache.get("engine") is engine and cache.get("span") == (lo, hi):
        return cache["shard"]
    cache.clear()
    if (hi - lo) * dbmm.shape[1] * 4 > engine.resident_budget(nq, k):
        return None
    cache.update(engine=engine, span=(lo, hi), shard=engine.upload_rows(dbmm, lo, hi))
    if hasattr(engine, "row_norm_bound"):           # (the CPU oracle engine of the tests has no prefiltered search)
        cache["row_norm_bound"] = engine.row_norm_bound(cache["shard"])
        cache["pf_image"] = engine.lazy_pf_image(cache["shard"], cache["row_norm_bound"])
    return cache["shard"]


# ------------------------------------------------------------------ dispatcher ---------
def run_dbsearch(inputs, db_name: str, tmp: str, device, topk: int, fastmode: bool, threads: int, mincos: float,
                 mintm: float, mincov: float, inputs_are_ca: bool = False, search_batchsize: int = 262144,
                 search_type: str = "IP", pdb_chain: Optional[str] = None, skip_tmalign: bool = False,
                 network=None, weights_path: Optional[str] = None):
    """Set up the encoder, open the database, search every input (dbsearch.py:475-551).
    Returns (search_results, all_search_results): one dict rank -> hit per input, twice."""
    if len(inputs) == 0:
        logger.error("No inputs were provided!")
        sys.exit(1)
    if not os.path.exists(tmp):
        os.mkdir(tmp)
    if network is None:
        network, device = network_setup(threads=threads, device=device, weights_path=weights_path)
    if not skip_tmalign and tm.find_tmalign() is None:
        logger.warning("no TM-align binary found (set $MERIZO_TMALIGN): running an embedding-only search; "
                       "TM-align columns are unavailable")
        skip_tmalign = True
    target_db = read_database(db_name=db_name, device=device, engine=network.engine)

    if target_db["faiss"]:
        if search_batchsize < 1:
            logger.error("search_batchsize must be >= 1.")
            sys.exit(1)
        return dbsearch_faiss(queries=inputs, target_dict=target_db, tmp=tmp, network=network, topk=topk,
                              mincov=mincov, mincos=mincos, mintm=mintm, fastmode=fastmode, device=device,
                              inputs_are_ca=inputs_are_ca, search_batchsize=search_batchsize, search_type=search_type,
                              pdb_chain=pdb_chain, skip_tmalign=skip_tmalign)

    query_dicts = _load_queries(inputs, inputs_are_ca, _chain_list(pdb_chain, len(inputs)))
    emb = sharded.embed_distributed(network, [qd["coords"] for qd in query_dicts])   # ragged launches, data-parallel over ranks
    batch = {"seq": [qd["seq"] for qd in query_dicts], "embedding": emb}
    top = search_query_against_db(batch, target_db, mincov, topk, engine=network.engine)   # one batched scan (+ exchange)
    top_s, top_i = top["scores"].cpu().numpy(), top["indices"].cpu().numpy()
    search_results, all_search_results = [], []
    if sharded.rank_world()[0] != 0:
        return [dict() for _ in query_dicts], [dict() for _ in query_dicts]      # rank 0 assembles the hit records
    for row, qd in enumerate(query_dicts):
        res, all_res = dbsearch(query=qd, target_dict=target_db, tmp=tmp, network=network, topk=topk, mincov=mincov,
                                mincos=mincos, mintm=mintm, fastmode=fastmode, device=device, inputs_are_ca=True,
                                skip_tmalign=skip_tmalign, _embedding=emb[row:row + 1], _topk=(top_s[row], top_i[row]))
        search_results.append(res)
        all_search_results.append(all_res)
    return search_results, all_search_results
