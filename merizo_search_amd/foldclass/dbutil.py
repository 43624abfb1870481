"""On-disk access to the two Foldclass database layouts.

faiss layout  (`<db>.json` + raw files; readers mirror programs/Foldclass/dbutil.py):
    json keys  dbfname_IP, DB_SIZE, DB_DIM, db_names_f, sif, sdf, cif, cdf, mif, mdf
    dbfname_IP headerless row-major float32 [DB_SIZE, DB_DIM], rows PRE-L2-NORMALISED  (dbutil.py:28-30)
    db_names_f 33-byte records: name left-justified in 32 bytes + '\n'                 (dbutil.py:107-108)
    sif/cif/mif int64 [N,2] (start, end) byte offsets into sdf (ASCII residues), cdf (float32 xyz,
               12 bytes per residue) and mdf (ASCII JSON)                               (dbutil.py:127-145)
pt layout     `<db>.pt` torch.save(FloatTensor[N,128]) RAW embeddings; `<db>.index` pickle of
              list[(path, float32[L,3], seq)]; optional `<db>.metadata` + `<db>.metadata.index`
              (makedb.py:85-91, dbsearch.py:50-64).

The reference ships no writer for the faiss layout; ``write_faiss_db`` is this package's own
 and is what createdb / the tests / the C1 fixture use.
"""
from __future__ import annotations

import json
import os
import pickle
from typing import Iterable, List, Optional, Sequence

import numpy as np

NAME_WIDTH = 32           # + '\n' = 33-byte records
NAME_RECORD = NAME_WIDTH + 1


# ------------------------------------------------------------------ readers ------------
def read_dbinfo(dbinfo_path: str) -> dict:
    with open(dbinfo_path, "r") as handle:
        return json.load(handle)


def db_memmap(filename: str, shape: tuple):
    """float32 [DB_SIZE, DB_DIM] view of `dbfname_IP` (row-major, no header)."""
    return np.memmap(filename, dtype="float32", mode="r", shape=tuple(shape))


def db_iterator(embeddings, batch_size: int):
    """Consecutive row blocks of at most batch_size rows (the reference's out-of-core streaming)."""
    for start in range(0, embeddings.shape[0], batch_size):
        yield embeddings[start:start + batch_size]


def startend_memmap(filename: str, n_entries: int):
    return np.memmap(filename, dtype="int64", mode="r", shape=(n_entries, 2))


def _fixed_records(idx: Sequence[int], mm, width: int) -> List[bytes]:
    """`width`-byte records number idx[*] from a bytes-like / mmap object, in the order given."""
    out = []
    for i in idx:
        i = int(i)
        out.append(bytes(mm[i * width:(i + 1) * width]))
    return out


def retrieve_names_by_idx(idx, mm, use_sorting: bool = False) -> np.ndarray:
    """Names of entries idx[*] from the 33-byte-record names file (order of `idx` preserved)."""
    return np.asarray([rec.decode().rstrip() for rec in _fixed_records(idx, mm, NAME_RECORD)])


def retrieve_start_end_by_idx(idx, mm, use_sorting: bool = False) -> np.ndarray:
    """(start, end) int64 pairs of entries idx[*] from an offsets file."""
    if len(idx) == 0:
        return np.zeros((0, 2), dtype=np.int64)
    return np.asarray([np.frombuffer(rec, dtype="int64") for rec in _fixed_records(idx, mm, 16)])


def retrieve_bytes(start, end, mm, typeconv=None):
    raw = bytes(mm[int(start):int(end)])
    return raw if typeconv is None else typeconv(raw)


def coord_conv(raw: bytes) -> np.ndarray:
    """12*L bytes -> float32 [L,3]."""
    flat = np.frombuffer(raw, dtype="float32")
    assert flat.size % 3 == 0
    return flat.reshape(-1, 3)


def ascii_conv(raw: bytes) -> str:
    return raw.decode("ascii")


# ------------------------------------------------------------------ faiss-layout writer -
def write_faiss_db(db_prefix: str, embeddings_norm: np.ndarray, names: Sequence[str], seqs: Sequence[str],
                   coords: Sequence[np.ndarray], metadata: Optional[Sequence[str]] = None) -> str:
    """Write a faiss-layout database next to `<db_prefix>.json`; returns the json path.

    embeddings_norm: float32 [N,128], rows already L2-normalised (the layout stores *_norm.db).
    File names follow the TED download (download_dbs.sh:30-38): <base>_raw_128d_norm.db,
    <base>_raw_128d.index_names, <base>_{seq,ca,metadata}.{db,index}; the json stores them
    relative to its own directory, as the reference resolves them (dbsearch.py:262-265).
    """
    emb = np.ascontiguousarray(embeddings_norm, dtype=np.float32)
    n = emb.shape[0]
    if not (len(names) == len(seqs) == len(coords) == n):
        raise ValueError("names / seqs / coords / embeddings disagree on the number of entries")
    base = os.path.basename(db_prefix)
    ddir = os.path.dirname(os.path.abspath(db_prefix))
    os.makedirs(ddir, exist_ok=True)
    files = {
        "dbfname_IP": f"{base}_raw_128d_norm.db", "db_names_f": f"{base}_raw_128d.index_names",
        "sif": f"{base}_seq.index", "sdf": f"{base}_seq.db", "cif": f"{base}_ca.index", "cdf": f"{base}_ca.db",
    }
    if metadata is not None:
        files.update({"mif": f"{base}_metadata.index", "mdf": f"{base}_metadata.db"})
    emb.tofile(os.path.join(ddir, files["dbfname_IP"]))
    with open(os.path.join(ddir, files["db_names_f"]), "wb") as handle:
        for name in names:
            raw = str(name).encode("ascii")
            if len(raw) > NAME_WIDTH:
                raise ValueError(f"entry name longer than {NAME_WIDTH} bytes: {name}")
            handle.write(raw.ljust(NAME_WIDTH) + b"\n")

    def blob_pair(index_key, data_key, blobs: Iterable[bytes]):
        offsets = np.zeros((n, 2), dtype=np.int64)
        pos = 0
        with open(os.path.join(ddir, files[data_key]), "wb") as handle:
            for row, raw in enumerate(blobs):
                handle.write(raw)
                offsets[row] = (pos, pos + len(raw))
                pos += len(raw)
        offsets.tofile(os.path.join(ddir, files[index_key]))

    blob_pair("sif", "sdf", (s.encode("ascii") for s in seqs))
    blob_pair("cif", "cdf", (np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 3).tobytes() for c in coords))
    if metadata is not None:
        blob_pair("mif", "mdf", (m.encode("ascii") for m in metadata))
    info = {"DB_SIZE": int(n), "DB_DIM": int(emb.shape[1])}
    info.update(files)
    path = db_prefix + ".json"
    with open(path, "w") as handle:
        json.dump(info, handle)
    return path


# ------------------------------------------------------------------ pt-layout helpers ---
def write_pt_db(db_prefix: str, embeddings_raw: np.ndarray, names: Sequence[str], coords: Sequence[np.ndarray],
                seqs: Sequence[str], metadata: Optional[Sequence[str]] = None) -> None:
    """`<db>.pt` + `<db>.index` exactly as run_createdb leaves them (makedb.py:85-91), plus the
    optional metadata pair read at dbsearch.py:59-62."""
    import torch

    torch.save(torch.from_numpy(np.ascontiguousarray(embeddings_raw, dtype=np.float32)), db_prefix + ".pt")
    with open(db_prefix + ".index", "wb") as handle:
        pickle.dump([(str(nm), np.asarray(c, dtype=np.float32), str(s)) for nm, c, s in zip(names, coords, seqs)], handle)
    if metadata is not None:
        offsets = np.zeros((len(metadata), 2), dtype=np.int64)
        pos = 0
        with open(db_prefix + ".metadata", "wb") as handle:
            for row, m in enumerate(metadata):
                raw = m.encode("ascii")
                handle.write(raw)
                offsets[row] = (pos, pos + len(raw))
                pos += len(raw)
        offsets.tofile(db_prefix + ".metadata.index")
