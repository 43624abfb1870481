"""BASELINE config C1 at its real size (TEST INFRASTRUCTURE): the shipped TED example database
(`examples/database/ted100.json`, 66,943 entries) rebuilt from tests/golden/ted100_full.npz.

The name file and the three offset files come out byte-identical to the shipped ones (sha256
checked); the four payload files the reference snapshot lacks are synthesised from a seed with
exactly the byte counts the shipped offsets demand (12 bytes of CA coordinates per residue, one
ASCII letter per residue, metadata JSON padded to the recorded length)."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ted100_full.npz")


def build(dirname: str, seed: int = 0, plant=None) -> str:
    """Write the database into `dirname`; returns the path prefix (`<dirname>/ted100`).
    plant: optional {row: float32[128] unit vector} written over the synthetic embedding rows."""
    g = np.load(GOLDEN)
    info = json.loads(str(g["ted100_json"]))
    seq_len = g["seq_len"].astype(np.int64)
    meta_len = g["meta_len"].astype(np.int64)
    n = seq_len.size
    assert n == info["DB_SIZE"] == 66943
    os.makedirs(dirname, exist_ok=True)
    path = lambda key: os.path.join(dirname, info[key])

    def offsets(lengths):
        end = np.cumsum(lengths)
        return np.stack([end - lengths, end], axis=1).astype(np.int64)

    offsets(seq_len).tofile(path("sif"))
    offsets(seq_len * 12).tofile(path("cif"))
    offsets(meta_len).tofile(path("mif"))
    with open(path("db_names_f"), "wb") as fh:
        fh.write(b"".join(nm.ljust(32) + b"\n" for nm in g["names"].tolist()))
    want = json.loads(str(g["sha256"]))
    for key, short in (("sif", "seq.index"), ("cif", "ca.index"), ("mif", "metadata.index"), ("db_names_f", "raw_128d.index_names")):
        got = hashlib.sha256(open(path(key), "rb").read()).hexdigest()
        assert got == want[short], f"{short}: rebuilt file differs from the shipped one"

    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((n, info["DB_DIM"])).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    for row, vec in (plant or {}).items():
        emb[row] = vec
    emb.tofile(path("dbfname_IP"))
    total = int(seq_len.sum())
    rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), size=total).tofile(path("sdf"))
    (rng.standard_normal((total, 3)) * 10).astype(np.float32).tofile(path("cdf"))
    with open(path("mdf"), "wb") as fh:
        for row, ln in enumerate(meta_len.tolist()):
            body = ('{"row": %d, "pad": "' % row).encode()
            fh.write(body + b"x" * (ln - len(body) - 2) + b'"}' if ln >= len(body) + 2 else b" " * ln)
    with open(os.path.join(dirname, "ted100.json"), "w") as fh:
        fh.write(str(g["ted100_json"]))
    return os.path.join(dirname, "ted100")


def check_search(prefix_out: str, planted_rows, k: int) -> None:
    """The TSV of `search M0 ... -k k`: the planted rows come first, names / lengths / metadata are the
    records of exactly those rows."""
    g = np.load(GOLDEN)
    rows = [l.split("\t") for l in open(prefix_out + "_search.tsv").read().splitlines()]
    assert rows[0] == ["query", "emb_rank", "target", "emb_score", "q_len", "t_len", "metadata"]
    body = rows[1:]
    assert len(body) == k and [r[1] for r in body] == [str(i) for i in range(k)]
    for rank, row in enumerate(planted_rows):
        r = body[rank]
        assert r[0] == "M0_ca" and r[2] == g["names"][row].decode() and r[4] == "31" and int(r[5]) == int(g["seq_len"][row])
        assert r[6].startswith('{"row": %d,' % row) and len(r[6]) == int(g["meta_len"][row])
    scores = [float(r[3]) for r in body]
    assert scores == sorted(scores, reverse=True) and scores[0] > 0.99
