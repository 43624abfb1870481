#!/usr/bin/env python
"""Compact fixture of the shipped TED example index (BASELINE config C1 at its real size).

Runs in the build container only (reads /root/reference).  The reference ships the name file and
the three offset files of `examples/database/ted100_9606_small/` (66,943 entries) but none of the
payloads.  The offset files are contiguous (start[i+1] == end[i]), so they are fully described by
the per-entry byte lengths; this script stores

    tests/golden/ted100_full.npz   seq_len u16[N], meta_len u16[N], names S32[N]   (compressed)
                                   + sha256 of each shipped file, and the shipped ted100.json text

from which tests/c1_case.py rebuilds the four shipped files byte for byte (checked against the
hashes) and synthesises payloads of exactly the byte counts those offsets demand.
"""
import hashlib
import json
import os

import numpy as np

REF = "/root/reference/examples/database"
SRC = os.path.join(REF, "ted100_9606_small")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ted100_full.npz")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    base = "ted100_9606_small"
    files = {k: os.path.join(SRC, f"{base}_{k}") for k in ("seq.index", "ca.index", "metadata.index", "raw_128d.index_names")}
    seq = np.fromfile(files["seq.index"], dtype=np.int64).reshape(-1, 2)
    ca = np.fromfile(files["ca.index"], dtype=np.int64).reshape(-1, 2)
    md = np.fromfile(files["metadata.index"], dtype=np.int64).reshape(-1, 2)
    n = seq.shape[0]
    for off in (seq, ca, md):
        assert off.shape == (n, 2) and off[0, 0] == 0 and np.array_equal(off[1:, 0], off[:-1, 1])
    assert np.array_equal(ca, seq * 12)
    seq_len = (seq[:, 1] - seq[:, 0]).astype(np.uint16)
    meta_len = (md[:, 1] - md[:, 0]).astype(np.uint16)
    assert (md[:, 1] - md[:, 0]).max() < 65536
    raw = open(files["raw_128d.index_names"], "rb").read()
    assert len(raw) == 33 * n
    names = np.frombuffer(raw, dtype="S33").astype("S33")
    assert all(x.endswith(b"\n") for x in names[:100])
    names32 = np.asarray([x[:32].rstrip() for x in names], dtype="S32")
    np.savez_compressed(OUT, seq_len=seq_len, meta_len=meta_len, names=names32,
                        sha256=json.dumps({k: sha(p) for k, p in files.items()}),
                        ted100_json=open(os.path.join(REF, "ted100.json")).read())
    print(OUT, os.path.getsize(OUT), "bytes,", n, "entries")


if __name__ == "__main__":
    main()
