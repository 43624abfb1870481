"""easy-search hand-off without the (out-of-scope) Merizo segmenter: domains from a given chopping.

The reference's easy-search passes Foldclass a list of per-domain dicts built by Merizo at
programs/Merizo/model/utils/utils.py:413-440:
    {'coords': float32 [L,3] CA, 'seq', 'name': '<pdb stem>_merizo_NN', 'dom_str', 'dom_conf', 'dom_plddt'}
Here the chopping is an input (from a `_segment.tsv`, a flag or a file) in the reference's own
syntax (format_dom_str, utils.py:511-541): domains separated by ',', discontinuous segments of a
domain joined by '_', residue ranges 'a-b' or single residues, in PDB residue numbering.
"""
from __future__ import annotations

import os
import re
from typing import List, Optional

import numpy as np

from .constants import three_to_single_aa


_SEG = re.compile(r"^(-?\d+)(?:-(-?\d+))?$")


def read_ca_records(pdbfile: str, pdb_chain: str = "A"):
    """CA records of one chain with residue numbers and B-factors -> dict of arrays."""
    resi, xyz, seq, bfac = [], [], [], []
    with open(pdbfile) as handle:
        for line in handle:
            if len(line) > 21 and line[21] == pdb_chain and line[:4] == "ATOM" and line[12:16] == " CA ":
                resi.append(int(line[22:26]))
                xyz.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
                seq.append(three_to_single_aa.get(line[17:20], "X"))
                bfac.append(float(line[60:66]) if len(line) >= 66 and line[60:66].strip() else 0.0)
    return {"resi": np.asarray(resi, dtype=np.int64), "coords": np.asarray(xyz, dtype=np.float64).astype(np.float32).reshape(-1, 3),
            "seq": "".join(seq), "b": np.asarray(bfac, dtype=np.float64)}


def parse_chopping(chopping: str) -> List[List[range]]:
    """'71-189,190-290_300-310,5' -> [[range(71,190)], [range(190,291), range(300,311)], [range(5,6)]]."""
    domains = []
    for dom in chopping.strip().split(","):
        dom = dom.strip()
        if not dom:
            continue
        segs = []
        for seg in dom.split("_"):
            m = _SEG.match(seg.strip())
            if m is None:
                raise ValueError(f"bad chopping segment '{seg}' in '{chopping}'")
            lo = int(m.group(1))
            hi = int(m.group(2)) if m.group(2) is not None else lo
            segs.append(range(lo, hi + 1))
        domains.append(segs)
    return domains


def domains_from_chopping(pdbfile: str, chopping: str, pdb_chain: str = "A", conf: Optional[float] = None) -> List[dict]:
    """The Merizo-style domain dicts for `pdbfile` under the given chopping (residues missing from
    the structure are skipped; empty domains are dropped)."""
    rec = read_ca_records(pdbfile, pdb_chain)
    stem = os.path.basename(pdbfile).replace(".pdb", "")
    out = []
    for num, (segs, dom_str) in enumerate(zip(parse_chopping(chopping), [d for d in chopping.strip().split(",") if d.strip()]), start=1):
        wanted = np.zeros(rec["resi"].shape, dtype=bool)
        for seg in segs:
            wanted |= (rec["resi"] >= seg.start) & (rec["resi"] < seg.stop)
        if not wanted.any():
            continue
        out.append({
            "coords": rec["coords"][wanted],
            "seq": "".join(np.asarray(list(rec["seq"]))[wanted]),
            "name": f"{stem}_merizo_{str(num).zfill(2)}",
            "dom_str": dom_str.strip(),
            "dom_conf": float(conf) if conf is not None else float("nan"),
            "dom_plddt": float(rec["b"][wanted].mean()),
        })
    return out


def segment_row(pdbfile: str, chopping: str, pdb_chain: str = "A", conf: float = float("nan"), runtime: float = 0.0) -> dict:
    """One `_segment.tsv` row (programs/utils.py:161-176) for a given chopping."""
    rec = read_ca_records(pdbfile, pdb_chain)
    doms = domains_from_chopping(pdbfile, chopping, pdb_chain)
    nres_dom = int(sum(len(d["seq"]) for d in doms))
    return {"name": pdbfile, "length": len(rec["seq"]), "nres_domain": nres_dom, "nres_non_domain": len(rec["seq"]) - nres_dom,
            "num_domains": len(doms), "conf": conf, "time": runtime, "dom_str": chopping}


def read_segment_tsv(path: str) -> dict:
    """{file stem: chopping} from a reference `_segment.tsv` (header optional)."""
    out = {}
    with open(path) as handle:
        for line in handle:
            cols = line.rstrip("\n").split("\t")
            if len(cols) < 8 or cols[0] == "filename":
                continue
            out[cols[0]] = cols[7]
    return out
