"""Golden vectors for the multi-domain combinatorial core, from the reference's own functions.

Run in the build container only (imports /root/reference):  python oracle/gen_golden_multidomain.py
Writes tests/golden/multidomain.json: inputs and outputs of domid2chainid_fn and
tmalign_submatrix_to_hits (programs/Foldclass/dbsearch_fulllength.py:36-39, :95-180).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/merizo_search")
from programs.Foldclass.dbsearch_fulllength import domid2chainid_fn, tmalign_submatrix_to_hits  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "multidomain.json")

names = ["cath-dompdb/2pi4A04.pdb", "xxx/AF-Q93009-F1-model_v4_TED02.pdb", "AF-A0A024R1R8-F1-model_v4_TED01", "1abcA01",
         "dir/sub/3w5hB12.pdb", "weird_name_b03.pdb", "nodigits.pdb", "x/AF-P12345-F1-model_v4_TED10", "endswithp07.pdb",
         "a/b/c/1xyzC_05.pdb", "q_TED.pdb", "7", "12"]
chains = [{"in": n, "out": domid2chainid_fn(n)} for n in names]

rng = np.random.default_rng(7)
cases = []
shapes = [(2, 2), (2, 3), (3, 3), (2, 5), (3, 4), (1, 1), (1, 3), (3, 2), (4, 4), (2, 2), (3, 5), (2, 4)]
for c, (nq, nh) in enumerate(shapes):
    for density in (0.35, 0.7, 1.0):
        m = np.round(rng.uniform(0.5, 1.0, size=(nq, nh)), 4)
        m[rng.uniform(size=(nq, nh)) > density] = 0.0
        qds = ["q%d_merizo_%02d" % (c, i + 1) for i in range(nq)]
        hds = [{"hd": "AF-T%d-F1-model_v4_TED%02d" % (c, j + 1), "hm": '{ "i": %d }' % j} for j in range(nh)]
        out = tmalign_submatrix_to_hits(m, qc="q%d" % c, hc="AF-T%d-F1-model_v4" % c, qds=qds, hds=hds)
        cases.append({"mtx": m.tolist(), "qc": "q%d" % c, "hc": "AF-T%d-F1-model_v4" % c, "qds": qds, "hds": hds,
                      "out": [list(t) for t in out]})

with open(OUT, "w") as fh:
    json.dump({"domid2chainid": chains, "submatrix": cases}, fh, indent=0)
print("wrote", OUT, len(chains), "names,", len(cases), "matrices,", sum(len(c["out"]) for c in cases), "mappings")
