"""One rank of the multi-rank driver tests (TEST INFRASTRUCTURE; started by tests/test_sharded_drivers.py).

    python dist_worker.py <repo> <port> <rank> <world> <oracle|hip> <workdir>

Joins a gloo process group (on a one-GPU box every `hip` rank uses cuda:0), runs the product
drivers -- run_dbsearch over the faiss-layout and the `.pt` database in <workdir>, resident and
forced-streaming -- and rank 0 saves what it would report: TSV text + raw score bits and rows.
"""
import os
import sys

repo, port, rank, world, kind, work = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
sys.path.insert(0, repo)
sys.path.insert(0, os.path.join(repo, "tests"))
import numpy as np
import torch
import torch.distributed as dist

if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
from merizo_search_amd.foldclass import dbsearch as ds, results, sharded

if kind == "hip":
    from merizo_search_amd.foldclass.network import network_setup
    torch.cuda.set_device(0)
    net, _dev = network_setup(device="cuda:0", allow_synthetic=True)
else:
    from oracle_engine import oracle_network
    net = oracle_network()

FMT = "query,emb_rank,target,emb_score,q_len,t_len,metadata".split(",")
case = np.load(os.path.join(work, "queries.npz"), allow_pickle=True)
queries = [dict(coords=c, seq=str(s), name=str(n)) for c, s, n in zip(case["coords"], case["seqs"], case["names"])]
k = int(case["k"])


def run(tag, db, mincov, budget=None):
    if budget is not None:
        net.engine.resident_budget = lambda nq=0, kk=0: budget
    res, _all = ds.run_dbsearch([dict(q) for q in queries], os.path.join(work, db), os.path.join(work, "tmp%d" % rank), "cuda", topk=k,
                                fastmode=False, threads=-1, mincos=-2.0, mintm=0.5, mincov=mincov, inputs_are_ca=True,
                                skip_tmalign=True, network=net, search_batchsize=int(case["batch"]))
    if budget is not None:
        del net.engine.resident_budget
    if sharded.rank_world()[0] != 0:
        assert all(len(r) == 0 for r in res)
        return
    results.write_search_results(res, os.path.join(work, f"{tag}_w{world}.tsv"), FMT, header=True)
    np.savez(os.path.join(work, f"{tag}_w{world}.npz"),
             scores=np.asarray([[np.float32(h["score"]) for h in r.values()] for r in res], dtype=np.float32),
             rows=np.asarray([[int(h["dbindex"]) for h in r.values()] for r in res], dtype=np.int64))


run("fa", "fa", 0.0)
run("fa_stream", "fa", 0.0, budget=0)
run("pt", "pt", 0.7)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print("rank", rank, "ok")
