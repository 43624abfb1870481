import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
out = bench.streamed_bench(torch, ops, syn, torch.device("cuda", 0), 10, lambda m: print(m, flush=True), sizes=(8_000_000, 45_625_000), nqs=(1, 256, 4096))
print(json.dumps(out["block_sweep"]))
