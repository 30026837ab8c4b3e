import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cogdl_amd import synth
from cogdl_amd.plan import csr2csc
for g in (synth.arxiv_like(seed=0, topology="rmat").to("cuda:0"), synth.reddit_like(seed=0, device="cuda:0")):
    for _ in range(3):
        csr2csc(g.rowptr, g.colind, g.num_nodes)
    torch.cuda.synchronize()
    print("done", g.nnz, flush=True)
