import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cogdl_amd import _lib, synth
from cogdl_amd.operators.fused_gat import gat_forward
from tools.ops_bench import timeit
DEV="cuda:0"; lib=_lib.hip()
g=synth.reddit_like(seed=0, device=DEV); n=g.num_nodes
for h,f,dt in ((1,48,torch.bfloat16),(1,64,torch.bfloat16),(2,32,torch.bfloat16),(1,44,torch.float32)):
    ar,ac=torch.randn(n,h,device=DEV),torch.randn(n,h,device=DEV); feat=torch.randn(n,h,f,device=DEV).to(dt)
    res=[]
    for kern in (0,1,2):
        for vcap in (0,8,4,2,1):
            lib.cogdl_hip_set_tuning(5,kern); lib.cogdl_hip_set_tuning(4,vcap)
            for p in (0.0,0.5):
                ms=timeit(lambda: gat_forward(ar,ac,g.rowptr,g.colind,0.2,feat,p,1),5)
                res.append("k%d/v%d/p%.0f %.2f"%(kern,vcap,p*10,ms))
    lib.cogdl_hip_set_tuning(5,0); lib.cogdl_hip_set_tuning(4,0)
    print("H=%d F=%d %s: %s"%(h,f,str(dt)[6:],"  ".join(res)),flush=True)
