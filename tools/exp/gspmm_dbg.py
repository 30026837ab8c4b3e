import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogdl_amd import xcdplan
from cogdl_amd.operators import ops
DEV="cuda:0"
gen = torch.Generator().manual_seed(11)
n, e, k = 5000, 120000, 16
row = torch.randint(0, n - n // 8, (e,), generator=gen); col = torch.randint(0, n, (e,), generator=gen)
for node, cnt in ((3, 129), (4, 20000), (17, 3000), (18, 700), (4999, 2)):
    row[torch.randperm(e, generator=gen)[:cnt]] = node
x, ef, w = torch.randn(n, k, generator=gen), torch.randn(e, k, generator=gen), torch.rand(e, generator=gen)
g = types.SimpleNamespace(edge_index=(row.to(DEV), col.to(DEV)), edge_weight=w.to(DEV))
res = {}
for mode in ("off", "force"):
    xcdplan.MODE = mode; ops.clear_plans()
    res[mode] = ops.s_mul_e_sum(g, x.to(DEV), ef.to(DEV), weight=True).cpu().double()
want = torch.zeros(n, k, dtype=torch.float64); msg = (x.double()[col] * ef.double()) * w.double().view(-1, 1)
want.index_add_(0, row, msg); scale = torch.zeros(n, k, dtype=torch.float64); scale.index_add_(0, row, msg.abs())
deg = torch.bincount(row, minlength=n)
for mode in res:
    err = ((res[mode] - want).abs() / scale.clamp(min=1e-30))
    worst = err.max(1).values
    top = torch.topk(worst, 6)
    print(mode, [(int(i), int(deg[i]), float(v)) for v, i in zip(top.values, top.indices)])
