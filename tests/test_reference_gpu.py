"""The REAL reference on the MI355X (round-1 verdict, "next" item 2): the unchanged cogdl package (staged copy,
oracle/_ref/pkg -- see oracle/Makefile) imported on the GPU box after cogdl_amd.install():

  1. cogdl.utils.spmm_utils.spmm's GPU branch (utils/spmm_utils.py:98-109) resolves to cogdl_amd's csrspmm;
  2. the unchanged GCNLayer / GATLayer / SAGELayer (+ MaxAggregator) classes run on cuda and reproduce the outputs
     and gradients the reference computed on CPU (tests/golden/{gcn,gat,sage}_layer.npz);
  3. BASELINE configs[0]/[1]: cogdl.experiment(model='gcn', dataset=<Cora- / arxiv-shaped NodeDataset>) trains a few
     epochs through the reference's Trainer on cuda:0; the loss trajectory matches the same experiment on the
     reference's CPU path (same seed) and every aggregation went through the HIP kernels (call counters).
Runs in a fresh interpreter so that the reference's modules never leak into the other tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import refpkg  # noqa: E402

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import json, os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from tools import refpkg
refpkg.setup(install=True)
import torch
import cogdl
from cogdl.data import Graph
from cogdl.layers import GCNLayer, GATLayer, SAGELayer
from cogdl.utils import spmm_utils
import cogdl_amd.operators.spmm as our_spmm

DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")
def gold(name): return dict(np.load(os.path.join(GOLD, name + ".npz")))
def T(a): return torch.from_numpy(np.ascontiguousarray(a))
def coo(z):
    rp = T(z["row_indptr"]); deg = rp[1:] - rp[:-1]
    return torch.repeat_interleave(torch.arange(deg.numel()), deg), T(z["col_indices"])

calls = {"spmm": 0}
_raw = our_spmm.csr_spmm_raw
def counted(*a, **k):
    calls["spmm"] += 1
    return _raw(*a, **k)
our_spmm.csr_spmm_raw = counted

report = {}
# ---- 1+2a. GCNLayer through the real dispatcher on cuda
z = gold("gcn_layer")
row, col = coo(z)
g = Graph(edge_index=(row, col), edge_weight=T(z["edge_weight"]), num_nodes=row.max().item() + 1).to(DEV)
layer = GCNLayer(32, 16).to(DEV)
with torch.no_grad():
    layer.linear.weight.copy_(T(z["W"])); layer.linear.bias.copy_(T(z["b"]))
x = T(z["x"]).to(DEV).requires_grad_()
out = layer(g, x)
assert spmm_utils.CONFIGS["fast_spmm"] is our_spmm.csrspmm, "the dispatcher's GPU branch did not resolve to cogdl_amd"
assert calls["spmm"] >= 1
np.testing.assert_allclose(out.detach().cpu().numpy(), z["out_train"], rtol=1e-4, atol=1e-5)
(out * T(z["G"]).to(DEV)).sum().backward()
np.testing.assert_allclose(x.grad.cpu().numpy(), z["grad_x"], rtol=1e-4, atol=1e-5)
np.testing.assert_allclose(layer.linear.weight.grad.cpu().numpy(), z["grad_W"], rtol=1e-4, atol=2e-5)
np.testing.assert_allclose(layer.linear.bias.grad.cpu().numpy(), z["grad_b"], rtol=1e-4, atol=2e-5)
report["gcn_layer"] = "ok"

# ---- 2b. GATLayer: attn_drop == 0 on a symmetric graph takes fused_gat_op (gat_layer.py:68), otherwise
#          edge_softmax + mh_spmm -- both through the real dispatcher, both against the reference's CPU output
z = gold("gat_layer")
row, col = coo(z)
g = Graph(edge_index=(row, col), num_nodes=row.max().item() + 1).to(DEV)
gat = GATLayer(16, 8, nhead=4, attn_drop=0.0, alpha=0.2).to(DEV)
with torch.no_grad():
    gat.W.copy_(T(z["W"])); gat.a_l.copy_(T(z["a_l"])); gat.a_r.copy_(T(z["a_r"]))
grads = {}
for leg, sym in (("fused", True), ("unfused", False)):
    g._adj.set_symmetric(sym)
    gat.zero_grad()
    x = T(z["x"]).to(DEV).requires_grad_()
    out = gat(g, x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=2e-4, atol=2e-5, err_msg=leg)
    (out * T(z["G"]).to(DEV)).sum().backward()
    grads[leg] = [t.detach().cpu().numpy() for t in (x.grad, gat.W.grad, gat.a_l.grad, gat.a_r.grad)]
assert spmm_utils.CONFIGS["fused_gat_func"].__module__ == "cogdl_amd.operators.fused_gat"
assert spmm_utils.CONFIGS["csr_edge_softmax"].__module__ == "cogdl_amd.operators.edge_softmax"
assert spmm_utils.CONFIGS["csrmhspmm"].__module__ == "cogdl_amd.operators.mhspmm"
for a, b in zip(grads["fused"], grads["unfused"]):
    np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)
report["gat_layer"] = "ok"

# ---- 2c. Graph.sample_adj(-1) + SAGELayer(mean) and MaxAggregator on cuda
z = gold("sage_layer")
rp = T(z["g_row_indptr"]); deg = rp[1:] - rp[:-1]
full = Graph(edge_index=(torch.repeat_interleave(torch.arange(deg.numel()), deg), T(z["g_col_indices"])),
             num_nodes=deg.numel())
nodes, block = full.sample_adj(T(z["batch"]), size=-1, replace=False)
assert np.array_equal(nodes.numpy(), z["nodes"]) and np.array_equal(block.col_indices.numpy(), z["block_col_indices"])
sage = SAGELayer(16, 12, aggr="mean").to(DEV)
with torch.no_grad():
    sage.fc.weight.copy_(T(z["fc_W"])); sage.fc.bias.copy_(T(z["fc_b"]))
    out = sage(block.to(DEV), T(z["x_src"]).to(DEV))
np.testing.assert_allclose(out.cpu().numpy(), z["out"], rtol=1e-4, atol=1e-5)
sage_max = SAGELayer(16, 12, aggr="max").to(DEV)
xs = T(z["x_src"]).to(DEV).requires_grad_()
o = sage_max(block.to(DEV), xs)
o.sum().backward()
assert torch.isfinite(o).all() and torch.isfinite(xs.grad).all()
import cogdl.layers.sage_layer as sl
report["sage_layer"] = "ok"

# ---- 2d. opt-in fused dispatcher front (install(fused_norm=True)): in_norm folded into the kernel, same SAGELayer output
import cogdl_amd
cogdl_amd.install(fused_norm=True)
import cogdl.layers.sage_layer as sage_mod
assert getattr(spmm_utils.spmm, "_cogdl_amd_fused", False) and getattr(sage_mod.spmm, "_cogdl_amd_fused", False)
with torch.no_grad():
    out_f = sage(block.to(DEV), T(z["x_src"]).to(DEV))
np.testing.assert_allclose(out_f.cpu().numpy(), z["out"], rtol=1e-4, atol=1e-5)
import cogdl_amd.fused
cogdl_amd.fused.uninstall()
assert not getattr(spmm_utils.spmm, "_cogdl_amd_fused", False)
report["fused_norm"] = "ok"

# ---- 3. experiment() through the reference's Trainer on cuda:0 (configs[0] shape, then configs[1] shape)
before = calls["spmm"]
ds = refpkg.cora_like(seed=0)
res_gpu, ms_gpu = refpkg.run_experiment(ds, model="gcn", epochs=6, cpu=False, seed=0)
assert calls["spmm"] - before >= 6 * 4, "Trainer epochs did not go through the HIP csr_spmm"
res_cpu, _ = refpkg.run_experiment(refpkg.cora_like(seed=0), model="gcn", epochs=6, cpu=True, seed=0)
report["cora_losses_gpu"], report["cora_losses_cpu"] = res_gpu["train_losses"], res_cpu["train_losses"]
report["cora_acc_gpu"], report["cora_acc_cpu"] = float(res_gpu["test_acc"]), float(res_cpu["test_acc"])
ds = refpkg.arxiv_like(seed=0)
res, ms = refpkg.run_experiment(ds, model="gcn", epochs=8, cpu=False, seed=0)
report["arxiv_losses"], report["arxiv_train_step_ms"] = res["train_losses"], ms
print("RESULT " + json.dumps(report))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(refpkg.STAGED, "cogdl")),
                    reason="staged reference package absent (make -C oracle ref in the build container)")
def test_unchanged_reference_layers_and_trainer_run_on_the_hip_operators():
    proc = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")]
    assert proc.returncode == 0 and lines, proc.stdout[-3000:] + proc.stderr[-5000:]
    rep = json.loads(lines[-1][7:])
    assert rep["gcn_layer"] == rep["gat_layer"] == rep["sage_layer"] == "ok"
    lg, lc = rep["cora_losses_gpu"], rep["cora_losses_cpu"]
    assert len(lg) == len(lc) == 6
    # dropout masks come from different generators on cpu / cuda: the trajectories agree in shape, not bitwise --
    # the first loss (before any update, dropout aside) and the downward trend are what is comparable
    assert lg[-1] < lg[0] and lc[-1] < lc[0]
    assert abs(lg[0] - lc[0]) < 0.35 * abs(lc[0]), (lg, lc)
    la = rep["arxiv_losses"]
    assert len(la) == 8 and la[-1] < la[0] and all(x == x for x in la)
    print("arxiv-shaped Trainer.train_step ms:", rep["arxiv_train_step_ms"])
