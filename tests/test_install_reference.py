"""Drop-in check against the REAL reference package (build container only: skipped where /root/reference is absent,
e.g. on the GPU box).  cogdl_amd.install() must make CogDL's own, unchanged code -- Graph, the spmm dispatcher,
Graph.sample_adj, coo2csr_index, GCNLayer -- run on the cogdl_amd operators.  CPU-only here, so what is exercised is
the host library (sampler, COO->CSR, CPU SpMM) plus the module plumbing; the GPU kernels behind the same names are
covered by the -m gpu suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("COGDL_REFERENCE", "/root/reference")

SCRIPT = r'''
import os, shutil, sys, tempfile
ROOT, REF = sys.argv[1], sys.argv[2]
scratch = tempfile.mkdtemp(prefix="cogdl_refcopy_")          # the reference writes into its own tree when imported
shutil.copytree(os.path.join(REF, "cogdl"), os.path.join(scratch, "cogdl"))
sys.dont_write_bytecode = True
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden", "_stubs"), scratch]
import torch
import cogdl_amd
served = cogdl_amd.install()
import cogdl
from cogdl.data import Graph
from cogdl.layers import GCNLayer
from cogdl.utils import spmm_utils, graph_utils
import cogdl.data.data as cdata
cogdl_amd.install()                                            # again, now that cogdl is imported: rebinds coo2csr_index

# 1. the operator modules CogDL sees are ours
import cogdl.operators.spmm as m_spmm, cogdl.operators.sample as m_sample, cogdl.operators.edge_softmax as m_es
assert m_spmm.__name__ == "cogdl_amd.operators.spmm" and m_sample.__name__ == "cogdl_amd.operators.sample"
assert m_es.__name__ == "cogdl_amd.operators.edge_softmax", m_es.__name__
assert graph_utils.coo2csr_index.__module__ == "cogdl_amd.graph_build"
assert cdata.coo2csr_index.__module__ == "cogdl_amd.graph_build"

# 2. Graph construction (COO -> CSR through our host operator) reproduces the documented example (graph.rst:53-61)
edges = torch.tensor([[0, 1], [1, 3], [2, 1], [4, 2], [0, 3]]).t().contiguous()
g = Graph(edge_index=edges, x=torch.randn(5, 8))
assert g.row_indptr.tolist() == [0, 2, 3, 4, 4, 5] and g.col_indices.tolist() == [1, 3, 3, 1, 2]

# 3. the dispatcher's CPU inference path runs our spmm_cpu and equals the scatter fallback
torch.manual_seed(0)
n = 300
ei = torch.randint(0, n, (2, 2500))
g = Graph(edge_index=ei, x=torch.randn(n, 16))
g.add_remaining_self_loops(); g.sym_norm()
x = torch.randn(n, 16)
with torch.no_grad():
    y = spmm_utils.spmm(g, x)
assert spmm_utils.CONFIGS["fast_spmm_cpu"] is m_spmm.spmm_cpu, "dispatcher did not pick up cogdl_amd's spmm_cpu"
row, col = g.edge_index
want = torch.zeros_like(x).index_add_(0, row, x[col] * g.edge_weight.unsqueeze(-1))
assert torch.allclose(y, want, rtol=1e-5, atol=1e-6)

# 4. an unchanged GCNLayer trains one step through the dispatcher
layer = GCNLayer(16, 4)
out = layer(g, x.requires_grad_())
out.sum().backward()
assert layer.linear.weight.grad is not None and out.shape == (n, 4)

# 5. Graph.sample_adj (unchanged data.py:792-832) on our sampler: seeds first, padded row_ptr, valid relabelling
batch = torch.tensor([5, 17, 3])
nodes, sub = g.sample_adj(batch, 4, replace=False)
assert nodes[:3].tolist() == [5, 17, 3]
rp, ci = sub.row_indptr, sub.col_indices
assert rp.numel() == nodes.numel() + 1 and int(ci.max()) < nodes.numel()
deg = g.row_indptr[batch + 1] - g.row_indptr[batch]
assert (rp[1:4] - rp[0:3]).tolist() == torch.clamp(deg, max=4).tolist()
nodes_all, sub_all = g.sample_adj(batch, -1)
for i, b in enumerate(batch.tolist()):
    mine = sorted(nodes_all[sub_all.col_indices[sub_all.row_indptr[i]:sub_all.row_indptr[i + 1]]].tolist())
    assert mine == sorted(g.col_indices[g.row_indptr[b]:g.row_indptr[b + 1]].tolist())

# 6. the message operators CogDL re-exports (cogdl/operators/__init__.py) are ours and pass the reference's own
#    exact-equality checks (tests/test_ops.py:59-110) on a reference Graph
import cogdl.operators as cops
assert cops.ops.__name__ == "cogdl_amd.operators.ops" and cops.s_add_e_sum.__module__ == "cogdl_amd.operators.ops"
tg = Graph(x=torch.randn(100, 10), edge_index=torch.randint(0, 100, (2, 200)))
ea = torch.randn(tg.num_edges, 10)
trow, tcol = tg.edge_index
msg = tg.x[tcol] * ea
want = torch.zeros(100, 10).scatter_add_(0, trow.view(-1, 1).expand(200, 10), msg)
assert (cops.s_mul_e_sum(tg, tg.x, ea) == want).all()
tdeg = torch.zeros(100).scatter_add_(0, trow, torch.ones(200))
tinv = tdeg.pow(-1); tinv[torch.isinf(tinv)] = 0
assert (cops.s_mul_e_mean(tg, tg.x, ea) == want * tinv.view(-1, 1)).all()
assert (cops.s_sub_t(tg, tg.x) == tg.x[tcol] - tg.x[trow]).all()
# 7. opt-in fused dispatcher front: CPU tensors / graphs without norm vectors are forwarded to the reference's own spmm
cogdl_amd.install(fused_norm=True)
import cogdl.layers.sage_layer as sage_mod
assert getattr(spmm_utils.spmm, "_cogdl_amd_fused", False) and getattr(sage_mod.spmm, "_cogdl_amd_fused", False)
with torch.no_grad():
    y2 = spmm_utils.spmm(g, x.detach())
assert torch.equal(y2, y)
nodes_b, block = g.sample_adj(torch.tensor([1, 2, 3]), -1)
block.row_norm()                                             # CSR-only graph: in_norm vector (data.py:248-252)
assert block.in_norm is not None
xb = torch.randn(nodes_b.numel(), 5)
with torch.no_grad():
    agg = sage_mod.spmm(block, xb)                           # CPU tensor -> the reference's path, in_norm applied there
deg = (block.row_indptr[1:] - block.row_indptr[:-1]).float()
want_b = torch.zeros(nodes_b.numel(), 5).index_add_(
    0, torch.repeat_interleave(torch.arange(deg.numel()), deg.long()), xb[block.col_indices])
want_b = want_b / deg.clamp(min=1).view(-1, 1)
assert torch.allclose(agg, want_b, rtol=1e-5, atol=1e-6)
import cogdl_amd.fused
cogdl_amd.fused.uninstall()
assert not getattr(spmm_utils.spmm, "_cogdl_amd_fused", False) and not getattr(sage_mod.spmm, "_cogdl_amd_fused", False)

# 8. the graph-preprocessing helpers are rebound everywhere they are held by name; CPU tensors keep the reference's results
import cogdl.utils.graph_utils as gu
for fn in ("add_remaining_self_loops", "symmetric_normalization", "row_normalization", "coo2csr_index"):
    assert getattr(gu, fn).__module__ == "cogdl_amd.graph_build", fn
assert cogdl.utils.row_normalization.__module__ == "cogdl_amd.graph_build"
orig = gu._cogdl_amd_orig_graph_build
r0, c0 = torch.randint(0, 50, (2, 400))
w0 = torch.rand(400)
(ra, ca), wa = gu.add_remaining_self_loops((r0, c0), w0, 1, 50)
(rb, cb), wb = orig["add_remaining_self_loops"]((r0, c0), w0, 1, 50)
assert torch.equal(ra, rb) and torch.equal(ca, cb) and torch.equal(wa, wb)
assert torch.equal(gu.symmetric_normalization(50, ra, ca, wa), orig["symmetric_normalization"](50, rb, cb, wb))
assert torch.equal(gu.row_normalization(50, ra, ca, wa), orig["row_normalization"](50, rb, cb, wb))

# 9. install(narrow_side=True): GCNLayer aggregates at the input width where it widens -- (A X) W + (A 1) b -- and keeps
#    the reference's order where it narrows; outputs and every gradient equal the unchanged layer's to fp32 reassociation
import copy
from cogdl.layers import GCNLayer
gN = Graph(edge_index=(r0, c0), edge_weight=w0, num_nodes=50)
gN.sym_norm()
for fin, fout in ((8, 32), (32, 8), (16, 16)):
    torch.manual_seed(fin)
    layer = GCNLayer(fin, fout, activation="relu", residual=True, dropout=0.0)
    twin = copy.deepcopy(layer)
    xN = torch.randn(50, fin, requires_grad=True)
    xT = xN.detach().clone().requires_grad_()
    want_o = layer(gN, xN)
    want_o.square().sum().backward()
    cogdl_amd.install(narrow_side=True)
    assert GCNLayer.forward.__module__ == "cogdl_amd.fused"
    got_o = twin(gN, xT)
    got_o.square().sum().backward()
    cogdl_amd.fused.uninstall_narrow_side()
    assert GCNLayer.forward.__module__ == "cogdl.layers.gcn_layer"
    assert torch.allclose(got_o, want_o, rtol=1e-5, atol=1e-6), (fin, fout)
    assert torch.allclose(xT.grad, xN.grad, rtol=1e-4, atol=1e-5), (fin, fout)
    for pa, pb in zip(twin.parameters(), layer.parameters()):
        assert torch.allclose(pa.grad, pb.grad, rtol=1e-4, atol=1e-5), (fin, fout)
# 10. install(big_graphs=True): the 64-bit front of the dispatcher is in place everywhere `spmm` is held by name; CPU tensors
#     (and GPU graphs below 2^31 edges) are forwarded to the reference's own function, bit for bit; uninstall restores it
cogdl_amd.install(big_graphs=True)
import cogdl_amd.big_dispatch as big_dispatch
import cogdl.layers.gcn_layer as gcn_mod
assert getattr(spmm_utils.spmm, "_cogdl_amd_big", False) and getattr(gcn_mod.spmm, "_cogdl_amd_big", False)
with torch.no_grad():
    y3 = spmm_utils.spmm(g, x.detach())
assert torch.equal(y3, y)
big_dispatch.BIG_EDGES = 1                                    # even "big" CPU graphs stay on the reference's path
with torch.no_grad():
    assert torch.equal(spmm_utils.spmm(g, x.detach()), y)
big_dispatch.BIG_EDGES = 2 ** 31 - 2 ** 20
big_dispatch.uninstall()
assert not getattr(spmm_utils.spmm, "_cogdl_amd_big", False) and not getattr(gcn_mod.spmm, "_cogdl_amd_big", False)
shutil.rmtree(scratch, ignore_errors=True)
print("INSTALL-OK", served)
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "cogdl")), reason="reference package not present")
def test_reference_package_runs_on_cogdl_amd_after_install():
    proc = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, REF], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "INSTALL-OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]
