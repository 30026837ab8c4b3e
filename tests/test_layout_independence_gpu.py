"""Results must not depend on where the allocator happens to put the operands (round-3 verdict, item 2: "a latent
out-of-bounds read invalidates bit-exact claims the day allocation layout changes").

One seeded sequence of every row-wise operator -- csr_spmm (weighted, hub rows), csr_sddmm, mhspmm, mhsddmm, fused GAT
forward / backward with and without attention dropout (one lane group and column-tiled), edge_softmax forward / backward
(flat and row kernels), scatter_max, csr2csc through the hand-written radix sort -- runs in three child interpreters:
  plain      the caching allocator's layout (operands carved out of big blocks, freed blocks reused by later outputs)
  nocache    PYTORCH_NO_CUDA_MEMORY_CACHING=1: every tensor its own hipMalloc -- no neighbours to read from silently
  serialize  AMD_SERIALIZE_KERNEL=3: one kernel at a time, the way a counter-collecting profiler dispatches them
Every operator is deterministic, so the SHA-256 of every output must be identical in all three; a fault in any mode
fails the test with the operator that was running (tools/gpu_round.sh `hunt` does the same at Reddit scale)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r'''
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import torch
from cogdl_amd import _lib, synth
from cogdl_amd.operators.edge_softmax import csr_edge_softmax
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func
from cogdl_amd.operators.mhspmm import mhsddmm_raw, mhspmm_raw
from cogdl_amd.operators.scatter_max import scatter_max
from cogdl_amd.operators.spmm import csr_sddmm_raw, csr_spmm_raw
from cogdl_amd.plan import csr2csc

DEV = "cuda:0"
def rand(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(DEV)
def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().contiguous().view(-1).view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()

out = {}
def run(name, fn):
    print("running", name, flush=True)
    r = fn()
    torch.cuda.synchronize()
    out[name] = digest(*(r if isinstance(r, (tuple, list)) else (r,)))

graphs = {"rmat": synth.scaled(40000, 14, seed=3, topology="rmat", norm="sym"),
          "hubs": synth.hub_csr(600, 600, hubs=((3, 129), (4, 1000), (17, 5000), (18, 257), (40, 128), (599, 3000)), seed=1)}
for gname, g in graphs.items():
    n, nnz = g.num_nodes, g.nnz
    rp, ci, w = g.rowptr.to(DEV), g.colind.to(DEV), g.weight.to(DEV)
    for f in (40, 64, 128):
        x = rand(n, f, seed=f)
        run("%s csr_spmm F%d" % (gname, f), lambda: csr_spmm_raw(rp, ci, w, x))
    run(gname + " csr_sddmm", lambda: csr_sddmm_raw(rp, ci, rand(n, 64, seed=1), rand(n, 64, seed=2)))
    att = rand(nnz, 8, seed=5)
    run(gname + " mhspmm", lambda: mhspmm_raw(rp, ci, att, rand(n, 8, 8, seed=6)))
    run(gname + " mhsddmm", lambda: mhsddmm_raw(rp, ci, rand(n, 8, 8, seed=7), rand(n, 8, 8, seed=6)))
    for h, f, p in ((8, 8, 0.0), (8, 8, 0.5), (1, 41, 0.5), (8, 64, 0.5), (6, 12, 0.0)):
        def gat():
            ar, ac, ft = (t.requires_grad_() for t in (rand(n, h, seed=11), rand(n, h, seed=12), rand(n, h, f, seed=13)))
            o = fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, p, seed=99)
            o.backward(rand(n, h, f, seed=14))
            return o, ar.grad, ac.grad, ft.grad
        run("%s gat H%dF%d p%.1f" % (gname, h, f, p), gat)
    for h, dt in ((8, torch.float32), (8, torch.bfloat16), (3, torch.float32), (1, torch.float32)):
        def es():
            v = rand(nnz, h, seed=21).to(dt).requires_grad_()
            o = csr_edge_softmax(rp, v)
            o.backward(rand(nnz, h, seed=22).to(dt))
            return o, v.grad
        run("%s edge_softmax H%d %s" % (gname, h, dt), es)
    run(gname + " scatter_max", lambda: scatter_max(rp, ci, rand(n, 64, seed=31)))
    def transpose():
        _lib.hip().cogdl_hip_set_tuning(10, 2)  # the hand-written radix sort at every size
        try:
            pl = csr2csc(rp, ci, n)
        finally:
            _lib.hip().cogdl_hip_set_tuning(10, 0)
        return pl.colptr, pl.rowind, pl.perm
    run(gname + " csr2csc", transpose)
print("RESULT " + json.dumps(out))
'''

MODES = {"plain": {}, "nocache": {"PYTORCH_NO_CUDA_MEMORY_CACHING": "1"}, "serialize": {"AMD_SERIALIZE_KERNEL": "3"}}


def test_every_operator_gives_the_same_bits_under_three_allocation_and_dispatch_regimes():
    results = {}
    for mode, extra in MODES.items():
        proc = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=600,
                              env=dict(os.environ, **extra))
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")]
        last = [ln for ln in proc.stdout.splitlines() if ln.startswith("running")][-1:]
        assert proc.returncode == 0 and lines, "%s: rc %d, last %s\n%s" % (mode, proc.returncode, last, proc.stderr[-3000:])
        results[mode] = json.loads(lines[-1][7:])
    ref = results["plain"]
    assert len(ref) >= 30
    for mode in ("nocache", "serialize"):
        diff = [k for k in ref if results[mode].get(k) != ref[k]]
        assert not diff, "%s differs from plain in: %s" % (mode, diff)
