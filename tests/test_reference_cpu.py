"""BASELINE.json configs[0]: `experiment(model='gcn', dataset=<Cora>)` on the reference's CPU path -- the plumbing case.
The REAL reference package (its Trainer, data wrappers, GCN model, spmm dispatcher; staged copy under oracle/_ref/pkg
or /root/reference) trains a Cora-shaped synthetic NodeDataset twice in fresh interpreters:
  A. untouched reference (its own JIT-built spmm_cpu / torch scatter path);
  B. after cogdl_amd.install(): the dispatcher's CPU inference goes through cogdl_amd's spmm_cpu (host library),
     Graph construction through cogdl_amd's coo2csr_index, sampling through cogdl_amd's sampler.
Both legs must produce the same training-loss trajectory and the same accuracies (the CPU operators are bit-exact
restatements), and leg B must really have been served by cogdl_amd."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import refpkg  # noqa: E402

SCRIPT = r'''
import json, os, sys
ROOT, INSTALL = sys.argv[1], sys.argv[2] == "1"
sys.path.insert(0, ROOT)
from tools import refpkg
refpkg.setup(install=INSTALL)
import torch
torch.set_num_threads(4)
ds = refpkg.cora_like(seed=0)
res, ms = refpkg.run_experiment(ds, model="gcn", epochs=6, cpu=True, seed=0)
from cogdl.utils import spmm_utils
served = getattr(spmm_utils.CONFIGS.get("fast_spmm_cpu"), "__module__", None)
import cogdl.data.data as cdata
print("RESULT " + json.dumps({"res": {k: (float(v) if not isinstance(v, list) else v) for k, v in res.items()},
                              "spmm_cpu": served, "coo2csr": cdata.coo2csr_index.__module__,
                              "sampler": cdata.sample_adj_c.__module__ if cdata.sample_adj_c is not None else None}))
'''


def _leg(install):
    env = dict(os.environ, TORCH_EXTENSIONS_DIR=os.path.join("/tmp", "cogdl_ref_torch_ext"))
    proc = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, "1" if install else "0"], capture_output=True, text=True,
                          timeout=900, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")]
    assert proc.returncode == 0 and lines, proc.stdout[-1500:] + proc.stderr[-3000:]
    return json.loads(lines[-1][7:])


@pytest.mark.skipif(not refpkg.available(), reason="reference package not present (run `make -C oracle ref`)")
def test_experiment_gcn_cora_shaped_cpu_matches_untouched_reference():
    ours = _leg(True)
    assert ours["spmm_cpu"] == "cogdl_amd.operators.spmm", ours
    assert ours["coo2csr"] == "cogdl_amd.graph_build" and ours["sampler"] == "cogdl_amd.operators.sample", ours
    ref = _leg(False)
    assert ref["coo2csr"] != "cogdl_amd.graph_build"
    lo, lr = ours["res"]["train_losses"], ref["res"]["train_losses"]
    assert len(lo) == len(lr) == 6 and lo[-1] < lo[0]
    for a, b in zip(lo, lr):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (lo, lr)
    for k in ("test_acc", "val_acc"):
        assert abs(ours["res"][k] - ref["res"][k]) < 1e-9, (ours["res"], ref["res"])


SAGE_SCRIPT = r'''
import json, os, sys, tempfile
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from tools import refpkg
refpkg.setup(install=True)
import torch
torch.set_num_threads(4)
import cogdl.data.data as cdata
served = cdata.sample_adj_c.__module__
log = tempfile.NamedTemporaryFile(prefix="cogdl_sampler_pids_", delete=False).name
inner = cdata.sample_adj_c
def traced(*a, **k):                      # which PROCESS samples: the reference's loaders fork 4 workers
    with open(log, "a") as f:
        f.write("%d\n" % os.getpid())
    return inner(*a, **k)
cdata.sample_adj_c = traced
ds = refpkg.node_dataset(6000, 30000, 32, 5, seed=0)
res, ms = refpkg.run_experiment(ds, model="graphsage", epochs=3, cpu=True, seed=0, batch_size=256)
pids = [int(x) for x in open(log).read().split()]
os.unlink(log)
print("RESULT " + json.dumps({"losses": res["train_losses"], "test_acc": float(res["test_acc"]), "sampler": served,
                              "calls": len(pids), "worker_calls": sum(p != os.getpid() for p in pids),
                              "worker_pids": len({p for p in pids if p != os.getpid()})}))
'''


@pytest.mark.skipif(not refpkg.available(), reason="reference package not present (run `make -C oracle ref`)")
def test_experiment_graphsage_cpu_samples_in_the_references_forked_loader_workers():
    """BASELINE.json configs[3]'s model on the reference's CPU path: experiment(model='graphsage') with the reference's
    own GraphSAGEDataWrapper (cogdl/wrappers/data_wrapper/node_classification/graphsage_dw.py:30-39: NeighborSampler
    loaders with num_workers=4) on top of cogdl_amd.install().  Graph.sample_adj (cogdl/data/data.py:792-832) then calls
    this library's host sampler INSIDE the forked worker processes -- HIP-free, fork-safe -- for the random
    training hops and the full-neighbourhood test pass (sizes=[-1], Graphsage.inference); training makes progress."""
    env = dict(os.environ, TORCH_EXTENSIONS_DIR=os.path.join("/tmp", "cogdl_ref_torch_ext"))
    proc = subprocess.run([sys.executable, "-c", SAGE_SCRIPT, ROOT], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")]
    assert proc.returncode == 0 and lines, proc.stdout[-1500:] + proc.stderr[-3000:]
    r = json.loads(lines[-1][7:])
    assert r["sampler"] == "cogdl_amd.operators.sample", r
    assert r["worker_calls"] > 0 and r["worker_pids"] >= 2, r      # sampled in several forked workers, not in the parent
    losses = r["losses"]
    assert len(losses) == 3 and all(l == l and abs(l) < 1e3 for l in losses) and losses[-1] < losses[0], r
    assert 0.0 <= r["test_acc"] <= 1.0
