"""The reference's OTHER entry paths on the HIP operators (round-4 verdict, "next" item 3):

  (a) `Trainer(fp16=True)` (cogdl/trainer/trainer.py:327,512-530: GradScaler + autocast; the dispatcher halves the edge
      weights, cogdl/utils/spmm_utils.py:104-105): GCN and GAT epochs through the unchanged Trainer on cuda:0, once on
      the HIP operators and once on the reference's own torch fallbacks (spmm_scatter / edge_softmax_val / per-head
      spmm, same GPU, same seed) -- loss trajectories within fp16 tolerance;
  (b) a model-zoo sweep: every reference model whose layers call spmm / edge_softmax / mh_spmm and that builds offline
      trains 2 epochs on a synthetic NodeDataset through cogdl.experiment() under install(), and again on the torch
      fallbacks: same losses (1e-4 relative on the first epoch, before any divergence of the weights), and the HIP entry
      points were really hit (per-entry-point call counters on the ctypes library object).
      Models this sweep cannot run, with the reason:  sagn -- its loss is NaN from the first epoch on this synthetic
      dataset on the reference's torch path as well;  srgcn, gtn, unet -- torch_sparse is not installed;  moe_gcn -- fmoe
      is not installed;  actgcn -- third_party/actnn is an empty submodule;  autognn -- optuna is a stub offline;
      gin, diffpool, sortpool, patchy_san, infograph -- graph-classification data wrappers (no node dataset);
      compgcn, rgcn -- knowledge-graph link prediction;  han -- heterogeneous dataset;  gcc, stgcn, stgat -- own datasets.
Follows /root/reference/tests/tasks/test_node_classification.py:47-100 (one short experiment per model).  Runs in a fresh
interpreter (the reference's modules never leak into the other tests)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import refpkg  # noqa: E402

pytestmark = pytest.mark.gpu

COMMON = r'''
import json, os, sys, collections
import numpy as np
if not hasattr(np, "int"):
    np.int = int  # (the reference still spells the alias numpy 1.24 removed)
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from tools import refpkg
refpkg.setup(install=True)
import torch
import cogdl
from cogdl.utils import spmm_utils
from cogdl_amd import _lib

# ---- per-entry-point call counters on the library object (every operator calls `lib.cogdl_hip_*` by attribute) ----
lib = _lib.hip()
COUNTS = collections.Counter()
def _counting(name, fn):
    def call(*a):
        COUNTS[name] += 1
        return fn(*a)
    return call
for _name in list(_lib.HIP_SIGNATURES):
    if _name.endswith("_workspace_bytes") or _name in ("cogdl_hip_set_tuning", "cogdl_hip_strerror", "cogdl_hip_abi_version",
                                                       "cogdl_hip_last_hip_error", "cogdl_hip_long_row_threshold",
                                                       "cogdl_hip_exact_row_edges", "cogdl_hip_csr_fingerprint"):
        continue
    setattr(lib, _name, _counting(_name, getattr(lib, _name)))

# ---- identical dropout masks in both legs (round-5 verdict, item 7).  torch's fused dropout kernel maps its random numbers to
# MEMORY positions: the contiguous [E, H] attention of the HIP operators and the transposed view the fallback stacks
# (cogdl/utils/spmm_utils.py:186-189; dropped at cogdl/layers/gat_layer.py:75) drew different masks from the same generator
# state.  Both legs now hand dropout the tensor in its LOGICAL (contiguous) layout, so element (e, h) of call k sees the
# same random number in both -- the default-argument attention models compare at the tolerance of every other model.
_torch_dropout = torch.nn.functional.dropout
def _dropout_logical_layout(input, p=0.5, training=True, inplace=False):
    if not input.is_contiguous():
        input, inplace = input.contiguous(), False
    return _torch_dropout(input, p, training, inplace)
torch.nn.functional.dropout = _dropout_logical_layout

# ---- and the same SOFTMAX in both legs.  With identical masks the default-argument gat / revgat legs still differed by 20 %
# (round 6, tools/exp/zoo_debug2.py): the fallback's edge_softmax_val (cogdl/utils/spmm_utils.py:149-152) HALVES every score
# while the largest exceeds 10 -- in place, which is not a shift: the result is the softmax of score / 2^k -- and a training
# forward (features scaled by 1 / (1 - 0.6) by the input dropout) produces such scores; evaluation and the *_nodrop legs do
# not, which is why those always agreed.  The operator the HIP kernel replaces, the reference's CUDA edge_softmax
# (cogdl/operators/edge_softmax/edge_softmax.cu:7-60), subtracts the row maximum: the true softmax.  The fallback leg
# therefore runs the reference's own lines WITHOUT that loop (everything else verbatim in meaning: exp, row sums through
# spmm, division); scores stay far below exp's fp32 range here (asserted).
def _edge_softmax_val_without_the_halving_loop(graph, edge_val):
    assert float(edge_val.max()) < 60.0
    with graph.local_graph():
        edge_val = torch.exp(edge_val)
        graph.edge_weight = edge_val
        x = torch.ones(graph.num_nodes, 1).to(edge_val.device)
        node_sum = spmm_utils.spmm(graph, x).squeeze()
        row = graph.edge_index[0]
        return edge_val / node_sum[row]
spmm_utils.edge_softmax_val = _edge_softmax_val_without_the_halving_loop

def use_fallback(on, fused_gat=True):
    """on: the dispatcher resolves nothing (flags set, callables None) -> spmm_scatter / edge_softmax_val / per-head spmm,
    the reference's own torch code on the same GPU.  off: resolve again -> the HIP operators (fused_gat=False: all but
    the fused GAT operator, so that GATLayer takes csr_edge_softmax + csrmhspmm)."""
    for k in ("spmm_flag", "mh_spmm_flag", "fused_gat_flag"):
        spmm_utils.CONFIGS[k] = bool(on)
    for k in ("fast_spmm", "csrmhspmm", "csr_edge_softmax", "fused_gat_func"):
        spmm_utils.CONFIGS[k] = None
    if not on:
        if fused_gat:
            spmm_utils.initialize_fused_gat()
        else:
            spmm_utils.CONFIGS["fused_gat_flag"] = True

def run(model, fallback, epochs=2, fused_gat=True, **kw):
    use_fallback(fallback, fused_gat)
    COUNTS.clear()
    ds = refpkg.node_dataset(2000, 10000, 32, 5, seed=1)
    res, ms = refpkg.run_experiment(ds, model=model, epochs=epochs, cpu=False, seed=0, **kw)
    return [float(l) for l in res["train_losses"]], dict(COUNTS)
'''

# (label, model, fused_gat on the HIP leg, model arguments).  The GAT legs run without dropout: torch's dropout kernel maps
# random numbers to elements differently for the contiguous [E, H] attention of the HIP operators and the transposed view
# the fallback stacks (utils/spmm_utils.py:186-189), so with dropout the two legs draw different masks -- both valid.
FP16 = [("gcn", "gcn", True, {}), ("gat_fused", "gat", True, {"dropout": 0.0, "attn_drop": 0.0}),
        ("gat_unfused", "gat", False, {"dropout": 0.0, "attn_drop": 0.0})]

FP16_SCRIPT = COMMON + r'''
report = {}
for label, model, fused, kw in json.loads(sys.argv[2]):
    hip_losses, hip_counts = run(model, False, epochs=6, fused_gat=fused, fp16=True, **kw)
    ref_losses, ref_counts = run(model, True, epochs=6, fp16=True, **kw)
    report[label] = {"hip": hip_losses, "fallback": ref_losses, "hip_counts": hip_counts, "fallback_counts": ref_counts}
print("RESULT " + json.dumps(report))
'''

NODROP = {"dropout": 0.0, "attn_drop": 0.0}
ZOO = [(m, m, True, {}) for m in
       ["gcn", "gat", "graphsage", "sage", "sgc", "gcnii", "ppnp", "mixhop", "drgat", "drgcn", "deepergcn", "disengcn", "grand",
        "dropedge_gcn", "pprgo", "sign", "gcnmix", "dgi", "mvgrl", "grace", "graphsaint", "gdc_gcn", "revgcn", "revgat", "revgen",
        "correct_smooth_mlp", "m3s", "gae", "vgae", "daegc", "unsup_graphsage", "agc"]]
ZOO += [("gat_nodrop_fused", "gat", True, NODROP), ("gat_nodrop_unfused", "gat", False, NODROP)]

ZOO_SCRIPT = COMMON + r'''
report = {}
for label, model, fused, kw in json.loads(sys.argv[2]):
    try:
        hip_losses, hip_counts = run(model, False, fused_gat=fused, **kw)
        ref_losses, ref_counts = run(model, True, **kw)
        report[label] = {"hip": hip_losses, "fallback": ref_losses, "hip_counts": hip_counts, "fallback_counts": ref_counts}
    except BaseException as e:
        import traceback
        report[label] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300]), "tb": traceback.format_exc()[-800:]}
print("RESULT " + json.dumps(report))
'''

# The HIP entry points whose calls prove that a model's aggregation ran on the library (not on a torch composition)
SPARSE_ENTRIES = ("cogdl_hip_csr_spmm", "cogdl_hip_csr_spmm_variant", "cogdl_hip_csr_spmm_acc", "cogdl_hip_csr_spmm_epilogue",
                  "cogdl_hip_mhspmm_eid", "cogdl_hip_mhspmm", "cogdl_hip_edge_softmax_fwd", "cogdl_hip_gat_fwd", "cogdl_hip_gspmm",
                  "cogdl_hip_scatter_max_fwd")
# Models that aggregate once, on the CPU, before training (pre-computed propagation) or not at all: no HIP call is expected
# on the training path -- they are in the sweep to show that install() does not break them.
NO_SPARSE_ON_GPU = {"correct_smooth_mlp", "sign"}  # (an MLP on features; SIGN propagates once, on the CPU, before training)
# (Rounds 4-5 compared gat / drgat / revgat at 50 % and revgen at 5e-3: "different dropout draws" and "sum order amplified by 14
#  GENConv layers".  Neither was the cause -- see COMMON: torch's layout-dependent dropout mask and, mostly, the halving loop of
#  the fallback's edge_softmax_val.  With both legs on the same mask and the same softmax every model compares at 1e-4; measured
#  in round 6: gat bit-equal, revgat / revgen / drgat 1e-7.)
LOOSE = {}


def _run(script, *args, timeout=2400):
    proc = subprocess.run([sys.executable, "-c", script, ROOT] + list(args), capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")]
    assert proc.returncode == 0 and lines, proc.stdout[-3000:] + proc.stderr[-5000:]
    return json.loads(lines[-1][7:])


needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(refpkg.STAGED, "cogdl")),
                               reason="staged reference package absent (make -C oracle ref in the build container)")


@needs_ref
def test_reference_trainer_fp16_on_the_hip_operators_matches_the_torch_fallback():
    rep = _run(FP16_SCRIPT, json.dumps(FP16))
    if os.environ.get("COGDL_AMD_ZOO_REPORT"):
        json.dump(rep, open(os.environ["COGDL_AMD_ZOO_REPORT"] + ".fp16", "w"))
    for model in [c[0] for c in FP16]:
        r = rep[model]
        hip, ref = r["hip"], r["fallback"]
        assert len(hip) == len(ref) == 6 and all(x == x and abs(x) < 1e4 for x in hip + ref), r
        assert sum(r["hip_counts"].get(k, 0) for k in SPARSE_ENTRIES) >= 6 * 2, r["hip_counts"]
        assert sum(r["fallback_counts"].get(k, 0) for k in SPARSE_ENTRIES) == 0, r["fallback_counts"]
        # half precision: 11 significand bits, and scatter_add's atomics sum in a different order every run
        assert abs(hip[0] - ref[0]) <= 5e-3 * abs(ref[0]), (model, hip, ref)
        for a, b in zip(hip, ref):
            assert abs(a - b) <= 3e-2 * max(1.0, abs(b)), (model, hip, ref)
        assert min(hip) < hip[0]
    assert rep["gat_fused"]["hip_counts"].get("cogdl_hip_gat_fwd", 0) >= 6 and rep["gat_unfused"]["hip_counts"].get("cogdl_hip_gat_fwd", 0) == 0
    assert rep["gat_unfused"]["hip_counts"].get("cogdl_hip_edge_softmax_fwd", 0) >= 6


def _run_chunks(script, items, n_chunks=3, timeout=2400):
    """The sweep in `n_chunks` interpreters side by side (an experiment is mostly host time: data wrappers, Trainer set-up,
    evaluation -- the GPU idles): a third of the wall time of one interpreter walking the whole list."""
    chunks = [items[i::n_chunks] for i in range(n_chunks)]
    procs = [subprocess.Popen([sys.executable, "-c", script, ROOT, json.dumps(c)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for c in chunks if c]
    rep = {}
    for proc in procs:
        out, err = proc.communicate(timeout=timeout)
        lines = [ln for ln in out.splitlines() if ln.startswith("RESULT ")]
        assert proc.returncode == 0 and lines, out[-3000:] + err[-5000:]
        rep.update(json.loads(lines[-1][7:]))
    return rep


@needs_ref
def test_model_zoo_trains_on_the_hip_operators_like_on_the_torch_fallback():
    rep = _run_chunks(ZOO_SCRIPT, ZOO)
    if os.environ.get("COGDL_AMD_ZOO_REPORT"):  # (calibration runs: the raw report beside the verdict)
        json.dump(rep, open(os.environ["COGDL_AMD_ZOO_REPORT"], "w"))
    failed = {m: r["error"] for m, r in rep.items() if "error" in r}
    assert not failed, failed
    summary = {}
    for model in [c[0] for c in ZOO]:
        r = rep[model]
        hip, ref = r["hip"], r["fallback"]
        sparse_calls = sum(r["hip_counts"].get(k, 0) for k in SPARSE_ENTRIES)
        summary[model] = (sparse_calls, hip[:2], ref[:2])
        assert sum(r["fallback_counts"].get(k, 0) for k in SPARSE_ENTRIES) == 0, (model, r["fallback_counts"])
        if model in NO_SPARSE_ON_GPU:
            continue
        assert sparse_calls >= 1, "%s never reached a HIP sparse operator: %s" % (model, r["hip_counts"])
        assert len(hip) == len(ref) and all(x == x for x in hip + ref), (model, hip, ref)
        if not hip:  # (agc: clustering, no Trainer.train_step -- its 55 propagation calls are what the counter shows)
            continue
        tol = LOOSE.get(model, 1e-4)
        assert abs(hip[0] - ref[0]) <= tol * max(1.0, abs(ref[0])), (model, hip, ref)
    print(json.dumps(summary))
