#!/usr/bin/env python3
"""The "GNN epoch time" of BASELINE.json's metric as the reference itself defines it: the wall time of one
cogdl.trainer.Trainer.train_step (cogdl/trainer/trainer.py:500-540; full-graph => one step = one epoch) of
`experiment(model='gcn', dataset=<ogbn-arxiv-shaped NodeDataset>)`, run by the REAL, unchanged reference package
(oracle/_ref/pkg) -- on cuda:0 on top of cogdl_amd.install(), and, for the baseline beside it, on the reference's own
CPU path (`cpu=True`, no install).  Prints one JSON object.  Measurement infrastructure, not product code.

    python tools/trainer_epoch.py gpu [epochs] [linear] [memo] [example]   |   python tools/trainer_epoch.py cpu [epochs] [example]

`example`: instead of the registered `gcn` model (2 layers, hidden 64) the model of the reference's ogbn-arxiv EXAMPLE
(examples/ogb/arxiv/gnn.py:10-44,148-156: 3 GCNLayers, hidden 256, batch-norm, relu, dropout 0.5; lr 0.01, no weight
decay) -- the configuration BASELINE.md section 3 timed at 5.94 s per training step on the CPU.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import refpkg  # noqa: E402


def example_gcn(in_feats, out_feats, hidden_size=256, num_layers=3, dropout=0.5):
    """examples/ogb/arxiv/gnn.py:10-44, restated (the examples directory is not part of the staged package): the
    reference's own GCNLayer with its dropout / batchnorm / activation arguments, sym_norm in forward."""
    import torch.nn as nn
    from cogdl.layers import GCNLayer
    from cogdl.models import BaseModel

    class GCN(BaseModel):
        def __init__(self):
            super().__init__()
            shapes = [in_feats] + [hidden_size] * (num_layers - 1) + [out_feats]
            last = num_layers - 1
            self.layers = nn.ModuleList([GCNLayer(shapes[i], shapes[i + 1], dropout=dropout if i != last else 0,
                                                  norm="batchnorm" if i != last else None,
                                                  activation="relu" if i != last else None) for i in range(num_layers)])

        def forward(self, graph):
            graph.sym_norm()
            h = graph.x
            for layer in self.layers:
                h = layer(graph, h)
            return h

    return GCN()


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "gpu"
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else (30 if mode == "gpu" else 3)
    linear = len(sys.argv) > 3 and "linear" in sys.argv[3:]
    memo = "memo" in sys.argv[3:]  # install(structure_memo=True): the Graph's int32 structure + its hash once per structure
    if mode == "cpu":
        # The untouched reference on its CPU path, exactly as in a GPU-less container: hide the GPU so that its
        # import-time `load(...spmm_kernel.cu)` JIT builds (hipify + hipcc, minutes, then `None`) are not attempted.
        os.environ["CUDA_VISIBLE_DEVICES"] = ""
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        import torch

        if torch.cuda.is_available():
            torch.cuda.is_available = lambda: False
    refpkg.setup(install=(mode == "gpu"), linear=linear)
    import torch

    if memo and mode == "gpu":
        import cogdl_amd

        cogdl_amd.install(linear=linear, structure_memo=True)

    if mode == "cpu":
        torch.set_num_threads(min(32, os.cpu_count() or 1))
    ds = refpkg.arxiv_like(seed=0)
    example = "example" in sys.argv[3:]
    fp16 = "fp16" in sys.argv[3:]  # the reference's own mixed precision: Trainer(fp16=True) = torch.cuda.amp.autocast + GradScaler
    extra = {"fp16": True} if fp16 else {}
    if example:
        model = example_gcn(ds.num_features, ds.num_classes)
        res, ms = refpkg.run_experiment(ds, model=model, epochs=epochs, cpu=(mode == "cpu"), seed=0, lr=0.01, weight_decay=0.0)
    else:
        res, ms = refpkg.run_experiment(ds, model="gcn", epochs=epochs, cpu=(mode == "cpu"), seed=0, **extra)
    steady = sorted(ms[min(5, len(ms) - 1):])  # the first epochs pay the CSR build, plan cache and allocator warm-up
    out = {"mode": mode, "epochs": epochs, "train_step_ms_median": steady[len(steady) // 2], "train_step_ms_min": steady[0],
           "train_step_ms_first": ms[0], "final_train_loss": res["train_losses"][-1],
           "val_acc": float(res.get("val_acc", float("nan"))),
           "threads": torch.get_num_threads() if mode == "cpu" else None,
           "linear": "cogdl_amd.linear" if linear else "torch", "structure_memo": bool(memo and mode == "gpu"), "fp16": fp16,
           "model": "examples/ogb/arxiv/gnn.py GCN: 3 layers, hidden 256, batchnorm, dropout 0.5" if example else "gcn (2 layers, hidden 64)"}
    print("TRAINER " + json.dumps(out))


if __name__ == "__main__":
    main()
