#!/usr/bin/env python3
"""The "GNN epoch time" of BASELINE.json's metric as the reference itself defines it: the wall time of one
cogdl.trainer.Trainer.train_step (cogdl/trainer/trainer.py:500-540; full-graph => one step = one epoch) of
`experiment(model='gcn', dataset=<ogbn-arxiv-shaped NodeDataset>)`, run by the REAL, unchanged reference package
(oracle/_ref/pkg) -- on cuda:0 on top of cogdl_amd.install(), and, for the baseline beside it, on the reference's own
CPU path (`cpu=True`, no install).  Prints one JSON object.  Measurement infrastructure, not product code.

    python tools/trainer_epoch.py gpu [epochs] [linear] [memo]   |   python tools/trainer_epoch.py cpu [epochs]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import refpkg  # noqa: E402


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
    res, ms = refpkg.run_experiment(ds, model="gcn", epochs=epochs, cpu=(mode == "cpu"), seed=0)
    steady = sorted(ms[min(5, len(ms) - 1):])  # the first epochs pay the CSR build, plan cache and allocator warm-up
    out = {"mode": mode, "epochs": epochs, "train_step_ms_median": steady[len(steady) // 2], "train_step_ms_min": steady[0],
           "train_step_ms_first": ms[0], "final_train_loss": res["train_losses"][-1],
           "val_acc": float(res.get("val_acc", float("nan"))),
           "threads": torch.get_num_threads() if mode == "cpu" else None,
           "linear": "cogdl_amd.linear" if linear else "torch", "structure_memo": bool(memo and mode == "gpu")}
    print("TRAINER " + json.dumps(out))


if __name__ == "__main__":
    main()
