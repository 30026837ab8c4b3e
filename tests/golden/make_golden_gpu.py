#!/usr/bin/env python3
"""Golden vectors that only a GPU can produce: outputs of the reference's OWN CUDA kernels, hipified and JIT-built for
gfx950 by the reference's own recipe (oracle/Makefile `ref` -> oracle/_ref/jit/, see oracle/build_ref_gpu_ops.py), run
on an MI355X.  Only kernels that are wave-size-agnostic (one thread per output element, no warp-level operation) are
legitimate here -- a warp-32 kernel recompiled for wave64 would not be the reference any more:

  scatter_max.npz   cogdl/operators/scatter_max/scatter_max.cu:5-28 (scatter_max_forward): out + argmax.
                    The kernel starts every non-empty row at FLT_MIN (smallest positive normal) and leaves max_id
                    UNINITIALISED when nothing beats it, so `argmax` is only meaningful where out > FLT_MIN; the file
                    carries `argmax_valid`.  Backward cannot be pinned: scatter_max_bp_cuda accumulates with atomicAdd
                    into torch::empty memory (scatter_max.cu:65-75).
  mhspmm.npz        cogdl/operators/spmm/multiheadSpmm.cu:6-51 (mhspmmSimple for f >= 32, mhspmm_1 below): sequential
                    CSR-order accumulation per output element; the device compiler contracts a*b+c into an FMA, so the
                    pin is 1e-6-relative, not bit-exact.

Run on the GPU box from the repo root:  python tests/golden/make_golden_gpu.py [outdir]   (default tests/golden; the
builder runs it under gpurun with outdir = gpurun_out/golden and commits the two files)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

FLT_MIN = np.float32(1.17549435e-38)


def ragged_csr(m, n_src, mean_deg, seed, hub=None):
    g = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, 2 * mean_deg + 1, (m,), generator=g)
    deg[torch.rand(m, generator=g) < 0.15] = 0
    if hub is not None:
        deg[hub[0]] = hub[1]
    rowptr = torch.zeros(m + 1, dtype=torch.long)
    torch.cumsum(deg, 0, out=rowptr[1:])
    colind = torch.randint(0, n_src, (int(rowptr[-1]),), generator=g)
    return rowptr.int(), colind.int()


def main(outdir):
    dev = "cuda:0"
    os.makedirs(outdir, exist_ok=True)
    sm = oracle.ref_gpu_op("scatter_max")
    cases = {}
    for name, (m, n_src, deg, k, mode) in {"pos_k16": (300, 200, 5, 16, "positive"), "mixed_k64": (500, 400, 8, 64, "mixed"),
                                            "ties_k7": (257, 50, 6, 7, "ties"), "hub_k128": (64, 3000, 4, 128, "mixed")}.items():
        rowptr, colind = ragged_csr(m, n_src, deg, seed=len(name) + k, hub=(3, 2500) if name.startswith("hub") else None)
        g = torch.Generator().manual_seed(k)
        feat = torch.randn(n_src, k, generator=g)
        if mode == "positive":
            feat = feat.abs() + 0.1
        elif mode == "ties":  # few distinct values: the first maximum in CSR order must win
            feat = torch.randint(-2, 3, (n_src, k), generator=g).float()
        out, arg = sm.scatter_max_fp(rowptr.to(dev), colind.to(dev), feat.to(dev))
        torch.cuda.synchronize()
        out, arg = out.cpu().numpy(), arg.cpu().numpy()
        cases[name + "_rowptr"], cases[name + "_colind"], cases[name + "_feat"] = rowptr.numpy(), colind.numpy(), feat.numpy()
        cases[name + "_out"], cases[name + "_argmax"] = out, arg
        cases[name + "_argmax_valid"] = out > FLT_MIN
    np.savez_compressed(os.path.join(outdir, "scatter_max.npz"), **cases)
    mh = oracle.ref_gpu_op("mhspmm")
    cases = {}
    for name, (v, deg, h, f) in {"h4_f8": (300, 6, 4, 8), "h8_f32": (200, 9, 8, 32), "h1_f64": (150, 5, 1, 64),
                                  "h3_f5": (120, 4, 3, 5)}.items():
        rowptr, colind = ragged_csr(v, v, deg, seed=h * 100 + f)
        g = torch.Generator().manual_seed(f)
        att, feat = torch.rand(colind.numel(), h, generator=g), torch.randn(v, h, f, generator=g)
        out = mh.mhspmm(rowptr.to(dev), colind.to(dev), att.to(dev), feat.to(dev))
        torch.cuda.synchronize()
        cases[name + "_rowptr"], cases[name + "_colind"] = rowptr.numpy(), colind.numpy()
        cases[name + "_att"], cases[name + "_feat"], cases[name + "_out"] = att.numpy(), feat.numpy(), out.cpu().numpy()
    np.savez_compressed(os.path.join(outdir, "mhspmm.npz"), **cases)
    print("wrote scatter_max.npz and mhspmm.npz to", outdir, "on", torch.cuda.get_device_name(0))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden"))
