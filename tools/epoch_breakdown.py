#!/usr/bin/env python3
"""Fold the rocprofv3 kernel statistics of tools/epoch_probe.py into per-epoch categories.
Usage: python tools/epoch_breakdown.py <kernel_stats.csv>"""
import csv
import sys

STEPS = 45  # EPOCHS + WARMUP of tools/epoch_probe.py
CATS = [
    ("cogdl_amd csr_spmm (4 launches: F=64 and F=40, fwd + bwd)", ("rowreduce_main_kernel<cogdl::SpmmOp", "rowreduce_combine_kernel<cogdl::SpmmOp")),
    ("cogdl_amd MFMA linear kernels (fwd, grad_input, split-K weight gradient + reduce)", ("linear_fwd_kernel", "linear_wgrad")),
    ("cogdl_amd structure fingerprint", ("fingerprint",)),
    # our rocPRIM kernels come from the system headers (namespace ROCPRIM_400200_NS here); torch bundles its own copy
    # (ROCPRIM_400001_NS) and uses it for nonzero / sort inside boolean-mask indexing -- those fall through to that row
    ("cogdl_amd csr2csc / gather (plan build, first epoch only)", ("csr2csc", "gather_rows_kernel", "rowind_from_perm", "colptr_from", "ROCPRIM_400200")),
    ("hipBLASLt / rocBLAS GEMMs", ("Cijk_", "gemm", "rocblas")),
    ("torch cross_entropy (log_softmax + nll_loss fwd/bwd)", ("softmax", "nll_loss")),
    ("torch boolean-mask indexing out[train_mask] fwd/bwd", ("index", "nonzero", "masked", "vectorized_gather", "rocprim", "scan", "Scan", "DeviceSelect", "radix", "Radix", "cub", "sort")),
    ("torch Adam (multi_tensor_apply)", ("multi_tensor_apply", "adam", "Adam")),
    ("torch dropout", ("dropout", "Dropout")),
]
rows = list(csv.DictReader(open(sys.argv[1])))
tot = {c[0]: [0.0, 0] for c in CATS}
other = [0.0, 0]
other_names = []
for r in rows:
    ns, calls = float(r["TotalDurationNs"]), int(r["Calls"])
    for name, keys in CATS:
        if any(k in r["Name"] for k in keys):
            tot[name][0] += ns
            tot[name][1] += calls
            break
    else:
        other[0] += ns
        other[1] += calls
        other_names.append((ns, r["Name"][:90]))
total = sum(v[0] for v in tot.values()) + other[0]
print("# GPU-busy time per training epoch of bench.py's GCN (arxiv-shaped, hidden 64, 40 classes; MFMA linear on):")
print("# rocprofv3 --kernel-trace --stats of tools/epoch_probe.py, totals / %d steps; %.1f us busy per epoch" % (STEPS, total / STEPS / 1e3))
for name, (ns, calls) in sorted(list(tot.items()) + [("torch elementwise / fill / copy / reduce (everything else)", other)], key=lambda t: -t[1][0]):
    if calls:
        print("%9.1f us  %5.1f%%  %5.1f launches  %s" % (ns / STEPS / 1e3, 100 * ns / total, calls / STEPS, name))
print("# largest kernels in 'everything else':")
for ns, n in sorted(other_names, reverse=True)[:6]:
    print("#   %8.1f us/epoch  %s" % (ns / STEPS / 1e3, n))
