"""The product never touches the oracle, the reference, or a CPU/PyTorch fallback for GPU ops."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def product_files():
    for d, _, files in os.walk(os.path.join(ROOT, "cogdl_amd")):
        if "build" in d.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                yield os.path.join(d, f)


def test_product_does_not_reference_oracle_or_reference_tree():
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|(oracle/)|(liboracle)|(/root/reference)", re.M)
    for p in product_files():
        src = open(p).read()
        m = bad.search(src)
        assert not m, "%s mentions %r" % (p, m.group(0))


def test_no_compat_layers():
    for p in product_files():
        src = open(p).read()
        for needle in ("__HIP_PLATFORM_AMD__", "import triton", "cuda_runtime.h", "hipify"):
            assert needle not in src, (p, needle)


def test_gpu_entry_points_fail_loudly_on_cpu_tensors():
    """No silent CPU route behind the GPU operators: handed CPU tensors (or, here, no GPU at all) every one of them
    raises BackendError -- only the operators the reference itself defines for CPU tensors (spmm_cpu, the host sampler,
    coo2csr, the graph-preprocessing helpers) compute anything on the host."""
    import pytest
    import torch

    from cogdl_amd import _lib, graph_build, pipeline
    from cogdl_amd.operators.edge_softmax import csr_edge_softmax
    from cogdl_amd.operators.mhspmm import csrmhspmm
    from cogdl_amd.operators.sample import sample_adj_padded
    from cogdl_amd.operators.scatter_max import scatter_max
    from cogdl_amd.operators.spmm import csrspmm, csrspmm_block, csrspmm_fused
    from cogdl_amd.plan import csr2csc

    rp = torch.tensor([0, 1, 2], dtype=torch.int32)
    ci = torch.tensor([1, 0], dtype=torch.int32)
    x = torch.ones(2, 4)
    calls = [
        lambda: csrspmm(rp, ci, x, None),
        lambda: csrspmm_fused(rp, ci, x, None, None, torch.ones(2)),
        lambda: csrspmm_block(rp, ci, x),
        lambda: csr_edge_softmax(rp, torch.ones(2, 1)),
        lambda: csrmhspmm(rp, ci, torch.ones(2, 1, 4), torch.ones(2, 1)),
        lambda: scatter_max(rp, ci, x),
        lambda: csr2csc(rp, ci, 2),
        lambda: graph_build.block_for_spmm(rp.long(), ci.long()),
        lambda: sample_adj_padded(rp.long(), ci.long(), torch.tensor([0]), 1),
        lambda: pipeline.gather_rows_by_id(x, torch.tensor([0])),
        lambda: pipeline.sample_blocks_padded(rp.long(), ci.long(), torch.tensor([0]), [1]),
    ]
    for i, call in enumerate(calls):
        with pytest.raises(_lib.BackendError):
            call()
            raise AssertionError("entry point %d computed something on CPU tensors" % i)
