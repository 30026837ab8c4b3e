#!/usr/bin/env python3
"""oracle/Makefile `ref`: JIT-build the reference's wave-size-agnostic CUDA operators for gfx950 exactly as the
reference's own loaders do (cogdl/operators/scatter_max.py:11-15, cogdl/operators/mhspmm.py:10-14) -- CHECKER
infrastructure: the modules are only ever loaded by tests/golden/make_golden_gpu.py on the GPU box.
usage: build_ref_gpu_ops.py <staged reference package dir> <output dir> <shim include dir>"""
import os
import sys

from torch.utils.cpp_extension import load

pkg, out, shim = (os.path.abspath(a) for a in sys.argv[1:4])
ops = os.path.join(pkg, "cogdl", "operators")


def build(name, sources, **kw):
    bd = os.path.join(out, name)
    os.makedirs(bd, exist_ok=True)
    load(name=name, sources=[os.path.join(ops, s) for s in sources], verbose=False, build_directory=bd, **kw)
    print("built", os.path.join(bd, name + ".so"))


build("scatter_max", ["scatter_max/scatter_max.cc", "scatter_max/scatter_max.cu"])
build("mhspmm", ["spmm/multiheadSpmm.cpp", "spmm/multiheadSpmm.cu"], extra_include_paths=[shim])
