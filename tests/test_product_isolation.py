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
