"""cogdl_amd -- MI355X (gfx950) sparse message-passing backend for CogDL's `cogdl.operators` API.

Scope is the one hot path of BASELINE.json's north_star: csr_spmm (fwd/bwd), csr2csc, sddmm,
edge_softmax, multi-head SpMM, scatter_max, fused GAT attention, sample_adj / COO->CSR, and a
vertex-sharded SpMM over RCCL; graphs of 2^31 edges and more run on one GPU through `csrspmm` with 64-bit row
pointers (cogdl_amd/bigcsr.py).  Everything else in CogDL is used as-is; `install()` slots these
operators underneath CogDL's unchanged dispatcher (cogdl/utils/spmm_utils.py) and Graph class.
"""
__version__ = "0.6.1"

from .install import install, uninstall  # noqa: F401
from .plan import transient_structures  # noqa: F401
