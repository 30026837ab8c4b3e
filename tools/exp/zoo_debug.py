import json, os, sys, collections
import numpy as np
if not hasattr(np, "int"):
    np.int = int
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import refpkg
refpkg.setup(install=True)
import torch
import cogdl
from cogdl.utils import spmm_utils
LOG = []
_orig = torch.nn.functional.dropout
def wrapped(input, p=0.5, training=True, inplace=False):
    c = input.is_contiguous()
    if not c:
        input, inplace = input.contiguous(), False
    st = torch.cuda.get_rng_state()
    off = int.from_bytes(bytes(st[8:16].tolist()), "little")
    out = _orig(input, p, training, inplace)
    if training and p > 0:
        m = (out != 0)
        idx = torch.arange(m.numel(), device=m.device).view(m.shape)
        LOG.append((tuple(input.shape), c, off, int(m.sum()), int((idx * m).sum() % 1000003), float(input.float().abs().sum())))
    return out
torch.nn.functional.dropout = wrapped
def use_fallback(on, fused_gat=True):
    for k in ("spmm_flag", "mh_spmm_flag", "fused_gat_flag"):
        spmm_utils.CONFIGS[k] = bool(on)
    for k in ("fast_spmm", "csrmhspmm", "csr_edge_softmax", "fused_gat_func"):
        spmm_utils.CONFIGS[k] = None
    if not on:
        spmm_utils.initialize_fused_gat()
for fb in (False, True):
    use_fallback(fb)
    LOG.clear()
    ds = refpkg.node_dataset(2000, 10000, 32, 5, seed=1)
    res, ms = refpkg.run_experiment(ds, model="gat", epochs=1, cpu=False, seed=0)
    print("fallback" if fb else "hip", res["train_losses"])
    for l in LOG[:8]:
        print("   ", l)
