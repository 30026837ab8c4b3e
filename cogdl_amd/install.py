"""Hook the HIP operators underneath an unmodified CogDL.

CogDL resolves its native operators by importing `cogdl.operators.<op>` lazily
(cogdl/utils/spmm_utils.py:21-40,137-146,241-248; cogdl/layers/sage_layer.py:22-25;
cogdl/data/data.py:18; cogdl/utils/graph_utils.py:7).  `install()` registers a meta-path finder
that serves those module names from `cogdl_amd.operators.<op>`, so GCNLayer / GATLayer /
SAGELayer and Graph pick the new path up without any source change.  Call it before
`import cogdl` (modules CogDL already imported are also rebound, and the dispatcher's CONFIGS
registry is reset so that it re-reads the names).

The alternative, equivalent integration is to replace each cogdl/operators/<op>.py by a
one-line shim (`from cogdl_amd.operators.<op> import *`) -- see INTEGRATION.md.
"""
import importlib
import importlib.abc
import importlib.util
import sys

REPLACED = ("spmm", "edge_softmax", "mhspmm", "scatter_max", "fused_gat", "sample", "ops")
_PREFIX = "cogdl.operators."


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_PREFIX) and fullname[len(_PREFIX):] in REPLACED:
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("cogdl_amd.operators." + spec.name[len(_PREFIX):])

    def exec_module(self, module):
        pass


_finder = None


def install(linear=False, fused_gat=True, fused_norm=False, narrow_side=False, fused_gat_dropout=False, structure_memo=False,
            metis=False, big_graphs=False):
    """Idempotent.  Returns the list of cogdl module names that are now served by cogdl_amd.
    fused_norm=True rebinds the dispatcher function `cogdl.utils.spmm_utils.spmm` itself (opt-in: that is no longer the
    unchanged dispatcher) to cogdl_amd.fused.spmm, which folds `out_norm * x` / `in_norm * x` into the kernel.
    fused_gat=True (default) also runs the dispatcher's own `initialize_fused_gat()` (utils/spmm_utils.py:241-248)
    once cogdl is imported and a GPU is present: nothing in the reference ever calls it, so GATLayer's fused branch
    (`check_fused_gat()`, layers/gat_layer.py:68) would otherwise stay dead even with a working fused operator.
    narrow_side=True rebinds GCNLayer.forward (opt-in, like fused_norm: no longer the unchanged layer) to
    cogdl_amd.fused's version, which aggregates at the input width where a layer widens ((A X) W instead of A (X W)).
    fused_gat_dropout=True rebinds GATLayer.forward (opt-in, same caveat) so that the branch the gat model takes by
    default -- attn_drop 0.5: leaky_relu(h_l[row] + h_r[col]) -> edge_softmax -> nn.Dropout -> mhspmm,
    layers/gat_layer.py:72-77 -- is one fused operator with the dropout mask regenerated from a seed (cogdl_amd/fused.py).
    `import cogdl` (or cogdl.layers) must have happened before; call install() again afterwards otherwise.
    structure_memo=True rebinds the two properties Graph.row_indptr / Graph.col_indices (opt-in, same caveat) so that the
    `.int()` copies the dispatcher makes on every call, and the content hash this library takes of them, happen once per
    structure (cogdl_amd/structure_memo.py).
    big_graphs=True rebinds the dispatcher function `spmm` (opt-in, like fused_norm) to cogdl_amd.big_dispatch's front: GPU
    graphs of 2^31 edges and more keep their int64 row pointer on the way to csrspmm (the reference's `.int()` wraps there).
    metis=True registers cogdl_amd.metis_compat as the module `metis` when the real package cannot be imported, so that
    ClusteredDataset / ClusteredLoader (cogdl/data/sampler.py:188-262) partition on the GPU instead of exiting.
    linear=True additionally routes torch.nn.functional.linear -- i.e. the unchanged nn.Linear inside every CogDL
    layer -- through cogdl_amd.linear (hand-written MFMA weight gradient for full-graph shapes)."""
    global _finder
    if _finder is None:
        _finder = _Finder()
        sys.meta_path.insert(0, _finder)
    for op in REPLACED:
        name = _PREFIX + op
        if name in sys.modules and not getattr(sys.modules[name], "__name__", "").startswith("cogdl_amd."):
            mod = importlib.import_module("cogdl_amd.operators." + op)
            sys.modules[name] = mod
            pkg = sys.modules.get("cogdl.operators")
            if pkg is not None:
                setattr(pkg, op, mod)
                if op == "ops":  # cogdl/operators/__init__.py:1-16 re-exports the s_* names from .ops
                    saved = pkg.__dict__.setdefault("_cogdl_amd_orig_ops", {})
                    for attr in dir(mod):
                        if attr.startswith("s_") and hasattr(pkg, attr):
                            saved.setdefault(attr, getattr(pkg, attr))
                            setattr(pkg, attr, getattr(mod, attr))
    _rebind_graph_build()
    if linear:
        from . import linear as _linear

        _linear.install()
    if fused_norm:
        from . import fused as _fused

        _fused.install()
    if narrow_side:
        from . import fused as _fused

        _fused.install_narrow_side()
    if fused_gat_dropout:
        from . import fused as _fused

        _import_target("cogdl.layers.gat_layer", "fused_gat_dropout")  # (the finder above already serves cogdl.operators.*)
        if not _fused.install_gat_dropout():
            raise _lib_error("install(fused_gat_dropout=True): GATLayer.forward could not be rebound")
    if big_graphs:
        from . import big_dispatch as _big

        _import_target("cogdl.utils.spmm_utils", "big_graphs")
        if not _big.install():
            raise _lib_error("install(big_graphs=True): cogdl.utils.spmm_utils.spmm could not be rebound")
    if structure_memo:
        from . import structure_memo as _memo

        _import_target("cogdl.data.data", "structure_memo")
        if not _memo.install():
            raise _lib_error("install(structure_memo=True): Graph.row_indptr / col_indices could not be rebound")
    if metis and "metis" not in sys.modules:
        try:
            importlib.import_module("metis")  # the real one wins where it exists
        except Exception:  # (ImportError, or the wrapper's RuntimeError when libmetis is missing)
            sys.modules["metis"] = importlib.import_module("cogdl_amd.metis_compat")
    su = sys.modules.get("cogdl.utils.spmm_utils")
    if su is not None:  # force the dispatcher to re-resolve the callables
        for k in ("spmm_flag", "mh_spmm_flag", "fused_gat_flag", "spmm_cpu_flag"):
            su.CONFIGS[k] = False
        for k in ("fast_spmm", "csrmhspmm", "csr_edge_softmax", "fused_gat_func", "fast_spmm_cpu"):
            su.CONFIGS[k] = None
        if fused_gat:
            import torch

            if torch.cuda.is_available():
                su.initialize_fused_gat()
    return [_PREFIX + op for op in REPLACED]




def _lib_error(msg):
    from ._lib import BackendError

    return BackendError(msg)


def _import_target(name, flag):
    """The opt-in rebinds patch a cogdl module: import it now (install() may be called before `import cogdl`) instead of
    silently doing nothing; an installation without cogdl gets an error naming the flag, not the slow path."""
    try:
        importlib.import_module(name)
    except ImportError as e:
        raise _lib_error("install(%s=True) needs the cogdl package (importing %s failed: %s)" % (flag, name, e)) from e


_GRAPH_BUILD_NAMES = ("coo2csr_index", "add_remaining_self_loops", "symmetric_normalization", "row_normalization")


def _rebind_graph_build():
    """CSR construction, self loops and normalisation stay on the GPU (cogdl_amd/graph_build.py; CPU
    tensors keep the reference's arithmetic).  Every cogdl module that holds one of the helpers by name is rebound
    (`from cogdl.utils import ...` copies the reference); effective for the modules already imported -- call install()
    again after `import cogdl` if it ran before (install() is idempotent)."""
    from . import graph_build

    for name, mod in list(sys.modules.items()):
        if mod is None or not (name == "cogdl" or name.startswith("cogdl.")):
            continue
        for fn in _GRAPH_BUILD_NAMES:
            cur = getattr(mod, fn, None)
            if cur is None or getattr(cur, "__module__", "") == "cogdl_amd.graph_build":
                continue
            if getattr(cur, "__module__", "") != "cogdl.utils.graph_utils":
                continue
            saved = mod.__dict__.setdefault("_cogdl_amd_orig_graph_build", {})
            saved.setdefault(fn, cur)
            setattr(mod, fn, getattr(graph_build, fn))


def uninstall():
    if "cogdl_amd.big_dispatch" in sys.modules:
        sys.modules["cogdl_amd.big_dispatch"].uninstall()
    if "cogdl_amd.structure_memo" in sys.modules:
        sys.modules["cogdl_amd.structure_memo"].uninstall()
    if "cogdl_amd.linear" in sys.modules:
        sys.modules["cogdl_amd.linear"].uninstall()
    if "cogdl_amd.fused" in sys.modules:
        sys.modules["cogdl_amd.fused"].uninstall()
    for name, mod in list(sys.modules.items()):
        if mod is not None and (name == "cogdl" or name.startswith("cogdl.")):
            for fn, orig in mod.__dict__.pop("_cogdl_amd_orig_graph_build", {}).items():
                setattr(mod, fn, orig)
    pkg = sys.modules.get("cogdl.operators")
    if pkg is not None:
        for attr, orig in pkg.__dict__.pop("_cogdl_amd_orig_ops", {}).items():
            setattr(pkg, attr, orig)
    global _finder
    if _finder is not None:
        sys.meta_path.remove(_finder)
        _finder = None
    for op in REPLACED:
        mod = sys.modules.get(_PREFIX + op)
        if mod is not None and getattr(mod, "__name__", "").startswith("cogdl_amd."):
            del sys.modules[_PREFIX + op]
