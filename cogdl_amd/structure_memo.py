"""Opt-in (`install(structure_memo=True)`): the int32 view of a Graph's CSR structure and its content fingerprint are
computed ONCE per structure instead of on every operator call.

CogDL's dispatcher hands the native operators fresh `.int()` copies of `graph.row_indptr` / `graph.col_indices` on every
call (cogdl/utils/spmm_utils.py:98-109, 172-188, 201-226): two cast kernels per call, and -- because the copies' addresses
never repeat -- this library has to hash their CONTENTS (one more kernel, an event, a pinned buffer) to find the cached
transpose for the backward pass.  In a full-graph epoch that is ~25 us of host time per SpMM call on a step that is
launch-bound on the host.

What is rebound: the two PROPERTIES `cogdl.data.Graph.row_indptr` / `.col_indices`.  They return the Graph's own int64
tensor wrapped in a thin torch.Tensor subclass (same storage, same version counter, torch functions disabled so every
operation on it yields plain tensors) whose only override is `.int()`: it returns a memoised int32 copy, keyed on the
identity of the int64 source (the tensor OBJECT, which the memo keeps alive, and its version counter) and kept on the
Adjacency object -- an in-place
change of the structure, a new tensor, `graph.to(device)` all miss the memo and rebuild it.  The int32 copy carries the
memo record, where the operators (cogdl_amd/operators/spmm.py, fused_gat.py, mhspmm.py) find the pair of tensors the
record vouches for and a Fingerprint computed once.  Nothing else about the Graph changes; `uninstall()` restores the
reference's properties."""
import sys

import torch

from .plan import Fingerprint

_ATTR = "__cogdl_amd_structure_memo__"  # (double underscores on both sides: outside Adjacency.keys, data.py:352-356)
_orig = {}


class StructureMemo:
    """rowptr32 / colind32 of ONE structure + its fingerprint (hashed on first use, then reused)."""
    __slots__ = ("src", "src_version", "rowptr32", "colind32", "_fp", "_fp_versions")

    def __init__(self):
        # The int64 SOURCE tensors are held, not just their addresses: a freed source's address can be handed to a new
        # tensor with the same shape and version 0 (graph.to("cpu"); graph.to("cuda")), which an address key would take
        # for the old one.  Identity of the object + its version counter cannot be recycled while the memo holds it.
        self.src = [None, None]
        self.src_version = [None, None]
        self.rowptr32 = self.colind32 = None
        self._fp = {}
        self._fp_versions = None  # (_version of rowptr32, colind32) the memoised hash was taken at

    def __reduce__(self):  # pickled with its Graph (Trainer.dist_train spawns ranks): events / pinned buffers stay behind
        return (StructureMemo, ())

    def fingerprint(self, rowptr, colind, n_cols):
        """The Fingerprint of (rowptr, colind) if those are the memoised tensors, else None."""
        if rowptr is not self.rowptr32 or colind is not self.colind32:
            return None
        versions = (rowptr._version, colind._version)
        if versions != self._fp_versions:  # an in-place edit of the cached int32 copies: the old hash (and the cached
            self._fp, self._fp_versions = {}, versions  # transpose it keys) no longer describes them
        fp = self._fp.get(n_cols)
        if fp is None:
            fp = Fingerprint(rowptr, colind, n_cols)
            if fp.event is not None:  # (made under hipGraph-capture replay: nothing was hashed -- not worth remembering,
                self._fp = {n_cols: fp}  # a later eager lookup would find no key behind it)
        return fp


class _StructIndex(torch.Tensor):
    """A Graph's int64 row_indptr / col_indices whose .int() is memoised (see the module docstring)."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def __new__(cls, base, memo, which):
        t = torch.Tensor._make_subclass(cls, base)
        t._memo, t._which, t._src = memo, which, base
        return t

    def int(self, *args, **kwargs):
        if args or kwargs:
            return self._src.int(*args, **kwargs)
        memo, which, base = self._memo, self._which, self._src
        cached = memo.rowptr32 if which == 0 else memo.colind32
        if cached is None or memo.src[which] is not base or memo.src_version[which] != base._version:
            cached = base.int()
            cached._cogdl_amd_struct = memo
            memo.src[which], memo.src_version[which] = base, base._version
            memo._fp = {}
            if which == 0:
                memo.rowptr32 = cached
            else:
                memo.colind32 = cached
        return cached


def lookup(rowptr, colind, n_cols):
    """The memoised Fingerprint for an operator call's (rowptr, colind), or None (then the caller hashes as before)."""
    memo = getattr(rowptr, "_cogdl_amd_struct", None)
    return memo.fingerprint(rowptr, colind, n_cols) if memo is not None else None


def _wrap(adj, base, which):
    if not torch.is_tensor(base) or base.dtype != torch.int64 or not base.is_cuda:
        return base  # CPU graphs: the reference's own path, untouched
    memo = adj.__dict__.get(_ATTR)
    if memo is None:
        memo = StructureMemo()
        adj.__dict__[_ATTR] = memo
    return _StructIndex(base, memo, which)


def install():
    mod = sys.modules.get("cogdl.data.data")
    if mod is None:
        return False
    cls = mod.Graph
    if cls in _orig:
        return True
    p_row, p_col = cls.__dict__["row_indptr"], cls.__dict__["col_indices"]
    _orig[cls] = (p_row, p_col)
    cls.row_indptr = property(lambda self: _wrap(self._adj, p_row.fget(self), 0), p_row.fset, p_row.fdel, p_row.__doc__)
    cls.col_indices = property(lambda self: _wrap(self._adj, p_col.fget(self), 1), p_col.fset, p_col.fdel, p_col.__doc__)
    return True


def uninstall():
    for cls, (p_row, p_col) in _orig.items():
        cls.row_indptr, cls.col_indices = p_row, p_col
    _orig.clear()
