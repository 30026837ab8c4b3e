"""cogdl_amd/structure_memo.py (install(structure_memo=True)): the memoised `.int()` of a Graph's CSR index tensors.
CPU part: the tensor subclass itself (identity, invalidation, plain results, pickling).  GPU part: the unchanged
dispatcher on a Graph takes ONE structure hash for many calls and gives the same results as without the memo."""
import pickle

import pytest
import torch

from cogdl_amd.structure_memo import StructureMemo, _StructIndex


def test_int_is_memoised_and_invalidated_by_in_place_changes():
    memo = StructureMemo()
    base = torch.tensor([0, 2, 3, 5], dtype=torch.int64)
    t = _StructIndex(base, memo, 0)
    a = t.int()
    assert a.dtype == torch.int32 and a.tolist() == [0, 2, 3, 5] and a is memo.rowptr32 and a._cogdl_amd_struct is memo
    assert _StructIndex(base, memo, 0).int() is a  # a second property access: the same copy, no cast
    base[3] = 6  # in-place change of the Graph's structure: version counter moves, the memo misses
    b = _StructIndex(base, memo, 0).int()
    assert b is not a and b.tolist() == [0, 2, 3, 6] and memo.rowptr32 is b
    other = torch.tensor([0, 1, 1, 2], dtype=torch.int64)  # a new tensor behind the property (graph.row_indptr = ...)
    c = _StructIndex(other, memo, 0).int()
    assert c.tolist() == [0, 1, 1, 2]
    # with arguments .int() is torch's own
    assert _StructIndex(base, memo, 0).int(memory_format=torch.contiguous_format).dtype == torch.int32


def test_a_recycled_address_is_not_mistaken_for_the_old_source():
    """graph.to("cpu"); graph.to("cuda") frees the int64 source and may get the SAME address back for different contents
    (same shape, version 0): the memo keys on the tensor object it holds, not on its address."""
    memo = StructureMemo()
    base = torch.tensor([0, 2, 3, 5], dtype=torch.int64)
    a = _StructIndex(base, memo, 0).int()
    addr = base.data_ptr()
    del base
    for _ in range(64):  # (the memo keeps the old source alive, so its address cannot come back; whatever does come is a new object)
        other = torch.tensor([0, 1, 1, 9], dtype=torch.int64)
        got = _StructIndex(other, memo, 0).int()
        assert got.tolist() == [0, 1, 1, 9] and other.data_ptr() != addr or got is not a
        addr = other.data_ptr()


def test_the_wrapper_is_an_ordinary_tensor_everywhere_else():
    memo = StructureMemo()
    base = torch.arange(5, dtype=torch.int64)
    t = _StructIndex(base, memo, 1)
    assert isinstance(t, torch.Tensor) and t.dtype == torch.int64 and t.data_ptr() == base.data_ptr()
    for r in (t + 1, t[1:], t.float(), t.to(torch.int16), torch.cat([t, t]), t[1:] - t[:-1], t.clone()):
        assert type(r) is torch.Tensor
    assert t.long() is t and t.to(torch.int64) is t  # (no-op conversions hand the same object back, as for any tensor)
    assert int(t[-1]) == 4 and t.tolist() == [0, 1, 2, 3, 4] and t.numel() == 5
    assert torch.equal(torch.repeat_interleave(torch.arange(4), t[1:] - t[:-1]), torch.arange(4))


def test_memo_pickles_empty():
    memo = StructureMemo()
    _StructIndex(torch.arange(3, dtype=torch.int64), memo, 0).int()
    back = pickle.loads(pickle.dumps(memo))
    assert isinstance(back, StructureMemo) and back.rowptr32 is None and back.colind32 is None


def test_fingerprint_memo_follows_the_versions_of_the_int32_copies(monkeypatch):
    """Round-4 advisor: the memoised Fingerprint was vouched for by tensor identity only -- an in-place edit of the cached
    int32 copy reused the stale hash (and the cached transpose it keys).  And a Fingerprint made under hipGraph-capture
    replay (nothing hashed: no event) must not be remembered for a later eager lookup."""
    import cogdl_amd.structure_memo as sm

    made = []

    class FakeFingerprint:
        def __init__(self, rowptr, colind, n_cols):
            self.event = object() if not made or made[-1] != "replay" else None
            made.append("fp")

    monkeypatch.setattr(sm, "Fingerprint", FakeFingerprint)
    memo = StructureMemo()
    rp = _StructIndex(torch.tensor([0, 1, 2], dtype=torch.int64), memo, 0).int()
    ci = _StructIndex(torch.tensor([1, 0], dtype=torch.int64), memo, 1).int()
    a = memo.fingerprint(rp, ci, 2)
    assert memo.fingerprint(rp, ci, 2) is a and made == ["fp"]
    ci[0] = 0  # in-place edit of the cached copy: its version moves, the hash is taken again
    b = memo.fingerprint(rp, ci, 2)
    assert b is not a and made == ["fp", "fp"] and memo.fingerprint(rp, ci, 2) is b
    rp.add_(0)
    made.append("replay")  # the next Fingerprint is built as under tape replay: event None
    c = memo.fingerprint(rp, ci, 2)
    assert c.event is None
    d = memo.fingerprint(rp, ci, 2)  # not memoised: a fresh one (with an event) is made for the eager lookup
    assert d is not c and d.event is not None and memo.fingerprint(rp, ci, 2) is d
    assert memo.fingerprint(rp.clone(), ci, 2) is None  # not the memoised tensors at all


def test_install_opt_in_rebinds_fail_loudly_without_their_target(monkeypatch):
    """Round-4 advisor: install(structure_memo=True) / install(fused_gat_dropout=True) before `import cogdl` silently did
    nothing.  They import their target module now; where cogdl is not importable at all they raise, naming the flag."""
    import importlib
    import sys

    import cogdl_amd
    from cogdl_amd import _lib

    if "cogdl" in sys.modules or importlib.util.find_spec("cogdl") is not None:
        pytest.skip("cogdl is importable here")
    for flag in ("structure_memo", "fused_gat_dropout"):
        with pytest.raises(_lib.BackendError, match=flag):
            cogdl_amd.install(**{flag: True})
    cogdl_amd.uninstall()
