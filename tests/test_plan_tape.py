"""The plan tape behind cogdl_amd.graphs.capture (no GPU: the cache and the fingerprint are stubbed): lookups are
recorded in call order and replayed in the same order; a captured step that asks for a different sequence is refused."""
import types

import pytest

from cogdl_amd import _lib, plan


class _FP:
    def __init__(self, m, nnz, n_cols, key):
        self.meta = (0, m, nnz, n_cols)
        self._k = key

    def key(self):
        return self.meta + (self._k,)


def _plan(m, nnz, n_cols):
    return types.SimpleNamespace(m=m, nnz=nnz, n_cols=n_cols, nbytes=lambda: 12 * nnz)


def test_record_then_replay_in_order(monkeypatch):
    cache = plan.PlanCache(budget_bytes=1 << 20)
    built = []

    def fake_csr2csc(rowptr, colind, n_cols):
        p = _plan(rowptr, colind, n_cols)  # (the stub passes m and nnz in place of the tensors)
        built.append(p)
        return p

    monkeypatch.setattr(plan, "csr2csc", fake_csr2csc)
    tape = plan.PlanTape()
    plan.set_tape(tape)
    try:
        a = cache.get(_FP(10, 30, 10, 111), 10, 30, 10)
        b = cache.get(_FP(7, 9, 5, 222), 7, 9, 5)
        a2 = cache.get(_FP(10, 30, 10, 111), 10, 30, 10)  # a cache hit is recorded, too
    finally:
        plan.set_tape(None)
    assert a2 is a and len(built) == 2 and tape.plans == [a, b, a]
    tape.mode, tape.pos = "replay", 0
    plan.set_tape(tape)
    try:
        # while capturing nothing is hashed: key() must not be called
        class NoKey(_FP):
            def key(self):
                raise AssertionError("fingerprint read during capture")

        assert cache.get(NoKey(10, 30, 10, 0), None, None, 10) is a
        assert cache.get(NoKey(7, 9, 5, 0), None, None, 5) is b
        with pytest.raises(_lib.BackendError):  # a different operator sequence than the recorded one
            cache.get(NoKey(8, 8, 8, 0), None, None, 8)
    finally:
        plan.set_tape(None)
    tape.pos = len(tape.plans)
    plan.set_tape(tape)
    try:
        with pytest.raises(_lib.BackendError):  # more lookups than recorded
            cache.get(_FP(10, 30, 10, 111), None, None, 10)
    finally:
        plan.set_tape(None)


def test_launch_decisions_are_recorded_and_replayed_by_call_site():
    """plan.taped_choice: without a tape the decision is taken; while recording it is taken and kept; while replaying the
    kept value comes back without deciding again -- and a step that asks from another call site, or more often, is refused."""
    calls = []

    def decide(v):
        def f():
            calls.append(v)
            return v
        return f

    assert plan.taped_choice("a", decide(1)) == 1 and calls == [1]
    tape = plan.PlanTape()
    plan.set_tape(tape)
    try:
        assert plan.recording() and not plan.replaying()
        assert plan.taped_choice("csr_spmm.forward", decide((7, "planA"))) == (7, "planA")
        assert plan.taped_choice("csr_spmm.backward", decide(None)) is None
    finally:
        plan.set_tape(None)
    assert tape.choices == [("csr_spmm.forward", (7, "planA")), ("csr_spmm.backward", None)] and calls == [1, (7, "planA"), None]
    tape.mode = "replay"
    plan.set_tape(tape)
    try:
        assert plan.replaying() and not plan.recording()
        assert plan.taped_choice("csr_spmm.forward", decide("never")) == (7, "planA")
        with pytest.raises(_lib.BackendError):  # another call site than the recorded one
            plan.taped_choice("fused_gat.backward", decide("never"))
        with pytest.raises(_lib.BackendError):  # more decisions than recorded
            plan.taped_choice("csr_spmm.forward", decide("never"))
    finally:
        plan.set_tape(None)
    assert "never" not in calls


def test_transient_structures_context_nests_and_resets():
    from cogdl_amd import plan, transient_structures

    assert not plan.transient()
    with transient_structures():
        assert plan.transient()
        with transient_structures():
            assert plan.transient()
        assert plan.transient()
    assert not plan.transient()
    try:
        with transient_structures():
            raise ValueError("boom")
    except ValueError:
        pass
    assert not plan.transient()  # left on the exception path too
