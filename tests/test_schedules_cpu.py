"""Plan-time schedules (round 6, second half), host logic only -- no kernel runs here:
  * cogdl_amd/xcdplan.py: an XCD's virtual rows in order of decreasing length -- the same virtual rows (content, records,
    owner XCD) as in (row, piece) order, only their slots differ;
  * cogdl_amd/bigcsr.py: window_degree_order -- a permutation, decreasing degree inside every window, windows in place;
  * xcdplan.ordered_wanted: the skew test and its preconditions.
The kernels that consume them are checked against the oracle in tests/test_xcd_gpu.py / tests/test_bigcsr_gpu.py."""
import numpy as np
import torch

from cogdl_amd import bigcsr, synth, xcdplan


def _virtual_rows(plan):
    vp, vd, vc, ei = (t.numpy() for t in (plan.vrowptr, plan.vdesc, plan.vcol, plan.eid))
    rows = {}
    for s in range(plan.n_slots):
        if vd[s, 0] >= 0:
            rows[(int(vd[s, 0]), int(vd[s, 1]))] = (tuple(vc[vp[s]:vp[s + 1]]), tuple(ei[vp[s]:vp[s + 1]]), (s // xcdplan.UNIT) % xcdplan.XCDS)
    return rows, np.diff(vp)


def test_slot_order_by_length_keeps_every_virtual_row(monkeypatch):
    g = synth.hub_csr(3000, 2500, base_deg=6, seed=3)
    monkeypatch.setattr(xcdplan, "SORT_BY_LENGTH", False)
    a, len_a = _virtual_rows(xcdplan.build(g.rowptr, g.colind, split=64, piece=32))
    monkeypatch.setattr(xcdplan, "SORT_BY_LENGTH", True)
    plan = xcdplan.build(g.rowptr, g.colind, split=64, piece=32)
    b, len_b = _virtual_rows(plan)
    assert a == b and sorted(len_a) == sorted(len_b)  # same (row, record) -> (columns, edge ids, owner XCD)
    assert int(len_b.sum()) == g.nnz
    per_xcd = len_b.reshape(-1, xcdplan.XCDS, xcdplan.UNIT)
    for x in range(xcdplan.XCDS):
        seq = per_xcd[:, x, :].reshape(-1)
        assert np.all(np.diff(seq) <= 0), x  # decreasing along the XCD's stream (padding slots, length 0, at its end)
    # what it buys: lane-slots filled when 8 lane groups share a wave (chunks of 8 edges)
    fill = lambda l: l.sum() / (np.ceil(l.reshape(-1, 8).max(1) / 8) * 64).sum()
    assert fill(len_b) > fill(len_a) + 0.1


def test_window_degree_order_is_a_windowed_permutation():
    g = synth.hub_csr(1000, 900, base_deg=5, seed=1)
    order = bigcsr.window_degree_order(g.rowptr.long(), window=128).numpy()
    assert np.array_equal(np.sort(order), np.arange(g.num_nodes))
    deg = g.degrees().numpy()[order]
    for w0 in range(0, g.num_nodes, 128):
        assert np.all(np.diff(deg[w0:w0 + 128]) <= 0) and np.all(order[w0:w0 + 128] // 128 == w0 // 128)
    assert bigcsr.window_degree_order(torch.zeros(1, dtype=torch.int64)).numel() == 0


class _Fp:
    def __init__(self, key):
        self._key = key

    def key(self):
        return self._key


def test_ordered_wanted_is_a_skew_test_behind_a_known_fingerprint(monkeypatch):
    monkeypatch.setattr(xcdplan, "MODE", "auto")
    monkeypatch.setattr(xcdplan, "ORDERED_MIN_EDGES", 1000)
    xcdplan._SKEW.clear()
    skewed = synth.hub_csr(3000, 2500, base_deg=4, seed=2)
    flat = synth.random_csr(3000, 2500, 8, seed=2, ragged=False)
    args = lambda g: (g.rowptr, g.num_nodes, g.nnz, 2500, 256)
    assert xcdplan.ordered_wanted(_Fp(("s",)), *args(skewed))
    assert not xcdplan.ordered_wanted(_Fp(("f",)), *args(flat))
    assert xcdplan._SKEW == {("s",): True, ("f",): False}  # one pass per structure, then cached
    assert not xcdplan.ordered_wanted(None, *args(skewed))            # no fingerprint
    assert not xcdplan.ordered_wanted(_Fp(None), *args(skewed))       # hash still in flight: the ordinary launch
    assert not xcdplan.ordered_wanted(_Fp(("s",)), skewed.rowptr, skewed.num_nodes, skewed.nnz, 1 << 24, 256)  # 24-bit row ids
    monkeypatch.setattr(xcdplan, "MODE", "off")
    assert not xcdplan.ordered_wanted(_Fp(("s",)), *args(skewed))
    xcdplan._SKEW.clear()
