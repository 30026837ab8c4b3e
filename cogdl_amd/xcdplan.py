"""XCD-partitioned column plans (round 6): the host side of `cogdl_hip_vrows` (include/cogdl_hip.h, csrc/rowreduce.h).

Every XCD of the MI355X has a private 4 MiB L2.  The gathered table of BASELINE configs[2] (233 k rows of 128-256 bytes)
fits none of them but fits the eight together -- if XCD x only ever gathers the columns it OWNS.  A plan gives every column
an owner (a multiplicative hash of its id: R-MAT / citation ids correlate with degree, `col % 8` leaves the XCDs 1.5x out of
balance), cuts every row of more than `split` edges into the sub-rows of its edges by owner (CSR order inside a sub-row,
pieces of at most `piece` edges) and lays these virtual rows out in units of 64 slots dealt round-robin to the XCDs -- the
order in which the hardware hands workgroups to them.  Shorter rows stay whole.  The reference has nothing like it (its
GE-SpMM kernels, cogdl/operators/spmm/spmm_kernel.cu:192-512, know one L2).

The plan is a property of the STRUCTURE: built once (a handful of sorts over the edge list, torch on the device: ~0.2 s at
1.1e8 edges), cached under the structure's content fingerprint next to the transpose (cogdl_amd/plan.py).  When it is used:
`wanted()` -- hub-heavy launches over cache-sized tables -- for the fused GAT operator and 16-bit csr_spmm; fp32 csr_spmm keeps
its bit-exact rows unless asked (`MODE = "force"` / COGDL_AMD_XCD=force), because a row cut into sub-rows is re-associated.
"""
import collections
import ctypes
import os

import torch

from . import _lib
from . import plan as _plan

UNIT = 64          # slots per unit (csrc/rowreduce.h: kVUnit)
XCDS = 8
SPLIT = 256        # rows of more than this many edges are cut by owner XCD (Reddit-shaped graph, fused GAT bf16 H8xF8 with dropout,
                   # fwd+bwd by split / piece: 64/256 6.21 ms, 128/256 6.02, 256/256 5.82, 256/512 6.15, 512/512 6.11: profiles/r06_gat_split.txt)
PIECE = 512        # edges per virtual row at most (with the virtual rows in order of length, same probe: 256/256 5.20 ms, 256/512 5.02,
                   # 512/512 5.24, 128/256 5.19, 256/128 5.47: profiles/r06_gat_split2.txt)
BIG_PARTS = 32     # (csrc/rowreduce.h: kVBigParts)
SORT_BY_LENGTH = os.environ.get("COGDL_AMD_XCD_SORT", "1") != "0"  # (A/B switch: the slot order inside an XCD's stream)
MODE = os.environ.get("COGDL_AMD_XCD", "auto")  # "auto" | "off" | "force" (every structure, every operator: tests)


class _VRowsStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("vrowptr", "vcol", "vdesc", "eid", "mrow", "mptr", "big")] + \
               [(n, ctypes.c_int64) for n in ("n_slots", "n_multi", "n_parts", "n_big", "nnz")]


class XcdPlan:
    __slots__ = ("vrowptr", "vcol", "vdesc", "eid", "mrow", "mptr", "big", "n_slots", "n_multi", "n_parts", "n_big", "nnz", "m",
                 "_struct", "_val_key", "_val_src", "_val_p")

    def __init__(self, vrowptr, vcol, vdesc, eid, mrow, mptr, n_parts, m):
        self.vrowptr, self.vcol, self.vdesc, self.eid, self.mrow, self.mptr = vrowptr, vcol, vdesc, eid, mrow, mptr
        self.n_slots, self.n_multi, self.n_parts = vdesc.shape[0], mrow.numel(), int(n_parts)
        # rows of more than BIG_PARTS parts (csrc/rowreduce.h: kVBigParts): merged by a whole workgroup each
        self.big = torch.nonzero((mptr[1:] - mptr[:-1]) > BIG_PARTS).flatten().int()
        self.n_big = self.big.numel()
        self.nnz, self.m = vcol.numel(), m
        self._val_key = self._val_src = self._val_p = None
        s = _VRowsStruct()
        for name in ("vrowptr", "vcol", "vdesc", "eid", "mrow", "mptr", "big"):
            t = getattr(self, name)
            setattr(s, name, t.data_ptr() if t is not None and t.numel() else None)
        s.n_slots, s.n_multi, s.n_parts, s.n_big, s.nnz = self.n_slots, self.n_multi, self.n_parts, self.n_big, self.nnz
        self._struct = s

    def ref(self):
        """const cogdl_hip_vrows * (the plan keeps the tensors -- and the struct -- alive)."""
        return ctypes.byref(self._struct)

    def nbytes(self):
        return 4 * sum(t.numel() for t in (self.vrowptr, self.vcol, self.vdesc, self.eid, self.mrow, self.mptr, self.big))

    def permuted_values(self, w):
        """w in plan order (w[eid]).  Constant edge weights are gathered once (memo keyed like CscPlan.transposed_values)."""
        src = w.detach()
        if w.requires_grad:
            return _plan.gather_rows(self.eid, src)
        key = _plan.tensor_key(w)
        if key != self._val_key or self._val_src is None:
            self._val_p = _plan.gather_rows(self.eid, src)
            self._val_key, self._val_src = key, src
        return self._val_p


def owner_of(col):
    """Owner XCD of a column id (int64 tensor): the top three bits of a 32-bit multiplicative hash."""
    return ((col * 2654435761) & 0xFFFFFFFF) >> 29


def build(rowptr, colind, eid_base=None, split=None, piece=None):
    """-> XcdPlan of the CSR structure (rowptr [m + 1], colind [nnz]; int32 device tensors).  eid_base: what position j of
    THIS structure means to a per-edge operand (a CSC view passes the transpose's perm); None: the position itself."""
    split = SPLIT if split is None else int(split)
    piece = PIECE if piece is None else int(piece)
    dev = rowptr.device
    m, nnz = rowptr.numel() - 1, colind.numel()
    rp = rowptr.long()
    deg = rp[1:] - rp[:-1]
    rows = torch.arange(m, device=dev)
    row = torch.repeat_interleave(rows, deg)
    col = colind.long()
    is_long = deg > split
    xcd_e = torch.where(is_long[row], owner_of(col), row % XCDS)
    key = xcd_e * max(m, 1) + row
    del xcd_e
    order = torch.argsort(key, stable=True)  # (XCD, row), CSR order inside
    gkey, cnt = torch.unique_consecutive(key[order], return_counts=True)
    del key
    empty = torch.nonzero(deg == 0).flatten()
    if empty.numel():  # rows without edges still own a (whole, empty) slot: their output has to be written
        gkey = torch.cat([gkey, (empty % XCDS) * max(m, 1) + empty])
        cnt = torch.cat([cnt, torch.zeros_like(empty)])
        gkey, o = torch.sort(gkey)  # (edge offsets below come from a cumulative sum: empty groups add nothing)
        cnt = cnt[o]
    # pieces of at most `piece` edges -- for the sub-rows of LONG rows only: a row of at most `split` edges stays ONE virtual
    # row whatever its length (sequential CSR order: fp32 csr_spmm's bit-exact rows, split = its exact-row bound > piece)
    n_p = torch.where(is_long[gkey % max(m, 1)], torch.clamp((cnt + piece - 1) // piece, min=1), torch.ones_like(cnt))
    n_groups = gkey.numel()
    vg = torch.repeat_interleave(torch.arange(n_groups, device=dev), n_p)
    n_v = vg.numel()
    first_v = torch.cumsum(n_p, 0) - n_p
    idx = torch.arange(n_v, device=dev) - first_v[vg]
    vlen = torch.where(n_p[vg] == 1, cnt[vg], torch.where(idx < n_p[vg] - 1, torch.full_like(idx, piece), cnt[vg] - piece * (n_p[vg] - 1)))
    vx, vr = (gkey // max(m, 1))[vg], (gkey % max(m, 1))[vg]
    # records: the virtual rows of a row with several parts, in (row, XCD, piece) order
    parts_of_row = torch.bincount(vr, minlength=m)
    multi = parts_of_row > 1
    ord2 = torch.argsort(vr, stable=True)  # (the list is (XCD, row, piece)-ordered: stable by row keeps (XCD, piece))
    is_multi_sorted = multi[vr[ord2]]
    rec = torch.full((n_v,), -1, dtype=torch.long, device=dev)
    rec_sorted = torch.cumsum(is_multi_sorted.long(), 0) - 1
    rec[ord2[is_multi_sorted]] = rec_sorted[is_multi_sorted]
    n_parts = int(is_multi_sorted.sum())
    mrow = torch.nonzero(multi).flatten()
    mptr = torch.zeros(mrow.numel() + 1, dtype=torch.long, device=dev)
    torch.cumsum(parts_of_row[mrow], 0, out=mptr[1:])
    # slots: XCD x's k-th virtual row -> unit (k // 64) * 8 + x
    counts = torch.bincount(vx, minlength=XCDS)
    per = int((int(counts.max()) + UNIT - 1) // UNIT * UNIT) if n_v else 0
    n_slots = XCDS * per
    start = torch.cumsum(counts, 0) - counts
    if SORT_BY_LENGTH and n_v:
        # An XCD's virtual rows in order of DECREASING length: the lane groups of a wave (and the waves of a workgroup) then
        # walk rows of the same length -- a wave lasts as long as its longest row, and in (row, piece) order neighbours
        # differ by an order of magnitude on a skewed graph (Reddit-shaped graph, fused GAT forward: 5.2 vector instructions per
        # edge executed where the loop body has 3.5: a third of all lane-slots masked off, profiles/r06_mem_pipeline.txt).
        # Longest first is also the better schedule for the launch's tail.  Which slot walks a virtual row changes nothing
        # in what it computes, and the records are merged by (row, XCD, piece) as before: results are bit-identical.
        by_len = torch.argsort(vx * (int(vlen.max()) + 1) + (int(vlen.max()) - vlen), stable=True)
        k = torch.empty(n_v, dtype=torch.long, device=dev)
        k[by_len] = torch.arange(n_v, device=dev) - start[vx[by_len]]
        del by_len
    else:
        k = torch.arange(n_v, device=dev) - start[vx]
    slot = ((k // UNIT) * XCDS + vx) * UNIT + k % UNIT
    vdesc = torch.full((n_slots, 2), -1, dtype=torch.int32, device=dev)
    vdesc[slot, 0] = vr.int()
    vdesc[slot, 1] = rec.int()
    lens = torch.zeros(n_slots, dtype=torch.long, device=dev)
    lens[slot] = vlen
    vrowptr = torch.zeros(n_slots + 1, dtype=torch.long, device=dev)
    torch.cumsum(lens, 0, out=vrowptr[1:])
    v_of_edge = torch.repeat_interleave(torch.arange(n_v, device=dev), vlen)
    eorder = torch.argsort(slot[v_of_edge], stable=True)
    final = order[eorder]
    vcol = colind[final].contiguous()
    eid = (final if eid_base is None else eid_base.long()[final]).int().contiguous()
    if nnz >= 2 ** 31 or n_slots >= 2 ** 31:
        raise _lib.BackendError("XCD plan: the structure exceeds int32 positions")
    return XcdPlan(vrowptr.int(), vcol, vdesc.contiguous(), eid, mrow.int(), mptr.int(), n_parts, m)


def wanted(m, nnz, n_src, row_bytes):
    """Should a launch over this structure, gathering rows of `row_bytes` from an [n_src, .] table, take a plan?  Hub-heavy
    launches (>= 64 edges per row on average, >= 8 M edges) over tables between one L2 and a few times the eight of them, rows
    of at least 128 bytes.  Measured on the Reddit-shaped graph (493 edges per row, profiles/r06_xcd_quick.txt), fused GAT
    forward + backward: H = 8 x F = 8 bf16 (128-byte rows, 30 MB) 7.31 -> 5.66 ms, fp32 (60 MB) 11.83 -> 7.73 ms; H = 1 x
    F = 48 bf16 (96-byte rows, 22 MB: one head per lane group, bound by instruction issue, not by the gathers) 6.07 -> 5.97 ms
    without and 6.56 -> 6.93 ms with dropout -- not taken (the operator pads such rows to 128 bytes: fused_gat._padded_width).
    Which operators ask: the fused GAT operator, and csr_spmm in fp32 (cut at its exact-row bound: operators/spmm.py)."""
    if MODE == "off" or _plan.transient():
        return False
    if n_src >= (1 << 24) or n_src * row_bytes >= (1 << 32):  # (the plan kernels address the table with 24 x 24 -> 32-bit offsets)
        return False
    if MODE == "force":
        return nnz > 0
    table = n_src * row_bytes
    return nnz >= (1 << 23) and nnz >= 64 * m and row_bytes >= 128 and (2 << 20) <= table <= (512 << 20)


ORDERED_MIN_EDGES = 1 << 18   # below: the launch is a few microseconds, nothing to balance
ORDERED_MAX_FILL = 0.5        # "skewed": eight consecutive rows fill less than this share of the lane-slots of their longest
_SKEW = {}


def ordered_wanted(fp, rowptr, m, nnz, n_src, row_bytes):
    """The second way into a plan (round 6): a SKEWED structure of any size whose fingerprint is ALREADY KNOWN on the host --
    memoised with the Graph (install(structure_memo=True)) or by tensor identity (plan.fingerprint_of), or a backward pass, which
    has waited for the hash anyway (the operators take it from the structure's second sighting on: plan.CscPlan.sightings).  What a
    plan buys there is its slot order: an XCD's virtual rows by decreasing length, so that the lane groups of a wave walk rows
    of one length, and hub rows as pieces merged by rowreduce_vcombine_kernel.  Measured on the arxiv-sized R-MAT graph (max
    degree 10^4; tools/exp/small_plan_ab.py, profiles/r06_small_plan_ab.txt), ordinary launch -> plan cut at the exact-row
    bound: fp32 F = 128 183 -> 144 us, F = 64 100 -> 63 us, F = 40 103 -> 60 us; bf16 134 -> 68, 96 -> 44, 95 -> 47 us; on
    the uniform graph of the same size +-3 % -- hence the skew test (one pass over the degrees per structure, cached): the
    share of lane-slots that eight consecutive rows fill, 0.73 on the uniform graph, 0.32 on the R-MAT one.
    A forward call whose hash is still in flight keeps the ordinary launch: waiting would drain the stream.
    Under cogdl_amd.graphs.capture the operators ask through plan.taped_choice: the recorded eager run waits for the key (it is
    not captured), the capture replays its decisions -- every launch of a captured step runs over the plan a known structure
    takes, forward calls included."""
    if MODE == "off" or _plan.transient() or fp is None or getattr(fp, "_key", None) is None:
        return False
    if nnz < ORDERED_MIN_EDGES or n_src >= (1 << 24) or n_src * row_bytes >= (1 << 32) or row_bytes >= (1 << 22):
        return False
    key = fp.key()
    hit = _SKEW.get(key)
    if hit is None:
        if len(_SKEW) > 256:
            _SKEW.clear()
        hit = _SKEW[key] = skewed(rowptr)
    return hit


def skewed(rowptr):
    """The skew test of ordered_wanted (one pass over the degrees + one host read): do eight consecutive rows fill less than
    ORDERED_MAX_FILL of the lane-slots of their longest?"""
    deg = (rowptr[1:] - rowptr[:-1]).float()
    pad = (-deg.numel()) % 8
    if pad:
        deg = torch.cat([deg, deg.new_zeros(pad)])
    fill = float(deg.sum() / (deg.view(-1, 8).max(1).values.sum() * 8).clamp(min=1.0))
    return fill < ORDERED_MAX_FILL


class _Cache:
    def __init__(self):
        self.budget = int(os.environ.get("COGDL_AMD_XCD_CACHE_MB", "8192")) << 20
        self.bytes = 0
        self.lru = collections.OrderedDict()

    def get(self, key, make):
        p = self.lru.get(key)
        if p is not None:
            self.lru.move_to_end(key)
            return p
        p = make()
        self.lru[key] = p
        self.bytes += p.nbytes()
        while self.bytes > self.budget and len(self.lru) > 1:
            _, old = self.lru.popitem(last=False)
            self.bytes -= old.nbytes()
        return p

    def clear(self):
        self.lru.clear()
        self.bytes = 0


XPLANS = _Cache()


def csr_plan(fp, rowptr, colind, split=None):
    """The plan of the structure itself, under its fingerprint (waits for the hash: one host synchronisation per call --
    `wanted()` only says yes to launches of hundreds of microseconds; with install(structure_memo=True) the key is memoised).
    split: rows of more than this many edges are cut by owner (default SPLIT; fp32 csr_spmm passes its exact-row bound)."""
    split = SPLIT if split is None else int(split)
    return XPLANS.get(("csr", split, PIECE) + fp.key(), lambda: build(rowptr, colind, split=split))


def csc_plan(fp, csc, split=None):
    """The plan of the transpose (a CscPlan): per-edge operands are indexed through its perm."""
    split = SPLIT if split is None else int(split)
    return XPLANS.get(("csc", split, PIECE) + fp.key(), lambda: build(csc.colptr, csc.rowind, eid_base=csc.perm, split=split))
