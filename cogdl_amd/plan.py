"""Graph plans: the cached CSC view (stable transpose) of a CSR structure.

Why a cache keyed on *content*: CogDL's dispatcher hands the operators fresh int32 copies of
the graph structure on every call (`graph.row_indptr.int()`, cogdl/utils/spmm_utils.py:106),
so pointer identity never repeats across calls.  The reference re-runs cuSPARSE csr2csc in
every backward (cogdl/operators/spmm.py:63-68) -- or skips the transpose when the caller
says `sym`, which is wrong for sampled blocks that merely default to symmetric
(cogdl/data/data.py:146).  Here the transpose is computed once per distinct structure:

  forward : a 64-bit content fingerprint of (rowptr, colind) is computed on the stream and
            copied to pinned host memory asynchronously (no stall);
  backward: the fingerprint (long since landed) keys an LRU of (colptr, rowind, perm).

perm[j] is the CSR position of CSC entry j, so per-call edge values are moved with one
gather.  The cache holds 8 bytes per edge; eviction is by byte budget.
"""
import collections
import contextlib
import weakref
import os

import torch

from . import _lib


class CscPlan:
    __slots__ = ("colptr", "rowind", "perm", "m", "n_cols", "nnz", "_val_key", "_val_src", "_val_t",
                 "_max_col_degree", "ready", "_ws", "sightings")

    def __init__(self, colptr, rowind, perm, m, n_cols, nnz):
        self.colptr, self.rowind, self.perm = colptr, rowind, perm
        self.m, self.n_cols, self.nnz = m, n_cols, nnz
        self._val_key, self._val_src, self._val_t = None, None, None
        self._max_col_degree = None
        self.ready, self._ws = None, None  # (csr2csc(stream=...): the event to wait for, the workspace kept until then)
        self.sightings = 1  # cache look-ups that found this structure (PlanCache._lookup): > 1 = a structure that comes back

    def has_hub_columns(self):
        """Does A^T have rows beyond the long-row threshold?  (One reduction + sync, once per plan.)  When it does not,
        the backward SpMM runs without the chunk-parallel path: no scratch, no combine launch."""
        if self._max_col_degree is None:
            self._max_col_degree = int((self.colptr[1:] - self.colptr[:-1]).max()) if self.n_cols else 0
        # (the threshold is re-read every call: tests move it through the tuning knobs)
        return self._max_col_degree > _lib.hip().cogdl_hip_long_row_threshold(self.nnz)

    def transposed_values(self, w):
        """w[perm].  Constant edge weights (CogDL passes the same graph.raw_edge_weight tensor every call,
        cogdl/utils/spmm_utils.py:102) are gathered once: the memo is keyed on the address, version counter and
        layout of `w` AND holds a reference to it, so the keyed storage cannot be freed and handed to a different
        tensor while the key is live (a fresh tensor at a recycled address would otherwise alias the key: every new
        tensor starts at version 0).  Weights that take part in autograd (learned / attention weights: a new
        tensor every step) are gathered per call and never memoised."""
        src = w.detach()
        if w.requires_grad:
            return gather_rows(self.perm, src)
        key = tensor_key(w)
        if key != self._val_key or self._val_src is None:
            self._val_t = gather_rows(self.perm, src)
            self._val_key, self._val_src = key, src  # src shares w's storage: pins the address
        return self._val_t

    def nbytes(self):
        return 4 * (self.colptr.numel() + self.rowind.numel() + self.perm.numel())


def tensor_key(t):
    """Identity of a tensor's contents as far as can be told without reading them: address, version counter, dtype
    and layout.  Only sound while the caller also keeps a reference to `t` (or a view of its storage)."""
    return (t.data_ptr(), t._version, t.dtype, tuple(t.shape), tuple(t.stride()), t.device.index)


def csr2csc(rowptr, colind, n_cols=None, padded=False, stream=None):
    """Stable transpose of the structure -> CscPlan (device tensors, int32).  padded: `colind` is a fixed-capacity
    buffer of which only the first rowptr[-1] entries (read on the device) are edges (cogdl_hip_csr2csc_padded).
    stream: a torch.cuda.Stream to run the transpose on, forked from the current stream here (it waits for what the
    current stream has queued so far); the plan then carries `ready`, the event to wait for before reading it, and
    keeps its workspace until it dies (the buffers are allocated on the CURRENT stream: the caller's ordinary stream
    owns them, the side stream only computes)."""
    dev = _lib.require_cuda(rowptr, colind)
    m = rowptr.numel() - 1
    nnz = colind.numel()
    n_cols = m if n_cols is None else int(n_cols)
    if m == 0:  # no rows: no edges whatever colind holds (a block whose target slots are all unused)
        return CscPlan(torch.zeros(n_cols + 1, dtype=torch.int32, device=dev), torch.zeros(nnz, dtype=torch.int32, device=dev),
                       torch.arange(nnz, dtype=torch.int32, device=dev), m, n_cols, nnz)
    colptr = torch.empty(n_cols + 1, dtype=torch.int32, device=dev)
    rowind = torch.empty(nnz, dtype=torch.int32, device=dev)
    perm = torch.empty(nnz, dtype=torch.int32, device=dev)
    lib = _lib.hip()
    query, fn = ((lib.cogdl_hip_csr2csc_padded_workspace_bytes, lib.cogdl_hip_csr2csc_padded) if padded
                 else (lib.cogdl_hip_csr2csc_workspace_bytes, lib.cogdl_hip_csr2csc))
    ws_bytes = query(m, n_cols, nnz)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream(dev))
        # the buffers belong to the CURRENT stream's pool but are written on `stream`: tell the caching allocator, so that
        # a plan dropped without anybody waiting for `ready` (a forward whose backward never runs) cannot have its memory
        # handed out again while the side-stream transpose is still writing it
        for t in (colptr, rowind, perm, ws):
            t.record_stream(stream)
    with _lib.on_device(dev):
        rc = fn(_lib.ptr(rowptr), _lib.ptr(colind), m, n_cols, nnz, _lib.ptr(colptr), _lib.ptr(rowind), _lib.ptr(perm),
                _lib.ptr(ws), ws_bytes, _lib.stream_of(rowptr) if stream is None else stream.cuda_stream)
    _lib.check(rc, "csr2csc")
    plan = CscPlan(colptr, rowind, perm, m, n_cols, nnz)
    if stream is not None:
        plan.ready = torch.cuda.Event()
        plan.ready.record(stream)
        plan._ws = ws  # (freed with the plan, i.e. after whoever waited for `ready` is done with it)
    return plan


def gather_rows(perm, src):
    """out[i] = src[perm[i]] along dim 0 (src: [E] or [E,H], 2- or 4-byte elements)."""
    dev = _lib.require_cuda(perm, src)
    src = src.contiguous()
    out = torch.empty_like(src)
    n = perm.numel()
    h = src.numel() // max(n, 1) if n else 0
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_gather_rows(_lib.ptr(perm), _lib.ptr(src), _lib.ptr(out), n, h,
                                              src.element_size(), _lib.stream_of(src))
    _lib.check(rc, "gather_rows")
    return out


FINGERPRINT_PARTS = 256  # COGDL_HIP_FINGERPRINT_PARTS


_PINNED_POOL = {}  # device index -> recycled ([FINGERPRINT_PARTS] int64 pinned buffer, event) pairs
_PENDING = []      # (device index, buffer, event) whose hash kernel had not finished when their Fingerprint died
_POOL_MAX = 64


def _reap_pending():
    """Move parked buffers whose kernel has completed back to the pool.  A buffer is never released to torch's
    pinned allocator while a kernel may still write it: the kernel holds a raw pointer the allocator knows nothing
    about, so a freed block could be re-issued (a DataLoader pin thread, a non_blocking copy) and scribbled on."""
    if not _PENDING:
        return
    still = []
    for dev_index, host, event in _PENDING:
        if event.query():
            pool = _PINNED_POOL.setdefault(dev_index, [])
            if len(pool) < _POOL_MAX:
                pool.append((host, event))
        else:
            still.append((dev_index, host, event))
    _PENDING[:] = still


class PlanTape:
    """hipGraph capture support.  While a stream is being captured nothing may be read back on the host, so the
    structure hash cannot key the plan lookup.  cogdl_amd.graphs.capture() therefore runs the step eagerly first in
    RECORD mode (every plan lookup of the step is appended to the tape, in call order) and then captures it in REPLAY
    mode (lookups are served from the tape in the same order, fingerprints are not computed): the captured step must
    issue the same operator sequence as the recorded one -- which graph capture requires anyway."""

    def __init__(self):
        self.plans, self.pos, self.mode = [], 0, "record"
        # the launches' own decisions (which of them take an XCD-partitioned plan, cogdl_amd/xcdplan.py, and which plan), in call
        # order: (kind, value) pairs written by taped_choice()
        self.choices, self.cpos = [], 0


_TAPE = None


def set_tape(tape):
    global _TAPE
    _TAPE = tape


def recording():
    """Is this the eager run whose lookups cogdl_amd.graphs.capture() records?  (It may wait for a structure hash: the run is
    not captured -- so every launch of a captured step can be given the plan a call with a known fingerprint would take.)"""
    return _TAPE is not None and _TAPE.mode == "record"


def replaying():
    return _TAPE is not None and _TAPE.mode == "replay"


def taped_choice(kind, decide):
    """A decision of an operator call that needs the structure's key (does the launch take an XCD-partitioned plan, and which):
    without a tape decide() is called; while recording it is called and its value appended to the tape; while a capture replays
    the tape the recorded value comes back (nothing may be hashed or read back then) -- `kind` names the call site, a captured
    step that asks in another order than the recorded one is refused."""
    tape = _TAPE
    if tape is None:
        return decide()
    if tape.mode == "replay":
        if tape.cpos >= len(tape.choices):
            raise _lib.BackendError("hipGraph capture: the captured step takes more launch decisions than the recorded one")
        k, value = tape.choices[tape.cpos]
        tape.cpos += 1
        if k != kind:
            raise _lib.BackendError("hipGraph capture: the captured step's operator sequence differs from the recorded one "
                                    "(decision %d: %s vs %s)" % (tape.cpos - 1, kind, k))
        return value
    value = decide()
    tape.choices.append((kind, value))
    return value


_TRANSIENT = 0


_EARLY_STREAM = None


def early_transpose_stream():
    """The side stream of the enclosing transient_structures(side_stream=...) context, or None."""
    return _EARLY_STREAM if _TRANSIENT > 0 else None


@contextlib.contextmanager
def transient_structures(side_stream=None):
    """Inside this context the SpMM operators treat every CSR structure they are called with as one they will never
    see again -- the sampled blocks of mini-batch training: nothing is hashed, cached or read back; a backward pass
    transposes its structure on the spot (cogdl_hip_csr2csc_padded, so `colind` may be a fixed-capacity buffer with
    unused slots behind rowptr[-1]).  That makes a whole sampled step free of host synchronisation (capturable in a
    hipGraph) and keeps a million one-off transposes out of the plan cache.  The mode is latched per call in forward:
    the backward of a call made inside the context is transient wherever it runs.
    side_stream: a torch.cuda.Stream on which the transposes are computed EARLY -- launched by the forward call, beside
    whatever follows it on the ordinary stream, instead of by the backward call in front of its SpMM (a sampled block's
    transpose is one small workgroup's work: 25-30 us on the critical path of a 0.35 ms step, nothing beside it).  The
    caller must let the ordinary stream wait for `side_stream` before it ends the step (inside a hipGraph capture an
    unjoined stream is an error) -- a backward pass does that, a forward-only call does not."""
    global _TRANSIENT, _EARLY_STREAM
    prev = _EARLY_STREAM
    if side_stream is not None:
        _EARLY_STREAM = side_stream
    _TRANSIENT += 1
    try:
        yield
    finally:
        _TRANSIENT -= 1
        _EARLY_STREAM = prev


def transient():
    return _TRANSIENT > 0


class Fingerprint:
    """A structure hash in flight: the kernel writes its per-workgroup partials straight into pinned host memory
    (device-visible on ROCm: no memset, no device-to-host copy kernel); an event guards them.  Buffers and events are
    recycled through a small per-device pool when the Fingerprint dies (a pinned allocation per forward call is
    ~10 us of host time, and a training epoch is launch-bound)."""
    __slots__ = ("host", "event", "meta", "_key")

    def __init__(self, rowptr, colind, n_cols):
        dev = rowptr.device
        m, nnz = rowptr.numel() - 1, colind.numel()
        self.meta = (dev.index, m, nnz, int(n_cols))
        self._key = None
        self.host = self.event = None
        if _TAPE is not None and _TAPE.mode == "replay":
            return  # capturing: the plan comes from the tape, nothing is hashed
        _reap_pending()
        pool = _PINNED_POOL.get(dev.index)
        if pool:
            self.host, self.event = pool.pop()
        else:
            self.host = torch.empty(FINGERPRINT_PARTS, dtype=torch.int64, pin_memory=True)
            with _lib.on_device(dev):
                self.event = torch.cuda.Event()
        stream = _lib.current_stream_obj(dev)
        with _lib.on_device(dev):
            rc = _lib.hip().cogdl_hip_csr_fingerprint(_lib.ptr(rowptr), _lib.ptr(colind), m, nnz,
                                                      self.host.data_ptr(), stream.cuda_stream)
            _lib.check(rc, "csr_fingerprint")
            self.event.record(stream)

    def key(self):
        if self._key is None:
            self.event.synchronize()
            self._key = self.meta + (int(self.host.sum()),)  # int64 sum wraps: the sum modulo 2^64
        return self._key

    def __del__(self):
        # Back to the pool once the kernel that writes the buffer is known to be done; otherwise the pair is parked
        # on _PENDING (keeping the pinned block allocated) until a later call finds its event complete.
        try:
            if self.host is None:
                return
            dev_index = self.meta[0]
            if self._key is not None or self.event.query():
                pool = _PINNED_POOL.setdefault(dev_index, [])
                if len(pool) < _POOL_MAX:
                    pool.append((self.host, self.event))
            else:
                _PENDING.append((dev_index, self.host, self.event))
        except Exception:  # interpreter shutdown / half-constructed object
            pass


# Identity memo (round 6): a caller that hands the operators THE SAME index tensor objects call after call (this library's own
# API users, tools/gat_bench.py -- not the reference dispatcher, whose `.int()` copies are new objects every time) need not
# have them hashed every time: 102 us per call at 1.1e8 edges, four calls per GAT training step.  Keyed on the two tensor
# OBJECTS (weak references: nothing is kept alive, and a dead object's id cannot vouch for its successor) and vouched for by
# their version counters, data pointers and sizes -- an in-place edit, a storage swap or a new tensor all miss.
_IDENT = collections.OrderedDict()
_IDENT_MAX = 16


def _ident_state(rowptr, colind):
    return (rowptr._version, colind._version, rowptr.data_ptr(), colind.data_ptr(), rowptr.numel(), colind.numel())


def known_fingerprint(rowptr, colind, n_cols):
    """The memoised Fingerprint of exactly these tensor objects (structure memo of the Graph, or the identity memo), or None --
    never hashes."""
    memo = getattr(rowptr, "_cogdl_amd_struct", None)
    if memo is not None:
        fp = memo.fingerprint(rowptr, colind, int(n_cols)) if (memo._fp.get(int(n_cols)) is not None) else None
        if fp is not None:
            return fp
    if _TAPE is not None:
        return None
    hit = _IDENT.get((id(rowptr), id(colind), int(n_cols)))
    if hit is not None and hit[0]() is rowptr and hit[1]() is colind and hit[2] == _ident_state(rowptr, colind):
        return hit[3]
    return None


def fingerprint_of(rowptr, colind, n_cols):
    """The structure's Fingerprint for an operator call: the memoised one when the tensors come from a Graph under
    `install(structure_memo=True)` (cogdl_amd/structure_memo.py: no hash kernel, no event, no pinned buffer per call) or are
    the very objects of an earlier call (identity memo), otherwise a fresh hash enqueued now."""
    memo = getattr(rowptr, "_cogdl_amd_struct", None)
    if memo is not None:
        fp = memo.fingerprint(rowptr, colind, int(n_cols))
        if fp is not None:
            return fp
    if _TAPE is not None:
        return Fingerprint(rowptr, colind, n_cols)
    key = (id(rowptr), id(colind), int(n_cols))
    hit = _IDENT.get(key)
    if hit is not None:
        if hit[0]() is rowptr and hit[1]() is colind and hit[2] == _ident_state(rowptr, colind):
            _IDENT.move_to_end(key)
            return hit[3]
        del _IDENT[key]
    fp = Fingerprint(rowptr, colind, n_cols)
    if fp.event is not None:
        try:
            _IDENT[key] = (weakref.ref(rowptr), weakref.ref(colind), _ident_state(rowptr, colind), fp)
        except TypeError:  # (an object that cannot be weakly referenced: no memo)
            return fp
        while len(_IDENT) > _IDENT_MAX:
            _IDENT.popitem(last=False)
    return fp


def clear_identity_memo():
    _IDENT.clear()


# Debug mode (COGDL_AMD_VERIFY_PLANS=1, or plan.VERIFY_HITS = True): a cache hit is trusted on sizes + a 64-bit content hash;
# a collision would hand the backward a WRONG transpose silently.  With the mode on every hit is checked against the
# structure of the call (O(nnz) torch work + one synchronisation per hit: for debugging, not for training).
VERIFY_HITS = os.environ.get("COGDL_AMD_VERIFY_PLANS", "0") == "1"


def verify_plan(plan, rowptr, colind):
    """Is `plan` the stable transpose of (rowptr, colind)?  Necessary and sufficient: the sizes agree, the columns of the
    CSR positions perm lists are exactly the plan's column runs (colind[perm] == repeat(arange(n_cols), diff(colptr))),
    perm ascends inside every column (stability; then it is a permutation), and rowind names the row that owns each
    position."""
    m, nnz = rowptr.numel() - 1, colind.numel()
    ok = plan.m == m and plan.nnz == nnz and plan.colptr.numel() == plan.n_cols + 1
    if ok and nnz:
        perm = plan.perm.long()
        counts = (plan.colptr[1:] - plan.colptr[:-1]).long()
        cols = torch.repeat_interleave(torch.arange(plan.n_cols, device=colind.device), counts)
        ok = cols.numel() == nnz and bool((colind.long()[perm] == cols).all())
        if ok and nnz > 1:
            same_col = cols[1:] == cols[:-1]
            ok = bool((perm[1:] > perm[:-1])[same_col].all())
        if ok:
            rows = torch.searchsorted(rowptr.long(), perm, right=True) - 1
            ok = bool((rows == plan.rowind.long()).all())
    if not ok:
        raise _lib.BackendError("plan cache: the cached transpose does not belong to this structure (m=%d, nnz=%d): a "
                                "fingerprint collision or a structure edited in place behind the cache" % (m, nnz))


class PlanCache:
    def __init__(self, budget_bytes=None):
        if budget_bytes is None:
            budget_bytes = int(os.environ.get("COGDL_AMD_PLAN_CACHE_MB", "4096")) << 20
        self.budget = budget_bytes
        self.bytes = 0
        self.lru = collections.OrderedDict()
        self.hits = 0
        self.misses = 0

    def get(self, fingerprint, rowptr, colind, n_cols):
        tape = _TAPE
        if tape is not None and tape.mode == "replay":
            if tape.pos >= len(tape.plans):
                raise _lib.BackendError("hipGraph capture: the captured step looks up more graph plans than the recorded one")
            plan = tape.plans[tape.pos]
            tape.pos += 1
            if (plan.m, plan.nnz, plan.n_cols) != fingerprint.meta[1:]:
                raise _lib.BackendError("hipGraph capture: the captured step's operator sequence differs from the "
                                        "recorded one (plan %d: %s vs %s)" % (tape.pos - 1, (plan.m, plan.nnz, plan.n_cols),
                                                                               fingerprint.meta[1:]))
            return plan
        plan = self._lookup(fingerprint, rowptr, colind, n_cols)
        if tape is not None:
            tape.plans.append(plan)
        return plan

    def _lookup(self, fingerprint, rowptr, colind, n_cols):
        key = fingerprint.key()
        plan = self.lru.get(key)
        if plan is not None:
            self.lru.move_to_end(key)
            self.hits += 1
            plan.sightings = getattr(plan, "sightings", 1) + 1
            if VERIFY_HITS:
                verify_plan(plan, rowptr, colind)
            return plan
        self.misses += 1
        plan = csr2csc(rowptr, colind, n_cols)
        if plan.nbytes() <= self.budget:
            self.lru[key] = plan
            self.bytes += plan.nbytes()
            while self.bytes > self.budget:
                _, old = self.lru.popitem(last=False)
                self.bytes -= old.nbytes()
        return plan

    def clear(self):
        self.lru.clear()
        self.bytes = 0


PLANS = PlanCache()
