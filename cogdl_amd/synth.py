"""Seeded synthetic graphs with the shapes of BASELINE.json's configs (no dataset is available
offline).  Preprocessing mirrors what CogDL does before SpMM sees a graph:

  symmetrise + coalesce           cogdl/datasets/ogb.py:50-55, planetoid_data.py:44-51
  add_remaining_self_loops        cogdl/data/data.py:175-191 (loops appended after the edges)
  stable COO->CSR                 cogdl/utils/graph_utils.py:133-142 -> sample.cpp:234-270
  sym_norm: w = d^-1/2[row] * d^-1/2[col], d = row sums of the unit weights   data.py:260-274

All generators are deterministic functions of `seed` (torch.Generator on CPU, so the same
graph is produced on every machine) and return CPU tensors; callers move them to the GPU.
"""
import math

import torch


class CSRGraph:
    """Plain container: int32 CSR (what reaches the operators) + fp32 weights."""

    def __init__(self, rowptr, colind, weight, num_nodes, n_cols=None):
        self.rowptr, self.colind, self.weight = rowptr, colind, weight
        self.num_nodes = num_nodes
        self.n_cols = num_nodes if n_cols is None else n_cols

    @property
    def nnz(self):
        return int(self.colind.numel())

    def to(self, device):
        return CSRGraph(self.rowptr.to(device), self.colind.to(device),
                        None if self.weight is None else self.weight.to(device), self.num_nodes, self.n_cols)

    def degrees(self):
        return (self.rowptr[1:] - self.rowptr[:-1]).long()


def coo_to_csr_stable(row, col, num_nodes):
    """Stable counting sort by row (edge order inside a row = COO order). -> rowptr(int64), perm."""
    perm = torch.sort(row, stable=True).indices
    counts = torch.bincount(row, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.long, device=row.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr, perm


def finalize(src, dst, num_nodes, symmetrise=True, self_loops=True, norm="sym"):
    """Directed pairs -> CogDL-preprocessed CSR graph (row = destination of aggregation)."""
    src, dst = src.long(), dst.long()
    if symmetrise:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    key = torch.unique(src * num_nodes + dst)  # coalesce: sorted by (row, col), duplicates removed
    row, col = key // num_nodes, key % num_nodes
    if self_loops:  # add_remaining_self_loops: keep existing loops, append the missing ones
        has_loop = torch.zeros(num_nodes, dtype=torch.bool, device=row.device)
        has_loop[row[row == col]] = True
        missing = torch.nonzero(~has_loop).flatten()
        row, col = torch.cat([row, missing]), torch.cat([col, missing])
    rowptr, perm = coo_to_csr_stable(row, col, num_nodes)
    row, col = row[perm], col[perm]
    w = torch.ones(row.numel(), dtype=torch.float32, device=row.device)
    if norm == "sym":
        deg = torch.zeros(num_nodes, dtype=torch.float32, device=row.device).scatter_add_(0, row, w)
        dinv = deg.pow(-0.5)
        dinv[torch.isinf(dinv)] = 0
        w = dinv[row] * w * dinv[col]
    elif norm == "row":
        deg = torch.zeros(num_nodes, dtype=torch.float32, device=row.device).scatter_add_(0, row, w)
        w = w / deg[row]
    elif norm is None:
        w = None
    return CSRGraph(rowptr.int(), col.int(), w, num_nodes)


def uniform_pairs(num_nodes, num_pairs, seed, device="cpu"):
    """device='cpu' (default) gives the same pairs on every machine; a GPU device is only for large benches."""
    g = torch.Generator(device=device).manual_seed(seed)
    src = torch.randint(0, num_nodes, (num_pairs,), generator=g, device=device)
    dst = torch.randint(0, num_nodes, (num_pairs,), generator=g, device=device)
    return src, dst


def rmat_pairs(num_nodes, num_pairs, seed, a=0.57, b=0.19, c=0.19, device="cpu"):
    """R-MAT (Chakrabarti et al.) edge list; ids outside [0, num_nodes) are folded back by modulo."""
    g = torch.Generator(device=device).manual_seed(seed)
    scale = max(1, math.ceil(math.log2(num_nodes)))
    src = torch.zeros(num_pairs, dtype=torch.long, device=device)
    dst = torch.zeros(num_pairs, dtype=torch.long, device=device)
    for _ in range(scale):
        r = torch.rand(num_pairs, generator=g, device=device)
        src_bit = (r >= a + b).long()
        dst_bit = ((r >= a) & (r < a + b) | (r >= a + b + c)).long()
        src = src * 2 + src_bit
        dst = dst * 2 + dst_bit
    return src % num_nodes, dst % num_nodes


# ----------------------------------------------------------------------------------------------
# Named workloads
def arxiv_like(seed=0, topology="uniform"):
    """ogbn-arxiv shape: N=169,343, 1,166,243 directed pairs -> symmetrise + loops + sym_norm
    (nnz ~ 2.50 M).  topology='uniform' is the worst-case-locality stand-in BASELINE.md measured;
    'rmat' is a power-law variant."""
    n, pairs = 169_343, 1_166_243
    src, dst = uniform_pairs(n, pairs, seed) if topology == "uniform" else rmat_pairs(n, pairs, seed)
    return finalize(src, dst, n)


REDDIT_NODES, REDDIT_UNDIRECTED = 232_965, 57_307_946


def reddit_like(seed=0, device="cpu", norm=None):
    """BASELINE.json configs[2]: Reddit's shape (SAINT format, cogdl/datasets/saint_data.py:15-64) -- N = 232,965,
    114,615,892 directed edges after symmetrisation, + one self loop per node = 114,848,857 nnz.  R-MAT pairs are
    drawn until there are enough distinct undirected pairs, then exactly 57,307,946 of them are kept (seeded
    selection), so the edge count is Reddit's regardless of how many duplicates R-MAT produced.  Power-law rows: the
    longest has ~10^5 edges, 82 % of the edges sit in rows of more than 1024.  On a GPU this takes a few seconds (on
    the CPU ~1 min); the graph depends on (seed, device type)."""
    n, target = REDDIT_NODES, REDDIT_UNDIRECTED
    keys = None
    draw = 0
    while keys is None or keys.numel() < target:
        src, dst = rmat_pairs(n, 100_000_000, seed * 7919 + draw, device=device)
        lo, hi = torch.minimum(src, dst), torch.maximum(src, dst)
        keep = lo != hi
        k = torch.unique(lo[keep] * n + hi[keep])
        del src, dst, lo, hi, keep
        keys = k if keys is None else torch.unique(torch.cat([keys, k]))
        draw += 1
    if keys.numel() > target:
        g = torch.Generator(device=device).manual_seed(seed + 1)
        keys = keys[torch.randperm(keys.numel(), generator=g, device=device)[:target]]
    return finalize(keys // n, keys % n, n, norm=norm)


def cora_like(seed=0):
    """Cora shape: N=2,708, 5,278 undirected pairs -> 10,556 directed + 2,708 loops."""
    n = 2_708
    src, dst = uniform_pairs(n, 5_400, seed)
    keep = src != dst
    return finalize(src[keep], dst[keep], n)


def scaled(num_nodes, avg_degree, seed=0, topology="uniform", **kw):
    pairs = int(num_nodes * avg_degree / 2)
    src, dst = uniform_pairs(num_nodes, pairs, seed) if topology == "uniform" else rmat_pairs(num_nodes, pairs, seed)
    return finalize(src, dst, num_nodes, **kw)


def random_csr(m, n_cols, nnz_per_row, seed=0, weighted=True, ragged=True):
    """Rectangular CSR with ragged rows (some empty, duplicates allowed): operator unit tests."""
    g = torch.Generator().manual_seed(seed)
    if ragged:
        deg = torch.randint(0, 2 * nnz_per_row + 1, (m,), generator=g)
        deg[torch.rand(m, generator=g) < 0.1] = 0
    else:
        deg = torch.full((m,), nnz_per_row, dtype=torch.long)
    rowptr = torch.zeros(m + 1, dtype=torch.long)
    torch.cumsum(deg, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colind = torch.randint(0, n_cols, (nnz,), generator=g)
    w = torch.randn(nnz, generator=g) if weighted else None
    return CSRGraph(rowptr.int(), colind.int(), w, m, n_cols)


def hub_csr(m, n_cols, base_deg=5, hubs=((3, 129), (4, 1000), (17, 5000), (18, 257), (40, 128)), seed=0, weighted=True):
    """Short ragged rows plus a few hub rows of given lengths (row id, degree): exercises the chunk-parallel
    long-row path (hubs longer than the threshold, next to each other, aligned or not with chunk borders)."""
    g = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, 2 * base_deg + 1, (m,), generator=g)
    for r, d in hubs:
        if r < m:
            deg[r] = d
    rowptr = torch.zeros(m + 1, dtype=torch.long)
    torch.cumsum(deg, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colind = torch.randint(0, n_cols, (nnz,), generator=g)
    w = torch.randn(nnz, generator=g) if weighted else None
    return CSRGraph(rowptr.int(), colind.int(), w, m, n_cols)


# ----------------------------------------------------------------------------------------------
# BASELINE.json configs[4] at FULL size on one GPU: ogbn-papers100M's shape (111,059,956 nodes, 1,615,685,872 directed
# citation pairs).  Built on the device in row buckets so that no single torch op sees 2^31 elements and the
# temporaries stay a few GB; returns a 64-bit CSR (int64 row pointers, int32 column ids) -- what cogdl_amd/bigcsr.py runs.
PAPERS_NODES, PAPERS_PAIRS = 111_059_956, 1_615_685_872


def rmat_pairs_i32(num_nodes, num_pairs, seed, device, chunk=1 << 27):
    """rmat_pairs in chunks, as int32 ids (num_nodes < 2^31): 8 bytes per pair instead of 16 + temporaries."""
    src = torch.empty(num_pairs, dtype=torch.int32, device=device)
    dst = torch.empty(num_pairs, dtype=torch.int32, device=device)
    for i, lo in enumerate(range(0, num_pairs, chunk)):
        n = min(chunk, num_pairs - lo)
        s, d = rmat_pairs(num_nodes, n, seed * 1_000_003 + i, device=device)
        src[lo:lo + n] = s
        dst[lo:lo + n] = d
        del s, d
    return src, dst


class BigCSRGraph:
    """64-bit CSR container: rowptr int64 [N+1], colind int32 [nnz], weight fp32 [nnz] (device tensors)."""

    def __init__(self, rowptr, colind, weight, num_nodes):
        self.rowptr, self.colind, self.weight, self.num_nodes = rowptr, colind, weight, num_nodes

    @property
    def nnz(self):
        return int(self.colind.numel())


def big_csr_from_pairs(src, dst, num_nodes, symmetrise, buckets=16):
    """int32 pairs on the device -> BigCSRGraph with row = aggregation target (dst), column ids ascending inside a row
    (the order CogDL's coalesce leaves, cogdl/datasets/ogb.py:50-55), multi-edges KEPT (so the edge count is the pair
    count: 1.6e9 directed, 3.2e9 symmetrised -- R-MAT draws duplicates, the dataset has none) and
    w = d^-1/2[row] * d^-1/2[col] with d = the row's edge count (sym_norm, cogdl/data/data.py:260-274)."""
    dev = src.device
    deg = torch.bincount(dst, minlength=num_nodes)
    if symmetrise:
        deg += torch.bincount(src, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    dinv = deg.to(torch.float32).pow_(-0.5)
    dinv[torch.isinf(dinv)] = 0
    del deg
    colind = torch.empty(nnz, dtype=torch.int32, device=dev)
    weight = torch.empty(nnz, dtype=torch.float32, device=dev)
    # buckets of about nnz / buckets edges each (cut by edge count, not by rows: R-MAT rows are skewed towards low ids)
    targets = torch.arange(1, buckets, device=dev, dtype=torch.int64) * (nnz // buckets)
    cuts = [0] + sorted(set(int(r) for r in (torch.searchsorted(rowptr, targets, right=True) - 1).tolist())) + [num_nodes]
    cuts = sorted(set(cuts))
    for r0, r1 in zip(cuts, cuts[1:]):
        sel = (dst >= r0) & (dst < r1)
        key = dst[sel].long() * num_nodes + src[sel].long()
        if symmetrise:
            sel = (src >= r0) & (src < r1)
            key = torch.cat([key, src[sel].long() * num_nodes + dst[sel].long()])
        del sel
        key = torch.sort(key).values
        e0, e1 = int(rowptr[r0]), int(rowptr[r1])
        assert key.numel() == e1 - e0
        row, col = key // num_nodes, key % num_nodes
        del key
        colind[e0:e1] = col
        weight[e0:e1] = dinv[row] * dinv[col]
        del row, col
    return BigCSRGraph(rowptr, colind, weight, num_nodes)


def papers100m_like(device, symmetrise=True, seed=0, num_nodes=PAPERS_NODES, num_pairs=PAPERS_PAIRS):
    """ogbn-papers100M's shape as one 64-bit CSR on `device`: directed (1.6e9 edges) or symmetrised as CogDL feeds it
    to GCN (3.2e9 edges > 2^31).  Smaller (num_nodes, num_pairs) give the same construction at test sizes."""
    src, dst = rmat_pairs_i32(num_nodes, num_pairs, seed, device)
    g = big_csr_from_pairs(src, dst, num_nodes, symmetrise)
    del src, dst
    if torch.device(device).type == "cuda":
        torch.cuda.empty_cache()
    return g
