"""A multilevel graph partitioner for the 1-D vertex sharding of configs[4] (SURVEY.md section 8e / section 7 "halo volume is
the risk"): the reference's only partitioner is METIS on the host (cogdl/data/sampler.py:188-243, ClusterGCN's loader; the
`metis` module is not even installable here), and the level order of `cogdl_amd.dist.bfs_order` recovers locality only
where communities are consecutive breadth-first levels.

The scheme is the classic multilevel one (METIS, KaHIP's "social" configuration), every phase made of data-parallel
passes over the edge list so that it runs on GPU tensors:

  coarsen     size-constrained LABEL PROPAGATION clustering (Raghavan et al. 2007; as a coarsening scheme: Meyerhenke,
              Sanders, Schulz 2014): every vertex adopts the label carrying the largest edge weight among its neighbours,
              random tie-breaking, a cluster may not outgrow a fraction of a part; clusters are contracted into vertices of
              the next level (edge weights summed, self loops dropped).  A few sweeps per level, levels until the graph is
              a few thousand vertices or stops shrinking.
  initial     on the coarsest graph (<= 2048 vertices, dense weight matrix): greedy graph growing (METIS's GGGP) -- a part grows
              from a seed by always taking the unassigned vertex with the heaviest connection to it until it holds its
              share of the weight -- from several seeds, each followed by the refinement below; the best cut is kept.
  uncoarsen   labels are projected down one level at a time and REFINED by size-constrained label propagation over the
              `world` part labels -- there every sweep is ONE weighted csr_spmm of this library:
              (A_w L)[v, p] = edge weight between v and part p, L the one-hot label matrix.

Vertex weight = degree + 1, so parts balance EDGES (what a rank's SpMM time follows).  Deterministic for a seed and a
device type.  Scale: every sweep sorts the level's edge list (torch.unique), fine for graphs of up to a few 10^8 edges on
one GPU; a papers100M-sized graph would be partitioned by a distributed variant of the same passes (not built)."""
import torch

COARSEST = 2048          # coarsen until at most this many vertices are left (or the graph stops shrinking)
CLUSTER_FRACTION = 0.1   # a cluster may hold at most this fraction of a part's weight
COARSEN_SWEEPS = 4
REFINE_SWEEPS = 10


class _Level:
    __slots__ = ("rowptr", "colind", "ew", "vw", "rows", "n")

    def __init__(self, rowptr, colind, ew, vw):
        self.rowptr, self.colind, self.ew, self.vw = rowptr, colind, ew, vw
        self.n = rowptr.numel() - 1
        self.rows = torch.repeat_interleave(torch.arange(self.n, device=rowptr.device), rowptr[1:] - rowptr[:-1])


def _default_spmm(rowptr32, colind32, weight, dense):
    from .operators.spmm import csr_spmm_raw

    return csr_spmm_raw(rowptr32, colind32, weight, dense)


def lp_cluster(level, max_weight, sweeps, gen):
    """Size-constrained label-propagation clustering of one level -> (cluster_of [n] int64 in [0, C), C)."""
    n, dev = level.n, level.rowptr.device
    labels = torch.arange(n, device=dev)
    if level.colind.numel() == 0:
        return labels, n
    for _ in range(sweeps):
        key = level.rows * n + labels[level.colind]
        ukey, inv = torch.unique(key, return_inverse=True)
        w = torch.zeros(ukey.numel(), dtype=torch.float32, device=dev).index_add_(0, inv, level.ew)
        w = w * (1.0 + 1e-3 * torch.rand(w.numel(), generator=gen, device=dev))  # random tie-breaking
        v, lab = ukey // n, ukey % n
        best_w = torch.zeros(n, dtype=torch.float32, device=dev).scatter_reduce(0, v, w, "amax", include_self=True)
        cand = torch.where(w >= best_w[v], lab, torch.full_like(lab, n))
        best = torch.full((n,), n, dtype=torch.long, device=dev).scatter_reduce(0, v, cand, "amin", include_self=True)
        best = torch.where(best >= n, labels, best)  # isolated vertices keep their label
        move = (best != labels) & (torch.rand(n, generator=gen, device=dev) < 0.5)
        cw = torch.zeros(n, dtype=torch.float32, device=dev).index_add_(0, labels, level.vw)
        inflow = torch.zeros(n, dtype=torch.float32, device=dev).index_add_(0, best[move], level.vw[move])
        room = (max_weight - cw).clamp(min=0.0)
        accept = (room / inflow.clamp(min=1e-9)).clamp(max=1.0)
        move &= torch.rand(n, generator=gen, device=dev) < accept[best]
        if not bool(move.any()):
            break
        labels = torch.where(move, best, labels)
    uniq, cluster_of = torch.unique(labels, return_inverse=True)
    return cluster_of, int(uniq.numel())


def contract(level, cluster_of, n_clusters):
    """The next level: clusters become vertices, parallel edges are merged (weights summed), self loops dropped."""
    dev = level.rowptr.device
    crow, ccol = cluster_of[level.rows], cluster_of[level.colind]
    keep = crow != ccol
    key = crow[keep] * n_clusters + ccol[keep]
    ukey, inv = torch.unique(key, return_inverse=True)  # sorted: by (row, col)
    ew = torch.zeros(ukey.numel(), dtype=torch.float32, device=dev).index_add_(0, inv, level.ew[keep])
    rows = ukey // n_clusters
    rowptr = torch.zeros(n_clusters + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_clusters), 0)
    vw = torch.zeros(n_clusters, dtype=torch.float32, device=dev).index_add_(0, cluster_of, level.vw)
    return _Level(rowptr, ukey % n_clusters, ew, vw)


def refine(level, labels, world, cap, sweeps, gen, spmm):
    """Size-constrained label propagation over the `world` part labels: one weighted SpMM per sweep."""
    n, dev = level.n, level.rowptr.device
    if level.colind.numel() == 0:
        return labels
    rp32, ci32 = level.rowptr.to(torch.int32).contiguous(), level.colind.to(torch.int32).contiguous()
    rows = torch.arange(n, device=dev)
    for _ in range(sweeps):
        onehot = torch.zeros(n, world, dtype=torch.float32, device=dev)
        onehot[rows, labels] = 1.0
        gain = spmm(rp32, ci32, level.ew, onehot)
        gain[rows, labels] += 1e-3  # staying put wins ties
        best = gain.argmax(dim=1)
        move = (best != labels) & (torch.rand(n, generator=gen, device=dev) < 0.5)
        size = torch.zeros(world, dtype=torch.float32, device=dev).index_add_(0, labels, level.vw)
        inflow = torch.zeros(world, dtype=torch.float32, device=dev).index_add_(0, best[move], level.vw[move])
        outflow = torch.zeros(world, dtype=torch.float32, device=dev).index_add_(0, labels[move], level.vw[move])
        room = (cap - size + 0.5 * outflow).clamp(min=0.0)
        accept = (room / inflow.clamp(min=1e-9)).clamp(max=1.0)
        move &= torch.rand(n, generator=gen, device=dev) < accept[best]
        if not bool(move.any()):
            break
        labels = torch.where(move, best, labels)
    return labels


def cut_weight(level, labels):
    return float(level.ew[labels[level.rows] != labels[level.colind]].sum())


def rebalance(level, labels, world, cap, gen, spmm, rounds=8):
    """Bring every part under `cap`: an overfull part gives up the vertices whose move to their best non-full part costs
    the least connection weight, just enough of them; then nothing but strictly feasible moves."""
    n, dev = level.n, level.rowptr.device
    if level.colind.numel() == 0:
        return labels
    rp32, ci32 = level.rowptr.to(torch.int32).contiguous(), level.colind.to(torch.int32).contiguous()
    rows = torch.arange(n, device=dev)
    for _ in range(rounds):
        size = torch.zeros(world, dtype=torch.float32, device=dev).index_add_(0, labels, level.vw)
        over = size > cap
        if not bool(over.any()):
            break
        onehot = torch.zeros(n, world, dtype=torch.float32, device=dev)
        onehot[rows, labels] = 1.0
        gain = spmm(rp32, ci32, level.ew, onehot)
        own = gain[rows, labels]
        open_parts = size < 0.97 * cap
        if not bool(open_parts.any()):
            open_parts = size <= size.min()
        alt_gain = gain.masked_fill(~open_parts.view(1, -1), -1.0)
        alt_gain[rows, labels] = -1.0
        alt = alt_gain.argmax(dim=1)
        loss = own - alt_gain[rows, alt]
        new = labels.clone()
        budget = (cap - size).clamp(min=0.0)  # what the receiving parts can still take in this round
        for p in torch.nonzero(over).flatten().tolist():
            idx = torch.nonzero(labels == p).flatten()
            order = idx[torch.argsort(loss[idx])]
            cum = torch.cumsum(level.vw[order], 0)
            excess = float(size[p] - cap)
            take = order[: int(torch.searchsorted(cum, torch.tensor(excess, device=dev))) + 1]
            # respect the receivers' room: drop the movers that would overfill their target
            tgt = alt[take]
            w_take = level.vw[take]
            for q in torch.unique(tgt).tolist():
                sel = take[tgt == q]
                c = torch.cumsum(level.vw[sel], 0)
                ok = sel[c <= budget[q]]
                new[ok] = q
                budget[q] = budget[q] - float(level.vw[ok].sum())
        if torch.equal(new, labels):
            break
        labels = new
    return labels


def force_balance(level, labels, world, cap, max_rounds=64):
    """What rebalance() could not place (hub neighbourhoods: every well-connected part is full as well) goes to the LIGHTEST
    parts, connection or not: an overfull part sheds its lightest vertices, one per open part and round, until no part
    exceeds `cap` or nothing fits any more (a single vertex heavier than a part's share stays where it is)."""
    dev = labels.device
    for _ in range(max_rounds):
        size = torch.zeros(world, dtype=torch.float32, device=dev).index_add_(0, labels, level.vw)
        over = size > cap
        if not bool(over.any()):
            break
        cand = torch.nonzero(over[labels]).flatten()  # vertices of overfull parts, lightest first within their part
        cand = cand[torch.argsort(level.vw[cand])]
        part_of = labels[cand]
        # per overfull part: only as many of its vertices as its excess needs (prefix by weight)
        order = torch.argsort(part_of, stable=True)
        cand, part_of = cand[order], part_of[order]
        w = level.vw[cand]
        cum = torch.cumsum(w, 0)
        start = torch.zeros(world, dtype=torch.float32, device=dev)
        first = torch.ones_like(part_of, dtype=torch.bool)
        first[1:] = part_of[1:] != part_of[:-1]
        start[part_of[first]] = (cum - w)[first]
        shed = (cum - start[part_of]) - w < (size - cap)[part_of]  # weight shed BEFORE this vertex is still short of the excess
        cand, w = cand[shed], w[shed]
        if cand.numel() == 0:
            break
        room = cap - size
        targets = torch.argsort(room, descending=True)  # the lightest parts first
        k = min(int(cand.numel()), int((room > 0).sum()))
        if k == 0:
            break
        heavy_first = torch.argsort(w, descending=True)[:k]  # heaviest movers to the roomiest parts
        mv, tg = cand[heavy_first], targets[:k]
        fits = level.vw[mv] <= room[tg]
        if not bool(fits.any()):
            break
        labels = labels.clone()
        labels[mv[fits]] = tg[fits]
    return labels


def initial_partition(level, world, cap, gen, spmm, tries=6):
    """Coarsest level: greedy graph growing from several seeds, each refined; the best cut wins."""
    dev = level.rowptr.device
    n = level.n
    total = float(level.vw.sum())
    dense = torch.zeros(n, n, dtype=torch.float32, device=dev)
    dense.index_put_((level.rows, level.colind), level.ew, accumulate=True)
    best, best_cut = None, None
    for t in range(tries):
        labels = torch.full((n,), world - 1, dtype=torch.long, device=dev)
        free = torch.ones(n, dtype=torch.bool, device=dev)
        for p in range(world - 1):
            if not bool(free.any()):
                break
            target = total / world
            cand = torch.nonzero(free).flatten()
            seed = cand[int(torch.randint(0, cand.numel(), (1,), generator=gen, device=dev))] if t else cand[torch.argmax(level.vw[cand])]
            conn = torch.zeros(n, dtype=torch.float32, device=dev)
            weight, v = 0.0, int(seed)
            while True:
                labels[v] = p
                free[v] = False
                weight += float(level.vw[v])
                conn += dense[v]
                if weight >= target or not bool(free.any()):
                    break
                score = torch.where(free, conn, torch.full_like(conn, -1.0))
                v = int(torch.argmax(score))
                if weight + float(level.vw[v]) > cap and weight > 0.5 * target:
                    break
        labels = refine(level, labels, world, cap, 3 * REFINE_SWEEPS, gen, spmm)
        labels = rebalance(level, labels, world, cap, gen, spmm)
        c = cut_weight(level, labels)
        if best_cut is None or c < best_cut:
            best, best_cut = labels, c
    return best


def multilevel_partition(rowptr, colind, world, seed=0, slack=1.03, spmm=None, info=None, balance="edges"):
    """labels [N] int64 in [0, world) for the symmetric graph (rowptr, colind); see the module docstring.
    spmm: (rowptr int32, colind int32, weight f32, dense [n, world]) -> [n, world] (default: cogdl_hip_csr_spmm; tests
    inject a CPU stand-in).  info: optional dict that receives the level sizes and the cut after every phase.
    balance: "edges" (vertex weight = degree + 1: what a shard's SpMM time follows) or "vertices" (unit weights: METIS's
    default, what ClusterGCN's equally sized clusters want)."""
    if balance not in ("edges", "vertices"):
        raise ValueError("balance must be 'edges' or 'vertices'")
    spmm = spmm or _default_spmm
    dev = rowptr.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    rp, ci = rowptr.to(torch.long), colind.to(torch.long)
    n = rp.numel() - 1
    if world <= 1 or n == 0:
        return torch.zeros(n, dtype=torch.long, device=dev)
    vw = (rp[1:] - rp[:-1]).to(torch.float32) + 1.0 if balance == "edges" else torch.ones(n, dtype=torch.float32, device=dev)
    if n <= 8 * world:
        # fewer than eight vertices per part: nothing for a multilevel scheme to work with (its coarsest level must hold
        # several vertices per part) -- contiguous weight-balanced blocks in the given order
        cum = torch.cumsum(vw, 0)
        return torch.clamp(((cum - 0.5 * vw) * (world / float(vw.sum()))).floor(), 0, world - 1).long()
    levels = [_Level(rp, ci, torch.ones(ci.numel(), dtype=torch.float32, device=dev), vw)]
    maps = []
    cap = slack * float(vw.sum()) / world
    while levels[-1].n > max(32 * world, 64):
        cur = levels[-1]
        cluster_of, c = lp_cluster(cur, CLUSTER_FRACTION * cap, COARSEN_SWEEPS, gen)
        if c > 0.9 * cur.n:  # no structure left to contract (R-MAT: one giant hub neighbourhood at its size limit)
            break
        if c < 8 * world and cur.n <= COARSEST:  # the next level would be too coarse to balance: partition this one
            break
        maps.append(cluster_of)
        levels.append(contract(cur, cluster_of, c))
    if info is not None:
        info["levels"] = [lv.n for lv in levels]
    if levels[-1].n > 4 * COARSEST:  # could not be coarsened far enough for the dense initial phase (no cluster structure):
        # contiguous weight-balanced blocks of the coarsest level's order are the start, refinement does the rest
        lv = levels[-1]
        cum = torch.cumsum(lv.vw, 0)
        labels = torch.clamp((cum - 0.5 * lv.vw) * world / float(lv.vw.sum()), max=world - 1e-3).long()
        labels = rebalance(lv, refine(lv, labels, world, cap, 3 * REFINE_SWEEPS, gen, spmm), world, cap, gen, spmm)
    else:
        labels = initial_partition(levels[-1], world, cap, gen, spmm)
    if info is not None:
        info["cut_coarsest"] = cut_weight(levels[-1], labels)
    for lv, cluster_of in zip(reversed(levels[:-1]), reversed(maps)):
        labels = refine(lv, labels[cluster_of], world, cap, REFINE_SWEEPS, gen, spmm)
        labels = rebalance(lv, labels, world, cap, gen, spmm)
    labels = force_balance(levels[0], labels, world, cap)
    if info is not None:
        info["cut"] = cut_weight(levels[0], labels)
        size = torch.zeros(world, dtype=torch.float32, device=dev).index_add_(0, labels, vw)
        info["imbalance"] = float(size.max()) * world / float(vw.sum())
    return labels
