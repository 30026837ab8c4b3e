"""`metis.part_graph` for CogDL's ClusterGCN loader on a box without METIS.

`cogdl.data.sampler.ClusteredDataset` (cogdl/data/sampler.py:188-243) does `import metis` and calls
`metis.part_graph(adjacency_list, n_cluster, seed=1)` once per dataset, on the host; the module is an optional
dependency that is absent from this image, and the loader exits the process without it.
`cogdl_amd.install(metis=True)` registers THIS module under the name `metis` when the real one cannot be imported, so
that the unchanged loader runs: the partition comes from `cogdl_amd.partitioner.multilevel_partition` (label-propagation
multilevel scheme, every sweep a csr_spmm of this library on the GPU) with unit vertex weights -- METIS's default
objective (equal vertex counts, small edge cut), not METIS's algorithm and not its cut quality on every graph
(DESIGN.md section 7 has the measurements).  Needs a GPU; there is no CPU route behind it.
"""
import numpy as np
import torch

from . import _lib
from .partitioner import multilevel_partition

__all__ = ["part_graph"]


def _csr_of(graph):
    """adjacency list (sequence of neighbour sequences, what the reference passes) or (xadj, adjncy) -> int64 numpy CSR"""
    if isinstance(graph, tuple) and len(graph) == 2 and not isinstance(graph[0], (list, tuple)) and np.ndim(graph[0]) == 1 \
            and np.ndim(graph[1]) == 1 and len(graph[0]) and int(np.asarray(graph[0])[-1]) == len(graph[1]):
        return np.asarray(graph[0], dtype=np.int64), np.asarray(graph[1], dtype=np.int64)
    if hasattr(graph, "adj") or hasattr(graph, "nodes"):
        raise _lib.BackendError("metis.part_graph (cogdl_amd): pass an adjacency list or (xadj, adjncy), not a networkx graph")
    lens = np.fromiter((len(a) for a in graph), dtype=np.int64, count=len(graph))
    xadj = np.zeros(len(graph) + 1, dtype=np.int64)
    np.cumsum(lens, out=xadj[1:])
    adjncy = np.concatenate([np.asarray(a, dtype=np.int64) for a in graph]) if len(graph) and xadj[-1] else np.zeros(0, np.int64)
    return xadj, adjncy


def part_graph(graph, nparts=2, tpwgts=None, ubvec=None, recursive=False, seed=0, device=None, **opts):
    """-> (edgecuts, parts): parts[v] in [0, nparts), edgecuts = undirected edges between different parts -- the return
    value of the `metis` package's function of the same name.  `graph`: adjacency list (list of neighbour arrays, one
    per vertex) or an (xadj, adjncy) pair; must be symmetric, as METIS requires.  tpwgts / ubvec are not supported
    (equal parts, 3 % slack); other METIS options are ignored."""
    if tpwgts is not None or ubvec is not None:
        raise _lib.BackendError("metis.part_graph (cogdl_amd): target part weights / imbalance vectors are not supported")
    nparts = int(nparts)
    if nparts < 1:
        raise _lib.BackendError("metis.part_graph: nparts must be >= 1")
    xadj, adjncy = _csr_of(graph)
    n = len(xadj) - 1
    if n == 0 or nparts == 1:
        return 0, [0] * n
    if adjncy.size and (adjncy.min() < 0 or adjncy.max() >= n):
        raise _lib.BackendError("metis.part_graph: a neighbour id lies outside [0, %d)" % n)
    if not torch.cuda.is_available():
        raise _lib.BackendError("metis.part_graph (cogdl_amd): the partitioner runs on the GPU; no device is present")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    rowptr = torch.from_numpy(xadj).to(dev)
    colind = torch.from_numpy(adjncy).to(dev)
    labels = multilevel_partition(rowptr, colind, nparts, seed=int(seed or 0), balance="vertices")
    rows = torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1])
    cut = int((labels[rows] != labels[colind]).sum()) // 2
    return cut, labels.cpu().tolist()
