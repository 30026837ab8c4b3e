"""cogdl_amd.operators -- same module layout and exported names as cogdl/operators/.

    spmm.py          csrspmm, spmm_cpu                      (cogdl/operators/spmm.py)
    edge_softmax.py  csr_edge_softmax                       (cogdl/operators/edge_softmax.py)
    mhspmm.py        csrmhspmm                              (cogdl/operators/mhspmm.py)
    scatter_max.py   scatter_max                            (cogdl/operators/scatter_max.py)
    fused_gat.py     fused_gat_func                         (cogdl/operators/fused_gat.py)
    sample.py        sample_adj_c, subgraph_c, coo2csr_cpu, coo2csr_cpu_index  (cogdl/operators/sample.py)

    ops.py           scatter_add, op_aggr, s_*_e_sum / s_*_e_mean (fused HIP), s_*_e, s_*_t   (cogdl/operators/ops.py)

Submodules are imported lazily: GPU modules load libcogdl_hip.so at import and raise if it
is missing; `sample` only needs libcogdl_host.so and is safe in forked CPU workers.
"""
