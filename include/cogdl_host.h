/*
 * cogdl_host.h -- C ABI of libcogdl_host.so: the HIP-free host operators of the message-
 * passing path (CSR construction, neighbour sampling, CPU SpMM).  They run inside forked
 * DataLoader worker processes in CogDL (cogdl/data/sampler.py:82-116), so this library
 * never touches the HIP runtime.  All index arrays are int64 (what cogdl/operators/sample/
 * sample.cpp uses: data_ptr<int64_t>, :8-10).  Return 0 on success, COGDL_HOST_E* otherwise.
 */
#ifndef COGDL_HOST_H
#define COGDL_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#if defined(COGDL_HOST_BUILD)
#define COGDL_HOST_API __attribute__((visibility("default")))
#else
#define COGDL_HOST_API
#endif

enum cogdl_host_status {
    COGDL_HOST_OK = 0,
    COGDL_HOST_EINVAL = 1,   /* null pointer / negative size */
    COGDL_HOST_ERANGE = 2,   /* an index lies outside [0, num_nodes) */
    COGDL_HOST_ECAP = 3      /* caller-provided output capacity too small */
};

COGDL_HOST_API const char *cogdl_host_strerror(int status);

/* coo2csr_cpu(row, col, val, num_nodes) -> (row_ptr, col_ind, out_val)
 * Replaces sampler.coo2csr_cpu (cogdl/operators/sample/sample.cpp:191-231): stable counting
 * sort by row (edges of a row keep their COO order).  val/out_val may be NULL. */
COGDL_HOST_API int cogdl_host_coo2csr(const int64_t *row, const int64_t *col, const float *val, int64_t nnz,
                                      int64_t num_nodes, int64_t *row_ptr, int64_t *col_ind, float *out_val);

/* coo2csr_cpu_index(row, col, num_nodes) -> (row_ptr, perm); perm[j] = COO position of CSR
 * entry j.  Replaces sampler.coo2csr_cpu_index (sample.cpp:234-270).  `row` must be
 * contiguous (the Python binding makes it so; the reference silently mis-reads strided views). */
COGDL_HOST_API int cogdl_host_coo2csr_index(const int64_t *row, int64_t nnz, int64_t num_nodes, int64_t *row_ptr,
                                            int64_t *perm);

/* sample_adj(indptr, indices, node_idx, num_neighbors, replace)
 *   -> (out_indptr[batch+1], out_indices[E'], out_nodes[N'], out_edges[E'])
 * Replaces sampler.sample_adj (sample.cpp:6-144).  Relabelling contract: seeds get ids
 * 0..batch-1 in order, new neighbours get the next id in discovery order (row by row, in
 * the order the row's sampled edges are emitted).
 *   num_neighbors < 0 : all neighbours, deterministic, identical to the reference.
 *   replace != 0      : num_neighbors uniform draws with replacement per seed.
 *   else              : min(deg, num_neighbors) distinct neighbours, uniform without
 *                       replacement (Floyd), emitted in ascending CSR position.
 * Randomness comes from an explicit 64-bit seed (splitmix64/xoshiro256**), NOT libc rand():
 * reproducible, thread- and fork-safe.  Capacities: cap_edges >= sum of per-seed counts,
 * cap_nodes >= batch + cap_edges always suffices.  out_counts = {N', E'}. */
COGDL_HOST_API int cogdl_host_sample_adj(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                         const int64_t *node_idx, int64_t batch, int64_t num_neighbors,
                                         int replace, uint64_t seed, int64_t *out_indptr, int64_t *out_indices,
                                         int64_t *out_nodes, int64_t *out_edges, int64_t cap_edges,
                                         int64_t cap_nodes, int64_t *out_counts);

/* The same with the picks (two of the three dependent random reads per sampled edge) split over `nthreads` host
 * OpenMP threads (the runtime a torch process already has workers of; nthreads = 1 -- what torch gives DataLoader
 * workers -- never enters a parallel region, so forked workers stay clear of OpenMP); the relabelling in discovery
 * order stays sequential.  Bit-identical results for every nthreads >= 1 (the random stream
 * of a seed row depends on (seed, row) only). */
COGDL_HOST_API int cogdl_host_sample_adj_mt(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                            const int64_t *node_idx, int64_t batch, int64_t num_neighbors,
                                            int replace, uint64_t seed, int64_t *out_indptr, int64_t *out_indices,
                                            int64_t *out_nodes, int64_t *out_edges, int64_t cap_edges,
                                            int64_t cap_nodes, int64_t *out_counts, int nthreads);

/* subgraph(indptr, indices, node_idx) -> induced subgraph in CSR, relabelled by position in
 * node_idx.  Replaces sampler.subgraph (sample.cpp:146-188).  out_counts = {E'}. */
COGDL_HOST_API int cogdl_host_subgraph(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                       const int64_t *node_idx, int64_t batch, int64_t *out_indptr,
                                       int64_t *out_indices, int64_t *out_edges, int64_t cap_edges,
                                       int64_t *out_counts);

/* csr_spmm_cpu(rowptr, colind, val, dense) -> out   (CPU tensors; CogDL's CPU inference path)
 * Replaces spmm_cpu.csr_spmm_cpu (cogdl/operators/spmm/spmm_cpu.cpp:6-58): rows in parallel
 * over `nthreads` host threads, each output element accumulated sequentially in CSR order
 * with fp32 multiply then add (bit-identical to the reference).  int32 indices, 64-bit offsets. */
COGDL_HOST_API int cogdl_host_csr_spmm_f32(const int32_t *rowptr, const int32_t *colind, const float *val,
                                           const float *dense, float *out, int64_t m, int64_t k, int nthreads);
/* The same with int64 row pointers: graphs of 2^31 edges and more on the host (the reference's `int key` / `int ik = i * k`
 * loop, spmm_cpu.cpp:24-33, overflows at 2^31 edges and already at 16.7 M rows x 128 columns).  Same arithmetic. */
COGDL_HOST_API int cogdl_host_csr_spmm_f32_i64(const int64_t *rowptr, const int32_t *colind, const float *val,
                                               const float *dense, float *out, int64_t m, int64_t k, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
