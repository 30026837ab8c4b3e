// host_ops.cpp -- libcogdl_host.so: HIP-free host operators of the message-passing path.
// Replaces cogdl/operators/sample/sample.cpp (sample_adj, subgraph, coo2csr_cpu,
// coo2csr_cpu_index) and cogdl/operators/spmm/spmm_cpu.cpp (csr_spmm_cpu) behind a C ABI
// (include/cogdl_host.h).  Must stay free of any HIP/ROCm dependency: CogDL calls the sampler
// from forked DataLoader workers (cogdl/data/sampler.py:82-116).
//
// Built with -ffp-contract=off and no -march: the CPU SpMM's `acc + v*b` is a rounded
// multiply then a rounded add, as in the reference's own x86-64 build.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/cogdl_host.h"

namespace {

// ---------------------------------------------------------------- random numbers
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    // uniform in [0, n), n > 0 (multiply-high; bias < n / 2^64)
    int64_t below(int64_t n) { return (int64_t)(((unsigned __int128)next() * (unsigned __int128)(uint64_t)n) >> 64); }
};

inline uint64_t stream_seed(uint64_t seed, uint64_t i) {
    SplitMix64 g(seed ^ (0xd1342543de82ef95ull * (i + 1)));
    return g.next();
}

// Choose `k` distinct positions out of [0, deg) uniformly; result ascending.
void sample_without_replacement(int64_t deg, int64_t k, SplitMix64 &rng, std::vector<int64_t> &pick) {
    pick.clear();
    if (k >= deg) {
        for (int64_t j = 0; j < deg; ++j) pick.push_back(j);
        return;
    }
    if (k * 4 >= deg) {  // dense: partial Fisher-Yates over the row's positions
        std::vector<int64_t> idx((size_t)deg);
        for (int64_t j = 0; j < deg; ++j) idx[(size_t)j] = j;
        for (int64_t j = 0; j < k; ++j) std::swap(idx[(size_t)j], idx[(size_t)(j + rng.below(deg - j))]);
        pick.assign(idx.begin(), idx.begin() + k);
    } else if (k <= 128) {  // sparse, small: Floyd with a linear membership scan
        for (int64_t j = deg - k; j < deg; ++j) {
            const int64_t t = rng.below(j + 1);
            pick.push_back(std::find(pick.begin(), pick.end(), t) == pick.end() ? t : j);
        }
    } else {  // sparse, large: Floyd with a hash set
        std::unordered_set<int64_t> seen;
        seen.reserve((size_t)k * 2);
        for (int64_t j = deg - k; j < deg; ++j) {
            const int64_t t = rng.below(j + 1);
            const int64_t v = seen.count(t) ? j : t;
            seen.insert(v);
            pick.push_back(v);
        }
    }
    std::sort(pick.begin(), pick.end());
}

// global id -> local id map of the samplers: a per-thread array that is all -1 between calls (only the entries a
// call touched are reset), so a call costs O(batch + sampled edges) instead of the O(num_nodes) fill the reference
// pays per mini-batch (sample.cpp:22: torch::full({N}, -1)).  Per thread => safe in DataLoader workers (threads or
// forked processes).
struct LocalIds {
    std::vector<int64_t> &map;
    const int64_t *touched;
    int64_t n_touched = 0;
    static std::vector<int64_t> &storage(int64_t num_nodes) {
        thread_local std::vector<int64_t> v;
        if ((int64_t)v.size() < num_nodes) v.resize((size_t)num_nodes, -1);
        return v;
    }
    LocalIds(int64_t num_nodes, const int64_t *touched_list) : map(storage(num_nodes)), touched(touched_list) {}
    ~LocalIds() {
        for (int64_t i = 0; i < n_touched; ++i) map[(size_t)touched[i]] = -1;
    }
};

bool in_range(const int64_t *a, int64_t n, int64_t hi) {
    for (int64_t i = 0; i < n; ++i)
        if (a[i] < 0 || a[i] >= hi) return false;
    return true;
}

// stable counting sort of edge positions by row; row_ptr gets the CSR offsets
template <typename Emit>
void counting_sort_by_row(const int64_t *row, int64_t nnz, int64_t num_nodes, int64_t *row_ptr, Emit emit) {
    std::fill(row_ptr, row_ptr + num_nodes + 1, int64_t(0));
    for (int64_t i = 0; i < nnz; ++i) ++row_ptr[row[i] + 1];
    for (int64_t r = 0; r < num_nodes; ++r) row_ptr[r + 1] += row_ptr[r];
    std::vector<int64_t> cursor(row_ptr, row_ptr + num_nodes);
    for (int64_t i = 0; i < nnz; ++i) emit(cursor[(size_t)row[i]]++, i);
}

// ---------------------------------------------------------------- CPU SpMM rows
constexpr int kPrefetchAhead = 8;
constexpr int kColBlock = 64;  // floats a row's accumulator holds in registers (8 AVX2 vectors)

// out[i, :] = sum_e val[e] * dense[colind[e], :] for the rows [r0, r1): every column's sum is taken in CSR edge order with
// a separately rounded multiply and add (-ffp-contract=off) -- the arithmetic of the reference's loop
// (operators/spmm/spmm_cpu.cpp:39-58 as it is shipped), bit for bit.  Columns are walked in blocks of kColBlock whose
// partial sums stay in registers across the row's edges (the plain loop stores and reloads the output row once per edge:
// a store-forwarding chain per column); the gathered rows are random 4k-byte reads out of a matrix far larger than the
// caches, so the row kPrefetchAhead edges ahead (across row boundaries: colind is one array) is requested on the first
// column block's pass.
// P = the row pointer type: int32 (what CogDL passes) or int64 (graphs of 2^31 edges and more; the reference's `int` loop,
// spmm_cpu.cpp:24-33, overflows there -- and at i * k >= 2^31 already)
template <typename P>
static inline __attribute__((always_inline)) void spmm_rows_body(const P *rowptr, const int32_t *colind, const float *val,
                                                                  const float *dense, float *out, int64_t r0, int64_t r1,
                                                                  int64_t k, int64_t nnz_total) {
    const int64_t k_blocked = k / kColBlock * kColBlock;
    for (int64_t i = r0; i < r1; ++i) {
        float *__restrict__ o = out + i * k;
        const P e0 = rowptr[i], e1 = rowptr[i + 1];
        for (int64_t cb = 0; cb < k_blocked; cb += kColBlock) {
            float acc[kColBlock];
            for (int t = 0; t < kColBlock; ++t) acc[t] = 0.f;
            for (P e = e0; e < e1; ++e) {
                const float *__restrict__ b = dense + (int64_t)colind[e] * k + cb;
                if (cb == 0 && e + kPrefetchAhead < nnz_total) {
                    const char *nb = (const char *)(dense + (int64_t)colind[e + kPrefetchAhead] * k);
                    for (int64_t off = 0; off < k * (int64_t)sizeof(float); off += 64) __builtin_prefetch(nb + off, 0, 1);
                }
                if (val) {
                    const float v = val[e];
                    for (int t = 0; t < kColBlock; ++t) acc[t] = acc[t] + v * b[t];
                } else {
                    for (int t = 0; t < kColBlock; ++t) acc[t] = acc[t] + b[t];
                }
            }
            for (int t = 0; t < kColBlock; ++t) o[cb + t] = acc[t];
        }
        if (k_blocked < k) {  // the last k % kColBlock columns: the plain loop
            for (int64_t t = k_blocked; t < k; ++t) o[t] = 0.f;
            for (P e = e0; e < e1; ++e) {
                const float *__restrict__ b = dense + (int64_t)colind[e] * k;
                if (k_blocked == 0 && e + kPrefetchAhead < nnz_total) {
                    const char *nb = (const char *)(dense + (int64_t)colind[e + kPrefetchAhead] * k);
                    for (int64_t off = 0; off < k * (int64_t)sizeof(float); off += 64) __builtin_prefetch(nb + off, 0, 1);
                }
                if (val) {
                    const float v = val[e];
                    for (int64_t t = k_blocked; t < k; ++t) o[t] = o[t] + v * b[t];
                } else {
                    for (int64_t t = k_blocked; t < k; ++t) o[t] = o[t] + b[t];
                }
            }
        }
    }
}

__attribute__((target_clones("avx2", "default"))) void spmm_rows(const int32_t *rowptr, const int32_t *colind,
                                                                  const float *val, const float *dense, float *out,
                                                                  int64_t r0, int64_t r1, int64_t k, int64_t nnz_total) {
    spmm_rows_body<int32_t>(rowptr, colind, val, dense, out, r0, r1, k, nnz_total);
}
__attribute__((target_clones("avx2", "default"))) void spmm_rows64(const int64_t *rowptr, const int32_t *colind,
                                                                    const float *val, const float *dense, float *out,
                                                                    int64_t r0, int64_t r1, int64_t k, int64_t nnz_total) {
    spmm_rows_body<int64_t>(rowptr, colind, val, dense, out, r0, r1, k, nnz_total);
}

// rows split over the threads at equal (edges + rows) boundaries
template <typename P, typename F>
int spmm_threads(const P *rowptr, const int32_t *colind, const float *val, const float *dense, float *out, int64_t m,
                 int64_t k, int nthreads, F rows_fn) {
    if (m < 0 || k < 0) return COGDL_HOST_EINVAL;
    if (m == 0 || k == 0) return COGDL_HOST_OK;
    if (!rowptr || !dense || !out) return COGDL_HOST_EINVAL;
    const int64_t nnz = rowptr[m];
    if (nthreads < 1) nthreads = 1;
    // a thread per ~64k edge-columns of work at most
    const int64_t work = (nnz + m) * k;
    int t = (int)std::min<int64_t>(nthreads, std::max<int64_t>(1, work / (1 << 16)));
    if (t <= 1) {
        rows_fn(rowptr, colind, val, dense, out, 0, m, k, nnz);
        return COGDL_HOST_OK;
    }
    std::vector<int64_t> cut((size_t)t + 1, 0);
    cut[(size_t)t] = m;
    for (int i = 1; i < t; ++i) {
        const int64_t target = (nnz + m) * i / t;  // balance edges + rows
        int64_t lo = 0, hi = m;
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            if ((int64_t)rowptr[mid] + mid < target) lo = mid + 1; else hi = mid;
        }
        cut[(size_t)i] = std::max(lo, cut[(size_t)i - 1]);
    }
    std::vector<std::thread> pool;
    for (int i = 0; i < t; ++i)
        pool.emplace_back(rows_fn, rowptr, colind, val, dense, out, cut[(size_t)i], cut[(size_t)i + 1], k, nnz);
    for (auto &th : pool) th.join();
    return COGDL_HOST_OK;
}

}  // namespace

extern "C" {

const char *cogdl_host_strerror(int status) {
    switch (status) {
        case COGDL_HOST_OK: return "ok";
        case COGDL_HOST_EINVAL: return "invalid argument";
        case COGDL_HOST_ERANGE: return "index out of range";
        case COGDL_HOST_ECAP: return "output capacity too small";
        default: return "unknown status";
    }
}

int cogdl_host_coo2csr(const int64_t *row, const int64_t *col, const float *val, int64_t nnz, int64_t num_nodes,
                       int64_t *row_ptr, int64_t *col_ind, float *out_val) {
    if (nnz < 0 || num_nodes < 0 || !row_ptr || (nnz > 0 && (!row || !col || !col_ind))) return COGDL_HOST_EINVAL;
    if (!in_range(row, nnz, num_nodes)) return COGDL_HOST_ERANGE;
    const bool with_val = val && out_val;
    counting_sort_by_row(row, nnz, num_nodes, row_ptr, [&](int64_t dst, int64_t src) {
        col_ind[dst] = col[src];
        if (with_val) out_val[dst] = val[src];
    });
    return COGDL_HOST_OK;
}

int cogdl_host_coo2csr_index(const int64_t *row, int64_t nnz, int64_t num_nodes, int64_t *row_ptr, int64_t *perm) {
    if (nnz < 0 || num_nodes < 0 || !row_ptr || (nnz > 0 && (!row || !perm))) return COGDL_HOST_EINVAL;
    if (!in_range(row, nnz, num_nodes)) return COGDL_HOST_ERANGE;
    counting_sort_by_row(row, nnz, num_nodes, row_ptr, [&](int64_t dst, int64_t src) { perm[dst] = src; });
    return COGDL_HOST_OK;
}

// The picks of seeds [s0, s1): edge positions -> out_edges, the neighbours' GLOBAL ids -> out_indices, both at the
// offsets out_indptr holds already.  Every sampled edge costs two dependent random reads here (indptr[seed],
// indices[edge]): the seeds go in blocks of kBlock with one pass per level, each pass prefetching what the next one
// reads, so that the misses of a block overlap.  Pure function of (seed word, seed row): threads can split the seeds
// any way they like.  Returns false on a neighbour id outside the graph.
static bool sample_picks(const int64_t *indptr, const int64_t *indices, int64_t num_nodes, const int64_t *node_idx,
                         int64_t s0, int64_t s1, int64_t num_neighbors, int replace, uint64_t seed,
                         const int64_t *out_indptr, int64_t *out_indices, int64_t *out_edges) {
    constexpr int64_t kBlock = 256;
    std::vector<int64_t> pick;
    for (int64_t b0 = s0; b0 < s1; b0 += kBlock) {
        const int64_t b1 = std::min(s1, b0 + kBlock);
        for (int64_t i = b0; i < b1; ++i) __builtin_prefetch(indptr + node_idx[i]);
        for (int64_t i = b0; i < b1; ++i) {  // level 1: row extents -> picks (edge positions), prefetch the neighbour ids
            const int64_t begin = indptr[node_idx[i]];
            const int64_t deg = indptr[node_idx[i] + 1] - begin;
            SplitMix64 rng(stream_seed(seed, (uint64_t)i));  // per-seed stream: result independent of batching
            pick.clear();
            if (num_neighbors < 0) {
                for (int64_t j = 0; j < deg; ++j) pick.push_back(j);
            } else if (replace) {
                if (deg > 0)
                    for (int64_t j = 0; j < num_neighbors; ++j) pick.push_back(rng.below(deg));
            } else {
                sample_without_replacement(deg, num_neighbors, rng, pick);
            }
            int64_t *dst = out_edges + out_indptr[i];
            for (size_t q = 0; q < pick.size(); ++q) {
                dst[q] = begin + pick[q];
                __builtin_prefetch(indices + begin + pick[q]);
            }
        }
        for (int64_t e = out_indptr[b0]; e < out_indptr[b1]; ++e) {  // level 2: the neighbour ids
            const int64_t src = indices[out_edges[e]];
            if (src < 0 || src >= num_nodes) return false;
            out_indices[e] = src;
        }
    }
    return true;
}

static int64_t sample_count(int64_t deg, int64_t num_neighbors, int replace) {
    if (num_neighbors < 0) return deg;
    if (replace) return deg > 0 ? num_neighbors : 0;
    return std::min(deg, num_neighbors);
}

int cogdl_host_sample_adj_mt(const int64_t *indptr, const int64_t *indices, int64_t num_nodes, const int64_t *node_idx,
                             int64_t batch, int64_t num_neighbors, int replace, uint64_t seed, int64_t *out_indptr,
                             int64_t *out_indices, int64_t *out_nodes, int64_t *out_edges, int64_t cap_edges,
                             int64_t cap_nodes, int64_t *out_counts, int nthreads) {
    if (!indptr || !out_indptr || !out_counts || batch < 0 || num_nodes < 0 || (batch > 0 && !node_idx))
        return COGDL_HOST_EINVAL;
    if (!in_range(node_idx, batch, num_nodes)) return COGDL_HOST_ERANGE;
    if (batch > cap_nodes) return COGDL_HOST_ECAP;
    // Three phases.  (1) how many edges every seed contributes -> out_indptr (one random read per seed, prefetched);
    // (2) the picks and the neighbours' global ids, written at their final offsets -- independent per seed, split over
    // `nthreads` OpenMP threads (nthreads = 1, what DataLoader workers get, never enters a parallel region);
    // (3) relabelling in discovery order, sequential by definition (the third dependent random read, the id map).
    // Picks, their order and the relabelling are exactly those of the one-seed-at-a-time loop of sample.cpp.
    constexpr int64_t kAhead = 16;
    out_indptr[0] = 0;
    for (int64_t i = 0; i < batch; ++i) {
        if (i + kAhead < batch) __builtin_prefetch(indptr + node_idx[i + kAhead]);
        const int64_t s = node_idx[i];
        out_indptr[i + 1] = out_indptr[i] + sample_count(indptr[s + 1] - indptr[s], num_neighbors, replace);
    }
    const int64_t n_edges = out_indptr[batch];
    if (n_edges > cap_edges) return COGDL_HOST_ECAP;
    // a thread per ~4096 sampled edges at most; seeds split at equal-edge boundaries
    int t = (int)std::min<int64_t>(std::max(nthreads, 1), std::max<int64_t>(1, n_edges / 4096));
    if (t <= 1) {
        if (!sample_picks(indptr, indices, num_nodes, node_idx, 0, batch, num_neighbors, replace, seed, out_indptr,
                          out_indices, out_edges))
            return COGDL_HOST_ERANGE;
    } else {
        std::vector<int64_t> cut((size_t)t + 1, batch);
        cut[0] = 0;
        for (int k = 1; k < t; ++k)
            cut[(size_t)k] = std::upper_bound(out_indptr, out_indptr + batch + 1, n_edges * k / t) - out_indptr - 1;
        std::vector<char> ok((size_t)t, 1);
        // OpenMP, not std::thread: the caller is a torch process whose OpenMP workers spin for a while after every
        // parallel torch operator -- fresh threads would fight them for the cores (measured: 2 std::threads made a
        // two-hop batch 3x SLOWER inside a torch loop), the OpenMP runtime hands the region to those same workers.
#pragma omp parallel for num_threads(t) schedule(static, 1)
        for (int k = 0; k < t; ++k)
            ok[(size_t)k] = sample_picks(indptr, indices, num_nodes, node_idx, cut[(size_t)k], cut[(size_t)k + 1],
                                         num_neighbors, replace, seed, out_indptr, out_indices, out_edges);
        for (char v : ok)
            if (!v) return COGDL_HOST_ERANGE;
    }
    // local id of every global node: -1 = not seen yet (out_nodes doubles as the list of touched entries)
    LocalIds ids(num_nodes, out_nodes);
    std::vector<int64_t> &local = ids.map;
    for (int64_t i = 0; i < batch; ++i) {
        local[(size_t)node_idx[i]] = i;
        out_nodes[i] = node_idx[i];
    }
    int64_t n_nodes = batch;
    ids.n_touched = batch;
    for (int64_t e = 0; e < n_edges; ++e) {
        if (e + kAhead < n_edges) __builtin_prefetch(&local[(size_t)out_indices[e + kAhead]], 1);
        int64_t &id = local[(size_t)out_indices[e]];
        if (id < 0) {
            if (n_nodes >= cap_nodes) return COGDL_HOST_ECAP;
            id = n_nodes;
            out_nodes[n_nodes++] = out_indices[e];
            ids.n_touched = n_nodes;
        }
        out_indices[e] = id;
    }
    out_counts[0] = n_nodes;
    out_counts[1] = n_edges;
    return COGDL_HOST_OK;
}

int cogdl_host_sample_adj(const int64_t *indptr, const int64_t *indices, int64_t num_nodes, const int64_t *node_idx,
                          int64_t batch, int64_t num_neighbors, int replace, uint64_t seed, int64_t *out_indptr,
                          int64_t *out_indices, int64_t *out_nodes, int64_t *out_edges, int64_t cap_edges,
                          int64_t cap_nodes, int64_t *out_counts) {
    return cogdl_host_sample_adj_mt(indptr, indices, num_nodes, node_idx, batch, num_neighbors, replace, seed, out_indptr,
                                    out_indices, out_nodes, out_edges, cap_edges, cap_nodes, out_counts, 1);
}

int cogdl_host_subgraph(const int64_t *indptr, const int64_t *indices, int64_t num_nodes, const int64_t *node_idx,
                        int64_t batch, int64_t *out_indptr, int64_t *out_indices, int64_t *out_edges,
                        int64_t cap_edges, int64_t *out_counts) {
    if (!indptr || !out_indptr || !out_counts || batch < 0 || num_nodes < 0 || (batch > 0 && !node_idx))
        return COGDL_HOST_EINVAL;
    if (!in_range(node_idx, batch, num_nodes)) return COGDL_HOST_ERANGE;
    LocalIds ids(num_nodes, node_idx);
    std::vector<int64_t> &local = ids.map;
    for (int64_t i = 0; i < batch; ++i) local[(size_t)node_idx[i]] = i;
    ids.n_touched = batch;
    int64_t n_edges = 0;
    out_indptr[0] = 0;
    // Per kept-or-not edge: one random read of the id map.  The neighbour ids themselves are sequential per row, so the
    // map entry of the edge kAhead positions further on is prefetched while this one is tested (and the next rows'
    // extents a few rows ahead): the misses overlap instead of being paid one at a time.
    constexpr int64_t kAhead = 16;
    for (int64_t i = 0; i < batch; ++i) {
        if (i + 4 < batch) {
            __builtin_prefetch(indptr + node_idx[i + 4]);
            if (i + 2 < batch) __builtin_prefetch(indices + indptr[node_idx[i + 2]]);
        }
        const int64_t node = node_idx[i];
        const int64_t row_end = indptr[node + 1];
        for (int64_t e = indptr[node]; e < row_end; ++e) {
            if (e + kAhead < row_end) {
                const int64_t nb = indices[e + kAhead];
                if (nb >= 0 && nb < num_nodes) __builtin_prefetch(&local[(size_t)nb]);
            }
            const int64_t id = local[(size_t)indices[e]];
            if (id < 0) continue;
            if (n_edges >= cap_edges) return COGDL_HOST_ECAP;
            out_indices[n_edges] = id;
            out_edges[n_edges++] = e;
        }
        out_indptr[i + 1] = n_edges;
    }
    out_counts[0] = n_edges;
    return COGDL_HOST_OK;
}

int cogdl_host_csr_spmm_f32(const int32_t *rowptr, const int32_t *colind, const float *val, const float *dense,
                            float *out, int64_t m, int64_t k, int nthreads) {
    return spmm_threads(rowptr, colind, val, dense, out, m, k, nthreads, spmm_rows);
}

int cogdl_host_csr_spmm_f32_i64(const int64_t *rowptr, const int32_t *colind, const float *val, const float *dense,
                                float *out, int64_t m, int64_t k, int nthreads) {
    return spmm_threads(rowptr, colind, val, dense, out, m, k, nthreads, spmm_rows64);
}

}  // extern "C"
