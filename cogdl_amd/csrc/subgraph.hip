// subgraph.hip -- node-induced subgraph of a GPU-resident CSR graph on gfx950.
// Same contract as sampler.subgraph (cogdl/operators/sample/sample.cpp:146-188), which the reference runs
// single-threaded on the CPU (and which cogdl/data/data.py:850-872 reaches after moving node_idx to the CPU):
//   row i of the result = row node_idx[i] of the graph restricted to the sources that are themselves in node_idx,
//   relabelled to their position in node_idx; a row's edges keep their CSR order; out_edges = the CSR positions.
// Pipeline (stream-ordered, no host round trip, no atomics on the outputs => deterministic):
//   1. assoc[num_nodes] = -1, then assoc[node_idx[i]] = i          (a duplicate id keeps its LAST position, like the
//                                                                    reference's index_copy_ on one thread)
//   2. one WAVE per result row counts the kept edges               -> exclusive scan -> out_indptr
//   3. one WAVE per result row writes them: every 64-edge chunk is compacted with a ballot / prefix popcount, so the
//      edges of a row come out in CSR order.
// Integer work, bound by the gather of assoc[] (4 bytes per edge of the selected rows); int64 in and out.
#include "common.h"

#include "scan.h"

namespace cogdl {

static size_t sg_align256(size_t v) { return (v + 255) / 256 * 256; }

// flags: bit 0 = a node id outside [0, num_nodes), bit 1 = a neighbour id outside, bit 2 = output capacity exceeded
__global__ void subgraph_assoc_kernel(const int64_t *__restrict__ node_idx, int64_t batch, int64_t num_nodes,
                                      int32_t *__restrict__ assoc, int *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const int64_t s = node_idx[i];
    if (s < 0 || s >= num_nodes) atomicOr(flags, 1);
    else atomicMax(assoc + s, (int32_t)i);  // duplicates: the last position wins, whatever the thread order
}

template <bool WRITE>
__global__ __launch_bounds__(256) void subgraph_rows_kernel(const int64_t *__restrict__ indptr,
                                                            const int64_t *__restrict__ indices,
                                                            const int64_t *__restrict__ node_idx, int64_t batch,
                                                            int64_t num_nodes, const int32_t *__restrict__ assoc,
                                                            int32_t *__restrict__ cnt,
                                                            const int64_t *__restrict__ out_indptr,
                                                            int64_t *__restrict__ out_indices,
                                                            int64_t *__restrict__ out_edges, int64_t cap_edges,
                                                            int *__restrict__ flags) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i > batch) return;
    if (i == batch) {  // cnt[batch] = 0: the scan then also yields the total
        if (!WRITE && lane == 0) cnt[batch] = 0;
        return;
    }
    const int64_t s = node_idx[i];
    int64_t start = 0, end = 0;
    if (s >= 0 && s < num_nodes) {
        start = indptr[s];
        end = indptr[s + 1];
    }
    int64_t written = WRITE ? out_indptr[i] : 0;
    int32_t total = 0;
    for (int64_t base = start; base < end; base += kWave) {
        const int64_t k = base + lane;
        int32_t local = -1;
        if (k < end) {
            const int64_t nb = indices[k];
            if (nb < 0 || nb >= num_nodes) atomicOr(flags, 2);
            else local = assoc[nb];
        }
        const unsigned long long keep = __ballot(local >= 0);
        if constexpr (WRITE) {
            if (local >= 0) {
                const int64_t pos = written + __popcll(keep & ((1ull << lane) - 1ull));
                if (pos >= cap_edges) {
                    atomicOr(flags, 4);
                } else {
                    out_indices[pos] = local;
                    out_edges[pos] = k;
                }
            }
            written += __popcll(keep);
        } else {
            total += __popcll(keep);
        }
    }
    if (!WRITE && lane == 0) cnt[i] = total;
}

__global__ void subgraph_counts_kernel(const int64_t *__restrict__ out_indptr, int64_t batch, int64_t cap_edges,
                                       const int *__restrict__ flags, int64_t *__restrict__ out_counts) {
    out_counts[0] = out_indptr[batch];
    out_counts[1] = *flags | (out_indptr[batch] > cap_edges ? 4 : 0);
}

struct SubgraphWs {
    int *flags;
    int32_t *assoc, *cnt;
    char *temp;
    size_t temp_bytes, total;
};

static SubgraphWs sg_carve(void *base, int64_t batch, int64_t num_nodes) {
    SubgraphWs w{};
    w.temp_bytes = device_scan_temp_bytes(batch + 1, sizeof(int64_t));
    char *p = (char *)base;
    auto take = [&](size_t bytes) {
        char *q = p;
        p += sg_align256(bytes);
        return q;
    };
    w.flags = (int *)take(256);
    w.assoc = (int32_t *)take((size_t)std::max<int64_t>(num_nodes, 1) * 4);
    w.cnt = (int32_t *)take((size_t)(batch + 1) * 4);
    w.temp = take(w.temp_bytes);
    w.total = (size_t)(p - (char *)base);
    return w;
}

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_subgraph_workspace_bytes(int64_t batch, int64_t num_nodes) {
    if (batch < 0 || num_nodes < 0) return 0;
    return sg_carve(nullptr, batch, num_nodes).total;
}

extern "C" int cogdl_hip_subgraph(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                  const int64_t *node_idx, int64_t batch, int64_t *out_indptr, int64_t *out_indices,
                                  int64_t *out_edges, int64_t cap_edges, int64_t *out_counts, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    if (batch < 0 || cap_edges < 0 || num_nodes < 0 || !out_indptr || !out_counts || !workspace) return COGDL_HIP_EINVAL;
    if (batch > 0 && (!indptr || !node_idx)) return COGDL_HIP_EINVAL;
    if (cap_edges > 0 && (!indices || !out_indices || !out_edges)) return COGDL_HIP_EINVAL;
    if (num_nodes > 0x7fffffff || batch > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    const SubgraphWs w = sg_carve(workspace, batch, num_nodes);
    if (workspace_bytes < w.total) return COGDL_HIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    auto fail = [](hipError_t e) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    };
    hipError_t e = fill_u32_async(w.flags, 0u, 1, s);
    if (e != hipSuccess) return fail(e);
    e = fill_u32_async(w.assoc, 0xffffffffu, (size_t)std::max<int64_t>(num_nodes, 1), s);  // -1 everywhere
    if (e != hipSuccess) return fail(e);
    if (batch > 0)
        hipLaunchKernelGGL(subgraph_assoc_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, s, node_idx, batch,
                           num_nodes, w.assoc, w.flags);
    const unsigned row_blocks = (unsigned)((batch + 1 + 3) / 4);
    hipLaunchKernelGGL((subgraph_rows_kernel<false>), dim3(row_blocks), dim3(256), 0, s, indptr, indices, node_idx, batch,
                       num_nodes, w.assoc, w.cnt, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr,
                       cap_edges, w.flags);
    const int rc_scan = device_exclusive_sum(w.cnt, out_indptr, batch + 1, w.temp, s);
    if (rc_scan != COGDL_HIP_OK) return rc_scan;
    if (batch > 0 && cap_edges > 0)
        hipLaunchKernelGGL((subgraph_rows_kernel<true>), dim3(row_blocks), dim3(256), 0, s, indptr, indices, node_idx, batch,
                           num_nodes, w.assoc, w.cnt, out_indptr, out_indices, out_edges, cap_edges, w.flags);
    hipLaunchKernelGGL(subgraph_counts_kernel, dim3(1), dim3(1), 0, s, out_indptr, batch, cap_edges, w.flags, out_counts);
    return launch_status();
}
