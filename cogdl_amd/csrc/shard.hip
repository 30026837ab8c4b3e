// shard.hip -- building one rank's shard of a 1-D vertex-partitioned CSR matrix on the GPU, and the breadth-first
// levels behind the locality reordering (BASELINE.json configs[4]: papers100M across 8 GPUs; SURVEY.md sections 7 / 8e).
// The reference has no counterpart: CogDL never partitions a graph across devices; its nearest relative is the METIS
// partitioning of ClusteredDataset (cogdl/data/sampler.py:188-243), which runs on the host.
//
// Rank p owns the rows [lo, hi) of A (all their incoming edges, GLOBAL column ids).  Its shard is
//     A_loc : columns inside [lo, hi), relabelled col - lo                        -> int32 CSR (+ weights)
//     A_rem : all other columns, relabelled into the sorted-unique HALO table     -> int32 CSR (+ weights)
//     halo  : the distinct remote column ids in ascending order (= grouped by owner: owners are contiguous id ranges)
// -- round 1-2 built this with torch glue (repeat_interleave, boolean masks, unique, searchsorted: several int64
// copies of a 4e8-entry edge list, 15-20 GB transient per shard).  Here: two passes over the edges, int32 scratch.
//   count : one WAVE per row counts its local / remote edges (ballots) and marks the remote columns in a table over
//           the global id space (plain stores of 1: every writer writes the same value); three exclusive scans
//           (rocPRIM) give the two row pointers and, over the marks, every remote column's halo index
//   fill  : one wave per row compacts its edges into the two blocks (ballot + prefix popcount: CSR order kept), remote
//           columns replaced by their halo index; one pass over the marks writes the halo table
// Output sizes are known after `count` (the host reads three numbers, once per shard -- building a shard is not a
// per-step operation), so the caller allocates exactly what `fill` writes.  Nothing else synchronises; errors are
// status codes, a column id outside [0, n_global) is reported through the counts' flag word.
#include "common.h"

#include "scan.h"

namespace cogdl {

static size_t sh_align256(size_t v) { return (v + 255) / 256 * 256; }

struct ShardWs {
    int32_t *cnt_loc, *cnt_rem;  // [n_local + 1] counts, then (in place after the scans) exclusive prefixes
    int32_t *mark;               // [n_global + 1] 1 = some edge of the shard has this remote column; then its scan
    char *temp;
    size_t temp_bytes, total;
};

static ShardWs shard_carve(void *base, int64_t n_local, int64_t n_global) {
    ShardWs w{};
    w.temp_bytes = std::max(device_scan_temp_bytes(n_local + 1, sizeof(int32_t)), device_scan_temp_bytes(n_global + 1, sizeof(int32_t)));
    char *p = (char *)base;
    auto take = [&](size_t bytes) {
        char *q = p;
        p += sh_align256(bytes);
        return q;
    };
    w.cnt_loc = (int32_t *)take((size_t)(n_local + 1) * 4);
    w.cnt_rem = (int32_t *)take((size_t)(n_local + 1) * 4);
    w.mark = (int32_t *)take((size_t)(n_global + 1) * 4);
    w.temp = take(w.temp_bytes);
    w.total = (size_t)(p - (char *)base) + 256;
    return w;
}

// One wave per row.  FILL = false: counts + marks; FILL = true: the two blocks.
template <bool FILL, typename W>
__global__ __launch_bounds__(256) void shard_rows_kernel(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
                                                         const W *__restrict__ weight, int64_t n_local, int64_t lo, int64_t hi,
                                                         int64_t n_global, int32_t *__restrict__ cnt_loc,
                                                         int32_t *__restrict__ cnt_rem, int32_t *__restrict__ mark,
                                                         int32_t *__restrict__ colind_loc, W *__restrict__ w_loc,
                                                         int32_t *__restrict__ colind_rem, W *__restrict__ w_rem,
                                                         int *__restrict__ flags, int64_t nnz) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_local) return;
    int64_t start = rowptr[r] - rowptr[0], end = rowptr[r + 1] - rowptr[0];  // (rowptr may be a slice of a global one)
    // The caller's row pointer is validated HERE, before anything is read through it: a row that runs backwards or out
    // of col[0, nnz), or a last row that does not end at nnz, raises flag bit 2 and is treated as empty (both passes
    // agree, so the counts and the fill stay consistent); the host raises on the flag word.
    if (start < 0 || end < start || end > nnz || (r == n_local - 1 && end != nnz)) {
        if (lane == 0) atomicOr(flags, 2);
        start = end = 0;
    }
    int64_t at_loc = FILL ? cnt_loc[r] : 0, at_rem = FILL ? cnt_rem[r] : 0;        // (FILL: the scanned counts)
    int32_t n_loc = 0, n_rem = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int64_t base = start; base < end; base += kWave) {
        const int64_t e = base + lane;
        int64_t c = -1;
        if (e < end) {
            c = col[e];
            if (c < 0 || c >= n_global) {
                atomicOr(flags, 1);
                c = -1;  // (dropped from both blocks; the caller raises on the flag)
            }
        }
        const bool is_loc = c >= lo && c < hi, is_rem = c >= 0 && !is_loc;
        const unsigned long long bl = __ballot(is_loc), br = __ballot(is_rem);
        if constexpr (FILL) {
            if (is_loc) {
                const int64_t pos = at_loc + __popcll(bl & lt);
                colind_loc[pos] = (int32_t)(c - lo);
                if (weight) w_loc[pos] = weight[e];
            } else if (is_rem) {
                const int64_t pos = at_rem + __popcll(br & lt);
                colind_rem[pos] = mark[c];  // (scanned: the halo index of column c)
                if (weight) w_rem[pos] = weight[e];
            }
            at_loc += __popcll(bl);
            at_rem += __popcll(br);
        } else {
            if (is_rem) mark[c] = 1;
            n_loc += __popcll(bl);
            n_rem += __popcll(br);
        }
    }
    if (!FILL && lane == 0) {
        cnt_loc[r] = n_loc;
        cnt_rem[r] = n_rem;
    }
}

__global__ void shard_copy2_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int32_t *__restrict__ oa,
                                   int32_t *__restrict__ ob, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        oa[i] = a[i];
        ob[i] = b[i];
    }
}

__global__ void shard_counts_kernel(const int32_t *__restrict__ cnt_loc, const int32_t *__restrict__ cnt_rem,
                                    const int32_t *__restrict__ mark, int64_t n_local, int64_t n_global,
                                    const int *__restrict__ flags, int64_t *__restrict__ counts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        counts[0] = cnt_loc[n_local];
        counts[1] = cnt_rem[n_local];
        counts[2] = mark[n_global];
        counts[3] = *flags;
    }
}

// halo_ids[mark_scan[v]] = v for the marked v (detected by a step in the scan); cut[q] = marked ids below bounds[q]
__global__ void shard_halo_kernel(const int32_t *__restrict__ scan, int64_t n_global, int64_t *__restrict__ halo_ids,
                                  const int64_t *__restrict__ bounds, int64_t n_bounds, int64_t *__restrict__ cut) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = tid; v < n_global; v += stride)
        if (scan[v + 1] != scan[v]) halo_ids[scan[v]] = v;
    for (int64_t q = tid; q < n_bounds; q += stride) {
        const int64_t b = bounds[q] < 0 ? 0 : (bounds[q] > n_global ? n_global : bounds[q]);
        cut[q] = scan[b];
    }
}

// ---- breadth-first levels: one level per launch (topology driven: every vertex at the current level relaxes its row).
// level[v] = -1 unvisited.  All writers of a vertex write the same value, `changed` is a flag: no atomics needed.
__global__ __launch_bounds__(256) void bfs_step_kernel(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
                                                       int64_t n, int32_t *__restrict__ level, int32_t cur,
                                                       int *__restrict__ changed) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // a wave takes 64 vertices at a time; each frontier vertex among them is then expanded by the whole wave
    for (int64_t v0 = wave0 * kWave; v0 < n; v0 += n_waves * kWave) {
        const int64_t v = v0 + lane;
        const bool front = v < n && level[v] == cur;
        unsigned long long todo = __ballot(front);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t u = v0 + src;
            const int64_t s = rowptr[u], e = rowptr[u + 1];
            for (int64_t k = s + lane; k < e; k += kWave) {
                const int64_t w = col[k];
                if (w >= 0 && w < n && level[w] < 0) {
                    level[w] = cur + 1;
                    *changed = 1;
                }
            }
        }
    }
}

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_shard_workspace_bytes(int64_t n_local, int64_t n_global) {
    if (n_local < 0 || n_global < 0) return 0;
    return shard_carve(nullptr, n_local, n_global).total;
}

static int shard_check(const int64_t *rowptr, const int64_t *col, int64_t n_local, int64_t nnz, int64_t lo, int64_t hi,
                       int64_t n_global, void *ws, size_t wsb) {
    if (n_local < 0 || nnz < 0 || n_global < 0 || lo < 0 || hi < lo || hi > n_global || hi - lo != n_local) return COGDL_HIP_EINVAL;
    if (!rowptr || (nnz > 0 && !col) || !ws) return COGDL_HIP_EINVAL;
    if (nnz > 0x7fffffff || n_global > 0x7ffffffe || n_local > 0x7ffffffe) return COGDL_HIP_ERANGE;
    if (!aligned_to(ws, 256)) return COGDL_HIP_EALIGN;
    if (wsb < cogdl_hip_shard_workspace_bytes(n_local, n_global)) return COGDL_HIP_EWORKSPACE;
    return COGDL_HIP_OK;
}

extern "C" int cogdl_hip_shard_count(const int64_t *rowptr, const int64_t *col, int64_t n_local, int64_t nnz, int64_t lo,
                                     int64_t hi, int64_t n_global, int64_t *counts, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    int rc = shard_check(rowptr, col, n_local, nnz, lo, hi, n_global, workspace, workspace_bytes);
    if (rc != COGDL_HIP_OK) return rc;
    if (!counts) return COGDL_HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const ShardWs w = shard_carve(workspace, n_local, n_global);
    int *flags = (int *)((char *)workspace + w.total - 256);
    auto fail = [](hipError_t e) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    };
    hipError_t e = fill_u32_async(flags, 0u, 1, s);
    if (e == hipSuccess) e = fill_u32_async(w.mark, 0u, (size_t)n_global + 1, s);
    if (e == hipSuccess) e = fill_u32_async(w.cnt_loc + n_local, 0u, 1, s);
    if (e == hipSuccess) e = fill_u32_async(w.cnt_rem + n_local, 0u, 1, s);
    if (e != hipSuccess) return fail(e);
    if (n_local > 0)
        hipLaunchKernelGGL((shard_rows_kernel<false, float>), dim3((unsigned)((n_local + 3) / 4)), dim3(256), 0, s, rowptr, col,
                           (const float *)nullptr, n_local, lo, hi, n_global, w.cnt_loc, w.cnt_rem, w.mark, (int32_t *)nullptr,
                           (float *)nullptr, (int32_t *)nullptr, (float *)nullptr, flags, nnz);
    // (in place: scan.h allows out == in)
    int rc_scan = device_exclusive_sum((const int32_t *)w.cnt_loc, w.cnt_loc, n_local + 1, w.temp, s);
    if (rc_scan == COGDL_HIP_OK) rc_scan = device_exclusive_sum((const int32_t *)w.cnt_rem, w.cnt_rem, n_local + 1, w.temp, s);
    if (rc_scan == COGDL_HIP_OK) rc_scan = device_exclusive_sum((const int32_t *)w.mark, w.mark, n_global + 1, w.temp, s);
    if (rc_scan != COGDL_HIP_OK) return rc_scan;
    hipLaunchKernelGGL(shard_counts_kernel, dim3(1), dim3(64), 0, s, w.cnt_loc, w.cnt_rem, w.mark, n_local, n_global, flags, counts);
    return launch_status();
}

extern "C" int cogdl_hip_shard_fill(const int64_t *rowptr, const int64_t *col, const float *weight, int64_t n_local,
                                    int64_t nnz, int64_t lo, int64_t hi, int64_t n_global, const int64_t *bounds,
                                    int64_t n_bounds, int32_t *rowptr_loc, int32_t *colind_loc, float *w_loc,
                                    int32_t *rowptr_rem, int32_t *colind_rem, float *w_rem, int64_t *halo_ids, int64_t *cut,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    int rc = shard_check(rowptr, col, n_local, nnz, lo, hi, n_global, workspace, workspace_bytes);
    if (rc != COGDL_HIP_OK) return rc;
    if (!rowptr_loc || !rowptr_rem || (n_bounds > 0 && (!bounds || !cut))) return COGDL_HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const ShardWs w = shard_carve(workspace, n_local, n_global);
    int *flags = (int *)((char *)workspace + w.total - 256);
    const unsigned cb = (unsigned)std::min<int64_t>((n_local + 256) / 256, 1 << 16);
    hipLaunchKernelGGL(shard_copy2_kernel, dim3(cb), dim3(256), 0, s, w.cnt_loc, w.cnt_rem, rowptr_loc, rowptr_rem, n_local + 1);
    if (n_local > 0)
        hipLaunchKernelGGL((shard_rows_kernel<true, float>), dim3((unsigned)((n_local + 3) / 4)), dim3(256), 0, s, rowptr, col,
                           weight, n_local, lo, hi, n_global, w.cnt_loc, w.cnt_rem, w.mark, colind_loc, w_loc, colind_rem, w_rem,
                           flags, nnz);
    const unsigned blocks = (unsigned)std::min<int64_t>((std::max(n_global, n_bounds) + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(shard_halo_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, s, w.mark, n_global, halo_ids, bounds,
                       n_bounds, cut);
    return launch_status();
}

extern "C" int cogdl_hip_bfs_step(const int64_t *rowptr, const int64_t *col, int64_t n, int32_t *level, int32_t cur,
                                  int *changed, void *stream) {
    if (n < 0 || cur < 0) return COGDL_HIP_EINVAL;
    if (n == 0) return COGDL_HIP_OK;
    if (!rowptr || !col || !level || !changed) return COGDL_HIP_EINVAL;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(bfs_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, n, level, cur, changed);
    return launch_status();
}
