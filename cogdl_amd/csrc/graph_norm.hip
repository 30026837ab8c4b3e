// graph_norm.hip -- graph preprocessing on the GPU: what CogDL's Graph does once per
// graph before the first SpMM, as a handful of HIP kernels instead of ~10 torch kernels each:
//   add_remaining_self_loops   cogdl/utils/graph_utils.py:40-70 (Adjacency.add_remaining_self_loops, data/data.py:175-191):
//                              drop the existing self loops, append one loop per node; a node that had a loop keeps that
//                              loop's weight (the LAST one in COO order, what a sequential index_put leaves), the others
//                              get fill_value.  Edges keep their COO order (stable compaction: scan of keep flags).
//   coo_norm_weights           symmetric_normalization / row_normalization (graph_utils.py:72-89): degrees = number of
//                              edges per row (get_degrees, :10-17), w = d^-1/2[col] * val * d^-1/2[row]  or  val / d[row],
//                              with 1/0 -> 0.  Integer atomics for the histogram (exact, order-independent).
// int64 indices in and out, like the tensors cogdl.data.Graph holds.  All stream-ordered, nothing synchronises.
#include "common.h"

#include "scan.h"

namespace cogdl {

static size_t gn_align256(size_t v) { return (v + 255) / 256 * 256; }

__global__ void gn_degree_kernel(const int64_t *__restrict__ row, int64_t nnz, int64_t n, int32_t *__restrict__ deg,
                                 int *__restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = row[e];
        if (r < 0 || r >= n) atomicOr(bad, 1);
        else atomicAdd(deg + r, 1);
    }
}

// mode 0: sym  w = dinv_sqrt[col] * val * dinv_sqrt[row];  mode 1: row  w = val * dinv[row]
__global__ void gn_apply_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                const float *__restrict__ val, int64_t nnz, int64_t n, const int32_t *__restrict__ deg,
                                int mode, float *__restrict__ out, int *__restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = row[e], c = col[e];
        if (r < 0 || r >= n || c < 0 || c >= n) {
            atomicOr(bad, 1);
            continue;
        }
        const float v = val ? val[e] : 1.f;
        const float dr = (float)deg[r];
        if (mode == 0) {
            const float dc = (float)deg[c];
            const float ir = dr > 0.f ? 1.f / sqrtf(dr) : 0.f, ic = dc > 0.f ? 1.f / sqrtf(dc) : 0.f;
            out[e] = (ic * v) * ir;  // the reference's association: row_sum_inv_sqrt[col] * val * row_sum_inv_sqrt[row]
        } else {
            out[e] = v * (dr > 0.f ? 1.f / dr : 0.f);
        }
    }
}

// ---- add_remaining_self_loops
__global__ void gn_loop_flags_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col, int64_t nnz,
                                     int64_t n, int32_t *__restrict__ keep, int32_t *__restrict__ loop_src,
                                     int *__restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= nnz; e += (int64_t)gridDim.x * blockDim.x) {
        if (e == nnz) {
            keep[e] = 0;  // the scan's last entry = number of kept edges
            continue;
        }
        const int64_t r = row[e], c = col[e];
        if (r < 0 || r >= n || c < 0 || c >= n) {
            atomicOr(bad, 1);
            keep[e] = 0;
            continue;
        }
        keep[e] = r != c ? 1 : 0;
        if (r == c) atomicMax(loop_src + r, (int32_t)e);  // the last loop of a node in COO order supplies its weight
    }
}

__global__ void gn_loop_scatter_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                       const float *__restrict__ val, int64_t nnz, int64_t n,
                                       const int32_t *__restrict__ keep, const int64_t *__restrict__ pos,
                                       const int32_t *__restrict__ loop_src, float fill, int64_t *__restrict__ out_row,
                                       int64_t *__restrict__ out_col, float *__restrict__ out_val,
                                       int64_t *__restrict__ out_count) {
    const int64_t kept = pos[nnz];
    const int64_t total = nnz + n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nnz) {
            if (keep[i]) {
                const int64_t p = pos[i];
                out_row[p] = row[i];
                out_col[p] = col[i];
                out_val[p] = val ? val[i] : 1.f;
            }
        } else {
            const int64_t v = i - nnz;
            const int32_t src = loop_src[v];
            out_row[kept + v] = v;
            out_col[kept + v] = v;
            out_val[kept + v] = (src >= 0 && val) ? val[src] : (src >= 0 ? 1.f : fill);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = kept + n;
}

static unsigned gn_blocks(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 32); }

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_coo_norm_weights_workspace_bytes(int64_t num_nodes) {
    return gn_align256((size_t)std::max<int64_t>(num_nodes, 1) * 4);
}

extern "C" int cogdl_hip_coo_norm_weights(const int64_t *row, const int64_t *col, const float *val, int64_t nnz,
                                          int64_t num_nodes, int mode, float *out, int *bad_flag, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    if (nnz < 0 || num_nodes < 0 || (mode != 0 && mode != 1) || !bad_flag) return COGDL_HIP_EINVAL;
    if (nnz == 0) return COGDL_HIP_OK;
    if (!row || !col || !out || !workspace) return COGDL_HIP_EINVAL;
    if (workspace_bytes < cogdl_hip_coo_norm_weights_workspace_bytes(num_nodes)) return COGDL_HIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int32_t *deg = (int32_t *)workspace;
    hipError_t e = fill_u32_async(deg, 0u, (size_t)std::max<int64_t>(num_nodes, 1), s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    hipLaunchKernelGGL(gn_degree_kernel, dim3(gn_blocks(nnz)), dim3(256), 0, s, row, nnz, num_nodes, deg, bad_flag);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gn_blocks(nnz)), dim3(256), 0, s, row, col, val, nnz, num_nodes, deg, mode, out,
                       bad_flag);
    return launch_status();
}

extern "C" size_t cogdl_hip_add_remaining_self_loops_workspace_bytes(int64_t nnz, int64_t num_nodes) {
    const size_t scan_t = device_scan_temp_bytes(nnz + 1, sizeof(int64_t));
    return gn_align256((size_t)(nnz + 1) * 4) + gn_align256((size_t)(nnz + 1) * 8) +
           gn_align256((size_t)std::max<int64_t>(num_nodes, 1) * 4) + gn_align256(scan_t) + 256;
}

extern "C" int cogdl_hip_add_remaining_self_loops(const int64_t *row, const int64_t *col, const float *val, int64_t nnz,
                                                  int64_t num_nodes, float fill_value, int64_t *out_row,
                                                  int64_t *out_col, float *out_val, int64_t *out_count, int *bad_flag,
                                                  void *workspace, size_t workspace_bytes, void *stream) {
    if (nnz < 0 || num_nodes < 0 || !out_count || !bad_flag || !workspace) return COGDL_HIP_EINVAL;
    if (nnz > 0 && (!row || !col)) return COGDL_HIP_EINVAL;
    if (nnz + num_nodes > 0 && (!out_row || !out_col || !out_val)) return COGDL_HIP_EINVAL;
    if (nnz > 0x7ffffffe || num_nodes > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (workspace_bytes < cogdl_hip_add_remaining_self_loops_workspace_bytes(nnz, num_nodes)) return COGDL_HIP_EWORKSPACE;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)workspace;
    int32_t *keep = (int32_t *)p;
    p += gn_align256((size_t)(nnz + 1) * 4);
    int64_t *pos = (int64_t *)p;
    p += gn_align256((size_t)(nnz + 1) * 8);
    int32_t *loop_src = (int32_t *)p;
    p += gn_align256((size_t)std::max<int64_t>(num_nodes, 1) * 4);
    hipError_t e = fill_u32_async(loop_src, 0xffffffffu, (size_t)std::max<int64_t>(num_nodes, 1), s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    hipLaunchKernelGGL(gn_loop_flags_kernel, dim3(gn_blocks(nnz + 1)), dim3(256), 0, s, row, col, nnz, num_nodes, keep,
                       loop_src, bad_flag);
    const int rc = device_exclusive_sum(keep, pos, nnz + 1, (void *)p, s);
    if (rc != COGDL_HIP_OK) return rc;
    hipLaunchKernelGGL(gn_loop_scatter_kernel, dim3(gn_blocks(nnz + num_nodes)), dim3(256), 0, s, row, col, val, nnz,
                       num_nodes, keep, pos, loop_src, fill_value, out_row, out_col, out_val, out_count);
    return launch_status();
}

// ---- a sampled block as the SpMM wants it, in ONE launch ------------------------------------------------------------
// The sampler speaks int64 (the reference's contract); the SpMM wants int32 indices, and the mean aggregator
// 1 / in-degree per row (Graph.row_norm, cogdl/data/data.py:240-258: 1/0 -> 0; a row without edges aggregates nothing
// either way).  Eager torch needs ~7 short kernels per block for this (two .int() copies, a slice copy, the degree
// difference, clamp, float, reciprocal) -- in a captured mini-batch step every one of them is a dependent graph node.
namespace cogdl {
__global__ void block_prepare_kernel(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ col, int64_t n_rows,
                                     int64_t n_slots, int32_t *__restrict__ rowptr32, int32_t *__restrict__ col32,
                                     float *__restrict__ inv_deg) {
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t0; i < n_slots; i += stride) col32[i] = (int32_t)col[i];
    for (int64_t i = t0; i <= n_rows; i += stride) {
        const int64_t a = row_ptr[i];
        rowptr32[i] = (int32_t)a;
        if (inv_deg && i < n_rows) {
            const int64_t d = row_ptr[i + 1] - a;
            inv_deg[i] = d > 0 ? 1.f / (float)d : 0.f;
        }
    }
}
}  // namespace cogdl

extern "C" int cogdl_hip_block_prepare(const int64_t *row_ptr, const int64_t *col, int64_t n_rows, int64_t n_slots,
                                       int32_t *rowptr32, int32_t *col32, float *inv_deg, void *stream) {
    if (n_rows < 0 || n_slots < 0 || !row_ptr || !rowptr32) return COGDL_HIP_EINVAL;
    if (n_slots > 0 && (!col || !col32)) return COGDL_HIP_EINVAL;
    if (n_slots > 0x7fffffff || n_rows > 0x7ffffffe) return COGDL_HIP_ERANGE;
    const int64_t work = std::max(n_slots, n_rows + 1);
    hipLaunchKernelGGL(block_prepare_kernel, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, row_ptr, col, n_rows, n_slots, rowptr32, col32, inv_deg);
    return launch_status();
}
