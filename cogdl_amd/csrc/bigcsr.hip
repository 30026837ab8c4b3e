// bigcsr.hip -- CSR structures with 64-bit row pointers (graphs of 2^31 edges and more) on gfx950.
//
// The reference cannot represent such a graph at all: its dispatcher casts the row pointers to int32
// (cogdl/utils/spmm_utils.py:106), csr_spmm_cpu walks `int` edge offsets and even `int ik = i * k`
// (cogdl/operators/spmm/spmm_cpu.cpp:24-33: overflows from 16.7 M rows at k = 128), and the CUDA kernels take `int nnz`.
// ogbn-papers100M as CogDL feeds it to GCN -- symmetrised and coalesced, cogdl/datasets/ogb.py:50-55 -- has 3.2e9 edges.
//
// Design: the row-wise kernels keep their 32-bit inner loops (edge offsets in `int`, one VGPR each: a 64-bit engine
// would cost every operator registers for the one graph size that needs it).  The rows are cut into SEGMENTS of
// about 2^29 edges (tuning key 15); a segment is an ordinary launch on a REBASED int32 copy of its row pointers with the
// colind / val / out pointers advanced to the segment's first edge / row -- 6 launches of ~0.1 s each for the
// symmetrised papers100M graph.  Column ids stay int32 (node ids < 2^31); every row * width offset is 64-bit already.
//
//   cogdl_hip_csr_segments        the cuts (binary searches of rowptr64 on the device, read back: plan time, synchronises)
//   cogdl_hip_csr_rebase_rowptr   the int32 row pointers of all segments, one launch
//   cogdl_hip_csr2csc_i64         stable transpose: every row segment is transposed by the 32-bit csr2csc (twice: once
//                                 for the column counts, once for the entries) and merged into the 64-bit CSC -- the
//                                 entries of column c from segment s land behind those of the segments before it, so the
//                                 order inside a column stays ascending CSR position (= ascending row), as cogdl_hip_csr2csc.
#include "common.h"

namespace cogdl {

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// cut j = the last row r with rowptr[r] <= j * max_edges (j = 0 .. n_cuts; cut 0 = row 0, cut n_cuts = row m)
__global__ void segment_cuts_kernel(const int64_t *__restrict__ rowptr, int64_t m, int64_t max_edges, int n_cuts,
                                    int64_t *__restrict__ rows, int64_t *__restrict__ edges) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_cuts) return;
    int64_t r;
    if (j == 0) r = 0;
    else if (j == n_cuts) r = m;
    else {
        const int64_t target = (int64_t)j * max_edges;
        int64_t lo = 0, hi = m + 1;  // invariant: rowptr[lo] <= target, (hi == m + 1 or rowptr[hi] > target)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (rowptr[mid] <= target) lo = mid; else hi = mid;
        }
        r = lo;
    }
    rows[j] = r;
    edges[j] = rowptr[r];
}

// A row pointer that decreases anywhere (or does not start at 0: segment_cuts_kernel stores rowptr[0] in edges[0]) makes
// the rebased 32-bit pointers negative or wrapped and the 32-bit kernels would index out of bounds: edges[0] = -1 marks it
// (plan time, one pass over rowptr; the host rejects edges[0] != 0).
__global__ void rowptr_monotone_kernel(const int64_t *__restrict__ rowptr, int64_t m, int64_t *__restrict__ edges) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x)
        bad |= rowptr[r + 1] < rowptr[r];
    if (bad) edges[0] = -1;
}

struct SegTable {
    int n;
    int64_t row[COGDL_HIP_MAX_SEGMENTS + 1];
    int64_t edge[COGDL_HIP_MAX_SEGMENTS + 1];
};

// out[row[s] + s + i] = rowptr[row[s] + i] - edge[s],  i = 0 .. rows of segment s (inclusive): m + n entries
__global__ void rebase_rowptr_kernel(const int64_t *__restrict__ rowptr, const SegTable t, int32_t *__restrict__ out) {
    const int64_t total = t.row[t.n] + t.n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int s = 0;
        while (s + 1 < t.n && i >= t.row[s + 1] + s + 1) ++s;
        const int64_t r = i - s;  // global row (row[s] <= r <= row[s + 1])
        out[i] = (int32_t)(rowptr[r] - t.edge[s]);
    }
}

// ---- transpose ------------------------------------------------------------------------------------------------------
// cnt[colind[e]] += 1 over ALL edges (64-bit atomics resolved in L2; ids outside [0, n_cols) are left to the segment
// transposes of pass 2, which reject them)
__global__ void count_columns_kernel(const int32_t *__restrict__ colind, int64_t nnz, int64_t n_cols,
                                     int64_t *__restrict__ cnt) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = colind[e];
        if (c >= 0 && c < n_cols) atomicAdd((unsigned long long *)(cnt + c), 1ull);
    }
}

// Exclusive scan of int64 in three launches (block sums, their scan by one workgroup, the blocks again): plan-time code,
// ~2 passes over n_cols * 8 bytes.
constexpr int kScanBlock = 256, kScanPer = 16, kScanTile = kScanBlock * kScanPer;

__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t *total) {  // 256 threads
    __shared__ int64_t wsum[4];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    int64_t x = v;
#pragma unroll
    for (int s = 1; s < kWave; s <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)x, s, kWave), hi = __shfl_up((uint32_t)((uint64_t)x >> 32), s, kWave);
        const int64_t y = (int64_t)(((uint64_t)hi << 32) | lo);
        if (lane >= s) x += y;
    }
    __syncthreads();  // (wsum may still be read by a previous call)
    if (lane == kWave - 1) wsum[w] = x;
    __syncthreads();
    int64_t base = 0;
    for (int q = 0; q < w; ++q) base += wsum[q];
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + x - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(const int64_t *__restrict__ in, int64_t n,
                                                                      int64_t *__restrict__ sums) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    int64_t acc = 0;
    for (int i = 0; i < kScanPer; ++i) {
        const int64_t idx = base + (int64_t)i * kScanBlock + threadIdx.x;
        if (idx < n) acc += in[idx];
    }
    int64_t total;
    (void)block_exclusive_scan(acc, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlock) void scan_sums_kernel(int64_t *__restrict__ sums, int64_t nb) {  // one workgroup
    int64_t carry = 0;
    for (int64_t base = 0; base < nb; base += kScanBlock) {
        const int64_t idx = base + threadIdx.x;
        const int64_t v = idx < nb ? sums[idx] : 0;
        int64_t total;
        const int64_t ex = block_exclusive_scan(v, &total);
        if (idx < nb) sums[idx] = carry + ex;
        carry += total;
    }
}

// out[i] = sums[block] + exclusive prefix inside the block; out[n] = the grand total (thread-contiguous runs of kScanPer)
// (`in` and `out` may be the same array: every thread reads its own entries before it writes them)
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const int64_t *in, int64_t n, const int64_t *__restrict__ sums,
                                                                 int64_t *out) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    int64_t v[kScanPer], acc = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        acc += v[i];
    }
    int64_t run = sums[blockIdx.x] + block_exclusive_scan(acc, nullptr);
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
        if (base + i == n - 1) out[n] = run;
    }
}

// One workgroup per run of 256 columns: the segment's entries of those columns are one contiguous range of its CSC;
// entry j of column c goes to colptr64[c] + prior[c] + (j - colptr_s[c]).  prior[c] (entries of the segments before) is
// advanced for the next segment.
template <typename V>
__global__ __launch_bounds__(256) void merge_segment_kernel(const int32_t *__restrict__ colptr_s,
                                                            const int32_t *__restrict__ rowind_s,
                                                            const int32_t *__restrict__ perm_s, int64_t n_cols,
                                                            int64_t row0, int64_t edge0,
                                                            const int64_t *__restrict__ colptr64,
                                                            int64_t *__restrict__ prior, int32_t *__restrict__ rowind,
                                                            int64_t *__restrict__ perm64, const V *__restrict__ val,
                                                            V *__restrict__ val_t) {
    __shared__ int32_t cp[257];
    __shared__ int64_t off[256];
    const int64_t c0 = (int64_t)blockIdx.x * 256;
    const int nc = (int)min((int64_t)256, n_cols - c0);
    for (int i = threadIdx.x; i <= nc; i += 256) cp[i] = colptr_s[c0 + i];  // nc + 1 <= 257 entries
    __syncthreads();
    if ((int)threadIdx.x < nc) {
        const int64_t c = c0 + threadIdx.x;
        const int64_t p = prior[c];
        off[threadIdx.x] = colptr64[c] + p - (int64_t)cp[threadIdx.x];
        prior[c] = p + (int64_t)(cp[threadIdx.x + 1] - cp[threadIdx.x]);
    }
    __syncthreads();
    const int begin = cp[0], end = cp[nc];
    for (int j = begin + (int)threadIdx.x; j < end; j += 256) {
        int lo = 0, hi = nc;  // the column with cp[lo] <= j < cp[lo + 1] (empty columns share their successor's offset)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cp[mid] <= j) lo = mid; else hi = mid;
        }
        const int64_t dst = (int64_t)j + off[lo];
        const int32_t e = perm_s[j];
        rowind[dst] = (int32_t)(rowind_s[j] + row0);
        if (perm64) perm64[dst] = edge0 + e;
        if (val_t) val_t[dst] = val[edge0 + e];
    }
}

template <typename E>
__global__ void gather_rows_i64_kernel(const int64_t *__restrict__ perm, const E *__restrict__ src, E *__restrict__ out,
                                       int64_t total, int64_t h) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / h, c = idx - i * h;
        out[idx] = src[perm[i] * h + c];
    }
}

static SegTable table_of(const cogdl_hip_segments *seg) {
    SegTable t{};
    t.n = seg->n;
    for (int i = 0; i <= seg->n; ++i) {
        t.row[i] = seg->row[i];
        t.edge[i] = seg->edge[i];
    }
    return t;
}

int segments_valid(const cogdl_hip_segments *seg) {
    if (!seg || seg->n < 0 || seg->n > COGDL_HIP_MAX_SEGMENTS) return COGDL_HIP_EINVAL;
    for (int s = 0; s < seg->n; ++s) {
        if (seg->row[s + 1] < seg->row[s] || seg->edge[s + 1] < seg->edge[s]) return COGDL_HIP_EINVAL;
        if (seg->edge[s + 1] - seg->edge[s] > COGDL_HIP_SEGMENT_MAX_EDGES || seg->row[s + 1] - seg->row[s] > 0x7fff0000)
            return COGDL_HIP_ERANGE;
    }
    return COGDL_HIP_OK;
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_csr_segments(const int64_t *rowptr, int64_t m, int64_t nnz, int64_t max_edges,
                                      cogdl_hip_segments *out, void *scratch, void *stream) {
    if (!out || m < 0 || nnz < 0 || max_edges < 0) return COGDL_HIP_EINVAL;
    out->n = 0;
    out->row[0] = 0;
    out->edge[0] = 0;
    if (m == 0) return COGDL_HIP_OK;
    if (!rowptr || !scratch) return COGDL_HIP_EINVAL;
    if (max_edges == 0) max_edges = g_tuning[kTuneSegmentEdges] > 0 ? g_tuning[kTuneSegmentEdges] : (int64_t)1 << 29;
    if (max_edges > ((int64_t)1 << 30)) max_edges = (int64_t)1 << 30;
    int64_t n_cuts = std::max<int64_t>(1, (nnz + max_edges - 1) / max_edges);
    if (n_cuts > COGDL_HIP_MAX_SEGMENTS) {  // (more edges than 64 segments of that size hold: larger segments)
        max_edges = (nnz + COGDL_HIP_MAX_SEGMENTS - 1) / COGDL_HIP_MAX_SEGMENTS;
        if (max_edges > ((int64_t)1 << 30)) return COGDL_HIP_ERANGE;
        n_cuts = (nnz + max_edges - 1) / max_edges;
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t *rows_d = (int64_t *)scratch, *edges_d = rows_d + COGDL_HIP_MAX_SEGMENTS + 1;
    hipLaunchKernelGGL(segment_cuts_kernel, dim3(1), dim3(128), 0, s, rowptr, m, max_edges, (int)n_cuts, rows_d, edges_d);
    hipLaunchKernelGGL(rowptr_monotone_kernel, dim3((unsigned)std::min<int64_t>((m + 255) / 256, 1 << 14)), dim3(256), 0, s,
                       rowptr, m, edges_d);
    int64_t host[2 * (COGDL_HIP_MAX_SEGMENTS + 1)];
    hipError_t e = hipMemcpyAsync(host, scratch, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    const int64_t *rows = host, *edges = host + COGDL_HIP_MAX_SEGMENTS + 1;
    if (edges[n_cuts] != nnz) return COGDL_HIP_EINVAL;  // rowptr[m] != nnz
    if (edges[0] != 0) return COGDL_HIP_EINVAL;         // rowptr[0] != 0, or rowptr decreases somewhere
    int n = 0;
    for (int64_t j = 1; j <= n_cuts; ++j) {  // (a row longer than max_edges owns several targets: equal cuts collapse)
        if (rows[j] == out->row[n]) continue;
        ++n;
        out->row[n] = rows[j];
        out->edge[n] = edges[j];
    }
    out->n = n;
    return segments_valid(out);
}

extern "C" int cogdl_hip_csr_rebase_rowptr(const int64_t *rowptr, const cogdl_hip_segments *seg, int32_t *rowptr32,
                                           void *stream) {
    int rc = segments_valid(seg);
    if (rc != COGDL_HIP_OK) return rc;
    if (seg->n == 0) return COGDL_HIP_OK;
    if (!rowptr || !rowptr32) return COGDL_HIP_EINVAL;
    const int64_t total = seg->row[seg->n] + seg->n;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(rebase_rowptr_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowptr, table_of(seg), rowptr32);
    return launch_status();
}

namespace {
struct TransposeLayout {
    size_t off_prior, off_sums, off_colptr_s, off_rowind_s, off_perm_s, off_inner, inner_bytes, total;
    int64_t n_scan_blocks;
};
TransposeLayout transpose_layout(const cogdl_hip_segments *seg, int64_t n_cols) {
    TransposeLayout L{};
    int64_t max_nnz = 0;
    size_t inner = 256;
    for (int s = 0; s < seg->n; ++s) {
        const int64_t nz = seg->edge[s + 1] - seg->edge[s];
        max_nnz = std::max(max_nnz, nz);
        inner = std::max(inner, cogdl_hip_csr2csc_workspace_bytes(seg->row[s + 1] - seg->row[s], n_cols, nz));
    }
    L.n_scan_blocks = (n_cols + kScanTile - 1) / kScanTile;
    size_t o = 0;
    L.off_prior = o;
    o += up256((size_t)(n_cols + 1) * 8);
    L.off_sums = o;
    o += up256((size_t)(L.n_scan_blocks + 1) * 8);
    L.off_colptr_s = o;
    o += up256((size_t)(n_cols + 1) * 4);
    L.off_rowind_s = o;
    o += up256((size_t)max_nnz * 4);
    L.off_perm_s = o;
    o += up256((size_t)max_nnz * 4);
    L.off_inner = o;
    L.inner_bytes = inner;
    L.total = o + up256(inner);
    return L;
}
}  // namespace

extern "C" size_t cogdl_hip_csr2csc_i64_workspace_bytes(const cogdl_hip_segments *seg, int64_t n_cols) {
    if (segments_valid(seg) != COGDL_HIP_OK || n_cols < 0) return 0;
    return transpose_layout(seg, n_cols).total;
}

extern "C" int cogdl_hip_csr2csc_i64(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                     int64_t n_cols, int64_t *colptr, int32_t *rowind, int64_t *perm, const void *val,
                                     void *val_t, int val_bytes, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = segments_valid(seg);
    if (rc != COGDL_HIP_OK) return rc;
    if (n_cols < 0 || !colptr) return COGDL_HIP_EINVAL;
    if ((val_t != nullptr) != (val != nullptr)) return COGDL_HIP_EINVAL;
    if (val && val_bytes != 4 && val_bytes != 2) return COGDL_HIP_EDTYPE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nnz = seg->n ? seg->edge[seg->n] : 0;
    hipError_t e = fill_u32_async(colptr, 0u, (size_t)(n_cols + 1) * 2, s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    if (nnz == 0 || n_cols == 0) return launch_status();
    if (!rowptr32 || !colind || !rowind || !workspace) return COGDL_HIP_EINVAL;
    const TransposeLayout L = transpose_layout(seg, n_cols);
    if (workspace_bytes < L.total) return COGDL_HIP_EWORKSPACE;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    char *ws = (char *)workspace;
    int64_t *prior = (int64_t *)(ws + L.off_prior), *sums = (int64_t *)(ws + L.off_sums);
    int32_t *colptr_s = (int32_t *)(ws + L.off_colptr_s), *rowind_s = (int32_t *)(ws + L.off_rowind_s);
    int32_t *perm_s = (int32_t *)(ws + L.off_perm_s);
    void *inner = ws + L.off_inner;
    auto transpose_segment = [&](int sgm) {
        const int64_t m_s = seg->row[sgm + 1] - seg->row[sgm], nz = seg->edge[sgm + 1] - seg->edge[sgm];
        return cogdl_hip_csr2csc(rowptr32 + seg->row[sgm] + sgm, colind + seg->edge[sgm], m_s, n_cols, nz, colptr_s, rowind_s,
                                 perm_s, inner, L.inner_bytes, stream);
    };
    // pass 1: column counts (a histogram over colind, accumulated in the output colptr), scanned in place
    hipLaunchKernelGGL(count_columns_kernel, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 16)), dim3(256), 0, s,
                       colind, nnz, n_cols, colptr);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)L.n_scan_blocks), dim3(kScanBlock), 0, s, colptr, n_cols, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanBlock), 0, s, sums, L.n_scan_blocks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)L.n_scan_blocks), dim3(kScanBlock), 0, s, colptr, n_cols, sums, colptr);
    // pass 2: the entries
    e = fill_u32_async(prior, 0u, (size_t)(n_cols + 1) * 2, s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    const unsigned mblocks = (unsigned)((n_cols + 255) / 256);
    for (int sgm = 0; sgm < seg->n; ++sgm) {
        if (seg->edge[sgm + 1] == seg->edge[sgm]) continue;
        if ((rc = transpose_segment(sgm)) != COGDL_HIP_OK) return rc;
        if (val_bytes == 2 && val)
            hipLaunchKernelGGL(merge_segment_kernel<uint16_t>, dim3(mblocks), dim3(256), 0, s, colptr_s, rowind_s, perm_s, n_cols,
                               seg->row[sgm], seg->edge[sgm], colptr, prior, rowind, perm, (const uint16_t *)val, (uint16_t *)val_t);
        else
            hipLaunchKernelGGL(merge_segment_kernel<uint32_t>, dim3(mblocks), dim3(256), 0, s, colptr_s, rowind_s, perm_s, n_cols,
                               seg->row[sgm], seg->edge[sgm], colptr, prior, rowind, perm, (const uint32_t *)val, (uint32_t *)val_t);
    }
    return launch_status();
}

extern "C" int cogdl_hip_gather_rows_i64(const int64_t *perm, const void *src, void *out, int64_t n, int64_t h,
                                         int elem_bytes, void *stream) {
    if (n < 0 || h < 0) return COGDL_HIP_EINVAL;
    if (n == 0 || h == 0) return COGDL_HIP_OK;
    if (!perm || !src || !out) return COGDL_HIP_EINVAL;
    const int64_t total = n * h;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 16);
    hipStream_t s = (hipStream_t)stream;
    if (elem_bytes == 4)
        hipLaunchKernelGGL(gather_rows_i64_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, perm, (const uint32_t *)src,
                           (uint32_t *)out, total, h);
    else if (elem_bytes == 2)
        hipLaunchKernelGGL(gather_rows_i64_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, perm, (const uint16_t *)src,
                           (uint16_t *)out, total, h);
    else
        return COGDL_HIP_EDTYPE;
    return launch_status();
}
