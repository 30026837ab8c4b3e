// transpose.hip -- csr2csc (stable CSR transpose), COO -> CSR index, row gathers and the structure hash for gfx950.
// Replaces the reference's cuSPARSE call (cogdl/operators/spmm/spmm_kernel.cu:514-532,
// cusparseCsr2cscEx2 ALG1: handle created per call and leaked, cudaMalloc/cudaFree per call).
// Every size is hand-written since round 5 (rounds 1-4 kept a rocPRIM radix/merge sort + row look-up pipeline for 16 k ..
// 256 k edge slots and for coo2csr_index):
//   * up to 16 k slots and columns: ONE single-workgroup launch, everything in LDS (radix_transpose.hip: small_transpose);
//   * above: the two-payload LSD radix sort of radix_transpose.hip (sorts the slots by column id, carrying the CSR
//     position and the row; 9-bit digits);
//   * coo2csr_index (stable sort of the edges by source row, cogdl/utils/graph_utils.py:133-142) IS that transpose of a
//     one-row matrix whose column ids are the rows: colptr = row_ptr, perm = the order -- no second sort in the library.
// All integer, HBM/L2 bound, allocation-free, stream-ordered, deterministic, hipGraph-capturable.
#include "common.h"

namespace cogdl {

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- COO -> CSR on the GPU (coo2csr_index) ------------------------------------------------------------------------
__global__ void keys_from_rows64(const int64_t *__restrict__ row, uint32_t *__restrict__ keys, int64_t nnz,
                                 int64_t num_nodes, int *__restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = row[i];
        if (r < 0 || r >= num_nodes) *bad = 1;  // reported by the caller's next look at `bad`; key clamped
        keys[i] = (uint32_t)(r < 0 ? 0 : (r >= num_nodes ? num_nodes - 1 : r));
    }
}

// colptr / perm of the one-row transpose, widened to the int64 the callers keep (cogdl/data/data.py: row_ptr, perm)
__global__ void widen_i32_to_i64_kernel(const int32_t *__restrict__ a, int64_t *__restrict__ a64, int64_t na,
                                        const int32_t *__restrict__ b, int64_t *__restrict__ b64, int64_t nb) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = tid; i < na; i += stride) a64[i] = a[i];
    for (int64_t i = tid; i < nb; i += stride) b64[i] = b[i];
}

// colptr of a structure without edges
__global__ void zero_i32_kernel(int32_t *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0;
}

template <typename E>
__global__ void gather_rows_kernel(const int32_t *__restrict__ perm, const E *__restrict__ src,
                                   E *__restrict__ out, int64_t total, int64_t h) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / h, c = idx - i * h;
        out[idx] = src[(int64_t)perm[i] * h + c];
    }
}

// 64-bit content hash of the CSR structure: sum over positions of a strong 64-bit mix of
// (position, value).  Commutative => any reduction order gives the same hash.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// One strong mix per 4 consecutive int32 values (position-keyed): ~1/4 of the 64-bit multiplies.
__device__ __forceinline__ uint64_t hash4(const int32_t *p, int64_t i, int64_t n, uint64_t salt) {
    uint32_t v[4];
    if (i + 4 <= n && (reinterpret_cast<uintptr_t>(p + i) & 15) == 0) {  // one 16-byte load (the usual case)
        const uint4 q = *reinterpret_cast<const uint4 *>(p + i);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i + j < n) ? (uint32_t)p[i + j] : 0x5bd1e995u;
    }
    const uint64_t a = ((uint64_t)v[1] << 32) | v[0], b = ((uint64_t)v[3] << 32) | v[2];
    return mix64(a * 0x9e3779b97f4a7c15ull + (b ^ salt) * 0xc2b2ae3d27d4eb4full + (uint64_t)i);
}

// Sum of hash4 over the groups i = first, first + stride, ... of p[0..n).  The bulk runs four unconditional 16-byte loads
// per trip (a load behind the alignment/tail branch of hash4 is waited for at the branch join: one round trip per group).
__device__ __forceinline__ uint64_t hash_stream(const int32_t *__restrict__ p, int64_t n, int64_t first, int64_t stride,
                                                uint64_t salt) {
    uint64_t acc = 0;
    int64_t i = first;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        for (; i + 3 * stride + 4 <= n; i += 4 * stride) {
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const uint4 *>(p + i + u * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t a = ((uint64_t)q[u].y << 32) | q[u].x, b = ((uint64_t)q[u].w << 32) | q[u].z;
                acc += mix64(a * 0x9e3779b97f4a7c15ull + (b ^ salt) * 0xc2b2ae3d27d4eb4full + (uint64_t)(i + u * stride));
            }
        }
    }
    for (; i < n; i += stride) acc += hash4(p, i, n, salt);
    return acc;
}

// One partial per workgroup, written with a plain store (the target may be host-mapped pinned memory): no memset,
// no atomics, no device-to-host copy behind it.  The fingerprint is the sum of the partials modulo 2^64.
__global__ __launch_bounds__(256) void csr_fingerprint_kernel(const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ colind, int64_t m,
                                                              int64_t nnz, unsigned long long *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    const int64_t tid4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    uint64_t acc = hash_stream(rowptr, m + 1, tid4, stride, 0xa5a5a5a5a5a5a5a5ull) + hash_stream(colind, nnz, tid4, stride, 0);
    // wave reduce, block reduce (integer add: order independent)
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)acc, s, kWave), hi = __shfl_xor((uint32_t)(acc >> 32), s, kWave);
        acc += ((uint64_t)hi << 32) | lo;
    }
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace cogdl

using namespace cogdl;

namespace cogdl {  // radix_transpose.hip
size_t radix_transpose_workspace_bytes(int64_t n_cols, int64_t nnz, bool padded);
int radix_transpose(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols, int64_t nnz, bool padded,
                    int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace, hipStream_t s);
bool small_transpose_covers(int64_t m, int64_t n_cols, int64_t nnz);
int small_transpose(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols, int64_t nnz, bool padded,
                    int32_t *colptr, int32_t *rowind, int32_t *perm, hipStream_t s);
}

extern "C" size_t cogdl_hip_csr2csc_workspace_bytes(int64_t m, int64_t n_cols, int64_t nnz) {
    (void)m;
    if (nnz <= 0) return 256;
    return radix_transpose_workspace_bytes(n_cols, nnz, false);
}

// Fixed-capacity blocks (sample_adj_padded): colind holds `nnz` slots of which only the first rowptr[m] are edges.  The
// surplus gets the key n_cols -- behind every real column, so colptr[n_cols] = rowptr[m] and no column reaches it.
extern "C" size_t cogdl_hip_csr2csc_padded_workspace_bytes(int64_t m, int64_t n_cols, int64_t nnz) {
    (void)m;
    if (nnz <= 0) return 256;
    return radix_transpose_workspace_bytes(n_cols, nnz, true);
}

static int csr2csc_impl(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols, int64_t nnz,
                        int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace, size_t workspace_bytes,
                        void *stream, bool padded) {
    if (m < 0 || n_cols < 0 || nnz < 0 || !colptr) return COGDL_HIP_EINVAL;
    if (nnz > COGDL_HIP_SEGMENT_MAX_EDGES) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)std::min<int64_t>((n_cols + 256) / 256, 4096)), dim3(256), 0, s, colptr,
                           n_cols + 1);
        return launch_status();
    }
    if (!rowptr || !colind || !rowind || !perm) return COGDL_HIP_EINVAL;
    if (m == 0) return padded ? COGDL_HIP_ERANGE : COGDL_HIP_EINVAL;  // edges without rows
    if (padded && n_cols >= 0x7fffffff) return COGDL_HIP_ERANGE;
    // Up to 16 k slots and columns (the sampled blocks of a mini-batch step): one single-workgroup launch, everything in
    // LDS (radix_transpose.hip: small_transpose_kernel); no workspace.
    // (tuning key 10 >= 2: the radix sort at every size -- tests and A/B runs)
    if (g_tuning[kTuneCsr2csc] < 2 && small_transpose_covers(m, n_cols, nnz)) return small_transpose(rowptr, colind, m, n_cols, nnz, padded, colptr, rowind, perm, s);
    if (!workspace) return COGDL_HIP_EINVAL;
    if (workspace_bytes < radix_transpose_workspace_bytes(n_cols, nnz, padded)) return COGDL_HIP_EWORKSPACE;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    return radix_transpose(rowptr, colind, m, n_cols, nnz, padded, colptr, rowind, perm, workspace, s);
}

extern "C" int cogdl_hip_csr2csc_padded(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols,
                                        int64_t nnz, int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    return csr2csc_impl(rowptr, colind, m, n_cols, nnz, colptr, rowind, perm, workspace, workspace_bytes, stream, true);
}

extern "C" int cogdl_hip_csr2csc(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols,
                                 int64_t nnz, int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    return csr2csc_impl(rowptr, colind, m, n_cols, nnz, colptr, rowind, perm, workspace, workspace_bytes, stream, false);
}

extern "C" int cogdl_hip_gather_rows(const int32_t *perm, const void *src, void *out, int64_t n, int64_t h,
                                     int elem_bytes, void *stream) {
    if (n < 0 || h < 0) return COGDL_HIP_EINVAL;
    if (n == 0 || h == 0) return COGDL_HIP_OK;
    if (!perm || !src || !out) return COGDL_HIP_EINVAL;
    const int64_t total = n * h;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    hipStream_t s = (hipStream_t)stream;
    // (Tried in round 3: the vectorised row gather of gather.hip -- 16-byte vectors, four independent id -> row chains in
    //  flight per lane -- instead of the element-wise kernel below: no faster (Reddit-sized [E, 8] fp32: 3.00 vs 3.00 ms,
    //  [E] fp32: 2.18 vs 2.03 ms).  A random gather of 4..32-byte rows is bound by the 64 / 128-byte line every row drags
    //  in, not by latency: 1.15e8 lines x 64 B = 7.3 GB of real traffic for 1.4 GB of useful bytes.)
    if (elem_bytes == 4)
        hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, perm, (const uint32_t *)src,
                           (uint32_t *)out, total, h);
    else if (elem_bytes == 2)
        hipLaunchKernelGGL(gather_rows_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, perm, (const uint16_t *)src,
                           (uint16_t *)out, total, h);
    else
        return COGDL_HIP_EDTYPE;
    return launch_status();
}

extern "C" int cogdl_hip_csr_fingerprint(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t nnz,
                                         uint64_t *out_parts, void *stream) {
    if (m < 0 || nnz < 0 || !rowptr || !out_parts) return COGDL_HIP_EINVAL;
    hipLaunchKernelGGL(csr_fingerprint_kernel, dim3(COGDL_HIP_FINGERPRINT_PARTS), dim3(256), 0, (hipStream_t)stream,
                       rowptr, colind, m, nnz, (unsigned long long *)out_parts);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Stable sort of POSITIONS by a 32-bit key (keys[i] < n_keys): perm_out[j] = the position holding the j-th smallest key, equal
// keys in ascending position -- the transpose of a one-row matrix without its colptr.  For the sampler's sort-based
// relabelling (sample.hip), which used rocPRIM's radix sort until round 5.  Workspace: rowptr2 | rowind (unused payload) |
// colptr (only for the single-workgroup kernel, which always writes it) | the transpose's own scratch.
namespace cogdl {
size_t sort_positions_workspace_bytes(int64_t n_keys, int64_t len) {
    if (len <= 0) return 256;
    size_t o = 256 + align_up((size_t)len * sizeof(int32_t), 256);
    if (small_transpose_covers(1, n_keys, len)) o += align_up((size_t)(n_keys + 1) * sizeof(int32_t), 256);
    return o + align_up(radix_transpose_workspace_bytes(n_keys, len, false), 256);
}

namespace {
__global__ void one_row_rowptr_kernel2(int32_t *rowptr2, int32_t nnz) {
    rowptr2[0] = 0;
    rowptr2[1] = nnz;
}
}  // namespace

int sort_positions_by_key(const uint32_t *keys, int64_t len, int64_t n_keys, int32_t *perm_out, void *workspace,
                          size_t workspace_bytes, hipStream_t s) {
    if (len <= 0) return COGDL_HIP_OK;
    if (!keys || !perm_out || !workspace) return COGDL_HIP_EINVAL;
    if (len > COGDL_HIP_SEGMENT_MAX_EDGES || n_keys >= 0x7fffffff) return COGDL_HIP_ERANGE;
    if (workspace_bytes < sort_positions_workspace_bytes(n_keys, len)) return COGDL_HIP_EWORKSPACE;
    char *ws = (char *)workspace;
    int32_t *rowptr2 = (int32_t *)ws;
    int32_t *rowind = (int32_t *)(ws + 256);
    char *rest = ws + 256 + align_up((size_t)len * sizeof(int32_t), 256);
    hipLaunchKernelGGL(one_row_rowptr_kernel2, dim3(1), dim3(1), 0, s, rowptr2, (int32_t)len);
    if (g_tuning[kTuneCsr2csc] < 2 && small_transpose_covers(1, n_keys, len)) {
        int32_t *colptr = (int32_t *)rest;
        return small_transpose(rowptr2, (const int32_t *)keys, 1, n_keys, len, false, colptr, rowind, perm_out, s);
    }
    if (small_transpose_covers(1, n_keys, len)) rest += align_up((size_t)(n_keys + 1) * sizeof(int32_t), 256);
    return radix_transpose(rowptr2, (const int32_t *)keys, 1, n_keys, len, false, nullptr, rowind, perm_out, rest, s);
}
}  // namespace cogdl

// coo2csr_index: the edges as ONE row of a [1 x num_nodes] matrix whose column ids are the source rows -- its stable
// transpose has colptr = row_ptr and perm = the stable order.  Workspace: keys | rowptr2 | colptr32 | rowind (unused
// payload) | perm32 | the transpose's own scratch.
namespace {
struct CooLayout {
    size_t off_keys, off_rowptr2, off_colptr, off_rowind, off_perm, off_inner, inner, total;
};
CooLayout coo_layout(int64_t nnz, int64_t num_nodes) {
    CooLayout L{};
    const size_t kb = align_up((size_t)std::max<int64_t>(nnz, 1) * sizeof(uint32_t), 256);
    size_t o = 0;
    L.off_keys = o;
    o += kb;
    L.off_rowptr2 = o;
    o += 256;
    L.off_colptr = o;
    o += align_up((size_t)(num_nodes + 1) * sizeof(int32_t), 256);
    L.off_rowind = o;
    o += kb;
    L.off_perm = o;
    o += kb;
    L.off_inner = o;
    L.inner = align_up(cogdl_hip_csr2csc_workspace_bytes(1, num_nodes, nnz), 256);
    L.total = o + L.inner;
    return L;
}
__global__ void one_row_rowptr_kernel(int32_t *rowptr2, int32_t nnz) {
    rowptr2[0] = 0;
    rowptr2[1] = nnz;
}
}  // namespace

extern "C" size_t cogdl_hip_coo2csr_index_workspace_bytes(int64_t nnz, int64_t num_nodes) {
    if (nnz <= 0) return 256;
    return coo_layout(nnz, num_nodes).total;
}

extern "C" int cogdl_hip_coo2csr_index(const int64_t *row, int64_t nnz, int64_t num_nodes, int64_t *row_ptr,
                                       int64_t *perm, int *bad_flag, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (nnz < 0 || num_nodes < 0 || !row_ptr || !bad_flag) return COGDL_HIP_EINVAL;
    if (nnz > COGDL_HIP_SEGMENT_MAX_EDGES || num_nodes >= 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (nnz > 0 && (!row || !perm || !workspace)) return COGDL_HIP_EINVAL;
    if (workspace_bytes < cogdl_hip_coo2csr_index_workspace_bytes(nnz, num_nodes)) return COGDL_HIP_EWORKSPACE;
    if (nnz > 0 && !aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    hipError_t e = fill_u32_async(bad_flag, 0u, 1, s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    if (nnz > 0 && num_nodes == 0) {  // every id is out of range
        e = fill_u32_async(bad_flag, 1u, 1, s);
        if (e == hipSuccess) e = fill_u32_async(row_ptr, 0u, 2, s);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return COGDL_HIP_ELAUNCH;
        }
        return launch_status();
    }
    if (nnz == 0) {
        e = fill_u32_async(row_ptr, 0u, (size_t)(num_nodes + 1) * 2, s);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return COGDL_HIP_ELAUNCH;
        }
        return launch_status();
    }
    const CooLayout L = coo_layout(nnz, num_nodes);
    char *ws = (char *)workspace;
    uint32_t *keys = (uint32_t *)(ws + L.off_keys);
    int32_t *rowptr2 = (int32_t *)(ws + L.off_rowptr2), *colptr32 = (int32_t *)(ws + L.off_colptr);
    int32_t *rowind = (int32_t *)(ws + L.off_rowind), *perm32 = (int32_t *)(ws + L.off_perm);
    const unsigned eblocks = (unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(keys_from_rows64, dim3(eblocks), dim3(256), 0, s, row, keys, nnz, num_nodes, bad_flag);
    hipLaunchKernelGGL(one_row_rowptr_kernel, dim3(1), dim3(1), 0, s, rowptr2, (int32_t)nnz);
    int rc = csr2csc_impl(rowptr2, (const int32_t *)keys, 1, num_nodes, nnz, colptr32, rowind, perm32, ws + L.off_inner, L.inner,
                          stream, false);
    if (rc != COGDL_HIP_OK) return rc;
    const unsigned blocks = (unsigned)std::min<int64_t>((std::max(nnz, num_nodes + 1) + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(widen_i32_to_i64_kernel, dim3(blocks), dim3(256), 0, s, colptr32, row_ptr, num_nodes + 1, perm32, perm, nnz);
    return launch_status();
}
