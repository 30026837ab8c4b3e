// transpose.hip -- csr2csc (stable CSR transpose) and row gathers for gfx950.
// Replaces the reference's cuSPARSE call (cogdl/operators/spmm/spmm_kernel.cu:514-532,
// cusparseCsr2cscEx2 ALG1: handle created per call and leaked, cudaMalloc/cudaFree per call)
// From 256 k edge slots on the transpose is the hand-written two-payload radix sort of radix_transpose.hip; below that
// (and under tuning key 10 = 1) this file's allocation-free, stream-ordered, deterministic rocPRIM pipeline:
//   1. stable LSD radix sort of (key = colind[e], value = e) restricted to the
//      ceil(log2(n_cols)) significant key bits (rocPRIM device primitive, header-only) -> perm
//   2. colptr from the sorted keys by boundary detection (no atomics)
//   3. rowind[j] = row owning CSR position perm[j] (binary search in rowptr, L2 resident)
// Traffic ~ nnz * (4+4) * 2 per radix pass + nnz * 12; all integer, HBM/L2 bound.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace cogdl {

static unsigned key_bits(int64_t n_cols) {
    unsigned b = 1;
    while (b < 32 && (int64_t(1) << b) < n_cols) ++b;
    return b;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// (Tried: a 9-bit-digit onesweep configuration -- two passes instead of three for column ids of up to 18 bits.  With 512
// bins the default rank algorithm needs 262 KB of LDS; the `match` algorithm fits but runs each pass slower: 4.6 ms vs
// 3.7 ms for the whole transpose of the Reddit-shaped graph.  rocPRIM's tuned default stays.)
// capturable: the merge sort at every size -- above 1 M keys rocPRIM's default is onesweep, whose hipMemsetAsync calls
// become memset nodes in a captured hipGraph, and those were seen not to replay (common.h: fill_u32_async).
using CapturableSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                        rocprim::default_config, (size_t)1 << 40>;
static hipError_t sort_pairs(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                             int32_t *perm_out, int64_t nnz, unsigned bits, hipStream_t s, bool capturable = false) {
    rocprim::counting_iterator<int32_t> iota(0);
    if (capturable)
        return rocprim::radix_sort_pairs<CapturableSortConfig>(temp, temp_bytes, keys_in, keys_out, iota, perm_out,
                                                               (size_t)nnz, 0u, bits, s);
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, iota, perm_out, (size_t)nnz, 0u, bits, s);
}

// colptr[c] = number of sorted keys < c (lower bound), one thread per column: fully
// parallel even when most columns are empty (sampled blocks), no atomics.
__global__ void colptr_from_sorted_keys(const uint32_t *__restrict__ keys, int32_t *__restrict__ colptr,
                                        int64_t nnz, int64_t n_cols) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= n_cols;
         c += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = nnz;  // first position whose key >= c
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < c) lo = mid + 1; else hi = mid;
        }
        colptr[c] = (int32_t)lo;
    }
}

// rowind[j] = the row that owns CSR position perm[j].  A full binary search of rowptr per element is 18-27 dependent
// L2 loads (measured: 1.9 ms of the 4.7 ms transpose of the Reddit-shaped graph).  A coarse table -- the row of every
// 128th edge, nnz/128 entries, L2 resident -- narrows the search to the rows that intersect one 128-edge block
// (one row for hubs, a few cache-line-adjacent rowptr entries otherwise).
constexpr int kCoarseShift = 7;

__device__ __forceinline__ int64_t row_search(const int32_t *__restrict__ rowptr, int64_t lo, int64_t hi, int64_t e) {
    while (hi - lo > 1) {  // invariant: rowptr[lo] <= e < rowptr[hi]; empty rows share their successor's offset
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void coarse_rows_kernel(const int32_t *__restrict__ rowptr, int64_t m, int64_t nnz,
                                   int32_t *__restrict__ coarse, int64_t n_coarse) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_coarse) return;
    const int64_t e = b << kCoarseShift;
    coarse[b] = e < nnz ? (int32_t)row_search(rowptr, 0, m, e) : (int32_t)(m - 1);
}

__global__ void rowind_from_perm(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ perm,
                                 const int32_t *__restrict__ coarse, int32_t *__restrict__ rowind, int64_t nnz,
                                 int64_t m) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nnz) return;
    const int32_t e = perm[j];
    const int64_t b = e >> kCoarseShift;
    // the row of edge b*128 is <= the wanted row, the row of edge (b+1)*128 is >= it
    rowind[j] = (int32_t)row_search(rowptr, coarse[b], min(m, (int64_t)coarse[b + 1] + 1), e);
}

// ---- COO -> CSR on the GPU (coo2csr_index) ------------------------------------------------------------------------
__global__ void keys_from_rows64(const int64_t *__restrict__ row, uint32_t *__restrict__ keys, int64_t nnz,
                                 int64_t num_nodes, int *__restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = row[i];
        if (r < 0 || r >= num_nodes) *bad = 1;  // reported by the caller's next look at `bad`; key clamped
        keys[i] = (uint32_t)(r < 0 ? 0 : (r >= num_nodes ? num_nodes - 1 : r));
    }
}

// row_ptr[r] (int64) = number of sorted keys < r; perm widened to int64 in the same launch
__global__ void rowptr64_and_perm64(const uint32_t *__restrict__ keys, const int32_t *__restrict__ perm32,
                                    int64_t *__restrict__ row_ptr, int64_t *__restrict__ perm, int64_t nnz,
                                    int64_t num_nodes) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t r = tid; r <= num_nodes; r += stride) {
        int64_t lo = 0, hi = nnz;  // first position whose key >= r
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < r) lo = mid + 1; else hi = mid;
        }
        row_ptr[r] = lo;
    }
    for (int64_t i = tid; i < nnz; i += stride) perm[i] = perm32[i];
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

static size_t rocprim_csr2csc_bytes(int64_t n_cols, int64_t nnz) {
    size_t temp = 0;
    (void)sort_pairs(nullptr, temp, nullptr, nullptr, nullptr, nnz, key_bits(n_cols), nullptr);
    return align_up((size_t)nnz * sizeof(uint32_t), 256) + align_up(temp, 256) +
           align_up((size_t)((nnz >> kCoarseShift) + 2) * sizeof(int32_t), 256) + 256;
}

extern "C" size_t cogdl_hip_csr2csc_workspace_bytes(int64_t m, int64_t n_cols, int64_t nnz) {
    (void)m;
    if (nnz <= 0) return 256;
    return std::max(rocprim_csr2csc_bytes(n_cols, nnz), radix_transpose_workspace_bytes(n_cols, nnz, false));
}

// Fixed-capacity blocks (sample_adj_padded): colind holds `nnz` slots of which only the first rowptr[m] are edges.  The
// surplus gets the key n_cols -- behind every real column, so colptr[n_cols] = rowptr[m] and no column reaches it.
__global__ void padded_keys_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind, int64_t m,
                                   int64_t nnz, uint32_t n_cols, uint32_t *__restrict__ keys) {
    const int64_t valid = rowptr[m];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x)
        keys[e] = e < valid ? (uint32_t)colind[e] : n_cols;
}

extern "C" size_t cogdl_hip_csr2csc_padded_workspace_bytes(int64_t m, int64_t n_cols, int64_t nnz) {
    size_t extra = 0;  // the merge sort's scratch where it exceeds the default algorithm's
    if (nnz > 0) {
        size_t t_default = 0, t_merge = 0;
        (void)sort_pairs(nullptr, t_default, nullptr, nullptr, nullptr, nnz, key_bits(n_cols + 1), nullptr);
        (void)sort_pairs(nullptr, t_merge, nullptr, nullptr, nullptr, nnz, key_bits(n_cols + 1), nullptr, true);
        if (t_merge > t_default) extra = align_up(t_merge - t_default, 256);
    }
    (void)m;
    return std::max(rocprim_csr2csc_bytes(n_cols + 1, nnz) + extra + align_up((size_t)std::max<int64_t>(nnz, 0) * sizeof(uint32_t), 256),
                    radix_transpose_workspace_bytes(n_cols, nnz, true));
}

static int csr2csc_impl(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols, int64_t nnz,
                        int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace, size_t workspace_bytes,
                        void *stream, bool padded);

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

static int csr2csc_impl(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols, int64_t nnz,
                        int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace, size_t workspace_bytes,
                        void *stream, bool padded) {
    if (m < 0 || n_cols < 0 || nnz < 0 || !colptr) return COGDL_HIP_EINVAL;
    if (nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        hipLaunchKernelGGL(colptr_from_sorted_keys, dim3(64), dim3(256), 0, s, (const uint32_t *)nullptr, colptr, nnz, n_cols);
        return launch_status();
    }
    if (!rowptr || !colind || !rowind || !perm || !workspace) return COGDL_HIP_EINVAL;
    if (workspace_bytes < (padded ? cogdl_hip_csr2csc_padded_workspace_bytes(m, n_cols, nnz)
                                  : cogdl_hip_csr2csc_workspace_bytes(m, n_cols, nnz)))
        return COGDL_HIP_EWORKSPACE;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    if (padded && (m == 0 || n_cols >= 0x7fffffff)) return COGDL_HIP_ERANGE;
    // The hand-written sort (radix_transpose.hip) from 256 k edge slots on: below that a transpose is ~10 launches either
    // way and rocPRIM's merge sort has the shorter ones (51 k slots: 69 vs 69 us alone, but inside the captured
    // mini-batch step +0.02 ms per step with the radix sort at every size: its 8192-slot tiles leave a 100 k-slot block
    // to 14 workgroups with a long serial chain each; 2.5 M slots: 198 vs 112 us the other way).
    const bool radix = g_tuning[kTuneCsr2csc] >= 2 || (g_tuning[kTuneCsr2csc] == 0 && nnz >= (1 << 18));
    if (radix && m > 0) {
        return radix_transpose(rowptr, colind, m, n_cols, nnz, padded, colptr, rowind, perm, workspace, s);
    }
    // Up to 16 k slots and columns (the sampled blocks of a mini-batch step): one single-workgroup launch, everything in
    // LDS (radix_transpose.hip: small_transpose_kernel).  tuning key 10 = 1 keeps the rocPRIM pipeline for A/B runs.
    if (g_tuning[kTuneCsr2csc] != 1 && small_transpose_covers(m, n_cols, nnz))
        return small_transpose(rowptr, colind, m, n_cols, nnz, padded, colptr, rowind, perm, s);
    const uint32_t *keys_in = (const uint32_t *)colind;
    if (padded) {  // (the key buffer sits in front of the ordinary layout)
        uint32_t *keys = (uint32_t *)workspace;
        workspace = (char *)workspace + align_up((size_t)nnz * sizeof(uint32_t), 256);
        hipLaunchKernelGGL(padded_keys_kernel, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 4096)), dim3(256), 0, s,
                           rowptr, colind, m, nnz, (uint32_t)n_cols, keys);
        keys_in = keys;
    }
    uint32_t *keys_sorted = (uint32_t *)workspace;
    char *temp = (char *)workspace + align_up((size_t)nnz * sizeof(uint32_t), 256);
    size_t temp_bytes = 0;
    const unsigned bits = key_bits(padded ? n_cols + 1 : n_cols);
    (void)sort_pairs(nullptr, temp_bytes, nullptr, nullptr, nullptr, nnz, bits, nullptr, padded);
    hipError_t e = sort_pairs(temp, temp_bytes, keys_in, keys_sorted, perm, nnz, bits, s, padded);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    const unsigned blocks = (unsigned)((nnz + 255) / 256);
    const unsigned cblocks = (unsigned)std::min<int64_t>((n_cols + 256) / 256, 1 << 20);
    int32_t *coarse = (int32_t *)(temp + align_up(temp_bytes, 256));
    const int64_t n_coarse = (nnz >> kCoarseShift) + 2;
    hipLaunchKernelGGL(coarse_rows_kernel, dim3((unsigned)((n_coarse + 255) / 256)), dim3(256), 0, s, rowptr, m, nnz, coarse,
                       n_coarse);
    hipLaunchKernelGGL(colptr_from_sorted_keys, dim3(cblocks), dim3(256), 0, s, keys_sorted, colptr, nnz, n_cols);
    hipLaunchKernelGGL(rowind_from_perm, dim3(blocks), dim3(256), 0, s, rowptr, perm, coarse, rowind, nnz, m);
    return launch_status();
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
extern "C" size_t cogdl_hip_coo2csr_index_workspace_bytes(int64_t nnz, int64_t num_nodes) {
    if (nnz <= 0) return 256;
    size_t temp = 0;
    (void)sort_pairs(nullptr, temp, nullptr, nullptr, nullptr, nnz, key_bits(num_nodes), nullptr);
    return 2 * align_up((size_t)nnz * sizeof(uint32_t), 256) + align_up((size_t)nnz * sizeof(int32_t), 256) +
           align_up(temp, 256) + 256;
}

extern "C" int cogdl_hip_coo2csr_index(const int64_t *row, int64_t nnz, int64_t num_nodes, int64_t *row_ptr,
                                       int64_t *perm, int *bad_flag, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (nnz < 0 || num_nodes < 0 || !row_ptr || !bad_flag) return COGDL_HIP_EINVAL;
    if (nnz > 0x7fffffff || num_nodes > 0xffffffffll) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (nnz > 0 && (!row || !perm || !workspace)) return COGDL_HIP_EINVAL;
    if (workspace_bytes < cogdl_hip_coo2csr_index_workspace_bytes(nnz, num_nodes)) return COGDL_HIP_EWORKSPACE;
    if (nnz > 0 && !aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    hipError_t e = fill_u32_async(bad_flag, 0u, 1, s);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    const size_t kb = align_up((size_t)nnz * sizeof(uint32_t), 256);
    uint32_t *keys = (uint32_t *)workspace;
    uint32_t *keys_sorted = (uint32_t *)((char *)workspace + kb);
    int32_t *perm32 = (int32_t *)((char *)workspace + 2 * kb);
    char *temp = (char *)workspace + 2 * kb + align_up((size_t)nnz * sizeof(int32_t), 256);
    const unsigned eblocks = (unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 16);
    if (nnz > 0) {
        hipLaunchKernelGGL(keys_from_rows64, dim3(eblocks), dim3(256), 0, s, row, keys, nnz, num_nodes, bad_flag);
        size_t temp_bytes = 0;
        const unsigned bits = key_bits(num_nodes);
        (void)sort_pairs(nullptr, temp_bytes, nullptr, nullptr, nullptr, nnz, bits, nullptr);
        e = sort_pairs(temp, temp_bytes, keys, keys_sorted, perm32, nnz, bits, s);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return COGDL_HIP_ELAUNCH;
        }
    }
    const unsigned blocks = (unsigned)std::min<int64_t>((std::max(nnz, num_nodes + 1) + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(rowptr64_and_perm64, dim3(blocks), dim3(256), 0, s, keys_sorted, perm32, row_ptr, perm, nnz,
                       num_nodes);
    return launch_status();
}
