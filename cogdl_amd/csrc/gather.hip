// gather.hip -- row gathers / scatters by node id:
//   gather_feature_rows   out[i, :] = src[ids[i], :]                   the mini-batch feature gather x[n_id]
//                         (cogdl/data/sampler.py:82-116, cogdl/models/nn/graphsage.py:86-99).  `src` may live in
//                         PINNED HOST memory: pinned allocations are mapped into the GPU's address space, so the
//                         kernel reads the selected rows straight over the host link ("zero copy") -- no host-side
//                         index_select into a staging buffer, no second pass over the batch, and the transfer overlaps
//                         whatever else the GPU does on other streams (the sampler of the next batch).
//   add_rows_at           out[ids[i], :] += src[i, :], ids DISTINCT     accumulation of returned halo gradients in the
//                         vertex-sharded SpMM backward (cogdl_amd/dist.py); plain read-modify-write, no atomics.
// Rows are moved as 16-byte vectors when the row size and the pointers allow, else as 4-byte words; consecutive lanes
// take consecutive vectors of a row, rows are dealt round-robin to lane groups, 4 independent vectors in flight per
// lane (a host-link read has ~2 us of latency).
#include "common.h"

namespace cogdl {

template <typename E, typename I, bool ADD>
__global__ __launch_bounds__(256) void rows_by_id_kernel(const I *__restrict__ ids, const E *__restrict__ src,
                                                         E *__restrict__ out, int64_t n, int64_t vecs_per_row,
                                                         int64_t n_other, int *__restrict__ bad_flag) {
    // a "vector" = one E; row i of the compact side <-> row ids[i] of the indexed side
    const int64_t total = n * vecs_per_row;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    constexpr int U = 4;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; base < total; base += stride * U) {
        // Unconditional phases: the U ids, then the U source vectors, then the stores.  (A load behind a per-lane branch is
        // waited for at the branch's join: the id -> row chain of the U slots used to run as 2 U serial round trips of ~2 us
        // each -- 18 us for a 15 k-row gather that moves 6 MB.)  Indices are clamped into range; what a slot past the end
        // or a bad id reads is ignored.
        int64_t idx[U], row_i[U], col_c[U], r[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = base + (int64_t)u * stride;
            const int64_t ic = min(idx[u], total - 1);
            row_i[u] = ic / vecs_per_row;
            col_c[u] = ic - row_i[u] * vecs_per_row;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = (int64_t)ids[row_i[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = idx[u] < total;
            if (ok[u] && (r[u] < 0 || r[u] >= n_other)) {
                if (bad_flag) atomicOr(bad_flag, 1);
                ok[u] = false;
            }
            r[u] = min(max(r[u], (int64_t)0), n_other - 1);
        }
        E v[U], o[ADD ? U : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (ADD) {
                v[u] = src[min(idx[u], total - 1)];
                o[u] = out[r[u] * vecs_per_row + col_c[u]];
            } else {
                v[u] = src[r[u] * vecs_per_row + col_c[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            if constexpr (ADD) {
                E acc = o[u];
                if constexpr (sizeof(E) == 16) {
                    float *of = reinterpret_cast<float *>(&acc);
                    const float *vf = reinterpret_cast<const float *>(&v[u]);
                    of[0] += vf[0];
                    of[1] += vf[1];
                    of[2] += vf[2];
                    of[3] += vf[3];
                } else {
                    *reinterpret_cast<float *>(&acc) += *reinterpret_cast<const float *>(&v[u]);
                }
                out[r[u] * vecs_per_row + col_c[u]] = acc;
            } else {
                out[idx[u]] = v[u];
            }
        }
    }
}

template <typename I, bool ADD>
static int launch_rows(const I *ids, const void *src, void *out, int64_t n, int64_t row_bytes, int64_t n_other,
                       int *bad_flag, hipStream_t s) {
    if (n < 0 || row_bytes < 0 || n_other < 0) return COGDL_HIP_EINVAL;
    if (n == 0 || row_bytes == 0) return COGDL_HIP_OK;
    if (!ids || !src || !out || n_other == 0) return COGDL_HIP_EINVAL;  // (rows asked of an empty table)
    if (row_bytes % 4 != 0 || !aligned_to(src, 4) || !aligned_to(out, 4)) return COGDL_HIP_EALIGN;
    const bool wide = row_bytes % 16 == 0 && aligned_to(src, 16) && aligned_to(out, 16);
    const int64_t vecs = row_bytes / (wide ? 16 : 4);
    const int64_t total = n * vecs;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 1023) / 1024, 256 * 16);
    if (wide)
        hipLaunchKernelGGL((rows_by_id_kernel<uint4, I, ADD>), dim3(blocks), dim3(256), 0, s, ids, (const uint4 *)src,
                           (uint4 *)out, n, vecs, n_other, bad_flag);
    else
        hipLaunchKernelGGL((rows_by_id_kernel<uint32_t, I, ADD>), dim3(blocks), dim3(256), 0, s, ids, (const uint32_t *)src,
                           (uint32_t *)out, n, vecs, n_other, bad_flag);
    return launch_status();
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_gather_feature_rows(const int64_t *ids, const void *src, void *out, int64_t n,
                                             int64_t row_bytes, int64_t n_src, int *bad_flag, void *stream) {
    return launch_rows<int64_t, false>(ids, src, out, n, row_bytes, n_src, bad_flag, (hipStream_t)stream);
}

extern "C" int cogdl_hip_gather_feature_rows_i32(const int32_t *ids, const void *src, void *out, int64_t n,
                                                 int64_t row_bytes, int64_t n_src, int *bad_flag, void *stream) {
    return launch_rows<int32_t, false>(ids, src, out, n, row_bytes, n_src, bad_flag, (hipStream_t)stream);
}

extern "C" int cogdl_hip_add_rows_at_f32(const int64_t *ids, const float *src, float *out, int64_t n, int64_t k,
                                         int64_t n_dst, int *bad_flag, void *stream) {
    return launch_rows<int64_t, true>(ids, src, out, n, k * 4, n_dst, bad_flag, (hipStream_t)stream);
}
