// sddmm.hip -- csr_sddmm for gfx950: out[e] = < d1[row(e),:], d2[colind[e],:] >.
// Replaces cogdl/operators/spmm/sddmm_kernel.cu:249-417 (16 edges per block, binary search
// of the row per 4 edges, warp=32 shuffles).  Here the row-group decomposition of spmm.hip
// is reused: a group of LPR lanes owns one CSR row, so row(e) is known without a search and
// d1[row,:] stays in registers for the whole row; each neighbour row of d2 is one coalesced
// vector load; per-edge dot products are reduced with a wave64 butterfly inside the group
// and written back as one coalesced store per chunk of LPR edges.
// HBM-bound: algorithmic bytes per edge = 4 (colind) + F*s (d2 row) + 4 (out); d1 once per row.
#include "common.h"

namespace cogdl {

template <int VEC, int LPR, int UNROLL, bool SINGLE_TILE>
__global__ __launch_bounds__(256) void csr_sddmm_kernel(const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ colind,
                                                        const float *__restrict__ d1,
                                                        const float *__restrict__ d2, float *__restrict__ out,
                                                        int64_t m, int k, XcdMap n_rowblocks) {
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int64_t row = rb * RPB + wave * RPW + sub;
    int start = 0, end = 0;
    if (row < m) {
        start = rowptr[row];
        end = rowptr[row + 1];
    }
    const float *a_row = d1 + (row < m ? row : 0) * (int64_t)k;
    float a0[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) a0[i] = 0.f;
    if (SINGLE_TILE && l * VEC < k) load_vec<float, VEC>(a_row + l * VEC, a0);

    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_c = (l < cnt) ? colind[base + l] : 0;
        float my_out = 0.f;
        for (int j = 0; j < cnt; j += UNROLL) {
            float p[UNROLL];
            int c[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                p[u] = 0.f;
                c[u] = __shfl(my_c, sub * LPR + min(j + u, cnt - 1), kWave);
            }
            if constexpr (SINGLE_TILE) {
                float b[UNROLL][VEC];
                const bool col_ok = l * VEC < k;
                const int c0 = col_ok ? l * VEC : 0;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) load_vec<float, VEC>(d2 + (int64_t)c[u] * k + c0, b[u]);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) p[u] = fmaf(a0[i], col_ok ? b[u][i] : 0.f, p[u]);
            } else {
                for (int c0 = l * VEC; c0 < k; c0 += LPR * VEC) {
                    float a[VEC], b[UNROLL][VEC];
                    load_vec<float, VEC>(a_row + c0, a);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) load_vec<float, VEC>(d2 + (int64_t)c[u] * k + c0, b[u]);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                        for (int i = 0; i < VEC; ++i) p[u] = fmaf(a[i], b[u][i], p[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const float r = group_sum<LPR>(p[u]);
                if (l == j + u) my_out = r;
            }
        }
        if (l < cnt) out[base + l] = my_out;
    }
}

template <int VEC, int LPR>
static int launch_sddmm(const int32_t *rowptr, const int32_t *colind, const float *d1, const float *d2,
                        float *out, int64_t m, int64_t k, hipStream_t s) {
    constexpr int RPB = (kWave / LPR) * 4;
    constexpr int UNROLL = 4;
    const int64_t n_rowblocks = (m + RPB - 1) / RPB;
    if (n_rowblocks == 0) return COGDL_HIP_OK;
    if (n_rowblocks > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
    dim3 grid(xcd_grid(make_xcd_map(n_rowblocks)));
    if (k <= (int64_t)LPR * VEC)
        hipLaunchKernelGGL((csr_sddmm_kernel<VEC, LPR, UNROLL, true>), grid, dim3(256), 0, s, rowptr, colind, d1, d2,
                           out, m, (int)k, make_xcd_map(n_rowblocks));
    else
        hipLaunchKernelGGL((csr_sddmm_kernel<VEC, LPR, UNROLL, false>), grid, dim3(256), 0, s, rowptr, colind, d1,
                           d2, out, m, (int)k, make_xcd_map(n_rowblocks));
    return launch_status();
}

template <int VEC>
static int dispatch_sddmm(const int32_t *rowptr, const int32_t *colind, const float *d1, const float *d2,
                          float *out, int64_t m, int64_t k, hipStream_t s) {
    const int64_t need = (k + VEC - 1) / VEC;
    if (need <= 8) return launch_sddmm<VEC, 8>(rowptr, colind, d1, d2, out, m, k, s);
    if (need <= 16) return launch_sddmm<VEC, 16>(rowptr, colind, d1, d2, out, m, k, s);
    if (need <= 32) return launch_sddmm<VEC, 32>(rowptr, colind, d1, d2, out, m, k, s);
    return launch_sddmm<VEC, 64>(rowptr, colind, d1, d2, out, m, k, s);
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_csr_sddmm(const int32_t *rowptr, const int32_t *colind, const float *d1,
                                   const float *d2, float *out, int64_t m, int64_t k, void *stream) {
    if (m < 0 || k < 0) return COGDL_HIP_EINVAL;
    if (m == 0) return COGDL_HIP_OK;
    if (!rowptr || !d1 || !d2 || !out) return COGDL_HIP_EINVAL;
    if (k > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (k == 0) return COGDL_HIP_EINVAL;
    if (k % 4 == 0 && aligned_to(d1, 16) && aligned_to(d2, 16)) return dispatch_sddmm<4>(rowptr, colind, d1, d2, out, m, k, s);
    if (k % 2 == 0 && aligned_to(d1, 8) && aligned_to(d2, 8)) return dispatch_sddmm<2>(rowptr, colind, d1, d2, out, m, k, s);
    return dispatch_sddmm<1>(rowptr, colind, d1, d2, out, m, k, s);
}
