// scatter_max.hip -- neighbourhood max aggregation (GraphSAGE MaxAggregator) for gfx950.
// Replaces cogdl/operators/scatter_max/scatter_max.cu:5-75.
// forward : out[r,c] = max_{e in row r} feat[colind[e],c], max_id[r,c] = colind of the FIRST
//           maximum in CSR order; empty row -> 0 / -1.  (The reference starts from FLT_MIN --
//           the smallest positive float -- so all-negative rows return 1.18e-38 with an
//           uninitialised argmax; that is not reproduced.)  NaN never wins (`v > acc` is false),
//           as in the reference's `acc < B`.
// backward: grad_src[max_id[r,c], c] += grad[r,c] into a buffer zeroed here (the reference
//           accumulates into torch::empty memory).  fp32 hardware atomics, one per element.
// Same row-group decomposition and coalesced row gathers as spmm.hip.
#include "common.h"

namespace cogdl {

template <int VEC, int LPR, int UNROLL>
__global__ __launch_bounds__(256) void scatter_max_fwd_kernel(const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ colind,
                                                              const float *__restrict__ x, float *__restrict__ out,
                                                              int32_t *__restrict__ max_id, int64_t m, int k,
                                                              XcdMap n_rowblocks) {
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int64_t row = rb * RPB + (threadIdx.x >> 6) * RPW + sub;
    const int col0 = ((int)blockIdx.y * LPR + l) * VEC;
    const bool col_ok = col0 < k;
    int start = 0, end = 0;
    if (row < m) {
        start = rowptr[row];
        end = rowptr[row + 1];
    }
    float acc[VEC];
    int id[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        acc[i] = 0.f;
        id[i] = -1;
    }
    const float *xcol = x + (col_ok ? col0 : 0);
    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_c = (l < cnt) ? colind[base + l] : 0;
        for (int j = 0; j < cnt; j += UNROLL) {
            float v[UNROLL][VEC];
            int c[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                c[u] = __shfl(my_c, sub * LPR + min(j + u, cnt - 1), kWave);
                load_vec<float, VEC>(xcol + (int64_t)c[u] * k, v[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (j + u < cnt) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const bool take = (id[i] < 0) || (v[u][i] > acc[i]);
                        acc[i] = take ? v[u][i] : acc[i];
                        id[i] = take ? c[u] : id[i];
                    }
                }
            }
        }
    }
    if (row < m && col_ok) {
        store_vec<float, VEC>(out + row * (int64_t)k + col0, acc);
#pragma unroll
        for (int i = 0; i < VEC; ++i) max_id[row * (int64_t)k + col0 + i] = id[i];
    }
}

__global__ void scatter_max_bwd_kernel(const float *__restrict__ grad, const int32_t *__restrict__ max_id,
                                       float *__restrict__ grad_src, int64_t total, int k) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = max_id[idx];
        if (id >= 0) atomicAdd(grad_src + (int64_t)id * k + (idx % k), grad[idx]);
    }
}

template <int VEC, int LPR>
static int launch_smax(const int32_t *rowptr, const int32_t *colind, const float *x, float *out, int32_t *max_id,
                       int64_t m, int64_t k, hipStream_t s) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int64_t nrb = (m + RPB - 1) / RPB;
    const int64_t tiles = (k + (int64_t)LPR * VEC - 1) / ((int64_t)LPR * VEC);
    if (nrb > 0x7fffffff / kXcds || tiles > 65535) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((scatter_max_fwd_kernel<VEC, LPR, 8>), dim3(xcd_grid(make_xcd_map(nrb)), (unsigned)tiles), dim3(256), 0, s,
                       rowptr, colind, x, out, max_id, m, (int)k, make_xcd_map(nrb));
    return launch_status();
}

template <int VEC>
static int dispatch_smax(const int32_t *rowptr, const int32_t *colind, const float *x, float *out, int32_t *max_id,
                         int64_t m, int64_t k, hipStream_t s) {
    const int64_t need = (k + VEC - 1) / VEC;
    if (need <= 8) return launch_smax<VEC, 8>(rowptr, colind, x, out, max_id, m, k, s);
    if (need <= 16) return launch_smax<VEC, 16>(rowptr, colind, x, out, max_id, m, k, s);
    if (need <= 32) return launch_smax<VEC, 32>(rowptr, colind, x, out, max_id, m, k, s);
    return launch_smax<VEC, 64>(rowptr, colind, x, out, max_id, m, k, s);
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_scatter_max_fwd(const int32_t *rowptr, const int32_t *colind, const float *feat,
                                         float *out, int32_t *max_id, int64_t m, int64_t k, void *stream) {
    if (m < 0 || k < 0) return COGDL_HIP_EINVAL;
    if (m == 0 || k == 0) return COGDL_HIP_OK;
    if (!rowptr || !feat || !out || !max_id) return COGDL_HIP_EINVAL;
    if (k > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (k % 4 == 0 && aligned_to(feat, 16) && aligned_to(out, 16)) return dispatch_smax<4>(rowptr, colind, feat, out, max_id, m, k, s);
    if (k % 2 == 0 && aligned_to(feat, 8) && aligned_to(out, 8)) return dispatch_smax<2>(rowptr, colind, feat, out, max_id, m, k, s);
    return dispatch_smax<1>(rowptr, colind, feat, out, max_id, m, k, s);
}

extern "C" int cogdl_hip_scatter_max_bwd(const float *grad, const int32_t *max_id, float *grad_src, int64_t m,
                                         int64_t k, int64_t n_src, void *stream) {
    if (m < 0 || k < 0 || n_src < 0) return COGDL_HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n_src > 0 && k > 0) {
        if (!grad_src) return COGDL_HIP_EINVAL;
        hipError_t e = hipMemsetAsync(grad_src, 0, (size_t)n_src * (size_t)k * sizeof(float), s);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return COGDL_HIP_ELAUNCH;
        }
    }
    if (m == 0 || k == 0) return COGDL_HIP_OK;
    if (!grad || !max_id) return COGDL_HIP_EINVAL;
    const int64_t total = m * k;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(scatter_max_bwd_kernel, dim3(blocks), dim3(256), 0, s, grad, max_id, grad_src, total, (int)k);
    return launch_status();
}
