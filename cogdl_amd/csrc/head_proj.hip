// head_proj.hip -- the attention projections of GATLayer.forward, h_l = (a_l * h).sum(-1) and h_r = (a_r * h).sum(-1)
// (cogdl/layers/gat_layer.py:65-66; h [N, H, F] in the layer's dtype, a_l / a_r [1, H, F] fp32 parameters), BOTH in one pass
// over h: torch runs them as two broadcast products ([N, H, F] fp32 temporaries) and two reductions -- 0.24 ms of the
// 11.2 ms GAT step of BASELINE configs[2], where one read of h (30 MB) is all there is to do.  One thread per (node, head):
// fp32 products, summed left to right (the library is built with -ffp-contract=off: product and sum are rounded separately,
// as in torch's expression; torch's own reduction order over F may differ -- the results agree to fp32 rounding).
#include "common.h"

namespace cogdl {

template <typename T, int FV>  // FV: F when it is a small compile-time width read as ONE vector (0: any F, element loop)
__global__ __launch_bounds__(256) void head_proj_fwd_kernel(const T *__restrict__ feat, const float *__restrict__ a_l,
                                                            const float *__restrict__ a_r, float *__restrict__ h_l,
                                                            float *__restrict__ h_r, int64_t n_rows, int heads, int f_dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (node, head) pair
    if (i >= n_rows * heads) return;
    const int h = (int)(i % heads);
    const float *al = a_l + (int64_t)h * f_dim, *ar = a_r + (int64_t)h * f_dim;
    float sl = 0.f, sr = 0.f;
    if constexpr (FV > 0) {
        float x[FV];
        constexpr int CH = (16 / (int)sizeof(T) < FV) ? 16 / (int)sizeof(T) : FV;  // elements per 16-byte (or narrower) load
#pragma unroll
        for (int c = 0; c < FV / CH; ++c) {
            float part[CH];
            load_vec<T, CH>(feat + i * FV + c * CH, part);
#pragma unroll
            for (int q = 0; q < CH; ++q) x[c * CH + q] = part[q];
        }
#pragma unroll
        for (int f = 0; f < FV; ++f) {
            sl = sl + al[f] * x[f];
            sr = sr + ar[f] * x[f];
        }
    } else {
        const T *p = feat + i * f_dim;
        for (int f = 0; f < f_dim; ++f) {
            const float x = to_f32<T>(p[f]);
            sl = sl + al[f] * x;
            sr = sr + ar[f] * x;
        }
    }
    h_l[i] = sl;
    h_r[i] = sr;
}

template <typename T>
static int head_proj_typed(const void *feat, const float *a_l, const float *a_r, float *h_l, float *h_r, int64_t n, int64_t h, int64_t f,
                           hipStream_t s) {
    const int64_t total = n * h;
    const unsigned grid = (unsigned)((total + 255) / 256);
    const T *x = (const T *)feat;
    const bool vec8 = f == 8 && aligned_to(feat, 8 * sizeof(T));
    const bool vec4 = f == 4 && aligned_to(feat, 4 * sizeof(T));
    if (vec8) hipLaunchKernelGGL((head_proj_fwd_kernel<T, 8>), dim3(grid), dim3(256), 0, s, x, a_l, a_r, h_l, h_r, n, (int)h, (int)f);
    else if (vec4) hipLaunchKernelGGL((head_proj_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, x, a_l, a_r, h_l, h_r, n, (int)h, (int)f);
    else hipLaunchKernelGGL((head_proj_fwd_kernel<T, 0>), dim3(grid), dim3(256), 0, s, x, a_l, a_r, h_l, h_r, n, (int)h, (int)f);
    return launch_status();
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_head_projection_fwd(const void *feat, int dtype, const float *a_l, const float *a_r, float *h_l, float *h_r,
                                             int64_t n_rows, int64_t heads, int64_t f_dim, void *stream) {
    if (n_rows < 0 || heads <= 0 || f_dim <= 0) return COGDL_HIP_EINVAL;
    if (n_rows == 0) return COGDL_HIP_OK;
    if (!feat || !a_l || !a_r || !h_l || !h_r) return COGDL_HIP_EINVAL;
    if (heads > 0x7fffffff || f_dim > 0x7fffffff || n_rows * heads > ((int64_t)0x7fffffff) * 256) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32: return head_proj_typed<float>(feat, a_l, a_r, h_l, h_r, n_rows, heads, f_dim, s);
        case COGDL_HIP_F16: return head_proj_typed<__half>(feat, a_l, a_r, h_l, h_r, n_rows, heads, f_dim, s);
        case COGDL_HIP_BF16: return head_proj_typed<__hip_bfloat16>(feat, a_l, a_r, h_l, h_r, n_rows, heads, f_dim, s);
        default: return COGDL_HIP_EDTYPE;
    }
}
