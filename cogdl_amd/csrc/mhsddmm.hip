// mhsddmm.hip -- multi-head SDDMM for gfx950: out[e,h] = < grad[row(e),h,:], feat[colind[e],h,:] >.
// Replaces cogdl/operators/spmm/multiheadSddmm.cu:6-113 (binary-searched rows, warp=32).
// Row-group decomposition as in spmm.hip/sddmm.hip: a group of LPR lanes owns one CSR row and
// the whole [H*F] feature row (VEC columns per lane, all inside one head); grad[row] stays in
// registers; each neighbour row is one coalesced load; the F/VEC lanes of a head reduce their
// partial dot with a butterfly and the head's first lane stores out[e,h].
#include "common.h"

namespace cogdl {

template <int VEC, int LPR, int UNROLL>
__global__ __launch_bounds__(256) void mhsddmm_kernel(const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ colind,
                                                      const float *__restrict__ grad,
                                                      const float *__restrict__ feat, float *__restrict__ out,
                                                      int64_t m, int heads, int fdim, int lph /* lanes per head */,
                                                      XcdMap n_rowblocks) {
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int64_t row = rb * RPB + (threadIdx.x >> 6) * RPW + sub;
    const int k = heads * fdim;
    const bool col_ok = l * VEC < k;
    const int c0 = col_ok ? l * VEC : 0;
    const int hd = c0 / fdim;
    int start = 0, end = 0;
    if (row < m) {
        start = rowptr[row];
        end = rowptr[row + 1];
    }
    float a[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] = 0.f;
    if (row < m && col_ok) load_vec<float, VEC>(grad + row * (int64_t)k + c0, a);

    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_c = (l < cnt) ? colind[base + l] : 0;
        for (int j = 0; j < cnt; j += UNROLL) {
            float b[UNROLL][VEC];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int c = __shfl(my_c, sub * LPR + min(j + u, cnt - 1), kWave);
                load_vec<float, VEC>(feat + (int64_t)c * k + c0, b[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float p = 0.f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) p = fmaf(a[i], col_ok ? b[u][i] : 0.f, p);
#pragma unroll
                for (int s = LPR / 2; s > 0; s >>= 1)
                    if (s < lph) p += __shfl_xor(p, s, kWave);
                if (j + u < cnt && col_ok && (l % lph) == 0) out[(int64_t)(base + j + u) * heads + hd] = p;
            }
        }
    }
}

// Any shape: a group of 64 lanes per row; lane t walks (edge, head) pairs and dots F terms serially.
__global__ __launch_bounds__(256) void mhsddmm_generic_kernel(const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ colind,
                                                              const float *__restrict__ grad,
                                                              const float *__restrict__ feat,
                                                              float *__restrict__ out, int64_t m, int heads,
                                                              int fdim, XcdMap n_rowblocks) {
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = rb * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int64_t lo = rowptr[row], hi = rowptr[row + 1];
    const int64_t k = (int64_t)heads * fdim;
    for (int64_t p = lo * heads + lane; p < hi * heads; p += kWave) {
        const int64_t e = p / heads;
        const int hd = (int)(p - e * heads);
        const float *a = grad + row * k + (int64_t)hd * fdim;
        const float *b = feat + (int64_t)colind[e] * k + (int64_t)hd * fdim;
        float acc = 0.f;
        for (int t = 0; t < fdim; ++t) acc = fmaf(a[t], b[t], acc);
        out[p] = acc;
    }
}

template <int VEC, int LPR>
static int launch_mhsddmm(const int32_t *rowptr, const int32_t *colind, const float *grad, const float *feat,
                          float *out, int64_t m, int64_t h, int64_t f, hipStream_t s) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int64_t nrb = (m + RPB - 1) / RPB;
    if (nrb > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((mhsddmm_kernel<VEC, LPR, 4>), dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, rowptr, colind, grad,
                       feat, out, m, (int)h, (int)f, (int)(f / VEC), make_xcd_map(nrb));
    return launch_status();
}

template <int VEC>
static int dispatch_mhsddmm(const int32_t *rowptr, const int32_t *colind, const float *grad, const float *feat,
                            float *out, int64_t m, int64_t h, int64_t f, hipStream_t s) {
    const int64_t need = h * f / VEC;
    if (need <= 8) return launch_mhsddmm<VEC, 8>(rowptr, colind, grad, feat, out, m, h, f, s);
    if (need <= 16) return launch_mhsddmm<VEC, 16>(rowptr, colind, grad, feat, out, m, h, f, s);
    if (need <= 32) return launch_mhsddmm<VEC, 32>(rowptr, colind, grad, feat, out, m, h, f, s);
    return launch_mhsddmm<VEC, 64>(rowptr, colind, grad, feat, out, m, h, f, s);
}

static bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_mhsddmm(const int32_t *rowptr, const int32_t *colind, const float *grad,
                                 const float *feat, float *out, int64_t v, int64_t h, int64_t f, void *stream) {
    if (v < 0 || h < 0 || f < 0) return COGDL_HIP_EINVAL;
    if (v == 0 || h == 0) return COGDL_HIP_OK;
    if (!rowptr || !grad || !feat || !out || f == 0) return COGDL_HIP_EINVAL;
    if (h * f > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    const bool al16 = aligned_to(grad, 16) && aligned_to(feat, 16);
    const bool al8 = aligned_to(grad, 8) && aligned_to(feat, 8);
    // fast path: the whole [H*F] row fits one group of <= 64 lanes and F/VEC is a power of two
    if (f % 4 == 0 && al16 && is_pow2(f / 4) && h * f / 4 <= kWave)
        return dispatch_mhsddmm<4>(rowptr, colind, grad, feat, out, v, h, f, s);
    if (f % 2 == 0 && al8 && is_pow2(f / 2) && h * f / 2 <= kWave)
        return dispatch_mhsddmm<2>(rowptr, colind, grad, feat, out, v, h, f, s);
    if (is_pow2(f) && h * f <= kWave) return dispatch_mhsddmm<1>(rowptr, colind, grad, feat, out, v, h, f, s);
    const int64_t nrb = (v + 3) / 4;
    if (nrb > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL(mhsddmm_generic_kernel, dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, rowptr, colind, grad, feat, out,
                       v, (int)h, (int)f, make_xcd_map(nrb));
    return launch_status();
}
