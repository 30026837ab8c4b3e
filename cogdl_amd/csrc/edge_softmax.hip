// edge_softmax.hip -- per-(destination row, head) softmax over incoming edges, gfx950.
// Replaces cogdl/operators/edge_softmax/edge_softmax.cu:7-98 (one 32 x H block per row, three
// passes over the row, H <= 32).
//
// Layout fact that shapes the kernel: values is [E, H] row-major and a CSR row owns a
// CONTIGUOUS run of deg*H floats.  For H a power of two <= 64 a group of LPR lanes (LPR a
// multiple of H) streams that run with unit stride -- lane l always sees head l % H -- so
// loads/stores are fully coalesced and the per-head reduction is a wave64 butterfly over the
// lane strides LPR/2 ... H.  One pass computes an online (max, sum) pair, a second pass
// (L1/L2-resident re-read of the same run) writes exp(v - max) / sum: algorithmic HBM bytes
// = E*H*4 read + E*H*4 written + rowptr.  Rows are packed 64/LPR per wave so short rows
// (arxiv: deg 15) do not idle most of a wave; LPR is chosen from the mean run length.
// Any other H (not a power of two, or > 64) takes the generic kernel: lanes own heads.
#include "common.h"

namespace cogdl {

struct MaxSum {
    float m, s;
};

__device__ __forceinline__ MaxSum combine(MaxSum a, MaxSum b) {
    const float m = fmaxf(a.m, b.m);
    // exp(-inf - -inf) guards: an empty partial has s == 0 and m == -inf
    const float sa = (a.s == 0.f) ? 0.f : a.s * expf(a.m - m);
    const float sb = (b.s == 0.f) ? 0.f : b.s * expf(b.m - m);
    return {m, sa + sb};
}

// Reduce over the lanes of a group that share the same head: strides LPR/2 ... H.
template <int LPR>
__device__ __forceinline__ MaxSum head_reduce_maxsum(MaxSum v, int h) {
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1) {
        if (s >= h) {
            MaxSum o{__shfl_xor(v.m, s, kWave), __shfl_xor(v.s, s, kWave)};
            v = combine(v, o);
        }
    }
    return v;
}
template <int LPR>
__device__ __forceinline__ float head_reduce_sum(float v, int h) {
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1)
        if (s >= h) v += __shfl_xor(v, s, kWave);
    return v;
}

template <int LPR, bool BACKWARD>
__global__ __launch_bounds__(256) void edge_softmax_pow2_kernel(const int32_t *__restrict__ rowptr,
                                                                const float *__restrict__ a,  // values | softmax
                                                                const float *__restrict__ g,  // unused | grad
                                                                float *__restrict__ out, int64_t m, int h,
                                                                XcdMap n_rowblocks) {
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int l = lane % LPR;
    const int64_t row = rb * RPB + (threadIdx.x >> 6) * RPW + lane / LPR;
    int64_t lo = 0, hi = 0;
    if (row < m) {
        lo = (int64_t)rowptr[row] * h;
        hi = (int64_t)rowptr[row + 1] * h;
    }
    if constexpr (!BACKWARD) {
        MaxSum acc{-INFINITY, 0.f};
        for (int64_t i = lo + l; i < hi; i += LPR) {
            const float v = a[i];
            const float mn = fmaxf(acc.m, v);
            acc.s = ((acc.s == 0.f) ? 0.f : acc.s * expf(acc.m - mn)) + expf(v - mn);
            acc.m = mn;
        }
        acc = head_reduce_maxsum<LPR>(acc, h);
        const float inv = 1.f / acc.s;
        for (int64_t i = lo + l; i < hi; i += LPR) out[i] = expf(a[i] - acc.m) * inv;
    } else {
        float dot = 0.f;
        for (int64_t i = lo + l; i < hi; i += LPR) dot = fmaf(a[i], g[i], dot);
        dot = head_reduce_sum<LPR>(dot, h);
        for (int64_t i = lo + l; i < hi; i += LPR) out[i] = a[i] * (g[i] - dot);
    }
}

// Generic H: one wave per row, lane owns heads lane, lane+64, ...; edges walked sequentially
// (loads are coalesced across heads).
template <bool BACKWARD>
__global__ __launch_bounds__(256) void edge_softmax_generic_kernel(const int32_t *__restrict__ rowptr,
                                                                   const float *__restrict__ a,
                                                                   const float *__restrict__ g,
                                                                   float *__restrict__ out, int64_t m, int h,
                                                                   XcdMap n_rowblocks) {
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = rb * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int64_t lo = rowptr[row], hi = rowptr[row + 1];
    for (int hd = lane; hd < h; hd += kWave) {
        if constexpr (!BACKWARD) {
            float mx = -INFINITY;
            for (int64_t e = lo; e < hi; ++e) mx = fmaxf(mx, a[e * h + hd]);
            float s = 0.f;
            for (int64_t e = lo; e < hi; ++e) s += expf(a[e * h + hd] - mx);
            const float inv = 1.f / s;
            for (int64_t e = lo; e < hi; ++e) out[e * h + hd] = expf(a[e * h + hd] - mx) * inv;
        } else {
            float dot = 0.f;
            for (int64_t e = lo; e < hi; ++e) dot = fmaf(a[e * h + hd], g[e * h + hd], dot);
            for (int64_t e = lo; e < hi; ++e) out[e * h + hd] = a[e * h + hd] * (g[e * h + hd] - dot);
        }
    }
}

template <int LPR, bool BACKWARD>
static int launch_pow2(const int32_t *rowptr, const float *a, const float *g, float *out, int64_t m, int h,
                       hipStream_t s) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int64_t nrb = (m + RPB - 1) / RPB;
    if (nrb > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((edge_softmax_pow2_kernel<LPR, BACKWARD>), dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, rowptr, a, g,
                       out, m, h, make_xcd_map(nrb));
    return launch_status();
}

template <bool BACKWARD>
static int edge_softmax_dispatch(const int32_t *rowptr, const float *a, const float *g, float *out, int64_t m,
                                 int64_t nnz, int64_t h, hipStream_t s) {
    if (m < 0 || h < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (m == 0 || h == 0 || nnz == 0) return COGDL_HIP_OK;
    if (!rowptr || !a || !out || (BACKWARD && !g)) return COGDL_HIP_EINVAL;
    if (h > 0x7fffffff) return COGDL_HIP_ERANGE;
    const bool pow2 = (h & (h - 1)) == 0 && h <= kWave;
    if (!pow2) {
        const int64_t nrb = (m + 3) / 4;
        if (nrb > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
        hipLaunchKernelGGL((edge_softmax_generic_kernel<BACKWARD>), dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, rowptr, a,
                           g, out, m, (int)h, make_xcd_map(nrb));
        return launch_status();
    }
    // lanes per row ~ mean run length (deg*H), at least H and 8, at most 64
    const int64_t run = (nnz * h + m - 1) / m;
    int lpr = 8;
    while (lpr < 64 && (lpr < h || lpr < run)) lpr <<= 1;
    switch (lpr) {
        case 8: return launch_pow2<8, BACKWARD>(rowptr, a, g, out, m, (int)h, s);
        case 16: return launch_pow2<16, BACKWARD>(rowptr, a, g, out, m, (int)h, s);
        case 32: return launch_pow2<32, BACKWARD>(rowptr, a, g, out, m, (int)h, s);
        default: return launch_pow2<64, BACKWARD>(rowptr, a, g, out, m, (int)h, s);
    }
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_edge_softmax_fwd(const int32_t *rowptr, const float *values, float *out, int64_t m,
                                          int64_t nnz, int64_t h, void *stream) {
    return edge_softmax_dispatch<false>(rowptr, values, nullptr, out, m, nnz, h, (hipStream_t)stream);
}

extern "C" int cogdl_hip_edge_softmax_bwd(const int32_t *rowptr, const float *softmax, const float *grad,
                                          float *grad_in, int64_t m, int64_t nnz, int64_t h, void *stream) {
    return edge_softmax_dispatch<true>(rowptr, softmax, grad, grad_in, m, nnz, h, (hipStream_t)stream);
}
