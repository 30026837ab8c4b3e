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
// Rows longer than the long-row threshold (hub nodes: Reddit has rows of 10^4..10^5 edges) are skipped by both
// kernels and recorded in the chunk table of rowreduce.h; three small kernels then treat them chunk-parallel, for
// any H: per-piece statistics ((max, sum) forward, <softmax, grad> backward) by a whole workgroup, a per-row merge
// in chunk order, and the normalising write of every piece.
#include "rowreduce.h"

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
                                                                XcdMap n_rowblocks, LongRows lr) {
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int l = lane % LPR;
    const int64_t row = rb * RPB + (threadIdx.x >> 6) * RPW + lane / LPR;
    int64_t lo = 0, hi = 0;
    if (row < m) {
        const int start = rowptr[row], end = rowptr[row + 1];
        if (mark_long_chunks<LPR>(lr, row, start, end, l)) return;  // long row: chunk-parallel kernels
        lo = (int64_t)start * h;
        hi = (int64_t)end * h;
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
                                                                   XcdMap n_rowblocks, LongRows lr) {
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = rb * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int64_t lo = rowptr[row], hi = rowptr[row + 1];
    if (mark_long_chunks<kWave>(lr, row, (int)lo, (int)hi, lane)) return;
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

// ---- long rows ---------------------------------------------------------------------------------------------
// Thread mapping of a piece (edges [lo, hi) of one row, values [lo*h, hi*h) contiguous): the 256 threads form
// R x C with C = the power of two >= min(h, 256); thread (r, ci) walks edges lo+r, lo+r+R, ... for the heads
// ci, ci+C, ... -- consecutive threads read consecutive heads (coalesced).  Records hold 2 floats per head.
__device__ __forceinline__ int es_cols(int h) {
    int c = 1;
    while (c < h && c < 256) c <<= 1;
    return c;
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void edge_softmax_long_stats_kernel(const int32_t *__restrict__ rowptr,
                                                                      const float *__restrict__ a,
                                                                      const float *__restrict__ g, int h,
                                                                      LongRows lr) {
    __shared__ float red_m[256], red_s[256];
    __shared__ int any_long;
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    const int64_t c_end = min(c_begin + lr.chunks_per_block, lr.n_chunks);
    if (!run_has_long_rows(lr, c_begin, c_end, &any_long)) return;
    const int C = es_cols(h), R = 256 / C;
    const int r = threadIdx.x / C, ci = threadIdx.x % C;
    for (int64_t c = c_begin; c < c_end; ++c) {
        for (int slot = 0; slot < 2; ++slot) {
            int32_t row;
            int lo, hi;
            if (!decode_piece(lr, rowptr, c, slot, row, lo, hi)) continue;
            float *rec = lr.partial + (2 * c + slot) * lr.rec_stride;
            for (int hd0 = 0; hd0 < h; hd0 += C) {
                const int hd = hd0 + ci;
                MaxSum acc{-INFINITY, 0.f};
                float dot = 0.f;
                if (hd < h) {
                    for (int64_t e = lo + r; e < hi; e += R) {
                        const float v = a[e * h + hd];
                        if constexpr (!BACKWARD) {
                            const float mn = fmaxf(acc.m, v);
                            acc.s = ((acc.s == 0.f) ? 0.f : acc.s * expf(acc.m - mn)) + expf(v - mn);
                            acc.m = mn;
                        } else {
                            dot = fmaf(v, g[e * h + hd], dot);
                        }
                    }
                }
                red_m[threadIdx.x] = BACKWARD ? dot : acc.m;
                red_s[threadIdx.x] = acc.s;
                __syncthreads();
                if (r == 0 && hd < h) {  // fixed order over r
                    for (int q = 1; q < R; ++q) {
                        if constexpr (!BACKWARD) acc = combine(acc, MaxSum{red_m[q * C + ci], red_s[q * C + ci]});
                        else dot += red_m[q * C + ci];
                    }
                    rec[2 * hd] = BACKWARD ? dot : acc.m;
                    rec[2 * hd + 1] = acc.s;
                }
                __syncthreads();
            }
        }
    }
}

// Per long row (found at its first full chunk c0): merge the piece records in chunk order; the row's final
// statistics replace the record of (c0, slot 0), where the apply kernel looks them up.
template <bool BACKWARD>
__global__ __launch_bounds__(256) void edge_softmax_long_merge_kernel(const int32_t *__restrict__ rowptr, int h,
                                                                      LongRows lr) {
    __shared__ int any_long;
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    const int64_t c_end = min(c_begin + lr.chunks_per_block, lr.n_chunks);
    if (!run_has_long_rows(lr, c_begin, c_end - 1, &any_long)) return;
    const int ch = lr.thresh;
    for (int64_t c = c_begin; c < c_end; ++c) {
        const int32_t row = lr.chunk_row[c];
        if (row < 0) continue;
        const int start = rowptr[row], end = rowptr[row + 1];
        if ((int64_t)(start + ch - 1) / ch != c) continue;  // not the row's first owned chunk
        const int64_t c_last = (end - 1) / ch;
        const bool head = (int64_t)start != c * ch;
        for (int hd = threadIdx.x; hd < h; hd += blockDim.x) {
            MaxSum acc{-INFINITY, 0.f};
            float dot = 0.f;
            if (head) {
                const float *p = lr.partial + (2 * (c - 1) + 1) * lr.rec_stride + 2 * hd;
                if constexpr (!BACKWARD) acc = combine(acc, MaxSum{p[0], p[1]});
                else dot += p[0];
            }
            for (int64_t q = c; q <= c_last; ++q) {
                const float *p = lr.partial + (2 * q) * lr.rec_stride + 2 * hd;
                if constexpr (!BACKWARD) acc = combine(acc, MaxSum{p[0], p[1]});
                else dot += p[0];
            }
            float *dst = lr.partial + (2 * c) * lr.rec_stride + 2 * hd;
            dst[0] = BACKWARD ? dot : acc.m;
            dst[1] = BACKWARD ? 0.f : 1.f / acc.s;
        }
    }
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void edge_softmax_long_apply_kernel(const int32_t *__restrict__ rowptr,
                                                                      const float *__restrict__ a,
                                                                      const float *__restrict__ g,
                                                                      float *__restrict__ out, int h, LongRows lr) {
    __shared__ int any_long;
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    const int64_t c_end = min(c_begin + lr.chunks_per_block, lr.n_chunks);
    if (!run_has_long_rows(lr, c_begin, c_end, &any_long)) return;
    const int C = es_cols(h), R = 256 / C;
    const int r = threadIdx.x / C, ci = threadIdx.x % C;
    const int ch = lr.thresh;
    for (int64_t c = c_begin; c < c_end; ++c) {
        for (int slot = 0; slot < 2; ++slot) {
            int32_t row;
            int lo, hi;
            if (!decode_piece(lr, rowptr, c, slot, row, lo, hi)) continue;
            const int64_t c0 = ((int64_t)rowptr[row] + ch - 1) / ch;  // the row's first full chunk holds its statistics
            const float *stat = lr.partial + (2 * c0) * lr.rec_stride;
            for (int hd = ci; hd < h; hd += C) {
                const float s0 = stat[2 * hd], s1 = stat[2 * hd + 1];
                for (int64_t e = lo + r; e < hi; e += R) {
                    const int64_t i = e * h + hd;
                    if constexpr (!BACKWARD) out[i] = expf(a[i] - s0) * s1;
                    else out[i] = a[i] * (g[i] - s0);
                }
            }
        }
    }
}

template <bool BACKWARD>
static void launch_long(const int32_t *rowptr, const float *a, const float *g, float *out, int h, const LongRows &lr,
                        hipStream_t s) {
    if (lr.n_chunks == 0) return;
    const unsigned gx = (unsigned)((lr.n_chunks + lr.chunks_per_block - 1) / lr.chunks_per_block);
    hipLaunchKernelGGL((edge_softmax_long_stats_kernel<BACKWARD>), dim3(gx), dim3(256), 0, s, rowptr, a, g, h, lr);
    hipLaunchKernelGGL((edge_softmax_long_merge_kernel<BACKWARD>), dim3(gx), dim3(256), 0, s, rowptr, h, lr);
    hipLaunchKernelGGL((edge_softmax_long_apply_kernel<BACKWARD>), dim3(gx), dim3(256), 0, s, rowptr, a, g, out, h, lr);
}

template <int LPR, bool BACKWARD>
static int launch_pow2(const int32_t *rowptr, const float *a, const float *g, float *out, int64_t m, int h,
                       const LongRows &lr, hipStream_t s) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int64_t nrb = (m + RPB - 1) / RPB;
    if (nrb > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((edge_softmax_pow2_kernel<LPR, BACKWARD>), dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, rowptr, a, g,
                       out, m, h, make_xcd_map(nrb), lr);
    launch_long<BACKWARD>(rowptr, a, g, out, h, lr, s);
    return launch_status();
}

template <bool BACKWARD>
static int edge_softmax_dispatch(const int32_t *rowptr, const float *a, const float *g, float *out, int64_t m,
                                 int64_t nnz, int64_t h, void *ws, size_t wsb, hipStream_t s) {
    if (m < 0 || h < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (m == 0 || h == 0 || nnz == 0) return COGDL_HIP_OK;
    if (!rowptr || !a || !out || (BACKWARD && !g)) return COGDL_HIP_EINVAL;
    if (h > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    LongRows lr{};
    lr.thresh = INT_MAX;
    if (ws) {
        const int t = pick_long_thresh(nnz);
        const int64_t n_chunks = n_chunks_for(nnz, t);
        if (wsb < rowreduce_workspace_bytes(nnz, 2 * h)) return COGDL_HIP_EWORKSPACE;
        if (!aligned_to(ws, 256)) return COGDL_HIP_EALIGN;
        lr.thresh = t;
        lr.chunk_row = (int32_t *)ws;
        lr.partial = (float *)((char *)ws + chunk_row_bytes(n_chunks));
        lr.n_chunks = n_chunks;
        lr.rec_stride = 2 * h;
        const int64_t max_wg = std::max(64, g_tuning[kTuneLongGrid]);
        lr.chunks_per_block = (int)((n_chunks + max_wg - 1) / max_wg);
    }
    const bool pow2 = (h & (h - 1)) == 0 && h <= kWave;
    if (!pow2) {
        const int64_t nrb = (m + 3) / 4;
        if (nrb > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
        hipLaunchKernelGGL((edge_softmax_generic_kernel<BACKWARD>), dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, rowptr, a,
                           g, out, m, (int)h, make_xcd_map(nrb), lr);
        launch_long<BACKWARD>(rowptr, a, g, out, (int)h, lr, s);
        return launch_status();
    }
    // lanes per row ~ mean run length (deg*H), at least H and 8, at most 64
    const int64_t run = (nnz * h + m - 1) / m;
    int lpr = 8;
    while (lpr < 64 && (lpr < h || lpr < run)) lpr <<= 1;
    switch (lpr) {
        case 8: return launch_pow2<8, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s);
        case 16: return launch_pow2<16, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s);
        case 32: return launch_pow2<32, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s);
        default: return launch_pow2<64, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s);
    }
}

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_edge_softmax_workspace_bytes(int64_t nnz, int64_t h) {
    if (nnz <= 0 || h <= 0) return 0;
    return rowreduce_workspace_bytes(nnz, 2 * h);
}

extern "C" int cogdl_hip_edge_softmax_fwd(const int32_t *rowptr, const float *values, float *out, int64_t m,
                                          int64_t nnz, int64_t h, void *workspace, size_t workspace_bytes,
                                          void *stream) {
    return edge_softmax_dispatch<false>(rowptr, values, nullptr, out, m, nnz, h, workspace, workspace_bytes,
                                        (hipStream_t)stream);
}

extern "C" int cogdl_hip_edge_softmax_bwd(const int32_t *rowptr, const float *softmax, const float *grad,
                                          float *grad_in, int64_t m, int64_t nnz, int64_t h, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    return edge_softmax_dispatch<true>(rowptr, softmax, grad, grad_in, m, nnz, h, workspace, workspace_bytes,
                                       (hipStream_t)stream);
}
