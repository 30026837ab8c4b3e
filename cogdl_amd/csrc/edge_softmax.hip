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
// Rows longer than the long-row threshold (hub nodes: Reddit has rows of 10^4..10^5 edges) are skipped by the row
// blocks; the LEADING workgroups of the same launch treat them chunk-parallel (rowreduce.h bookkeeping), for any H:
// per-piece statistics ((max, sum) forward, <softmax, grad> backward) by a whole workgroup; a second kernel merges a
// row's piece records in chunk order (each piece workgroup redoes that tiny merge itself) and writes the piece.
#include "rowreduce.h"

#include <type_traits>

namespace cogdl {

// exp of a non-positive difference (value - running max): the hardware exponential (v_exp_f32 of x * log2 e, 1 ulp)
// -- its relative error |x| * 2^-24 is below 2e-6 wherever the result is not negligible; the libm expf costs ~10x the
// instructions and made the forward VALU-bound (2.25 exponentials per element).
__device__ __forceinline__ float es_exp(float x) { return __expf(x); }

struct MaxSum {
    float m, s;
};

__device__ __forceinline__ MaxSum combine(MaxSum a, MaxSum b) {
    const float m = fmaxf(a.m, b.m);
    // exp(-inf - -inf) guards: an empty partial has s == 0 and m == -inf
    const float sa = (a.s == 0.f) ? 0.f : a.s * es_exp(a.m - m);
    const float sb = (b.s == 0.f) ? 0.f : b.s * es_exp(b.m - m);
    return {m, sa + sb};
}

// Reduce over the lanes of a group that share the same head: strides LPR/2 ... H.
template <int LPR>
__device__ __forceinline__ MaxSum head_reduce_maxsum(MaxSum v, int h) {
    // butterfly of the maxima first, ONE rescale of the lane's sum to the common maximum, then plain sums
    float m = v.m;
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1)
        if (s >= h) m = fmaxf(m, __shfl_xor(m, s, kWave));
    float sum = (v.s == 0.f) ? 0.f : v.s * es_exp(v.m - m);
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1)
        if (s >= h) sum += __shfl_xor(sum, s, kWave);
    return {m, sum};
}
template <int LPR>
__device__ __forceinline__ float head_reduce_sum(float v, int h) {
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1)
        if (s >= h) v += __shfl_xor(v, s, kWave);
    return v;
}

// ---- strided element walks, kEsUnroll independent loads in flight -------------------------------------------------
// Every kernel below walks elements idx = i0, i0 + stride, ... < end of `a` (and `g`).  The loads of a batch are
// issued back to back; the forward statistic takes ONE rescale per batch (batch max first), i.e. U+1 exps per U
// elements instead of the 2U of an element-wise online softmax.
constexpr int kEsUnroll = 4;

// (T = float, __half or __hip_bfloat16: values are read and written in their own type, all arithmetic is fp32)
template <typename T>
__device__ __forceinline__ MaxSum maxsum_strided(const T *__restrict__ a, int64_t i0, int64_t end, int64_t stride) {
    if (i0 + stride >= end) {  // at most one element for this lane (short rows: the common case on small graphs)
        if (i0 < end) return {to_f32<T>(a[i0]), 1.f};
        return {-INFINITY, 0.f};
    }
    MaxSum acc{-INFINITY, 0.f};
    for (; i0 < end; i0 += stride * kEsUnroll) {
        float v[kEsUnroll];
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            v[u] = (idx < end) ? to_f32<T>(a[idx]) : -INFINITY;
        }
        float bm = v[0];
#pragma unroll
        for (int u = 1; u < kEsUnroll; ++u) bm = fmaxf(bm, v[u]);
        const float mn = fmaxf(acc.m, bm);
        float ssum = 0.f;
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) ssum += es_exp(v[u] - mn);  // masked slots: exp(-inf) == 0
        acc.s = ((acc.s == 0.f) ? 0.f : acc.s * es_exp(acc.m - mn)) + ssum;
        acc.m = mn;
    }
    return acc;
}

template <typename T>
__device__ __forceinline__ float dot_strided(const T *__restrict__ a, const T *__restrict__ g, int64_t i0,
                                             int64_t end, int64_t stride) {
    float dot = 0.f;
    for (; i0 < end; i0 += stride * kEsUnroll) {
        float v[kEsUnroll], w[kEsUnroll];
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            const bool ok = idx < end;
            v[u] = ok ? to_f32<T>(a[idx]) : 0.f;
            w[u] = ok ? to_f32<T>(g[idx]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) dot = fmaf(v[u], w[u], dot);
    }
    return dot;
}

// out = exp(a - mx) * inv   (forward)      out = a * (g - dot)   (backward; mx carries dot)
template <bool BACKWARD, typename T>
__device__ __forceinline__ void write_strided(const T *__restrict__ a, const T *__restrict__ g,
                                              T *__restrict__ out, int64_t i0, int64_t end, int64_t stride,
                                              float mx, float inv) {
    if (i0 + stride >= end) {
        if (i0 < end) {
            const float av = to_f32<T>(a[i0]);
            float r;
            if constexpr (BACKWARD) r = av * (to_f32<T>(g[i0]) - mx);
            else r = es_exp(av - mx) * inv;
            out[i0] = from_f32<T>(r);
        }
        return;
    }
    for (; i0 < end; i0 += stride * kEsUnroll) {
        float v[kEsUnroll], w[kEsUnroll];
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            const bool ok = idx < end;
            v[u] = ok ? to_f32<T>(a[idx]) : 0.f;
            if constexpr (BACKWARD) w[u] = ok ? to_f32<T>(g[idx]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            if (idx < end) out[idx] = from_f32<T>(BACKWARD ? v[u] * (w[u] - mx) : es_exp(v[u] - mx) * inv);
        }
    }
}

// ---- the same walks with 16-byte lanes (H a multiple of 4: every row's run starts 16-byte aligned) ----------------
// Lane state: one (max, sum) / one dot per component, i.e. per head (4 l + c) % H.
struct MaxSum4 {
    float m[4], s[4];
};
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void get4(const float4 &q, float (&v)[4]) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }

__device__ __forceinline__ MaxSum4 maxsum_strided4(const float *__restrict__ a, int64_t i0, int64_t end, int64_t stride) {
    MaxSum4 acc;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc.m[c] = -INFINITY;
        acc.s[c] = 0.f;
    }
    if (i0 + stride >= end) {  // at most one vector for this lane
        if (i0 < end) {
            get4(ld4(a + i0), acc.m);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc.s[c] = 1.f;
        }
        return acc;
    }
    for (; i0 < end; i0 += stride * kEsUnroll) {
        float v[kEsUnroll][4];
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            if (idx < end) get4(ld4(a + idx), v[u]);
            else
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = -INFINITY;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float bm = v[0][c];
#pragma unroll
            for (int u = 1; u < kEsUnroll; ++u) bm = fmaxf(bm, v[u][c]);
            const float mn = fmaxf(acc.m[c], bm);
            float ssum = 0.f;
#pragma unroll
            for (int u = 0; u < kEsUnroll; ++u) ssum += es_exp(v[u][c] - mn);
            acc.s[c] = ((acc.s[c] == 0.f) ? 0.f : acc.s[c] * es_exp(acc.m[c] - mn)) + ssum;
            acc.m[c] = mn;
        }
    }
    return acc;
}

__device__ __forceinline__ void dot_strided4(const float *__restrict__ a, const float *__restrict__ g, int64_t i0,
                                             int64_t end, int64_t stride, float (&dot)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dot[c] = 0.f;
    for (; i0 < end; i0 += stride * kEsUnroll) {
        float v[kEsUnroll][4], w[kEsUnroll][4];
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            if (idx < end) {
                get4(ld4(a + idx), v[u]);
                get4(ld4(g + idx), w[u]);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = w[u][c] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) dot[c] = fmaf(v[u][c], w[u][c], dot[c]);
    }
}

template <bool BACKWARD>
__device__ __forceinline__ void write_strided4(const float *__restrict__ a, const float *__restrict__ g,
                                               float *__restrict__ out, int64_t i0, int64_t end, int64_t stride,
                                               const float (&mx)[4], const float (&inv)[4]) {
    for (; i0 < end; i0 += stride * kEsUnroll) {
        float v[kEsUnroll][4], w[kEsUnroll][4];
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            if (idx < end) {
                get4(ld4(a + idx), v[u]);
                if constexpr (BACKWARD) get4(ld4(g + idx), w[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kEsUnroll; ++u) {
            const int64_t idx = i0 + u * stride;
            if (idx < end) {
                float4 o;
                if constexpr (BACKWARD) o = make_float4(v[u][0] * (w[u][0] - mx[0]), v[u][1] * (w[u][1] - mx[1]),
                                                        v[u][2] * (w[u][2] - mx[2]), v[u][3] * (w[u][3] - mx[3]));
                else o = make_float4(es_exp(v[u][0] - mx[0]) * inv[0], es_exp(v[u][1] - mx[1]) * inv[1],
                                     es_exp(v[u][2] - mx[2]) * inv[2], es_exp(v[u][3] - mx[3]) * inv[3]);
                *reinterpret_cast<float4 *>(out + idx) = o;
            }
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

// 16-byte-lane variant of the piece statistics (H a power of two in [4, 64], 16-byte aligned operands): thread
// (r, ci) reads the heads 4 ci .. 4 ci + 3 of the edges lo + r, lo + r + R, ... with R = 256 / (H / 4) -- four times the
// bytes in flight of the 4-byte walk, which is what bounds a piece (a hub row's piece is a handful of dependent batches).
template <bool BACKWARD>
__device__ __forceinline__ void edge_softmax_long_stats_block4(const int32_t *__restrict__ rowptr,
                                                               const float *__restrict__ a,
                                                               const float *__restrict__ g, int64_t m, int h,
                                                               const LongRows &lr) {
    __shared__ float red_m[4 * kWave], red_s[4 * kWave];  // [wave][head]
    __shared__ int32_t tbl[kMaxChunksPerBlock + 1];
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    if (c_begin >= lr.n_chunks) return;
    const int n = (int)min((int64_t)lr.chunks_per_block, lr.n_chunks - c_begin);
    if (!build_chunk_table(lr, rowptr, m, c_begin, n + 1, tbl)) return;
    const int C = h / 4, R = 256 / C;  // C <= 16 lanes across the heads
    const int r = threadIdx.x / C, ci = threadIdx.x % C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    for (int t = 0; t < n; ++t) {
        const int64_t c = c_begin + t;
        for (int slot = 0; slot < 2; ++slot) {
            int32_t row;
            int lo, hi;
            if (!decode_piece(lr, rowptr, tbl, c, t, slot, row, lo, hi)) continue;
            float *rec = lr.partial + (2 * c + slot) * lr.rec_stride;
            const int64_t i0 = ((int64_t)lo + r) * h + 4 * ci, iend = (int64_t)hi * h, istr = (int64_t)R * h;
            MaxSum4 acc;
            float dot[4];
            if constexpr (!BACKWARD) acc = maxsum_strided4(a, i0, iend, istr);
            else dot_strided4(a, g, i0, iend, istr, dot);
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // lanes with equal ci: strides C .. 32 inside the wave
                if constexpr (!BACKWARD) {
                    const MaxSum v = head_reduce_maxsum<kWave>(MaxSum{acc.m[q], acc.s[q]}, C);
                    acc.m[q] = v.m;
                    acc.s[q] = v.s;
                } else {
                    dot[q] = head_reduce_sum<kWave>(dot[q], C);
                }
            }
            if (lane < C) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    red_m[wave * h + 4 * ci + q] = BACKWARD ? dot[q] : acc.m[q];
                    red_s[wave * h + 4 * ci + q] = BACKWARD ? 0.f : acc.s[q];
                }
            }
            __syncthreads();
            if ((int)threadIdx.x < h) {  // the four wave partials in wave order
                MaxSum tot{red_m[threadIdx.x], red_s[threadIdx.x]};
                float dsum = red_m[threadIdx.x];
                for (int q = 1; q < 4; ++q) {
                    if constexpr (!BACKWARD) tot = combine(tot, MaxSum{red_m[q * h + threadIdx.x], red_s[q * h + threadIdx.x]});
                    else dsum += red_m[q * h + threadIdx.x];
                }
                rec[2 * threadIdx.x] = BACKWARD ? dsum : tot.m;
                rec[2 * threadIdx.x + 1] = tot.s;
            }
            __syncthreads();
        }
    }
}

template <bool BACKWARD, typename T = float>
__device__ __forceinline__ void edge_softmax_long_stats_block(const int32_t *__restrict__ rowptr,
                                                              const T *__restrict__ a,
                                                              const T *__restrict__ g, int64_t m, int h,
                                                              const LongRows &lr) {
    __shared__ float red_m[256], red_s[256];
    __shared__ int32_t tbl[kMaxChunksPerBlock + 1];
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    if (c_begin >= lr.n_chunks) return;
    const int n = (int)min((int64_t)lr.chunks_per_block, lr.n_chunks - c_begin);
    if (!build_chunk_table(lr, rowptr, m, c_begin, n + 1, tbl)) return;
    const int C = es_cols(h), R = 256 / C;
    const int r = threadIdx.x / C, ci = threadIdx.x % C;
    for (int t = 0; t < n; ++t) {
        const int64_t c = c_begin + t;
        for (int slot = 0; slot < 2; ++slot) {
            int32_t row;
            int lo, hi;
            if (!decode_piece(lr, rowptr, tbl, c, t, slot, row, lo, hi)) continue;
            float *rec = lr.partial + (2 * c + slot) * lr.rec_stride;
            for (int hd0 = 0; hd0 < h; hd0 += C) {
                const int hd = hd0 + ci;
                MaxSum acc{-INFINITY, 0.f};
                float dot = 0.f;
                if (hd < h) {
                    const int64_t i0 = ((int64_t)lo + r) * h + hd, iend = (int64_t)hi * h, istr = (int64_t)R * h;
                    if constexpr (!BACKWARD) acc = maxsum_strided(a, i0, iend, istr);
                    else dot = dot_strided(a, g, i0, iend, istr);
                }
                // Reduce over the R threads that share head `hd`: a butterfly inside each wave over the lane strides
                // >= C (lanes with equal ci), then the (at most 4) wave partials through LDS in wave order.
                if (C < kWave) {
#pragma unroll
                    for (int sft = kWave / 2; sft > 0; sft >>= 1) {
                        if (sft >= C) {
                            if constexpr (!BACKWARD) {
                                MaxSum o{__shfl_xor(acc.m, sft, kWave), __shfl_xor(acc.s, sft, kWave)};
                                acc = combine(acc, o);
                            } else {
                                dot += __shfl_xor(dot, sft, kWave);
                            }
                        }
                    }
                }
                // after the butterfly every lane of a wave holds the wave's value for its ci
                const int per_wave = (C < kWave) ? C : kWave;          // distinct ci values inside one wave
                const int n_part = (C < kWave) ? 4 : R;                // partials per head to merge through LDS
                const int part = (C < kWave) ? (threadIdx.x >> 6) : r;  // which partial this thread holds
                const bool writer = (C < kWave) ? ((threadIdx.x & (kWave - 1)) < per_wave) : true;
                if (writer) {
                    red_m[part * C + ci] = BACKWARD ? dot : acc.m;
                    red_s[part * C + ci] = acc.s;
                }
                __syncthreads();
                if (threadIdx.x < C && hd0 + (int)threadIdx.x < h) {  // thread ci finishes head hd0 + ci
                    MaxSum tot{red_m[threadIdx.x], red_s[threadIdx.x]};
                    float dsum = red_m[threadIdx.x];
                    for (int q = 1; q < n_part; ++q) {
                        if constexpr (!BACKWARD) tot = combine(tot, MaxSum{red_m[q * C + threadIdx.x], red_s[q * C + threadIdx.x]});
                        else dsum += red_m[q * C + threadIdx.x];
                    }
                    rec[2 * (hd0 + threadIdx.x)] = BACKWARD ? dsum : tot.m;
                    rec[2 * (hd0 + threadIdx.x) + 1] = tot.s;
                }
                __syncthreads();
            }
        }
    }
}

// Second launch: every piece workgroup merges the records of ITS row in chunk order (a row of 10^5 edges has ~100
// records of 2H floats: cheaper than a third launch) and writes the piece's outputs.
template <bool BACKWARD, bool VEC4, typename T = float>
__global__ __launch_bounds__(256) void edge_softmax_long_apply_kernel(const int32_t *__restrict__ rowptr,
                                                                      const T *__restrict__ a,
                                                                      const T *__restrict__ g,
                                                                      T *__restrict__ out, int64_t m, int h,
                                                                      LongRows lr) {
    static_assert(!VEC4 || std::is_same<T, float>::value, "16-byte lanes: fp32 values only");
    __shared__ int32_t tbl[kMaxChunksPerBlock + 1];
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    if (c_begin >= lr.n_chunks) return;
    const int n = (int)min((int64_t)lr.chunks_per_block, lr.n_chunks - c_begin);
    if (!build_chunk_table(lr, rowptr, m, c_begin, n + 1, tbl)) return;
    const int C = VEC4 ? h / 4 : es_cols(h), R = 256 / C;
    const int r = threadIdx.x / C, ci = threadIdx.x % C;
    const int ch = lr.thresh;
    // A hub row spans many consecutive chunks of this workgroup's run: its merged statistics are computed once (per
    // head by one thread, records in chunk order) and kept in LDS while the pieces of the row are written.
    constexpr int kStatCap = 1024;  // heads cached in LDS; beyond that every thread merges for itself
    __shared__ float st_a[kStatCap], st_b[kStatCap];
    int32_t cached_row = -1;
    for (int t = 0; t < n; ++t) {
        const int64_t c = c_begin + t;
        for (int slot = 0; slot < 2; ++slot) {
            int32_t row;
            int lo, hi;
            if (!decode_piece(lr, rowptr, tbl, c, t, slot, row, lo, hi)) continue;
            const int start = rowptr[row], end = rowptr[row + 1];
            const int64_t c0 = ((int64_t)start + ch - 1) / ch, c_last = ((int64_t)end - 1) / ch;
            const bool head = (int64_t)start != c0 * ch;  // head piece: chunk c0-1, slot 1
            auto merged = [&](int hd, float &sa, float &sb) {
                MaxSum acc{-INFINITY, 0.f};
                float dot = 0.f;
                if (head) {
                    const float *p = lr.partial + (2 * (c0 - 1) + 1) * lr.rec_stride + 2 * hd;
                    if constexpr (!BACKWARD) acc = combine(acc, MaxSum{p[0], p[1]});
                    else dot += p[0];
                }
                for (int64_t q = c0; q <= c_last; ++q) {
                    const float *p = lr.partial + (2 * q) * lr.rec_stride + 2 * hd;
                    if constexpr (!BACKWARD) acc = combine(acc, MaxSum{p[0], p[1]});
                    else dot += p[0];
                }
                sa = BACKWARD ? dot : acc.m;
                sb = BACKWARD ? 0.f : 1.f / acc.s;
            };
            if (row != cached_row && h <= kStatCap) {  // workgroup-uniform
                __syncthreads();  // the previous row's readers are done
                for (int hd = threadIdx.x; hd < h; hd += blockDim.x) merged(hd, st_a[hd], st_b[hd]);
                __syncthreads();
                cached_row = row;
            }
            if constexpr (VEC4 && std::is_same<T, float>::value) {  // H a power of two in [4, 64]: 16-byte lanes, heads 4 ci .. 4 ci + 3
                float sa4[4], sb4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sa4[q] = st_a[4 * ci + q];
                    sb4[q] = st_b[4 * ci + q];
                }
                write_strided4<BACKWARD>(a, g, out, ((int64_t)lo + r) * h + 4 * ci, (int64_t)hi * h, (int64_t)R * h, sa4, sb4);
                continue;
            }
            for (int hd = ci; hd < h; hd += C) {
                float sa, sb;
                if (h <= kStatCap) {
                    sa = st_a[hd];
                    sb = st_b[hd];
                } else {
                    merged(hd, sa, sb);
                }
                write_strided<BACKWARD>(a, g, out, ((int64_t)lo + r) * h + hd, (int64_t)hi * h, (int64_t)R * h, sa, sb);
            }
        }
    }
}

template <int LPR, bool BACKWARD>
__global__ __launch_bounds__(256) void edge_softmax_pow2_kernel(const int32_t *__restrict__ rowptr,
                                                                const float *__restrict__ a,  // values | softmax
                                                                const float *__restrict__ g,  // unused | grad
                                                                float *__restrict__ out, int64_t m, int h,
                                                                XcdMap n_rowblocks, LongRows lr, int long_vec4) {
    if (blockIdx.x < lr.n_long_blocks) {
        if (long_vec4) edge_softmax_long_stats_block4<BACKWARD>(rowptr, a, g, m, h, lr);
        else edge_softmax_long_stats_block<BACKWARD>(rowptr, a, g, m, h, lr);
        return;
    }
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x - lr.n_long_blocks, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int l = lane % LPR;
    const int64_t row = rb * RPB + (threadIdx.x >> 6) * RPW + lane / LPR;
    int64_t lo = 0, hi = 0;
    if (row < m) {
        const int start = rowptr[row], end = rowptr[row + 1];
        if (end - start > lr.thresh) return;  // long row: the long-row workgroups compute it
        lo = (int64_t)start * h;
        hi = (int64_t)end * h;
    }
    if constexpr (!BACKWARD) {
        const MaxSum acc = head_reduce_maxsum<LPR>(maxsum_strided(a, lo + l, hi, LPR), h);
        write_strided<false>(a, g, out, lo + l, hi, LPR, acc.m, 1.f / acc.s);
    } else {
        const float dot = head_reduce_sum<LPR>(dot_strided(a, g, lo + l, hi, LPR), h);
        write_strided<true>(a, g, out, lo + l, hi, LPR, dot, 0.f);
    }
}

// H a power of two in [4, 64] and 16-byte aligned operands: 16-byte lanes.  Lane l of the group reads the elements
// 4 l .. 4 l + 3 (+ multiples of 4 LPR) of the row's run, i.e. always the heads (4 l + c) % H; lanes with equal l % (H/4)
// share their heads, so the per-head reduction is a butterfly over the lane strides >= H/4.
template <int LPR, bool BACKWARD>
__global__ __launch_bounds__(256) void edge_softmax_vec4_kernel(const int32_t *__restrict__ rowptr,
                                                                const float *__restrict__ a,
                                                                const float *__restrict__ g, float *__restrict__ out,
                                                                int64_t m, int h, XcdMap n_rowblocks, LongRows lr, int long_vec4) {
    if (blockIdx.x < lr.n_long_blocks) {
        if (long_vec4) edge_softmax_long_stats_block4<BACKWARD>(rowptr, a, g, m, h, lr);
        else edge_softmax_long_stats_block<BACKWARD>(rowptr, a, g, m, h, lr);
        return;
    }
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x - lr.n_long_blocks, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int l = lane % LPR;
    const int64_t row = rb * RPB + (threadIdx.x >> 6) * RPW + lane / LPR;
    int64_t lo = 0, hi = 0;
    if (row < m) {
        const int start = rowptr[row], end = rowptr[row + 1];
        if (end - start > lr.thresh) return;  // long row: the long-row workgroups compute it
        lo = (int64_t)start * h;
        hi = (int64_t)end * h;
    }
    const int period = h / 4;  // lanes per pass over the heads
    if constexpr (!BACKWARD) {
        MaxSum4 acc = maxsum_strided4(a, lo + 4 * l, hi, 4 * LPR);
        float mx[4], inv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const MaxSum r = head_reduce_maxsum<LPR>(MaxSum{acc.m[c], acc.s[c]}, period);
            mx[c] = r.m;
            inv[c] = 1.f / r.s;
        }
        write_strided4<false>(a, g, out, lo + 4 * l, hi, 4 * LPR, mx, inv);
    } else {
        float dot[4], zero[4] = {0.f, 0.f, 0.f, 0.f};
        dot_strided4(a, g, lo + 4 * l, hi, 4 * LPR, dot);
#pragma unroll
        for (int c = 0; c < 4; ++c) dot[c] = head_reduce_sum<LPR>(dot[c], period);
        write_strided4<true>(a, g, out, lo + 4 * l, hi, 4 * LPR, dot, zero);
    }
}

// Generic H: one wave per row, lane owns heads lane, lane+64, ...; edges walked sequentially
// (loads are coalesced across heads).
template <bool BACKWARD, typename T = float>
__global__ __launch_bounds__(256) void edge_softmax_generic_kernel(const int32_t *__restrict__ rowptr,
                                                                   const T *__restrict__ a,
                                                                   const T *__restrict__ g,
                                                                   T *__restrict__ out, int64_t m, int h,
                                                                   XcdMap n_rowblocks, LongRows lr, int long_vec4) {
    if (blockIdx.x < lr.n_long_blocks) {
        if constexpr (std::is_same<T, float>::value) {
            if (long_vec4) {
                edge_softmax_long_stats_block4<BACKWARD>(rowptr, a, g, m, h, lr);
                return;
            }
        }
        edge_softmax_long_stats_block<BACKWARD, T>(rowptr, a, g, m, h, lr);
        return;
    }
    const int64_t rb = xcd_remap(blockIdx.x - lr.n_long_blocks, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = rb * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int64_t lo = rowptr[row], hi = rowptr[row + 1];
    if (hi - lo > lr.thresh) return;
    for (int hd = lane; hd < h; hd += kWave) {
        if constexpr (!BACKWARD) {
            const MaxSum acc = maxsum_strided<T>(a, lo * h + hd, hi * h, h);
            write_strided<false, T>(a, g, out, lo * h + hd, hi * h, h, acc.m, 1.f / acc.s);
        } else {
            const float dot = dot_strided<T>(a, g, lo * h + hd, hi * h, h);
            write_strided<true, T>(a, g, out, lo * h + hd, hi * h, h, dot, 0.f);
        }
    }
}

template <bool BACKWARD>
static void launch_long(const int32_t *rowptr, const float *a, const float *g, float *out, int64_t m, int h,
                        const LongRows &lr, hipStream_t s, bool vec4 = false) {
    if (lr.n_long_blocks == 0) return;
    if (vec4)
        hipLaunchKernelGGL((edge_softmax_long_apply_kernel<BACKWARD, true>), dim3(lr.n_long_blocks), dim3(256), 0, s, rowptr,
                           a, g, out, m, h, lr);
    else
        hipLaunchKernelGGL((edge_softmax_long_apply_kernel<BACKWARD, false>), dim3(lr.n_long_blocks), dim3(256), 0, s, rowptr,
                           a, g, out, m, h, lr);
}

template <int LPR, bool BACKWARD>
static int launch_vec4(const int32_t *rowptr, const float *a, const float *g, float *out, int64_t m, int h,
                       const LongRows &lr, hipStream_t s, int lv4) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int64_t nrb = (m + RPB - 1) / RPB;
    if (!grid_fits(make_xcd_map(nrb), 4096)) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((edge_softmax_vec4_kernel<LPR, BACKWARD>), dim3(lr.n_long_blocks + xcd_grid(make_xcd_map(nrb))), dim3(256),
                       0, s, rowptr, a, g, out, m, h, make_xcd_map(nrb), lr, lv4);
    launch_long<BACKWARD>(rowptr, a, g, out, m, h, lr, s, lv4 != 0);
    return launch_status();
}

template <int LPR, bool BACKWARD>
static int launch_pow2(const int32_t *rowptr, const float *a, const float *g, float *out, int64_t m, int h,
                       const LongRows &lr, hipStream_t s, int lv4) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int64_t nrb = (m + RPB - 1) / RPB;
    if (!grid_fits(make_xcd_map(nrb), 4096)) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((edge_softmax_pow2_kernel<LPR, BACKWARD>), dim3(lr.n_long_blocks + xcd_grid(make_xcd_map(nrb))), dim3(256),
                       0, s, rowptr, a, g, out, m, h, make_xcd_map(nrb), lr, lv4);
    launch_long<BACKWARD>(rowptr, a, g, out, m, h, lr, s, lv4 != 0);
    return launch_status();
}

// A piece of the long-row path is T edges x H values for a whole workgroup: with few heads the generic threshold makes
// pieces too small to amortise their two launches' bookkeeping (measured, reddit-shaped graph, H = 1: T 1024 -> 573 us,
// 2048 -> 469 us; arxiv R-MAT, H = 1: T 128 -> 128 us, 512 -> 72 us; H = 8 is best at the generic threshold).
static int es_thresh_scale(int64_t h) { return h >= 8 ? 1 : (h >= 4 ? 2 : 4); }

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
        if (wsb < rowreduce_workspace_bytes(nnz, 2 * h, es_thresh_scale(h))) return COGDL_HIP_EWORKSPACE;
        if (!aligned_to(ws, 256)) return COGDL_HIP_EALIGN;
        plan_long_rows(lr, nnz, es_thresh_scale(h));
        lr.partial = (float *)((char *)ws + kFoundBytes);
        lr.rec_stride = 2 * h;
    }
    const bool pow2 = (h & (h - 1)) == 0 && h <= kWave;
    if (!pow2) {
        const int64_t nrb = (m + 3) / 4;
        if (!grid_fits(make_xcd_map(nrb), 4096)) return COGDL_HIP_ERANGE;
        hipLaunchKernelGGL((edge_softmax_generic_kernel<BACKWARD>), dim3(lr.n_long_blocks + xcd_grid(make_xcd_map(nrb))), dim3(256),
                           0, s, rowptr, a, g, out, m, (int)h, make_xcd_map(nrb), lr, 0);
        launch_long<BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s);
        return launch_status();
    }
    // lanes per row ~ mean run length (deg*H), at least H and 8, at most 64
    const int64_t run = (nnz * h + m - 1) / m;
    // 16-byte lanes: always for the backward (its reduction is a plain sum); for the forward only on long runs -- with
    // one vector per lane every lane holds 4 different heads, and the (max, sum) butterfly then spans 4x more lane
    // strides than the 4-byte layout (measured, arxiv-shaped graph: H = 8 forward 101 -> 117 us, backward 82 -> 66 us;
    // reddit-shaped: forward 2.02 -> 1.89 ms).
    const bool vec4_pays = BACKWARD || run >= 1024;
    const bool vec4_layout = h >= 4 && aligned_to(a, 16) && aligned_to(out, 16) && (!BACKWARD || aligned_to(g, 16));
    // tuning key 7 (experiments): bit 0 = 4-byte lanes in the row kernels, bit 1 = 4-byte lanes in the hub-row path
    const bool vec4_ok = vec4_layout && (g_tuning[kTuneEsScalar] & 1) == 0;
    const int lv4 = (vec4_layout && (g_tuning[kTuneEsScalar] & 2) == 0) ? 1 : 0;  // hub rows: 16-byte lanes whenever
    if (vec4_ok && vec4_pays) {                                                   // the layout allows (bytes in flight)
        int lpr4 = 8;  // 16-byte lanes: a quarter of the lanes for the same run, at least H/4
        while (lpr4 < 64 && (lpr4 * 4 < h || lpr4 * 4 < run)) lpr4 <<= 1;
        switch (lpr4) {
            case 8: return launch_vec4<8, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
            case 16: return launch_vec4<16, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
            case 32: return launch_vec4<32, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
            default: return launch_vec4<64, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
        }
    }
    int lpr = 8;
    while (lpr < 64 && (lpr < h || lpr < run)) lpr <<= 1;
    switch (lpr) {
        case 8: return launch_pow2<8, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
        case 16: return launch_pow2<16, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
        case 32: return launch_pow2<32, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
        default: return launch_pow2<64, BACKWARD>(rowptr, a, g, out, m, (int)h, lr, s, lv4);
    }
}

// 2-byte values outside the flat kernel's coverage (H not a power of two, H > 64, no workspace, unaligned operands): the
// generic row kernel + the scalar hub-row path, instantiated for the value type -- values are read and written natively,
// no fp32 copies of the [E, H] tensors (rounds 1-3 converted them outside the kernel).
template <bool BACKWARD, typename T>
static int edge_softmax_dispatch16(const int32_t *rowptr, const T *a, const T *g, T *out, int64_t m, int64_t nnz, int64_t h,
                                   void *ws, size_t wsb, hipStream_t s) {
    LongRows lr{};
    lr.thresh = INT_MAX;
    if (ws) {
        if (wsb < rowreduce_workspace_bytes(nnz, 2 * h, es_thresh_scale(h))) return COGDL_HIP_EWORKSPACE;
        if (!aligned_to(ws, 256)) return COGDL_HIP_EALIGN;
        plan_long_rows(lr, nnz, es_thresh_scale(h));
        lr.partial = (float *)((char *)ws + kFoundBytes);
        lr.rec_stride = 2 * h;
    }
    const int64_t nrb = (m + 3) / 4;
    if (!grid_fits(make_xcd_map(nrb), 4096)) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((edge_softmax_generic_kernel<BACKWARD, T>), dim3(lr.n_long_blocks + xcd_grid(make_xcd_map(nrb))), dim3(256),
                       0, s, rowptr, a, g, out, m, (int)h, make_xcd_map(nrb), lr, 0);
    if (lr.n_long_blocks > 0)
        hipLaunchKernelGGL((edge_softmax_long_apply_kernel<BACKWARD, false, T>), dim3(lr.n_long_blocks), dim3(256), 0, s, rowptr,
                           a, g, out, m, (int)h, lr);
    return launch_status();
}

}  // namespace cogdl

// edge_softmax_flat.hip
namespace cogdl {
bool es_flat_covers(int64_t h, int dtype, const void *a, const void *g, const void *out);
size_t es_flat_workspace_bytes(int64_t nnz, int64_t h);
int es_flat_launch(bool bwd, const int32_t *rowptr, const void *a, const void *g, void *out, int64_t m, int64_t nnz,
                   int64_t h, int dtype, void *ws, size_t wsb, hipStream_t s);
}  // namespace cogdl

using namespace cogdl;

// Tuning key 7, bit 2: force the row kernels of this file (experiments / A-B tests of the flat kernel).
static bool es_use_flat(int64_t h, int dtype, const void *a, const void *g, const void *out, const void *ws) {
    return ws != nullptr && (g_tuning[kTuneEsScalar] & 4) == 0 && es_flat_covers(h, dtype, a, g, out);
}

extern "C" size_t cogdl_hip_edge_softmax_workspace_bytes(int64_t nnz, int64_t h) {
    if (nnz <= 0 || h <= 0) return 0;
    const size_t rows = rowreduce_workspace_bytes(nnz, 2 * h, es_thresh_scale(h));
    const bool pow2 = (h & (h - 1)) == 0 && h <= kWave;
    return pow2 ? std::max(rows, es_flat_workspace_bytes(nnz, h)) : rows;
}

template <bool BACKWARD>
static int edge_softmax_entry(const int32_t *rowptr, const void *a, const void *g, void *out, int64_t m, int64_t nnz,
                              int64_t h, int dtype, void *ws, size_t wsb, hipStream_t s) {
    if (m < 0 || h < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (dtype != COGDL_HIP_F32 && dtype != COGDL_HIP_F16 && dtype != COGDL_HIP_BF16) return COGDL_HIP_EDTYPE;
    if (m == 0 || h == 0 || nnz == 0) return COGDL_HIP_OK;
    if (!rowptr || !a || !out || (BACKWARD && !g)) return COGDL_HIP_EINVAL;
    if (h > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (es_use_flat(h, dtype, a, g, out, ws)) return es_flat_launch(BACKWARD, rowptr, a, g, out, m, nnz, h, dtype, ws, wsb, s);
    // the row kernels (H not a power of two, H > 64, unaligned operands, or no workspace)
    if (dtype == COGDL_HIP_F16)
        return edge_softmax_dispatch16<BACKWARD, __half>(rowptr, (const __half *)a, (const __half *)g, (__half *)out, m, nnz, h, ws,
                                                         wsb, s);
    if (dtype == COGDL_HIP_BF16)
        return edge_softmax_dispatch16<BACKWARD, __hip_bfloat16>(rowptr, (const __hip_bfloat16 *)a, (const __hip_bfloat16 *)g,
                                                                 (__hip_bfloat16 *)out, m, nnz, h, ws, wsb, s);
    return edge_softmax_dispatch<BACKWARD>(rowptr, (const float *)a, (const float *)g, (float *)out, m, nnz, h, ws, wsb, s);
}

extern "C" int cogdl_hip_edge_softmax_fwd(const int32_t *rowptr, const void *values, void *out, int64_t m,
                                          int64_t nnz, int64_t h, int dtype, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    return edge_softmax_entry<false>(rowptr, values, nullptr, out, m, nnz, h, dtype, workspace, workspace_bytes,
                                     (hipStream_t)stream);
}

extern "C" int cogdl_hip_edge_softmax_bwd(const int32_t *rowptr, const void *softmax, const void *grad,
                                          void *grad_in, int64_t m, int64_t nnz, int64_t h, int dtype,
                                          void *workspace, size_t workspace_bytes, void *stream) {
    return edge_softmax_entry<true>(rowptr, softmax, grad, grad_in, m, nnz, h, dtype, workspace, workspace_bytes,
                                    (hipStream_t)stream);
}
