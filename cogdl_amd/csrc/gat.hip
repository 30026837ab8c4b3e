// gat.hip -- fused GAT attention + aggregation for gfx950 (fused_gat_func of cogdl/operators/fused_gat.py:14-41;
// the reference binds dgNN's fused_gatconv, whose source is absent from the tree, so the semantics are those
// of the unfused layer path cogdl/layers/gat_layer.py:73-77):
//     s[e,h] = LeakyReLU(attn_row[row(e),h] + attn_col[col[e],h])
//     a[e,h] = softmax over the edges of row(e)
//     out[v,h,:] = sum_e a[e,h] * feat[col[e],h,:]
// Forward: ONE pass over the edges with an online softmax (running max / rescaled sum, flash-attention style),
// so the [E,H] score/attention tensors are never materialised: algorithmic bytes per edge =
// 4 (colind) + 4H (attn_col row) + H*F*s (feature row); per node 2*4H + H*F*s.  Row-group decomposition of
// spmm.hip: a group of LPR lanes owns one destination row, each lane VEC columns inside one head.
// Backward (fp32): with D[v,h] = <g[v,h,:], out[v,h,:]>,
//     dL/ds[e,h]      = a[e,h] * (<g[row,h,:], feat[col,h,:]> - D[row,h]) * LeakyReLU'(.)
//     grad_attn_row[v,h] = sum over the row's edges            (row pass over the CSR, gathers feat)
//     grad_attn_col[u,h] = sum over the column's edges         (column pass over the CSC, gathers g)
//     grad_feat[u,h,:]   = sum over the column's edges a * g[row,h,:]        (same column pass)
// a[e,h] is recomputed from the saved row max / row sum (edge_max, edge_sum), exactly what the reference's
// FusedGATFunction keeps in ctx (operators/fused_gat.py:20).  No atomics anywhere: deterministic.
#include "common.h"

namespace cogdl {

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// Sum over the `lph` lanes (power of two, aligned) that hold one head.
template <int LPR>
__device__ __forceinline__ float head_sum(float v, int lph) {
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1)
        if (s < lph) v += __shfl_xor(v, s, kWave);
    return v;
}

template <typename T, int VEC, int LPR, int UNROLL>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ colind,
                                                      const float *__restrict__ attn_row,
                                                      const float *__restrict__ attn_col, const T *__restrict__ feat,
                                                      float slope, T *__restrict__ out, float *__restrict__ edge_max,
                                                      float *__restrict__ edge_sum, int64_t m, int heads, int fdim,
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
    const int col0 = ((int)blockIdx.y * LPR + l) * VEC;
    const bool col_ok = col0 < k;
    const int cc = col_ok ? col0 : 0;
    const int hd = cc / fdim;
    int start = 0, end = 0;
    if (row < m) {
        start = rowptr[row];
        end = rowptr[row + 1];
    }
    const float ar = (row < m) ? attn_row[row * heads + hd] : 0.f;
    float mx = -INFINITY, lsum = 0.f;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_c = (l < cnt) ? colind[base + l] : 0;
        for (int j = 0; j < cnt; j += UNROLL) {
            float v[UNROLL][VEC];
            float ac[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int c = __shfl(my_c, sub * LPR + min(j + u, cnt - 1), kWave);
                ac[u] = attn_col[(int64_t)c * heads + hd];
                load_vec<T, VEC>(feat + (int64_t)c * k + cc, v[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (j + u < cnt) {
                    const float s = leaky(ar + ac[u], slope);
                    const float mn = fmaxf(mx, s);
                    const float scale = (lsum == 0.f) ? 0.f : expf(mx - mn);
                    const float p = expf(s - mn);
                    lsum = lsum * scale + p;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, v[u][i], acc[i] * scale);
                    mx = mn;
                }
            }
        }
    }
    if (row < m && col_ok) {
        const float inv = (lsum > 0.f) ? 1.f / lsum : 0.f;  // empty row -> zeros
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] *= inv;
        store_vec<T, VEC>(out + row * (int64_t)k + col0, acc);
        if (col0 % fdim == 0) {
            edge_max[row * heads + hd] = mx;
            edge_sum[row * heads + hd] = lsum;
        }
    }
}

// Row pass of the backward: D[v,h] and grad_attn_row[v,h].  The whole [H*F] row must fit one group.
template <int VEC, int LPR, int UNROLL>
__global__ __launch_bounds__(256) void gat_bwd_row_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind, const float *__restrict__ attn_row,
    const float *__restrict__ attn_col, const float *__restrict__ feat, float slope,
    const float *__restrict__ edge_max, const float *__restrict__ edge_sum, const float *__restrict__ out,
    const float *__restrict__ grad_out, float *__restrict__ dvec, float *__restrict__ grad_attn_row, int64_t m,
    int heads, int fdim, int lph, XcdMap n_rowblocks) {
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
    const int cc = col_ok ? l * VEC : 0;
    const int hd = cc / fdim;
    int start = 0, end = 0;
    if (row < m) {
        start = rowptr[row];
        end = rowptr[row + 1];
    }
    float g[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
    float d = 0.f, ar = 0.f, mx = 0.f, inv = 0.f;
    if (row < m && col_ok) {
        load_vec<float, VEC>(grad_out + row * (int64_t)k + cc, g);
        float o[VEC];
        load_vec<float, VEC>(out + row * (int64_t)k + cc, o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) d = fmaf(g[i], o[i], d);
        ar = attn_row[row * heads + hd];
        mx = edge_max[row * heads + hd];
        const float ls = edge_sum[row * heads + hd];
        inv = ls > 0.f ? 1.f / ls : 0.f;
    }
    d = head_sum<LPR>(d, lph);
    float gacc = 0.f;
    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_c = (l < cnt) ? colind[base + l] : 0;
        for (int j = 0; j < cnt; j += UNROLL) {
            float v[UNROLL][VEC];
            float ac[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int c = __shfl(my_c, sub * LPR + min(j + u, cnt - 1), kWave);
                ac[u] = attn_col[(int64_t)c * heads + hd];
                load_vec<float, VEC>(feat + (int64_t)c * k + cc, v[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) dot = fmaf(g[i], col_ok ? v[u][i] : 0.f, dot);
                dot = head_sum<LPR>(dot, lph);
                if (j + u < cnt) {
                    const float pre = ar + ac[u];
                    const float a = expf(leaky(pre, slope) - mx) * inv;
                    gacc += a * (dot - d) * (pre > 0.f ? 1.f : slope);
                }
            }
        }
    }
    if (row < m && col_ok && cc % fdim == 0) {
        dvec[row * heads + hd] = d;
        grad_attn_row[row * heads + hd] = gacc;
    }
}

// Column pass of the backward over the CSC (colptr, rowind): grad_feat[u,h,:] and grad_attn_col[u,h].
template <int VEC, int LPR, int UNROLL>
__global__ __launch_bounds__(256) void gat_bwd_col_kernel(
    const int32_t *__restrict__ colptr, const int32_t *__restrict__ rowind, const float *__restrict__ attn_row,
    const float *__restrict__ attn_col, const float *__restrict__ feat, float slope,
    const float *__restrict__ edge_max, const float *__restrict__ edge_sum, const float *__restrict__ dvec,
    const float *__restrict__ grad_out, float *__restrict__ grad_feat, float *__restrict__ grad_attn_col,
    int64_t n_src, int heads, int fdim, int lph, XcdMap n_rowblocks) {
    constexpr int RPW = kWave / LPR;
    constexpr int RPB = RPW * 4;
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int64_t u_node = rb * RPB + (threadIdx.x >> 6) * RPW + sub;
    const int k = heads * fdim;
    const bool col_ok = l * VEC < k;
    const int cc = col_ok ? l * VEC : 0;
    const int hd = cc / fdim;
    int start = 0, end = 0;
    if (u_node < n_src) {
        start = colptr[u_node];
        end = colptr[u_node + 1];
    }
    float f[VEC], acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = acc[i] = 0.f;
    float ac = 0.f;
    if (u_node < n_src && col_ok) {
        load_vec<float, VEC>(feat + u_node * (int64_t)k + cc, f);
        ac = attn_col[u_node * heads + hd];
    }
    float gacc = 0.f;
    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int my_r = (l < cnt) ? rowind[base + l] : 0;
        for (int j = 0; j < cnt; j += UNROLL) {
            float g[UNROLL][VEC];
            float ar[UNROLL], mx[UNROLL], ls[UNROLL], dd[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int r = __shfl(my_r, sub * LPR + min(j + u, cnt - 1), kWave);
                const int64_t rh = (int64_t)r * heads + hd;
                ar[u] = attn_row[rh];
                mx[u] = edge_max[rh];
                ls[u] = edge_sum[rh];
                dd[u] = dvec[rh];
                load_vec<float, VEC>(grad_out + (int64_t)r * k + cc, g[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) dot = fmaf(f[i], col_ok ? g[u][i] : 0.f, dot);
                dot = head_sum<LPR>(dot, lph);
                if (j + u < cnt) {
                    const float pre = ar[u] + ac;
                    const float a = expf(leaky(pre, slope) - mx[u]) / ls[u];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a, g[u][i], acc[i]);
                    gacc += a * (dot - dd[u]) * (pre > 0.f ? 1.f : slope);
                }
            }
        }
    }
    if (u_node < n_src && col_ok) {
        store_vec<float, VEC>(grad_feat + u_node * (int64_t)k + cc, acc);
        if (cc % fdim == 0) grad_attn_col[u_node * heads + hd] = gacc;
    }
}

static bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

template <typename T, int VEC, int LPR>
static int launch_fwd(const int32_t *rowptr, const int32_t *colind, const float *ar, const float *ac, const T *feat,
                      float slope, T *out, float *emax, float *esum, int64_t v, int64_t h, int64_t f, hipStream_t s) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int64_t nrb = (v + RPB - 1) / RPB;
    const int64_t tiles = (h * f + (int64_t)LPR * VEC - 1) / ((int64_t)LPR * VEC);
    if (nrb > 0x7fffffff / kXcds || tiles > 65535) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((gat_fwd_kernel<T, VEC, LPR, 8>), dim3(xcd_grid(make_xcd_map(nrb)), (unsigned)tiles), dim3(256), 0, s, rowptr,
                       colind, ar, ac, feat, slope, out, emax, esum, v, (int)h, (int)f, make_xcd_map(nrb));
    return launch_status();
}

template <typename T, int VEC>
static int dispatch_fwd(const int32_t *rowptr, const int32_t *colind, const float *ar, const float *ac, const T *feat,
                        float slope, T *out, float *emax, float *esum, int64_t v, int64_t h, int64_t f,
                        hipStream_t s) {
    const int64_t need = (h * f + VEC - 1) / VEC;
    if (need <= 8) return launch_fwd<T, VEC, 8>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
    if (need <= 16) return launch_fwd<T, VEC, 16>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
    if (need <= 32) return launch_fwd<T, VEC, 32>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
    return launch_fwd<T, VEC, 64>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
}

template <typename T>
static int gat_fwd_typed(const int32_t *rowptr, const int32_t *colind, const float *ar, const float *ac,
                         const void *feat_, float slope, void *out_, float *emax, float *esum, int64_t v, int64_t h,
                         int64_t f, hipStream_t s) {
    const T *feat = (const T *)feat_;
    T *out = (T *)out_;
    constexpr int MAXV = 16 / sizeof(T);
    int vec = 1;
    for (int w = 1; w <= MAXV; w <<= 1) {
        if (f % w != 0 || !aligned_to(feat, w * sizeof(T)) || !aligned_to(out, w * sizeof(T))) break;
        vec = w;
        if (w * sizeof(T) >= 4 && h * f <= (int64_t)w * kWave) break;
    }
    switch (vec) {
        case 8:
            if constexpr (MAXV >= 8) return dispatch_fwd<T, 8>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
        case 4: return dispatch_fwd<T, 4>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
        case 2: return dispatch_fwd<T, 2>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
        default: return dispatch_fwd<T, 1>(rowptr, colind, ar, ac, feat, slope, out, emax, esum, v, h, f, s);
    }
}

struct BwdArgs {
    const int32_t *rowptr, *colind, *colptr, *rowind;
    const float *ar, *ac, *feat, *emax, *esum, *out, *gout;
    float slope;
    float *gfeat, *gar, *gac, *dvec;
    int64_t v, n_src, h, f;
};

template <int VEC, int LPR>
static int launch_bwd(const BwdArgs &b, hipStream_t s) {
    constexpr int RPB = (kWave / LPR) * 4;
    const int lph = (b.h == 1) ? LPR : (int)(b.f / VEC);  // one head: reduce over the whole (zero-padded) group
    const int64_t nrb = (b.v + RPB - 1) / RPB, ncb = (b.n_src + RPB - 1) / RPB;
    if (nrb > 0x7fffffff / kXcds || ncb > 0x7fffffff / kXcds) return COGDL_HIP_ERANGE;
    if (nrb)
        hipLaunchKernelGGL((gat_bwd_row_kernel<VEC, LPR, 4>), dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, b.rowptr, b.colind,
                           b.ar, b.ac, b.feat, b.slope, b.emax, b.esum, b.out, b.gout, b.dvec, b.gar, b.v, (int)b.h,
                           (int)b.f, lph, make_xcd_map(nrb));
    if (ncb)
        hipLaunchKernelGGL((gat_bwd_col_kernel<VEC, LPR, 4>), dim3(xcd_grid(make_xcd_map(ncb))), dim3(256), 0, s, b.colptr, b.rowind,
                           b.ar, b.ac, b.feat, b.slope, b.emax, b.esum, b.dvec, b.gout, b.gfeat, b.gac, b.n_src,
                           (int)b.h, (int)b.f, lph, make_xcd_map(ncb));
    return launch_status();
}

template <int VEC>
static int dispatch_bwd(const BwdArgs &b, hipStream_t s) {
    const int64_t need = (b.h * b.f + VEC - 1) / VEC;
    if (need <= 8) return launch_bwd<VEC, 8>(b, s);
    if (need <= 16) return launch_bwd<VEC, 16>(b, s);
    if (need <= 32) return launch_bwd<VEC, 32>(b, s);
    return launch_bwd<VEC, 64>(b, s);
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_gat_fwd(const int32_t *rowptr, const int32_t *colind, const float *attn_row,
                                 const float *attn_col, const void *feat, float negative_slope, void *out,
                                 float *edge_max, float *edge_sum, int64_t v, int64_t h, int64_t f, int dtype,
                                 void *stream) {
    if (v < 0 || h < 0 || f < 0) return COGDL_HIP_EINVAL;
    if (v == 0 || h == 0 || f == 0) return COGDL_HIP_OK;
    if (!rowptr || !attn_row || !attn_col || !feat || !out || !edge_max || !edge_sum) return COGDL_HIP_EINVAL;
    if (h * f > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32:
            return gat_fwd_typed<float>(rowptr, colind, attn_row, attn_col, feat, negative_slope, out, edge_max,
                                        edge_sum, v, h, f, s);
        case COGDL_HIP_F16:
            return gat_fwd_typed<__half>(rowptr, colind, attn_row, attn_col, feat, negative_slope, out, edge_max,
                                         edge_sum, v, h, f, s);
        case COGDL_HIP_BF16:
            return gat_fwd_typed<__hip_bfloat16>(rowptr, colind, attn_row, attn_col, feat, negative_slope, out,
                                                 edge_max, edge_sum, v, h, f, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

extern "C" size_t cogdl_hip_gat_bwd_workspace_bytes(int64_t v, int64_t h) {
    return (size_t)(v > 0 ? v : 0) * (size_t)(h > 0 ? h : 0) * sizeof(float) + 256;
}

// Returns COGDL_HIP_EINVAL for shapes the fused backward does not cover (the [H*F] row must fit one group of
// 64 lanes * 4 columns, and F/VEC must be a power of two unless H == 1): callers then use the unfused
// edge_softmax + mhspmm operators instead.
extern "C" int cogdl_hip_gat_bwd(const int32_t *rowptr, const int32_t *colind, const int32_t *colptr,
                                 const int32_t *rowind, const float *attn_row, const float *attn_col,
                                 const float *feat, float negative_slope, const float *edge_max,
                                 const float *edge_sum, const float *out, const float *grad_out, float *grad_feat,
                                 float *grad_attn_row, float *grad_attn_col, void *workspace, size_t workspace_bytes,
                                 int64_t v, int64_t n_src, int64_t h, int64_t f, void *stream) {
    if (v < 0 || n_src < 0 || h <= 0 || f <= 0) return COGDL_HIP_EINVAL;
    if (!rowptr || !colptr || !attn_row || !attn_col || !feat || !edge_max || !edge_sum || !out || !grad_out ||
        !grad_feat || !grad_attn_row || !grad_attn_col || !workspace)
        return COGDL_HIP_EINVAL;
    if (workspace_bytes < cogdl_hip_gat_bwd_workspace_bytes(v, h)) return COGDL_HIP_EWORKSPACE;
    BwdArgs b{rowptr, colind, colptr, rowind, attn_row, attn_col, feat, edge_max, edge_sum, out, grad_out,
              negative_slope, grad_feat, grad_attn_row, grad_attn_col, (float *)workspace, v, n_src, h, f};
    hipStream_t s = (hipStream_t)stream;
    const int64_t k = h * f;
    const bool al16 = aligned_to(feat, 16) && aligned_to(out, 16) && aligned_to(grad_out, 16) && aligned_to(grad_feat, 16);
    const bool al8 = aligned_to(feat, 8) && aligned_to(out, 8) && aligned_to(grad_out, 8) && aligned_to(grad_feat, 8);
    auto ok = [&](int vec) { return f % vec == 0 && k / vec <= kWave && (h == 1 || pow2(f / vec)); };
    if (ok(1)) return dispatch_bwd<1>(b, s);
    if (ok(2) && al8) return dispatch_bwd<2>(b, s);
    if (ok(4) && al16) return dispatch_bwd<4>(b, s);
    return COGDL_HIP_EINVAL;
}
