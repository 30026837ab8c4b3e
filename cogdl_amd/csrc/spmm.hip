// spmm.hip -- csr_spmm / mhspmm for gfx950 (MI355X).  Replaces the reference's GE-SpMM style CUDA
// kernels (cogdl/operators/spmm/spmm_kernel.cu:7-512 and multiheadSpmm.cu:6-51: warp=32,
// shared-memory colind tiles) with a wave64 design; the arithmetic follows the reference CPU
// operator (cogdl/operators/spmm/spmm_cpu.cpp:24-35) so fp32 results are bit-identical to it.
//
// Work decomposition ("row groups"):
//   * a group of LPR lanes owns one CSR row and VEC consecutive feature columns per lane, so one
//     gathered neighbour row x[col,:] is ONE fully coalesced vector load of LPR*VEC*sizeof(T)
//     bytes (F=128 fp32: 64 lanes x float2 = 512 B);
//   * a wave holds 64/LPR consecutive rows, a 256-thread workgroup 4 waves => its output tile is
//     one contiguous span of memory;
//   * the group reads its row's colind/val in coalesced chunks of LPR edges and broadcasts them
//     lane-to-lane (ds_bpermute; v_readlane -> SGPR address when LPR == 64), so every index is
//     fetched from memory once;
//   * the gather is issued UNROLL rows at a time (independent loads in flight); accumulation
//     stays strictly in CSR edge order per output element: acc = acc + w*x with separate fp32
//     multiply and add -- no cross-lane reduction over edges;
//   * workgroups are remapped so each XCD (private L2) walks a contiguous range of rows.
// Skewed degree distributions ("long rows"): a row longer than `long_thresh` edges would
// serialise one lane group for longer than the rest of the launch takes.  Such rows are skipped
// by the main kernel (it records them per edge-chunk), their edges are processed chunk-parallel
// by a second kernel (every group of a workgroup takes a contiguous slice of the chunk, partial
// sums are combined in a FIXED order through LDS and a small fp32 workspace) and a third kernel
// adds the per-chunk partials in chunk order.  No atomics: results are deterministic; rows
// <= long_thresh keep the reference's exact summation order, longer rows differ from it only by
// re-association (<= 1e-6 relative).
// HBM-bound: algorithmic bytes per edge = 4 (colind) + s_w + F*s, per row 4 + F*s.
#include <climits>

#include "common.h"

namespace cogdl {

constexpr int kDefaultUnroll = 8;

template <bool EXACT>
__device__ __forceinline__ float mul_add(float acc, float w, float v) {
    // The library is compiled with -ffp-contract=off: `acc + w * v` is a rounded multiply
    // followed by a rounded add, exactly what the reference's x86-64 build executes.
    if constexpr (EXACT) return acc + w * v;
    else return fmaf(w, v, acc);
}

template <typename T>
struct SpmmArgs {
    const int32_t *rowptr;
    const int32_t *colind;
    const T *val;        // WMODE 1: one weight per edge
    const float *att;    // WMODE 2: [E, heads] fp32
    const T *x;
    T *out;
    int64_t m;
    int k;               // feature width (heads * fdim for WMODE 2)
    int fdim;
    int64_t n_rowblocks;
    int long_thresh;     // rows with more edges take the chunk-parallel path (INT_MAX: disabled)
    int acc_mode;        // != 0: out += A x
    int32_t *chunk_row;  // [n_chunks][2]: long row intersecting the chunk (slot 0: the row that contains the
                         //   chunk's first edge; slot 1: a row that starts inside the chunk), -1 = none
    float *partial;      // [n_chunks][2][k] fp32 partial sums
    int64_t nnz;
};

// WMODE: 0 = unweighted (csr_spmm_no_edge_value), 1 = one weight per edge (val, dtype T),
//        2 = multi-head (mhspmm): x is [N, H, fdim], weight att[e, head] in fp32, head = column / fdim.
// Accumulate edges [start, end) of one row into acc, in order.  All lanes of the group execute it.
template <typename T, int VEC, int LPR, int UNROLL, int WMODE, bool EXACT>
__device__ __forceinline__ void accumulate_edges(const SpmmArgs<T> &a, int start, int end, int sub, int l,
                                                 const T *xcol, int heads, int hd, float (&acc)[VEC]) {
    constexpr bool WEIGHTED = WMODE != 0;
    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        int my_c = 0;
        float my_w = 0.f;
        if (l < cnt) {
            my_c = a.colind[base + l];
            if constexpr (WMODE == 1) my_w = to_f32<T>(a.val[base + l]);
        }
        for (int j = 0; j < cnt; j += UNROLL) {
            float v[UNROLL][VEC];
            float w[UNROLL];
            // Issue all UNROLL gathers back to back (no branches: a masked tail slot re-reads
            // the row's last valid neighbour, an L1 hit, and is zeroed by the selects below).
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int jj = min(j + u, cnt - 1);
                int c;
                if constexpr (LPR == kWave) {
                    c = __builtin_amdgcn_readlane(my_c, jj);
                    w[u] = (WMODE == 1) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), jj)) : 1.f;
                } else {
                    c = __shfl(my_c, sub * LPR + jj, kWave);
                    w[u] = (WMODE == 1) ? __shfl(my_w, sub * LPR + jj, kWave) : 1.f;
                }
                if constexpr (WMODE == 2) w[u] = a.att[(int64_t)(base + jj) * heads + hd];  // 4*H-byte run per edge
                load_vec<T, VEC>(xcol + (int64_t)c * a.k, v[u]);
            }
            // Strictly in CSR order.  acc + 0*0 == acc exactly (acc is never -0), so masked
            // slots are no-ops; selecting v (not only w) to zero keeps inf/nan out.
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const bool valid = (j + u) < cnt;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float vv = valid ? v[u][i] : 0.f;
                    if constexpr (WEIGHTED) acc[i] = mul_add<EXACT>(acc[i], valid ? w[u] : 0.f, vv);
                    else acc[i] = acc[i] + vv;
                }
            }
        }
    }
}

template <typename T, int VEC, int LPR, int UNROLL, int WMODE, bool EXACT>
__global__ __launch_bounds__(256) void csr_spmm_rowgroup_kernel(const SpmmArgs<T> a) {
    constexpr int RPW = kWave / LPR;  // rows per wave
    constexpr int RPB = RPW * 4;      // rows per 256-thread workgroup
    const int64_t rb = xcd_remap(blockIdx.x, a.n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int64_t row = rb * RPB + wave * RPW + sub;
    const int col0 = ((int)blockIdx.y * LPR + l) * VEC;
    const bool col_ok = col0 < a.k;

    int start = 0, end = 0;
    if (row < a.m) {
        start = a.rowptr[row];
        end = a.rowptr[row + 1];
    }
    if constexpr (LPR == kWave) {  // whole wave on one row: make the loop bounds scalar
        start = __builtin_amdgcn_readfirstlane(start);
        end = __builtin_amdgcn_readfirstlane(end);
    }
    if (end - start > a.long_thresh) {
        // Long row: leave it to the chunk-parallel kernels, just record which chunks it touches.
        if (blockIdx.y == 0) {
            const int ch = a.long_thresh;  // chunk length == threshold
            const int c_first = start / ch, c_last = (end - 1) / ch;
            for (int c = c_first + l; c <= c_last; c += LPR)
                a.chunk_row[2 * c + ((c == c_first && start != c * ch) ? 1 : 0)] = (int32_t)row;
        }
        return;
    }
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (a.acc_mode && row < a.m && col_ok) load_vec<T, VEC>(a.out + row * (int64_t)a.k + col0, acc);
    const int cc = col_ok ? col0 : 0;  // lanes past the last column read column 0 and never store
    const int heads = (WMODE == 2) ? a.k / a.fdim : 1;
    accumulate_edges<T, VEC, LPR, UNROLL, WMODE, EXACT>(a, start, end, sub, l, a.x + cc, heads,
                                                        (WMODE == 2) ? cc / a.fdim : 0, acc);
    if (row < a.m && col_ok) store_vec<T, VEC>(a.out + row * (int64_t)a.k + col0, acc);
}

// One workgroup per edge chunk [c*ch, (c+1)*ch): for each long row intersecting the chunk the
// 256/LPR groups take contiguous slices of the intersection, then group partials are summed in
// group order through LDS and written to partial[c][slot][:].
template <typename T, int VEC, int LPR, int UNROLL, int WMODE>
__global__ __launch_bounds__(256) void csr_spmm_longrow_partial_kernel(const SpmmArgs<T> a, int64_t n_chunks) {
    constexpr int G = 256 / LPR;  // groups per workgroup
    __shared__ float red[G][LPR * VEC];
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int g = (threadIdx.x >> 6) * (kWave / LPR) + sub;
    const int col0 = ((int)blockIdx.y * LPR + l) * VEC;
    const bool col_ok = col0 < a.k;
    const int cc = col_ok ? col0 : 0;
    const int heads = (WMODE == 2) ? a.k / a.fdim : 1;
    const int hd = (WMODE == 2) ? cc / a.fdim : 0;
    const int ch = a.long_thresh;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        for (int slot = 0; slot < 2; ++slot) {
            const int32_t row = a.chunk_row[2 * c + slot];  // workgroup-uniform
            if (row < 0) continue;
            const int lo = max(a.rowptr[row], (int)(c * ch));
            const int hi = (int)min((int64_t)a.rowptr[row + 1], (c + 1) * ch);
            const int per = (hi - lo + G - 1) / G;
            int s = min(lo + g * per, hi), e = min(s + per, hi);
            if constexpr (LPR == kWave) {  // one wave per slice: keep the loop bounds scalar
                s = __builtin_amdgcn_readfirstlane(s);
                e = __builtin_amdgcn_readfirstlane(e);
            }
            float acc[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
            accumulate_edges<T, VEC, LPR, UNROLL, WMODE, true>(a, s, e, sub, l, a.x + cc, heads, hd, acc);
#pragma unroll
            for (int i = 0; i < VEC; ++i) red[g][l * VEC + i] = acc[i];
            __syncthreads();
            if (g == 0 && col_ok) {
                float *dst = a.partial + ((int64_t)(2 * c + slot)) * a.k + col0;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float sum = red[0][l * VEC + i];
                    for (int q = 1; q < G; ++q) sum += red[q][l * VEC + i];
                    dst[i] = sum;
                }
            }
            __syncthreads();
        }
    }
}

// For every long row (found through the chunk in which it starts) add its per-chunk partials in
// chunk order and write (or accumulate into) the output row.  One wave per chunk.
template <typename T>
__global__ __launch_bounds__(64) void csr_spmm_longrow_combine_kernel(const SpmmArgs<T> a, int64_t n_chunks) {
    const int ch = a.long_thresh;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        for (int slot = 0; slot < 2; ++slot) {
            const int32_t row = a.chunk_row[2 * c + slot];
            if (row < 0) continue;
            const int start = a.rowptr[row], end = a.rowptr[row + 1];
            if (start < c * ch) continue;  // the row started in an earlier chunk: that chunk owns the combine
            const int64_t c_last = (end - 1) / ch;
            for (int col = threadIdx.x; col < a.k; col += kWave) {
                float acc = a.acc_mode ? to_f32<T>(a.out[(int64_t)row * a.k + col]) : 0.f;
                acc += a.partial[(2 * c + slot) * (int64_t)a.k + col];
                for (int64_t cc = c + 1; cc <= c_last; ++cc) acc += a.partial[(2 * cc) * (int64_t)a.k + col];
                a.out[(int64_t)row * a.k + col] = from_f32<T>(acc);
            }
        }
    }
}

static inline int64_t n_chunks_for(int64_t nnz, int thresh) { return (nnz + thresh - 1) / thresh; }

// Threshold above which a row is split: the sequential time of a row of T edges (~T/UNROLL gather
// round trips) should stay a small fraction of the whole launch (~nnz / 13 GEdges/s).
static inline int pick_long_thresh(int64_t nnz) {
    int t = 128;
    while (t < 4096 && (int64_t)t * 16384 < nnz) t <<= 1;
    return t;
}

template <typename T, int VEC, int LPR, int UNROLL, int WMODE, bool EXACT>
static int launch_rowgroup(SpmmArgs<T> a, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    constexpr int RPB = (kWave / LPR) * 4;
    a.n_rowblocks = (a.m + RPB - 1) / RPB;
    const int64_t tiles = ((int64_t)a.k + (int64_t)LPR * VEC - 1) / ((int64_t)LPR * VEC);
    if (a.n_rowblocks == 0) return COGDL_HIP_OK;
    if (a.n_rowblocks > 0x7fffffff / kXcds || tiles > 65535) return COGDL_HIP_ERANGE;
    a.long_thresh = INT_MAX;
    int64_t n_chunks = 0;
    if (workspace && a.nnz > 0) {
        const int t = pick_long_thresh(a.nnz);
        n_chunks = n_chunks_for(a.nnz, t);
        const size_t need = (size_t)n_chunks * 2 * sizeof(int32_t) + 256 + (size_t)n_chunks * 2 * a.k * sizeof(float);
        if (workspace_bytes < need) return COGDL_HIP_EWORKSPACE;
        if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
        a.long_thresh = t;
        a.chunk_row = (int32_t *)workspace;
        const size_t off = ((size_t)n_chunks * 2 * sizeof(int32_t) + 255) / 256 * 256;
        a.partial = (float *)((char *)workspace + off);
        hipError_t e = hipMemsetAsync(a.chunk_row, 0xff, (size_t)n_chunks * 2 * sizeof(int32_t), stream);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return COGDL_HIP_ELAUNCH;
        }
    }
    dim3 grid(xcd_grid(a.n_rowblocks), (unsigned)tiles);
    hipLaunchKernelGGL((csr_spmm_rowgroup_kernel<T, VEC, LPR, UNROLL, WMODE, EXACT>), grid, dim3(256), 0, stream, a);
    if (n_chunks > 0) {
        const unsigned gx = (unsigned)std::min<int64_t>(n_chunks, 4096);
        hipLaunchKernelGGL((csr_spmm_longrow_partial_kernel<T, VEC, LPR, UNROLL, WMODE>), dim3(gx, (unsigned)tiles),
                           dim3(256), 0, stream, a, n_chunks);
        hipLaunchKernelGGL((csr_spmm_longrow_combine_kernel<T>), dim3(gx), dim3(64), 0, stream, a, n_chunks);
    }
    return launch_status();
}

// (VEC, LPR) choice: as many lanes per row as the row has VEC-wide columns (whole-wave rows are the
// fastest: scalar column broadcast, no inter-row divergence inside a wave), VEC as small as that allows.
template <typename T, int VEC, int WMODE>
static int dispatch_lpr(const SpmmArgs<T> &a, void *ws, size_t wsb, hipStream_t s) {
    const int64_t need = ((int64_t)a.k + VEC - 1) / VEC;
    if (need <= 4) return launch_rowgroup<T, VEC, 4, kDefaultUnroll, WMODE, true>(a, ws, wsb, s);
    if (need <= 8) return launch_rowgroup<T, VEC, 8, kDefaultUnroll, WMODE, true>(a, ws, wsb, s);
    if (need <= 16) return launch_rowgroup<T, VEC, 16, kDefaultUnroll, WMODE, true>(a, ws, wsb, s);
    if (need <= 32) return launch_rowgroup<T, VEC, 32, kDefaultUnroll, WMODE, true>(a, ws, wsb, s);
    return launch_rowgroup<T, VEC, 64, kDefaultUnroll, WMODE, true>(a, ws, wsb, s);
}

// Vector width: every lane's VEC columns must stay inside one row (k % VEC == 0), inside one head for
// mhspmm (fdim % VEC == 0), and naturally aligned.  Among the legal widths take the SMALLEST that
// still lets one group of <= 64 lanes cover the row (k / VEC <= 64), else the widest.
template <typename T, int WMODE>
static int spmm_auto(const SpmmArgs<T> &a, void *ws, size_t wsb, hipStream_t s) {
    constexpr int MAXV = 16 / sizeof(T);
    const int64_t unit = (WMODE == 2) ? a.fdim : a.k;
    auto legal = [&](int v) {
        return v <= MAXV && unit % v == 0 && aligned_to(a.x, v * sizeof(T)) && aligned_to(a.out, v * sizeof(T));
    };
    int vec = 1;
    for (int v = 1; v <= MAXV; v <<= 1) {
        if (!legal(v)) break;
        vec = v;
        if (v * sizeof(T) >= 4 && (int64_t)a.k <= (int64_t)v * kWave) break;  // at least 4 B per lane
    }
    switch (vec) {
        case 8:
            if constexpr (MAXV >= 8) return dispatch_lpr<T, 8, WMODE>(a, ws, wsb, s);
        case 4: return dispatch_lpr<T, 4, WMODE>(a, ws, wsb, s);
        case 2: return dispatch_lpr<T, 2, WMODE>(a, ws, wsb, s);
        default: return dispatch_lpr<T, 1, WMODE>(a, ws, wsb, s);
    }
}

template <typename T>
static int spmm_typed(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x, void *out,
                      int64_t m, int64_t k, int64_t nnz, int acc_mode, void *ws, size_t wsb, hipStream_t s) {
    if (!aligned_to(x, sizeof(T)) || !aligned_to(out, sizeof(T))) return COGDL_HIP_EALIGN;
    SpmmArgs<T> a{};
    a.rowptr = rowptr; a.colind = colind; a.val = (const T *)val; a.att = nullptr;
    a.x = (const T *)x; a.out = (T *)out; a.m = m; a.k = (int)k; a.fdim = (int)k; a.nnz = nnz; a.acc_mode = acc_mode;
    return val ? spmm_auto<T, 1>(a, ws, wsb, s) : spmm_auto<T, 0>(a, ws, wsb, s);
}

static int check_args(const int32_t *rowptr, const void *x, const void *out, int64_t m, int64_t k, int64_t nnz) {
    if (m < 0 || k < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (m == 0 || k == 0) return COGDL_HIP_OK;
    if (!rowptr || !x || !out) return COGDL_HIP_EINVAL;
    if (k > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    return COGDL_HIP_OK;
}

static int csr_spmm_entry(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x, void *out,
                          int64_t m, int64_t k, int64_t nnz, int dtype, int acc_mode, void *ws, size_t wsb,
                          void *stream) {
    int rc = check_args(rowptr, x, out, m, k, nnz);
    if (rc != COGDL_HIP_OK || m == 0 || k == 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32: return spmm_typed<float>(rowptr, colind, val, x, out, m, k, nnz, acc_mode, ws, wsb, s);
        case COGDL_HIP_F16: return spmm_typed<__half>(rowptr, colind, val, x, out, m, k, nnz, acc_mode, ws, wsb, s);
        case COGDL_HIP_BF16:
            return spmm_typed<__hip_bfloat16>(rowptr, colind, val, x, out, m, k, nnz, acc_mode, ws, wsb, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

template <typename T>
static int mhspmm_typed(const int32_t *rowptr, const int32_t *colind, const float *att, const void *feat, void *out,
                        int64_t v, int64_t h, int64_t f, int64_t nnz, void *ws, size_t wsb, hipStream_t s) {
    if (!aligned_to(feat, sizeof(T)) || !aligned_to(out, sizeof(T))) return COGDL_HIP_EALIGN;
    SpmmArgs<T> a{};
    a.rowptr = rowptr; a.colind = colind; a.val = nullptr; a.att = att;
    a.x = (const T *)feat; a.out = (T *)out; a.m = v; a.k = (int)(h * f); a.fdim = (int)f; a.nnz = nnz; a.acc_mode = 0;
    return spmm_auto<T, 2>(a, ws, wsb, s);
}

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_csr_spmm_workspace_bytes(int64_t nnz, int64_t k) {
    if (nnz <= 0 || k <= 0) return 0;
    const int64_t n_chunks = n_chunks_for(nnz, pick_long_thresh(nnz));
    return (size_t)n_chunks * 2 * sizeof(int32_t) + 256 + (size_t)n_chunks * 2 * (size_t)k * sizeof(float);
}

extern "C" int cogdl_hip_csr_spmm_long_row_threshold(int64_t nnz) { return pick_long_thresh(nnz); }

extern "C" int cogdl_hip_csr_spmm(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                  void *out, int64_t m, int64_t k, int64_t nnz, int dtype, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    return csr_spmm_entry(rowptr, colind, val, x, out, m, k, nnz, dtype, 0, workspace, workspace_bytes, stream);
}

extern "C" int cogdl_hip_csr_spmm_acc(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                      void *out, int64_t m, int64_t k, int64_t nnz, int dtype, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    return csr_spmm_entry(rowptr, colind, val, x, out, m, k, nnz, dtype, 1, workspace, workspace_bytes, stream);
}

extern "C" int cogdl_hip_mhspmm(const int32_t *rowptr, const int32_t *colind, const float *att, const void *feat,
                                void *out, int64_t v, int64_t h, int64_t f, int64_t nnz, int dtype, void *workspace,
                                size_t workspace_bytes, void *stream) {
    if (v < 0 || h < 0 || f < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (v == 0 || h == 0 || f == 0) return COGDL_HIP_OK;
    if (!rowptr || !att || !feat || !out) return COGDL_HIP_EINVAL;
    if (h * f > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32:
            return mhspmm_typed<float>(rowptr, colind, att, feat, out, v, h, f, nnz, workspace, workspace_bytes, s);
        case COGDL_HIP_F16:
            return mhspmm_typed<__half>(rowptr, colind, att, feat, out, v, h, f, nnz, workspace, workspace_bytes, s);
        case COGDL_HIP_BF16:
            return mhspmm_typed<__hip_bfloat16>(rowptr, colind, att, feat, out, v, h, f, nnz, workspace,
                                                workspace_bytes, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

// Benchmark hook: explicit (VEC, LPR, UNROLL, arithmetic) choices for fp32 weighted, k % 4 == 0.
extern "C" int cogdl_hip_csr_spmm_variant(const int32_t *rowptr, const int32_t *colind, const void *val,
                                          const void *x, void *out, int64_t m, int64_t k, int64_t nnz, int dtype,
                                          int variant, void *workspace, size_t workspace_bytes, void *stream) {
    if (variant < 0)
        return cogdl_hip_csr_spmm(rowptr, colind, val, x, out, m, k, nnz, dtype, workspace, workspace_bytes, stream);
    int rc = check_args(rowptr, x, out, m, k, nnz);
    if (rc != COGDL_HIP_OK || m == 0 || k == 0) return rc;
    if (dtype != COGDL_HIP_F32 || !val) return COGDL_HIP_EDTYPE;
    if (k % 4 != 0 || !aligned_to(x, 16) || !aligned_to(out, 16)) return COGDL_HIP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    SpmmArgs<float> a{};
    a.rowptr = rowptr; a.colind = colind; a.val = (const float *)val; a.x = (const float *)x; a.out = (float *)out;
    a.m = m; a.k = (int)k; a.fdim = (int)k; a.nnz = nnz;
#define V(id, VEC, LPR, UNR, EX) \
    case id: return launch_rowgroup<float, VEC, LPR, UNR, 1, EX>(a, workspace, workspace_bytes, s);
    switch (variant) {
        V(0, 4, 32, 8, true)
        V(1, 4, 32, 4, true)
        V(2, 4, 16, 8, true)
        V(3, 2, 64, 8, true)
        V(4, 2, 64, 4, true)
        V(5, 2, 64, 8, false)
        V(6, 2, 32, 8, true)
        V(7, 4, 64, 8, true)
        V(8, 1, 64, 8, true)
        V(9, 1, 64, 16, true)
        V(10, 2, 64, 16, true)
        default: return COGDL_HIP_EINVAL;
    }
#undef V
}
