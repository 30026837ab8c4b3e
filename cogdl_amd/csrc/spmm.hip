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
    XcdMap rowblocks;
    int long_thresh;     // rows with more edges take the chunk-parallel path (INT_MAX: disabled)
    int acc_mode;        // != 0: out += A x
    int rows_seq;        // consecutive rows walked one after the other by each lane group
    int32_t *chunk_row;  // [n_chunks]: the LONG row that contains edge c*ch (the chunk's first edge), else -1.
                         //   Every entry is (re)written by the main kernel on every launch: no memset needed.
    float *partial;      // [n_chunks][2][k] fp32 partial sums (slot 0: the row owning the chunk's first edge,
                         //   slot 1: the head of a long row that starts inside the chunk)
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
    constexpr int RPW = kWave / LPR;  // row groups per wave
    constexpr int GPB = RPW * 4;      // row groups per 256-thread workgroup
    const int64_t rb = xcd_remap(blockIdx.x, a.rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int col0 = ((int)blockIdx.y * LPR + l) * VEC;
    const bool col_ok = col0 < a.k;
    const int cc = col_ok ? col0 : 0;  // lanes past the last column read column 0 and never store
    const int heads = (WMODE == 2) ? a.k / a.fdim : 1;
    const int hd = (WMODE == 2) ? cc / a.fdim : 0;
    const int64_t row0 = (rb * GPB + wave * RPW + sub) * a.rows_seq;

    for (int rs = 0; rs < a.rows_seq; ++rs) {
        const int64_t row = row0 + rs;
        int start = 0, end = 0;
        if (row < a.m) {
            start = a.rowptr[row];
            end = a.rowptr[row + 1];
        }
        if constexpr (LPR == kWave) {  // whole wave on one row: make the loop bounds scalar
            start = __builtin_amdgcn_readfirstlane(start);
            end = __builtin_amdgcn_readfirstlane(end);
        }
        if (a.chunk_row && blockIdx.y == 0 && end > start) {
            // Own the chunk_row entries of the chunks whose first edge lies in this row: c*ch in [start, end).
            const int ch = a.long_thresh;
            const int32_t mark = (end - start > ch) ? (int32_t)row : -1;
            for (int c = (start + ch - 1) / ch + l; (int64_t)c * ch < end; c += LPR) a.chunk_row[c] = mark;
        }
        if (end - start > a.long_thresh) continue;  // long row: the chunk-parallel kernels compute it
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        if (a.acc_mode && row < a.m && col_ok) load_vec<T, VEC>(a.out + row * (int64_t)a.k + col0, acc);
        accumulate_edges<T, VEC, LPR, UNROLL, WMODE, EXACT>(a, start, end, sub, l, a.x + cc, heads, hd, acc);
        if (row < a.m && col_ok) store_vec<T, VEC>(a.out + row * (int64_t)a.k + col0, acc);
    }
}

// Chunk c = edges [c*ch, (c+1)*ch).  A long row (> ch edges) intersecting it is either the row owning the
// chunk's first edge (chunk_row[c], slot 0) or a row that starts inside the chunk -- which then owns the
// NEXT chunk's first edge (chunk_row[c+1], slot 1).  For each piece the 256/LPR groups of the workgroup take
// contiguous slices, group partials are summed in group order through LDS and written to partial[c][slot][:].
// Workgroups own contiguous runs of chunks and skip runs without long rows after one coalesced look.
template <typename T, int VEC, int LPR, int UNROLL, int WMODE>
__global__ __launch_bounds__(256) void csr_spmm_longrow_partial_kernel(const SpmmArgs<T> a, int64_t n_chunks,
                                                                       int chunks_per_block) {
    constexpr int G = 256 / LPR;  // groups per workgroup
    __shared__ float red[G][LPR * VEC];
    __shared__ int any_long;
    const int64_t c_begin = (int64_t)blockIdx.x * chunks_per_block;
    const int64_t c_end = min(c_begin + chunks_per_block, n_chunks);
    if (threadIdx.x == 0) any_long = 0;
    __syncthreads();
    for (int64_t c = c_begin + threadIdx.x; c <= c_end && c < n_chunks; c += blockDim.x)  // includes c_end: slot 1
        if (a.chunk_row[c] >= 0) any_long = 1;
    __syncthreads();
    if (!any_long) return;
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
    for (int64_t c = c_begin; c < c_end; ++c) {
        for (int slot = 0; slot < 2; ++slot) {
            int32_t row = -1;  // workgroup-uniform
            if (slot == 0) row = a.chunk_row[c];
            else if (c + 1 < n_chunks) {
                row = a.chunk_row[c + 1];
                if (row >= 0 && (int64_t)a.rowptr[row] <= c * ch) row = -1;  // same row as slot 0, not a new head
            }
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

// For every long row add its per-chunk partials in chunk order and write (or accumulate into) the output
// row.  The row is combined by the wave that finds it at its FIRST full chunk (the row's head piece, if any,
// sits in slot 1 of the chunk before).
template <typename T>
__global__ __launch_bounds__(256) void csr_spmm_longrow_combine_kernel(const SpmmArgs<T> a, int64_t n_chunks,
                                                                       int chunks_per_block) {
    __shared__ int any_long;
    const int64_t c_begin = (int64_t)blockIdx.x * chunks_per_block;
    const int64_t c_end = min(c_begin + chunks_per_block, n_chunks);
    if (threadIdx.x == 0) any_long = 0;
    __syncthreads();
    for (int64_t c = c_begin + threadIdx.x; c < c_end; c += blockDim.x)
        if (a.chunk_row[c] >= 0) any_long = 1;
    __syncthreads();
    if (!any_long) return;
    const int ch = a.long_thresh;
    const int lane = threadIdx.x & (kWave - 1);
    for (int64_t c = c_begin + (threadIdx.x >> 6); c < c_end; c += 4) {  // one wave per chunk
        const int32_t row = a.chunk_row[c];
        if (row < 0) continue;
        const int start = a.rowptr[row], end = a.rowptr[row + 1];
        if ((int64_t)(start + ch - 1) / ch != c) continue;  // not the row's first owned chunk
        const int64_t c_last = (end - 1) / ch;
        const bool head = (int64_t)start != c * ch;          // a head piece lives in chunk c-1, slot 1
        for (int col = lane; col < a.k; col += kWave) {
            float acc = a.acc_mode ? to_f32<T>(a.out[(int64_t)row * a.k + col]) : 0.f;
            if (head) acc += a.partial[(2 * (c - 1) + 1) * (int64_t)a.k + col];
            for (int64_t q = c; q <= c_last; ++q) acc += a.partial[(2 * q) * (int64_t)a.k + col];
            a.out[(int64_t)row * a.k + col] = from_f32<T>(acc);
        }
    }
}

static inline int64_t n_chunks_for(int64_t nnz, int thresh) { return (nnz + thresh - 1) / thresh; }

// Threshold above which a row is split: the sequential time of a row of T edges (~T/UNROLL gather
// round trips) should stay a small fraction of the whole launch (~nnz / 13 GEdges/s).
static inline int pick_long_thresh(int64_t nnz) {
    if (g_tuning[kTuneLongThresh] > 0) return g_tuning[kTuneLongThresh];
    int t = 256;
    while (t < 4096 && (int64_t)t * 16384 < nnz) t <<= 1;
    return t;
}

static inline size_t chunk_row_bytes(int64_t n_chunks) { return ((size_t)(n_chunks + 1) * sizeof(int32_t) + 255) / 256 * 256; }

template <typename T, int VEC, int LPR, int UNROLL, int WMODE, bool EXACT>
static int launch_rowgroup(SpmmArgs<T> a, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    a.rows_seq = std::max(1, std::min(64, g_tuning[kTuneRowsSeq]));
    const int64_t RPB = (kWave / LPR) * 4 * a.rows_seq;
    const int64_t n_rowblocks = (a.m + RPB - 1) / RPB;
    const int64_t tiles = ((int64_t)a.k + (int64_t)LPR * VEC - 1) / ((int64_t)LPR * VEC);
    if (n_rowblocks == 0) return COGDL_HIP_OK;
    if (n_rowblocks > 0x7fffffff / (kXcds * 64) || tiles > 65535) return COGDL_HIP_ERANGE;
    a.rowblocks = make_xcd_map(n_rowblocks);
    a.long_thresh = INT_MAX;
    a.chunk_row = nullptr;
    int64_t n_chunks = 0;
    if (workspace && a.nnz > 0) {
        const int t = pick_long_thresh(a.nnz);
        n_chunks = n_chunks_for(a.nnz, t);
        const size_t need = chunk_row_bytes(n_chunks) + (size_t)n_chunks * 2 * a.k * sizeof(float);
        if (workspace_bytes < need) return COGDL_HIP_EWORKSPACE;
        if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
        a.long_thresh = t;
        a.chunk_row = (int32_t *)workspace;
        a.partial = (float *)((char *)workspace + chunk_row_bytes(n_chunks));
    }
    dim3 grid(xcd_grid(a.rowblocks), (unsigned)tiles);
    hipLaunchKernelGGL((csr_spmm_rowgroup_kernel<T, VEC, LPR, UNROLL, WMODE, EXACT>), grid, dim3(256), 0, stream, a);
    if (n_chunks > 0) {
        // a bounded number of workgroups, each owning a contiguous run of chunks (cheap to skip when no row is long)
        const int64_t max_wg = std::max(64, g_tuning[kTuneLongGrid]);
        const int cpb = (int)((n_chunks + max_wg - 1) / max_wg);
        const unsigned gx = (unsigned)((n_chunks + cpb - 1) / cpb);
        hipLaunchKernelGGL((csr_spmm_longrow_partial_kernel<T, VEC, LPR, UNROLL, WMODE>), dim3(gx, (unsigned)tiles),
                           dim3(256), 0, stream, a, n_chunks, cpb);
        hipLaunchKernelGGL((csr_spmm_longrow_combine_kernel<T>), dim3(gx), dim3(256), 0, stream, a, n_chunks, cpb);
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

// Vector width and lanes per row.  Every lane's VEC columns must stay inside one row (k % VEC == 0), inside
// one head for mhspmm (fdim % VEC == 0), and be naturally aligned.  Measured on MI355X (arxiv-shaped graph):
// narrow rows (k/4 <= 16 lanes) are fastest with 16-byte lanes and several rows per wave (less per-row
// overhead); from 64 columns up a whole wave per row wins (scalar column broadcast, no intra-wave length
// divergence), so the vector is narrowed until the row fills 64 lanes.
template <typename T, int WMODE>
static int spmm_auto(const SpmmArgs<T> &a, void *ws, size_t wsb, hipStream_t s) {
    constexpr int MAXV = 16 / sizeof(T);
    const int64_t unit = (WMODE == 2) ? a.fdim : a.k;
    auto legal = [&](int v) {
        return v <= MAXV && unit % v == 0 && aligned_to(a.x, v * sizeof(T)) && aligned_to(a.out, v * sizeof(T));
    };
    int vec = 1;
    while (vec * 2 <= MAXV && legal(vec * 2)) vec *= 2;                                   // widest legal
    while (vec > 1 && (vec / 2) * sizeof(T) >= 4 && (int64_t)a.k / vec > 16 && (int64_t)a.k / (vec / 2) <= kWave) vec /= 2;
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
    return chunk_row_bytes(n_chunks) + (size_t)n_chunks * 2 * (size_t)k * sizeof(float);
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
