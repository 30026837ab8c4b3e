// gat_tiled.hip -- fused GAT backward for the shapes one lane group does not cover (gat_op.h, TILED = true): rows of more
// than 64 lanes x 16 bytes (8 heads x 64 features, 16 x 64, ...) and heads whose width is not a power-of-two number of
// lanes (6 heads x 12 features).  The reference's FusedGATFunction.backward has no shape limit
// (cogdl/operators/fused_gat.py:28-40).
//
// The [H*F] row is cut into column tiles of LPR * VEC columns (blockIdx.y, like csr_spmm).  What crosses tiles are the
// per-(row, head) scalars: D = <g, out>, the dot product <g, sum_e c_e d_e feat_e> and, in the column pass,
// <feat, sum_e c_e d_e g_e>.  Every tile reduces its lanes of a head with a segmented shuffle sum (seg_sum: a head is any
// contiguous run of lanes), writes one partial per (tile, row, head) and a finish kernel adds the partials of the tiles
// a head touches in tile order -- no atomics, deterministic.  The per-head edge sums (sum_e c_e, sum_e c_e D) are identical
// in every tile; the tile that holds the head's first column writes them.
//   row pass (tiles)  ->  gat_finish_kernel<ROW>: D, grad_attn_row  ->  column pass (tiles)  ->  gat_finish_kernel<COL>
#include "gat_op.h"

namespace cogdl {

// out[r,h] = sum_{t in tiles of head h} pdot[t,r,h]  -  (ROW ? D[r,h] * hsum[r,h] : hsum[r,h]);   ROW also writes D.
template <bool ROW>
__global__ __launch_bounds__(256) void gat_finish_kernel(const float *__restrict__ pdot, const float *__restrict__ pd,
                                                         const float *__restrict__ hsum, float4 *__restrict__ stats,
                                                         const float *__restrict__ attn_row, const float *__restrict__ edge_max,
                                                         const float *__restrict__ edge_sum, float *__restrict__ out,
                                                         int64_t rows, int heads, int fdim, int tile_cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * heads) return;
    const int h = (int)(i % heads);
    const int t0 = (h * fdim) / tile_cols, t1 = ((h + 1) * fdim - 1) / tile_cols;
    float dot = 0.f, d = 0.f;
    for (int t = t0; t <= t1; ++t) {
        dot += pdot[(int64_t)t * rows * heads + i];
        if constexpr (ROW) d += pd[(int64_t)t * rows * heads + i];
    }
    if constexpr (ROW) {
        const float ls = edge_sum[i];
        stats[i] = make_float4(attn_row[i], gat_neg_max_log(edge_max[i], ls), d, ls > 0.f ? 1.f / ls : 0.f);  // (the column pass's per-edge record)
        out[i] = dot - d * hsum[i];
    } else {
        out[i] = dot - hsum[i];
    }
}

template <typename T, int VEC, int LPR, bool DROP>
static int launch_tiled(const GatBwdArgs &b, int64_t tiles, hipStream_t s) {
    GatDrop row_drop = b.drop;
    row_drop.eid = nullptr;
    const int tile_cols = LPR * VEC;
    GatBwdRowOp<T, VEC, LPR, 4, DROP, true> row_op{b.ar, b.ac, (const T *)b.feat, b.emax, b.esum, (const T *)b.out,
                                                   (const T *)b.gout, b.stats, b.gar, b.slope, (int)b.h, (int)b.f, 0,
                                                   row_drop, GatTiles{b.t_pdot, b.t_pd, b.t_hsum, b.v}};
    int rc = launch_rowreduce(row_op, b.rowptr, b.colind, b.v, b.nnz, tiles, b.ws_row, b.wsb_row, s);
    if (rc != COGDL_HIP_OK) return rc;
    if (b.v > 0) {
        hipLaunchKernelGGL(gat_finish_kernel<true>, dim3((unsigned)((b.v * b.h + 255) / 256)), dim3(256), 0, s, b.t_pdot,
                           b.t_pd, b.t_hsum, b.stats, b.ar, b.emax, b.esum, b.gar, b.v, (int)b.h, (int)b.f, tile_cols);
        if ((rc = launch_status()) != COGDL_HIP_OK) return rc;
    }
    GatBwdColOp<T, VEC, LPR, 4, DROP, true> col_op{b.ar, b.ac, (const T *)b.feat, b.stats,
                                                   (const T *)b.gout, (T *)b.gfeat, b.gac, b.slope, (int)b.h, (int)b.f,
                                                   0, b.drop, GatTiles{b.t_pdot, nullptr, b.t_hsum, b.n_src}};
    rc = launch_rowreduce(col_op, b.colptr, b.rowind, b.n_src, b.nnz, tiles, b.ws_col, b.wsb_col, s);
    if (rc != COGDL_HIP_OK) return rc;
    if (b.n_src > 0) {
        hipLaunchKernelGGL(gat_finish_kernel<false>, dim3((unsigned)((b.n_src * b.h + 255) / 256)), dim3(256), 0, s,
                           b.t_pdot, nullptr, b.t_hsum, nullptr, nullptr, nullptr, nullptr, b.gac, b.n_src, (int)b.h, (int)b.f,
                           tile_cols);
        rc = launch_status();
    }
    return rc;
}

template <typename T, int VEC, bool DROP>
static int tiled_lpr(const GatBwdArgs &b, const GatBwdGeometry &g, hipStream_t s) {
    if (g.lpr == 16) return launch_tiled<T, VEC, 16, DROP>(b, g.tiles, s);
    return launch_tiled<T, VEC, 64, DROP>(b, g.tiles, s);
}

template <typename T, bool DROP>
static int tiled_vec(const GatBwdArgs &b, const GatBwdGeometry &g, hipStream_t s) {
    switch (g.vec) {
        case 8:
            if constexpr (sizeof(T) == 2) return tiled_lpr<T, 8, DROP>(b, g, s);
            return COGDL_HIP_EINVAL;
        case 4: return tiled_lpr<T, 4, DROP>(b, g, s);
        case 2: return tiled_lpr<T, 2, DROP>(b, g, s);
        default: return tiled_lpr<T, 1, DROP>(b, g, s);
    }
}

template <bool DROP>
static int tiled_any(const GatBwdArgs &b, const GatBwdGeometry &g, int dtype, hipStream_t s) {
    switch (dtype) {
        case COGDL_HIP_F32: return tiled_vec<float, DROP>(b, g, s);
        case COGDL_HIP_F16: return tiled_vec<__half, DROP>(b, g, s);
        default: return tiled_vec<__hip_bfloat16, DROP>(b, g, s);
    }
}

int gat_bwd_tiled(const GatBwdArgs &b, const GatBwdGeometry &g, int dtype, bool drop, hipStream_t s) {
    return drop ? tiled_any<true>(b, g, dtype, s) : tiled_any<false>(b, g, dtype, s);
}

}  // namespace cogdl
