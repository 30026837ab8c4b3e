// gat.hip -- C ABI of the fused GAT attention + aggregation (functors: gat_op.h) and the plain (no dropout, one lane
// group per row) instantiations.  gat_drop.hip holds the attention-dropout instantiations, gat_tiled.hip the backward
// for wide rows / heads that are not a power-of-two number of lanes.
#include "gat_op.h"

using namespace cogdl;

namespace {

int check_fwd(const int32_t *rowptr, const float *attn_row, const float *attn_col, const void *feat, const void *out,
              const float *edge_max, const float *edge_sum, int64_t v, int64_t h, int64_t f, int64_t nnz) {
    if (v < 0 || h < 0 || f < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (v == 0 || h == 0 || f == 0) return -1;  // nothing to do
    if (!rowptr || !attn_row || !attn_col || !feat || !out || !edge_max || !edge_sum) return COGDL_HIP_EINVAL;
    if (h * f > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    return COGDL_HIP_OK;
}

// Layout of the backward workspace:
//   [row-pass records {attn_row, max, 1/sum, D}: v*h float4][TILED: pdot tiles*max(v,n_src)*h | pd tiles*v*h | hsum max(v,n_src)*h]
//   [long-row scratch of the row pass][... of the column pass]
struct BwdLayout {
    size_t d, pdot, pd, hsum, row, col, total_min, total;
};
size_t pad256(size_t b) { return (b + 255) / 256 * 256; }
BwdLayout bwd_layout(const GatBwdGeometry &g, int64_t v, int64_t n_src, int64_t h, int64_t nnz) {
    BwdLayout L{};
    L.d = gat_stats_bytes(v, h);
    if (g.tiled) {
        const size_t rows = (size_t)std::max<int64_t>(std::max(v, n_src), 0);
        L.pdot = pad256((size_t)g.tiles * rows * (size_t)h * sizeof(float));
        L.pd = pad256((size_t)g.tiles * (size_t)std::max<int64_t>(v, 0) * (size_t)h * sizeof(float));
        L.hsum = pad256(rows * (size_t)h * sizeof(float));
    }
    L.total_min = L.d + L.pdot + L.pd + L.hsum;
    if (nnz > 0) {
        L.row = rowreduce_workspace_bytes(nnz, g.tiles * (int64_t)(g.vec + 1) * g.lpr);
        L.col = rowreduce_workspace_bytes(nnz, g.tiles * (int64_t)(2 * g.vec + 1) * g.lpr);
    }
    L.total = L.total_min + L.row + L.col;
    return L;
}

int gat_bwd_entry(const int32_t *rowptr, const int32_t *colind, const int32_t *colptr, const int32_t *rowind,
                  const float *attn_row, const float *attn_col, const void *feat, float negative_slope,
                  const float *edge_max, const float *edge_sum, const void *out, const void *grad_out, void *grad_feat,
                  float *grad_attn_row, float *grad_attn_col, void *workspace, size_t workspace_bytes, int64_t v,
                  int64_t n_src, int64_t h, int64_t f, int64_t nnz, int dtype, const GatDrop *drop, void *stream) {
    if (v < 0 || n_src < 0 || h <= 0 || f <= 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (dtype != COGDL_HIP_F32 && dtype != COGDL_HIP_F16 && dtype != COGDL_HIP_BF16) return COGDL_HIP_EDTYPE;
    if (h * f > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (drop && h > 64) return COGDL_HIP_EUNSUPPORTED;  // keep bits of an edge: one 64-bit word
    const int elem_bytes = dtype == COGDL_HIP_F32 ? 4 : 2;
    if (!rowptr || !colptr || !attn_row || !attn_col || !feat || !edge_max || !edge_sum || !out || !grad_out ||
        !grad_feat || !grad_attn_row || !grad_attn_col || !workspace)
        return COGDL_HIP_EINVAL;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EWORKSPACE;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) |
                           reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_feat);
    const int align = (bits % 16 == 0) ? 16 : (bits % 8 == 0) ? 8 : (bits % 4 == 0) ? 4 : 2;
    if (align < elem_bytes) return COGDL_HIP_EALIGN;
    const GatBwdGeometry g = gat_bwd_geometry(h, f, align, elem_bytes);
    if (g.tiles > 65535) return COGDL_HIP_ERANGE;
    const BwdLayout L = bwd_layout(g, v, n_src, h, nnz);
    if (workspace_bytes < L.total_min) return COGDL_HIP_EWORKSPACE;
    GatBwdArgs b{};
    b.rowptr = rowptr, b.colind = colind, b.colptr = colptr, b.rowind = rowind;
    b.ar = attn_row, b.ac = attn_col, b.feat = feat, b.emax = edge_max, b.esum = edge_sum;
    b.out = out, b.gout = grad_out, b.slope = negative_slope;
    b.gfeat = grad_feat, b.gar = grad_attn_row, b.gac = grad_attn_col;
    b.v = v, b.n_src = n_src, b.h = h, b.f = f, b.nnz = nnz;
    char *w = (char *)workspace;
    b.stats = (float4 *)w;
    b.t_pdot = (float *)(w + L.d);
    b.t_pd = (float *)(w + L.d + L.pdot);
    b.t_hsum = (float *)(w + L.d + L.pdot + L.pd);
    // the long-row scratch is used only when the caller's workspace covers all of it
    if (nnz > 0 && workspace_bytes >= L.total) {
        b.ws_row = w + L.total_min;
        b.wsb_row = L.row;
        b.ws_col = w + L.total_min + L.row;
        b.wsb_col = L.col;
    }
    if (drop) b.drop = *drop;
    hipStream_t s = (hipStream_t)stream;
    if (g.tiled) return gat_bwd_tiled(b, g, dtype, drop != nullptr, s);
    if (drop) return gat_bwd_drop(b, g, dtype, s);
    return gat_bwd_any<false>(b, g, dtype, s);
}

GatDrop make_drop(float p, uint64_t seed, const int32_t *perm) {
    GatDrop d{};
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32);
    drop_params(p, d.thresh, d.scale);
    d.eid = perm;
    return d;
}

__global__ void edge_dropout_mask_kernel(float *__restrict__ mask, int64_t nnz, int heads, GatDrop d) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint64_t bits = drop_keep_bits(d.seed_lo, d.seed_hi, (uint32_t)e, heads, d.thresh);
    for (int h = 0; h < heads; ++h) mask[e * heads + h] = ((bits >> h) & 1ull) ? d.scale : 0.f;
}

}  // namespace

extern "C" size_t cogdl_hip_gat_fwd_workspace_bytes(int64_t nnz, int64_t h, int64_t f, int dtype) {
    if (nnz <= 0 || h <= 0 || f <= 0) return 0;
    const RowGeometry g = gat_fwd_geometry(h, f, dtype == COGDL_HIP_F32 ? 4 : 2, 16);
    return rowreduce_workspace_bytes(nnz, g.tiles * (g.vec + 2) * g.lpr);
}

extern "C" int cogdl_hip_gat_fwd(const int32_t *rowptr, const int32_t *colind, const float *attn_row,
                                 const float *attn_col, const void *feat, float negative_slope, void *out,
                                 float *edge_max, float *edge_sum, int64_t v, int64_t h, int64_t f, int64_t nnz,
                                 int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    const int rc = check_fwd(rowptr, attn_row, attn_col, feat, out, edge_max, edge_sum, v, h, f, nnz);
    if (rc != COGDL_HIP_OK) return rc < 0 ? COGDL_HIP_OK : rc;
    GatFwdArgs a{rowptr, colind, attn_row, attn_col, feat, negative_slope, out, edge_max, edge_sum, v, h, f, nnz,
                 GatDrop{}};
    return gat_fwd_any<false>(a, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int cogdl_hip_gat_dropout_fwd(const int32_t *rowptr, const int32_t *colind, const float *attn_row,
                                         const float *attn_col, const void *feat, float negative_slope, float p,
                                         uint64_t seed, void *out, float *edge_max, float *edge_sum, int64_t v,
                                         int64_t h, int64_t f, int64_t nnz, int dtype, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    const int rc = check_fwd(rowptr, attn_row, attn_col, feat, out, edge_max, edge_sum, v, h, f, nnz);
    if (rc != COGDL_HIP_OK) return rc < 0 ? COGDL_HIP_OK : rc;
    if (!(p >= 0.f && p <= 1.f)) return COGDL_HIP_EINVAL;
    if (h > 64) return COGDL_HIP_EUNSUPPORTED;
    GatFwdArgs a{rowptr, colind, attn_row, attn_col, feat, negative_slope, out, edge_max, edge_sum, v, h, f, nnz,
                 make_drop(p, seed, nullptr)};
    return gat_fwd_drop(a, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t cogdl_hip_gat_bwd_workspace_bytes(int64_t v, int64_t n_src, int64_t h, int64_t f, int64_t nnz,
                                                    int dtype) {
    if (h <= 0 || f <= 0) return 256;
    const GatBwdGeometry g = gat_bwd_geometry(h, f, 16, dtype == COGDL_HIP_F32 ? 4 : 2);
    return bwd_layout(g, v, n_src, h, nnz).total + 256;
}

extern "C" int cogdl_hip_gat_bwd(const int32_t *rowptr, const int32_t *colind, const int32_t *colptr,
                                 const int32_t *rowind, const float *attn_row, const float *attn_col,
                                 const void *feat, float negative_slope, const float *edge_max,
                                 const float *edge_sum, const void *out, const void *grad_out, void *grad_feat,
                                 float *grad_attn_row, float *grad_attn_col, void *workspace, size_t workspace_bytes,
                                 int64_t v, int64_t n_src, int64_t h, int64_t f, int64_t nnz, int dtype,
                                 void *stream) {
    return gat_bwd_entry(rowptr, colind, colptr, rowind, attn_row, attn_col, feat, negative_slope, edge_max, edge_sum,
                         out, grad_out, grad_feat, grad_attn_row, grad_attn_col, workspace, workspace_bytes, v, n_src,
                         h, f, nnz, dtype, nullptr, stream);
}

extern "C" int cogdl_hip_gat_dropout_bwd(const int32_t *rowptr, const int32_t *colind, const int32_t *colptr,
                                         const int32_t *rowind, const int32_t *perm, const float *attn_row,
                                         const float *attn_col, const void *feat, float negative_slope, float p,
                                         uint64_t seed, const float *edge_max, const float *edge_sum, const void *out,
                                         const void *grad_out, void *grad_feat, float *grad_attn_row,
                                         float *grad_attn_col, void *workspace, size_t workspace_bytes, int64_t v,
                                         int64_t n_src, int64_t h, int64_t f, int64_t nnz, int dtype, void *stream) {
    if (!(p >= 0.f && p <= 1.f) || (!perm && nnz > 0)) return COGDL_HIP_EINVAL;
    const GatDrop d = make_drop(p, seed, perm);
    return gat_bwd_entry(rowptr, colind, colptr, rowind, attn_row, attn_col, feat, negative_slope, edge_max, edge_sum,
                         out, grad_out, grad_feat, grad_attn_row, grad_attn_col, workspace, workspace_bytes, v, n_src,
                         h, f, nnz, dtype, &d, stream);
}

// ---- XCD-partitioned plans (rowreduce.h: virtual rows) ------------------------------------------------------------------
extern "C" size_t cogdl_hip_gat_fwd_xcd_workspace_bytes(int64_t n_parts, int64_t h, int64_t f, int dtype) {
    if (h <= 0 || f <= 0) return 256;
    const RowGeometry g = gat_fwd_geometry(h, f, dtype == COGDL_HIP_F32 ? 4 : 2, 16);
    return vrows_workspace_bytes(n_parts, g.tiles * (g.vec + 2) * g.lpr);
}

extern "C" int cogdl_hip_gat_fwd_xcd(const cogdl_hip_vrows *plan, const float *attn_row, const float *attn_col,
                                     const void *feat, float negative_slope, float p, uint64_t seed, void *out,
                                     float *edge_max, float *edge_sum, int64_t v, int64_t h, int64_t f, int dtype,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    int rc = vrows_valid(plan);
    if (rc != COGDL_HIP_OK) return rc;
    const int32_t dummy = 0;
    rc = check_fwd(&dummy, attn_row, attn_col, feat, out, edge_max, edge_sum, v, h, f, plan->nnz);
    if (rc != COGDL_HIP_OK) return rc < 0 ? COGDL_HIP_OK : rc;
    if (!(p >= 0.f && p <= 1.f)) return COGDL_HIP_EINVAL;
    // (the plan kernels address the gathered tables with 24 x 24 -> 32-bit offsets, common.h: gather_row; the caller
    //  vouches for the column ids -- < 2^24 source rows, tables below 4 GiB: cogdl_amd/xcdplan.py wanted())
    if (h * f * 4 >= (1 << 24)) return COGDL_HIP_EUNSUPPORTED;
    GatFwdArgs a{nullptr, nullptr, attn_row, attn_col, feat, negative_slope, out, edge_max, edge_sum, v, h, f, plan->nnz,
                 GatDrop{}, plan};
    if (p > 0.f) {
        if (h > 64) return COGDL_HIP_EUNSUPPORTED;
        if (!plan->eid && plan->nnz > 0) return COGDL_HIP_EINVAL;  // (the mask is a function of the CSR position)
        a.drop = make_drop(p, seed, plan->eid);
        return gat_fwd_drop(a, dtype, workspace, workspace_bytes, (hipStream_t)stream);
    }
    return gat_fwd_any<false>(a, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

namespace {
struct BwdXcdLayout {
    size_t d, row, col, total;
};
BwdXcdLayout bwd_xcd_layout(const GatBwdGeometry &g, int64_t v, int64_t h, int64_t parts_row, int64_t parts_col) {
    BwdXcdLayout L{};
    L.d = gat_stats_bytes(v, h);
    L.row = vrows_workspace_bytes(parts_row, g.tiles * (int64_t)(g.vec + 1) * g.lpr);
    L.col = vrows_workspace_bytes(parts_col, g.tiles * (int64_t)(2 * g.vec + 1) * g.lpr);
    L.total = L.d + L.row + L.col;
    return L;
}
}  // namespace

extern "C" size_t cogdl_hip_gat_bwd_xcd_workspace_bytes(int64_t v, int64_t h, int64_t f, int64_t n_parts_row,
                                                        int64_t n_parts_col, int dtype) {
    if (h <= 0 || f <= 0) return 256;
    const GatBwdGeometry g = gat_bwd_geometry(h, f, 16, dtype == COGDL_HIP_F32 ? 4 : 2);
    return bwd_xcd_layout(g, v, h, n_parts_row, n_parts_col).total + 256;
}

extern "C" int cogdl_hip_gat_bwd_xcd(const cogdl_hip_vrows *plan_csr, const cogdl_hip_vrows *plan_csc,
                                     const float *attn_row, const float *attn_col, const void *feat, float negative_slope,
                                     float p, uint64_t seed, const float *edge_max, const float *edge_sum, const void *out,
                                     const void *grad_out, void *grad_feat, float *grad_attn_row, float *grad_attn_col,
                                     void *workspace, size_t workspace_bytes, int64_t v, int64_t n_src, int64_t h,
                                     int64_t f, int dtype, void *stream) {
    int rc = vrows_valid(plan_csr);
    if (rc == COGDL_HIP_OK) rc = vrows_valid(plan_csc);
    if (rc != COGDL_HIP_OK) return rc;
    if (v < 0 || n_src < 0 || h <= 0 || f <= 0 || plan_csr->nnz != plan_csc->nnz) return COGDL_HIP_EINVAL;
    if (dtype != COGDL_HIP_F32 && dtype != COGDL_HIP_F16 && dtype != COGDL_HIP_BF16) return COGDL_HIP_EDTYPE;
    if (h * f > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (!(p >= 0.f && p <= 1.f)) return COGDL_HIP_EINVAL;
    const bool drop = p > 0.f;
    if (drop && h > 64) return COGDL_HIP_EUNSUPPORTED;
    if (drop && plan_csr->nnz > 0 && (!plan_csr->eid || !plan_csc->eid)) return COGDL_HIP_EINVAL;
    // (24 x 24 -> 32-bit table offsets, common.h: gather_row -- the widest gathered row is the float4 record per head)
    if (std::max(v, n_src) >= (1 << 24) || h * std::max<int64_t>(f * 4, 16) >= (1 << 24) ||
        std::max(v, n_src) * h * std::max<int64_t>(f * 4, 16) >= ((int64_t)1 << 32))
        return COGDL_HIP_EUNSUPPORTED;
    const int elem_bytes = dtype == COGDL_HIP_F32 ? 4 : 2;
    if (!attn_row || !attn_col || !feat || !edge_max || !edge_sum || !out || !grad_out || !grad_feat || !grad_attn_row ||
        !grad_attn_col || !workspace)
        return COGDL_HIP_EINVAL;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EWORKSPACE;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) |
                           reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_feat);
    const int align = (bits % 16 == 0) ? 16 : (bits % 8 == 0) ? 8 : (bits % 4 == 0) ? 4 : 2;
    if (align < elem_bytes) return COGDL_HIP_EALIGN;
    const GatBwdGeometry g = gat_bwd_geometry(h, f, align, elem_bytes);
    if (g.tiled) return COGDL_HIP_EUNSUPPORTED;  // (column tiles keep the ordinary entry)
    const BwdXcdLayout L = bwd_xcd_layout(g, v, h, plan_csr->n_parts, plan_csc->n_parts);
    if (workspace_bytes < L.total) return COGDL_HIP_EWORKSPACE;
    GatBwdArgs b{};
    b.ar = attn_row, b.ac = attn_col, b.feat = feat, b.emax = edge_max, b.esum = edge_sum;
    b.out = out, b.gout = grad_out, b.slope = negative_slope;
    b.gfeat = grad_feat, b.gar = grad_attn_row, b.gac = grad_attn_col;
    b.v = v, b.n_src = n_src, b.h = h, b.f = f, b.nnz = plan_csr->nnz;
    char *w = (char *)workspace;
    b.stats = (float4 *)w;
    b.ws_row = w + L.d, b.wsb_row = L.row;
    b.ws_col = w + L.d + L.row, b.wsb_col = L.col;
    b.vr_row = plan_csr, b.vr_col = plan_csc;
    hipStream_t s = (hipStream_t)stream;
    if (drop) {
        b.drop = make_drop(p, seed, nullptr);
        return gat_bwd_drop(b, g, dtype, s);
    }
    return gat_bwd_any<false>(b, g, dtype, s);
}

extern "C" int cogdl_hip_edge_dropout_mask(int64_t nnz, int64_t h, float p, uint64_t seed, float *mask, void *stream) {
    if (nnz < 0 || h < 0 || !(p >= 0.f && p <= 1.f)) return COGDL_HIP_EINVAL;
    if (nnz == 0 || h == 0) return COGDL_HIP_OK;
    if (!mask) return COGDL_HIP_EINVAL;
    if (h > 64) return COGDL_HIP_EUNSUPPORTED;
    if (nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL(edge_dropout_mask_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mask, nnz, (int)h, make_drop(p, seed, nullptr));
    return launch_status();
}

// The same mask computed on the host by the same header code (philox.h is __host__ __device__): lets the CPU test suite
// pin the generator against the published Philox4x32-10 known-answer vectors without a GPU.
extern "C" int cogdl_hip_edge_dropout_mask_host(int64_t nnz, int64_t h, float p, uint64_t seed, float *mask) {
    if (nnz < 0 || h < 0 || !(p >= 0.f && p <= 1.f)) return COGDL_HIP_EINVAL;
    if (nnz == 0 || h == 0) return COGDL_HIP_OK;
    if (!mask) return COGDL_HIP_EINVAL;
    if (h > 64) return COGDL_HIP_EUNSUPPORTED;
    if (nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    const GatDrop d = make_drop(p, seed, nullptr);
    for (int64_t e = 0; e < nnz; ++e) {
        const uint64_t bits = drop_keep_bits(d.seed_lo, d.seed_hi, (uint32_t)e, (int)h, d.thresh);
        for (int64_t k = 0; k < h; ++k) mask[e * h + k] = ((bits >> k) & 1ull) ? d.scale : 0.f;
    }
    return COGDL_HIP_OK;
}
