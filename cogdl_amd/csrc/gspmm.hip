// gspmm.hip -- "source (op) edge feature, then aggregate" for gfx950: the s_{add,sub,mul}_e_{sum,mean} family and
// scatter_add of cogdl/operators/ops.py:4-103.  The reference composes them from torch ops on a COO graph
// (ops.py:43-52): it materialises the [E, F] message tensor  msg = (n_feat[col] OP e_feat) * weight  and scatters it
// with scatter_add_ (atomics on a GPU: non-deterministic).  Here the message never exists in memory:
//
//     out[v, :] = scale_v * SUM_{j in row v}  w[id_j] * ( x[colind[j], :]  OP  efeat[id_j, :] )         id_j = eid[j] | j
//
// over the destination-sorted (CSR) view of the edges; `eid` maps a CSR position to the caller's edge id when the
// caller's edges are not row-sorted (stable sort => the edges of a row keep the caller's order, so the per-element
// sum has the same association as a sequential scatter_add_ over the COO list).  Same engine as csr_spmm
// (rowreduce.h): a group of LPR lanes per row, coalesced vector gathers of the source row AND of the edge-feature
// row, strictly sequential accumulation, hub rows split over workgroups and merged in a fixed order.  No atomics.
// fp32 only (the reference operators are fp32 torch code).
// HBM-bound: algorithmic bytes per edge = 4 (colind) [+ 4 eid] [+ 4 w] + F*4 (source row) + F*4 | 4 (edge feature).
#include "rowreduce.h"

namespace cogdl {

template <int VEC_, int LPR_, int UNROLL_>
struct GspmmOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_;
    static constexpr bool kReduce = true;
    static constexpr int kLds = 0;
    const int32_t *rowptr;  // for the mean
    const int32_t *eid;     // CSR position -> edge id of efeat / weight (NULL: identity)
    const float *x;         // [n_src, k] (NULL: the message is the edge feature alone -> scatter_add)
    const float *ef;        // [E, k] or [E] (NULL: the message is the source row alone)
    const float *w;         // [E] or NULL
    float *out;             // [m, k]
    int k;
    int op;         // COGDL_HIP_GSPMM_{ADD,SUB,MUL,WMUL}
    int ef_scalar;  // efeat is [E]: one value per edge, broadcast over the columns
    int mean;

    struct Ctx {
        int col0;
        bool col_ok;
        const float *xcol, *efcol;
    };
    struct State { float acc[VEC]; };
    struct LaneVals {
        int id;
        float w, es;
    };
    struct Batch {
        float v[UNROLL][VEC];
        float e[UNROLL][VEC];
        float w[UNROLL];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int tile) const {
        Ctx c;
        c.col0 = (tile * LPR + l) * VEC;
        c.col_ok = c.col0 < k;
        const int cc = c.col_ok ? c.col0 : 0;
        c.xcol = x ? x + cc : nullptr;
        c.efcol = (ef && !ef_scalar) ? ef + cc : nullptr;
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &, int64_t, bool) const {}
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { init_zero(s); }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &lv, int64_t e) const {
        lv.id = eid ? eid[e] : (int)e;
        lv.w = w ? w[lv.id] : 1.f;
        lv.es = (ef && ef_scalar) ? ef[lv.id] : 0.f;
    }
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &lv, int sub,
                                          int jj) const {
        b.w[u] = group_bcast<LPR>(lv.w, sub, jj);
        if (c.xcol) load_vec<float, VEC>(c.xcol + (int64_t)col * k, b.v[u]);
        if (c.efcol) {
            const int id = group_bcast<LPR>(lv.id, sub, jj);
            load_vec<float, VEC>(c.efcol + (int64_t)id * k, b.e[u]);
        } else {
            const float es = group_bcast<LPR>(lv.es, sub, jj);
#pragma unroll
            for (int i = 0; i < VEC; ++i) b.e[u][i] = es;
        }
    }
    // msg = (src OP e) * w, then out += msg: the reference's torch expression (ops.py:19-26,49-51), each step rounded
    // to fp32 (the library is built with -ffp-contract=off).  Multiplying by w == 1.f is exact.
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float msg;
            if (!c.xcol) msg = b.e[u][i];
            else if (!ef) msg = b.v[u][i];
            else if (op == COGDL_HIP_GSPMM_ADD) msg = b.v[u][i] + b.e[u][i];
            else if (op == COGDL_HIP_GSPMM_SUB) msg = b.v[u][i] - b.e[u][i];
            else if (op == COGDL_HIP_GSPMM_WMUL) msg = (w ? b.v[u][i] * b.w[u] : b.v[u][i]) * b.e[u][i];  // weight first
            else msg = b.v[u][i] * b.e[u][i];
            if (w && op != COGDL_HIP_GSPMM_WMUL) msg = msg * b.w[u];
            s.acc[i] = s.acc[i] + (valid ? msg : 0.f);
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &, State &, int, int, int, int, int, float *, const LaneVals &) const {}
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t row, bool ok) const {
        if (!(ok && c.col_ok)) return;
        float r[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) r[i] = s.acc[i];
        if (mean) {  // op_aggr "mean" (ops.py:31-37): out * deg^-1 with 1/0 -> 0
            const int deg = rowptr[row + 1] - rowptr[row];
            const float inv = deg > 0 ? 1.0f / (float)deg : 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) r[i] = r[i] * inv;
        }
        store_vec<float, VEC>(out + row * (int64_t)k + c.col0, r);
    }
    __device__ __forceinline__ void pack(const State &s, float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) rec[i] = s.acc[i];
    }
    __device__ __forceinline__ void unpack(State &s, const float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = rec[i];
    }
    __device__ __forceinline__ void merge(const Ctx &, State &a, const State &b) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) a.acc[i] += b.acc[i];
    }
};

struct GspmmArgs {
    const int32_t *rowptr, *colind, *eid;
    const float *x, *ef, *w;
    float *out;
    int64_t m, nnz;
    int k, op, ef_scalar, mean;
    const cogdl_hip_vrows *vr = nullptr;  // XCD-partitioned, length-ordered plan of the destination-sorted view (rowreduce.h)
};

constexpr int kGspmmUnroll = 4;  // two gathers per edge: half of csr_spmm's unroll keeps the same loads in flight

template <int VEC, int LPR>
static int launch_gspmm(const GspmmArgs &a, void *ws, size_t wsb, hipStream_t s) {
    GspmmOp<VEC, LPR, kGspmmUnroll> op{a.rowptr, a.eid, a.x, a.ef, a.w, a.out, a.k, a.op, a.ef_scalar, a.mean};
    const int64_t tiles = ((int64_t)a.k + (int64_t)LPR * VEC - 1) / ((int64_t)LPR * VEC);
    if (a.vr) {  // plan position j -> edge id plan->eid[j] (the plan was built over the sorted view with the view's perm as base)
        op.eid = a.vr->eid;
        return launch_rowreduce_vrows(op, a.vr, tiles, ws, wsb, s);
    }
    return launch_rowreduce(op, a.rowptr, a.colind, a.m, a.nnz, tiles, ws, wsb, s);
}

template <int VEC>
static int gspmm_lpr(const GspmmArgs &a, int lpr, void *ws, size_t wsb, hipStream_t s) {
    switch (lpr) {
        case 4: return launch_gspmm<VEC, 4>(a, ws, wsb, s);
        case 8: return launch_gspmm<VEC, 8>(a, ws, wsb, s);
        case 16: return launch_gspmm<VEC, 16>(a, ws, wsb, s);
        case 32: return launch_gspmm<VEC, 32>(a, ws, wsb, s);
        default: return launch_gspmm<VEC, 64>(a, ws, wsb, s);
    }
}

static int gspmm_alignment(const GspmmArgs &a) {
    uintptr_t v = reinterpret_cast<uintptr_t>(a.out);
    if (a.x) v |= reinterpret_cast<uintptr_t>(a.x);
    if (a.ef && !a.ef_scalar) v |= reinterpret_cast<uintptr_t>(a.ef);
    return (v % 16 == 0) ? 16 : (v % 8 == 0) ? 8 : 4;
}

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_gspmm_workspace_bytes(int64_t nnz, int64_t k) {
    if (nnz <= 0 || k <= 0) return 0;
    const RowGeometry g = spmm_geometry(k, k, 4, 16);
    return rowreduce_workspace_bytes(nnz, g.tiles * g.vec * g.lpr);
}

extern "C" int cogdl_hip_gspmm(const int32_t *rowptr, const int32_t *colind, const int32_t *eid, const float *x,
                               const float *efeat, int efeat_is_scalar, const float *weight, int op, int mean,
                               float *out, int64_t m, int64_t k, int64_t nnz, void *workspace,
                               size_t workspace_bytes, void *stream) {
    if (m < 0 || k < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (m == 0 || k == 0) return COGDL_HIP_OK;
    if (!rowptr || !out || (!x && !efeat) || (nnz > 0 && !colind)) return COGDL_HIP_EINVAL;
    if (op < COGDL_HIP_GSPMM_ADD || op > COGDL_HIP_GSPMM_WMUL) return COGDL_HIP_EINVAL;
    if (op == COGDL_HIP_GSPMM_WMUL && (!x || !efeat)) return COGDL_HIP_EINVAL;
    if (!x && efeat_is_scalar) return COGDL_HIP_EINVAL;  // scatter_add of a scalar per edge: pass it as [E, 1], k = 1
    if (k > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    GspmmArgs a{rowptr, colind, eid, x, efeat, weight, out, m, nnz, (int)k, op, efeat_is_scalar ? 1 : 0, mean ? 1 : 0};
    if (!aligned_to(out, 4) || (x && !aligned_to(x, 4)) || (efeat && !aligned_to(efeat, 4))) return COGDL_HIP_EALIGN;
    const RowGeometry g = spmm_geometry(k, k, 4, gspmm_alignment(a));
    // The workspace was sized for the 16-byte-aligned geometry; a narrower one needs at most as many floats.
    hipStream_t s = (hipStream_t)stream;
    switch (g.vec) {
        case 4: return gspmm_lpr<4>(a, g.lpr, workspace, workspace_bytes, s);
        case 2: return gspmm_lpr<2>(a, g.lpr, workspace, workspace_bytes, s);
        default: return gspmm_lpr<1>(a, g.lpr, workspace, workspace_bytes, s);
    }
}

// The same operator over an XCD-partitioned plan of the destination-sorted view (cogdl_amd/xcdplan.py, cut at the exact-row
// bound: rows up to it are one virtual row each, summed in the caller's edge order exactly as above; longer rows are summed in
// pieces -- which the ordinary launch does too, rowreduce.h long-row path).  What the plan buys on skewed graphs is its slot
// order (rows of one length per wave) and hub rows merged by whole workgroups.  rowptr: the view's row pointer (for `mean`).
extern "C" size_t cogdl_hip_gspmm_xcd_workspace_bytes(int64_t n_parts, int64_t k) {
    if (k <= 0) return 256;
    const RowGeometry g = spmm_geometry(k, k, 4, 16);
    return vrows_workspace_bytes(n_parts, g.tiles * g.vec * g.lpr);
}

extern "C" int cogdl_hip_gspmm_xcd(const cogdl_hip_vrows *plan, const int32_t *rowptr, const float *x, const float *efeat,
                                   int efeat_is_scalar, const float *weight, int op, int mean, float *out, int64_t m, int64_t k,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    int rc = vrows_valid(plan);
    if (rc != COGDL_HIP_OK) return rc;
    if (m < 0 || k < 0) return COGDL_HIP_EINVAL;
    if (m == 0 || k == 0) return COGDL_HIP_OK;
    if (!rowptr || !out || (!x && !efeat)) return COGDL_HIP_EINVAL;
    if (op < COGDL_HIP_GSPMM_ADD || op > COGDL_HIP_GSPMM_WMUL) return COGDL_HIP_EINVAL;
    if (op == COGDL_HIP_GSPMM_WMUL && (!x || !efeat)) return COGDL_HIP_EINVAL;
    if (!x && efeat_is_scalar) return COGDL_HIP_EINVAL;
    if (k > 0x7fffffff) return COGDL_HIP_ERANGE;
    GspmmArgs a{rowptr, nullptr, nullptr, x, efeat, weight, out, m, plan->nnz, (int)k, op, efeat_is_scalar ? 1 : 0, mean ? 1 : 0, plan};
    if (!aligned_to(out, 4) || (x && !aligned_to(x, 4)) || (efeat && !aligned_to(efeat, 4))) return COGDL_HIP_EALIGN;
    const RowGeometry g = spmm_geometry(k, k, 4, gspmm_alignment(a));
    hipStream_t s = (hipStream_t)stream;
    switch (g.vec) {
        case 4: return gspmm_lpr<4>(a, g.lpr, workspace, workspace_bytes, s);
        case 2: return gspmm_lpr<2>(a, g.lpr, workspace, workspace_bytes, s);
        default: return gspmm_lpr<1>(a, g.lpr, workspace, workspace_bytes, s);
    }
}

// ---- per-edge gradients of the family (the backward of src_op_e_aggr_coo, cogdl/operators/ops.py:43-52) ----------------
// For every edge e = (r <- c) of the caller's COO list, from the upstream gradient G [m, k]:
//     g   = G[r, :] * scale[r]                       (scale: 1 / deg for op2 == mean, NULL for sum)
//     gw[e]    = sum_k (x[c, k] OP ef[e, k]) * g[k]                               (gradient of the edge weight)
//     ge[e, :] = { MUL: (g * w[e]) * x[c, :] ;  ADD: g * w[e] ;  SUB: -(g * w[e]) }   (gradient of the edge feature;
//                                                                  summed over k into ge[e] when ef is one scalar per edge)
// with autograd's own roundings in autograd's order (g, then * w, then * x), so a [E, k] result equals the torch
// composition bit for bit.  No [E, k] temporaries: the gathered rows live in registers.  One lane group per edge,
// edge-parallel (no reduction over edges); the sums over k are per-lane column-order partials + one butterfly.
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void gspmm_edge_grad_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                                              const float *__restrict__ G, const float *__restrict__ scale,
                                                              const float *__restrict__ w, const float *__restrict__ x,
                                                              const float *__restrict__ ef, int ef_scalar, int op,
                                                              float *__restrict__ ge, float *__restrict__ gw, int64_t n_edges,
                                                              int k) {
    constexpr int GPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t g0 = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
    for (int64_t e = g0; e < n_edges; e += (int64_t)gridDim.x * GPB) {
        const int64_t r = row[e], c = col[e];
        const float sc = scale ? scale[r] : 1.f;
        const float we = w ? w[e] : 1.f;
        const float es = (ef && ef_scalar) ? ef[e] : 0.f;
        float acc_w = 0.f, acc_e = 0.f;
        for (int c0 = l * VEC; c0 < k; c0 += LPR * VEC) {
            float gv[VEC], xv[VEC], ev[VEC], out[VEC];
            load_vec<float, VEC>(G + r * (int64_t)k + c0, gv);
            if (x) load_vec<float, VEC>(x + c * (int64_t)k + c0, xv);
            if (ef && !ef_scalar) load_vec<float, VEC>(ef + e * (int64_t)k + c0, ev);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float g = scale ? gv[i] * sc : gv[i];
                if (gw) {
                    const float efv = ef_scalar ? es : ev[i];
                    const float m = op == COGDL_HIP_GSPMM_MUL ? xv[i] * efv : (op == COGDL_HIP_GSPMM_SUB ? xv[i] - efv : xv[i] + efv);
                    acc_w = acc_w + m * g;
                }
                const float gm = w ? g * we : g;
                out[i] = op == COGDL_HIP_GSPMM_MUL ? gm * xv[i] : (op == COGDL_HIP_GSPMM_SUB ? -gm : gm);
                acc_e = acc_e + out[i];
            }
            if (ge && !ef_scalar) store_vec<float, VEC>(ge + e * (int64_t)k + c0, out);
        }
        if (gw) {
            const float t = group_sum<LPR>(acc_w);
            if (l == 0) gw[e] = t;
        }
        if (ge && ef_scalar) {
            const float t = group_sum<LPR>(acc_e);
            if (l == 0) ge[e] = t;
        }
    }
}

template <int VEC>
static int launch_edge_grad(int lpr, dim3 grid, hipStream_t s, const int64_t *row, const int64_t *col, const float *G,
                            const float *scale, const float *w, const float *x, const float *ef, int ef_scalar, int op, float *ge,
                            float *gw, int64_t n_edges, int k) {
#define EG(L) hipLaunchKernelGGL((gspmm_edge_grad_kernel<VEC, L>), grid, dim3(256), 0, s, row, col, G, scale, w, x, ef, ef_scalar, op, ge, gw, n_edges, k)
    switch (lpr) {
        case 4: EG(4); break;
        case 8: EG(8); break;
        case 16: EG(16); break;
        case 32: EG(32); break;
        default: EG(64); break;
    }
#undef EG
    return launch_status();
}

extern "C" int cogdl_hip_gspmm_edge_grad(const int64_t *row, const int64_t *col, const float *grad, const float *scale,
                                         const float *weight, const float *x, const float *efeat, int efeat_is_scalar, int op,
                                         float *grad_efeat, float *grad_weight, int64_t n_edges, int64_t k, void *stream) {
    if (n_edges < 0 || k < 0) return COGDL_HIP_EINVAL;
    if (n_edges == 0 || k == 0 || (!grad_efeat && !grad_weight)) return COGDL_HIP_OK;
    if (!row || !col || !grad) return COGDL_HIP_EINVAL;
    if (op < COGDL_HIP_GSPMM_ADD || op > COGDL_HIP_GSPMM_MUL) return COGDL_HIP_EINVAL;
    if ((op == COGDL_HIP_GSPMM_MUL || grad_weight) && !x) return COGDL_HIP_EINVAL;
    if (grad_weight && !efeat) return COGDL_HIP_EINVAL;
    if (efeat_is_scalar && !efeat) return COGDL_HIP_EINVAL;
    if (k > 0x7fffffff) return COGDL_HIP_ERANGE;
    uintptr_t v = reinterpret_cast<uintptr_t>(grad);
    if (x) v |= reinterpret_cast<uintptr_t>(x);
    if (efeat && !efeat_is_scalar) v |= reinterpret_cast<uintptr_t>(efeat);
    if (grad_efeat && !efeat_is_scalar) v |= reinterpret_cast<uintptr_t>(grad_efeat);
    if (v % 4 != 0) return COGDL_HIP_EALIGN;
    const int vec = (k % 4 == 0 && v % 16 == 0) ? 4 : (k % 2 == 0 && v % 8 == 0) ? 2 : 1;
    const int64_t need = (k + vec - 1) / vec;
    int lpr = 4;
    while (lpr < kWave && lpr < need) lpr <<= 1;
    const int64_t groups_per_block = 256 / lpr;
    const unsigned blocks = (unsigned)std::min<int64_t>((n_edges + groups_per_block - 1) / groups_per_block, 1 << 16);
    hipStream_t s = (hipStream_t)stream;
    switch (vec) {
        case 4: return launch_edge_grad<4>(lpr, dim3(blocks), s, row, col, grad, scale, weight, x, efeat, efeat_is_scalar ? 1 : 0, op, grad_efeat, grad_weight, n_edges, (int)k);
        case 2: return launch_edge_grad<2>(lpr, dim3(blocks), s, row, col, grad, scale, weight, x, efeat, efeat_is_scalar ? 1 : 0, op, grad_efeat, grad_weight, n_edges, (int)k);
        default: return launch_edge_grad<1>(lpr, dim3(blocks), s, row, col, grad, scale, weight, x, efeat, efeat_is_scalar ? 1 : 0, op, grad_efeat, grad_weight, n_edges, (int)k);
    }
}
