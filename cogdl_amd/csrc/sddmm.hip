// sddmm.hip -- csr_sddmm for gfx950: out[e] = < d1[row(e),:], d2[colind[e],:] >.
// Replaces cogdl/operators/spmm/sddmm_kernel.cu:249-417 (16 edges per block, binary search
// of the row per 4 edges, warp=32 shuffles).  Here the row-group engine of rowreduce.h is
// reused: a group of LPR lanes owns one CSR row, so row(e) is known without a search and
// d1[row,:] stays in registers for the whole row; each neighbour row of d2 is one coalesced
// vector load; per-edge dot products are reduced with a wave64 butterfly inside the group
// and written back as one coalesced store per chunk of LPR edges.  Rows longer than the
// long-row threshold are cut into chunks processed by whole workgroups (per-edge outputs:
// nothing to merge).
// HBM-bound: algorithmic bytes per edge = 4 (colind) + F*s (d2 row) + 4 (out); d1 once per row.
#include "rowreduce.h"

namespace cogdl {

template <int VEC_, int LPR_, int UNROLL_, bool SINGLE_TILE>
struct SddmmOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = 1;
    static constexpr bool kReduce = false;
    static constexpr int kLds = 0;
    const float *d1, *d2;
    float *out;
    int k;

    struct Ctx {
        int l, c0;
        bool col_ok;
        const float *a_row;
        float a0[VEC];
    };
    struct State { float my_out; };
    struct LaneVals {};
    struct Batch {
        float b[UNROLL][VEC];
        int c[UNROLL];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int) const {
        Ctx c;
        c.l = l;
        c.col_ok = l * VEC < k;
        c.c0 = c.col_ok ? l * VEC : 0;
        c.a_row = d1;
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t row, bool ok) const {
        c.a_row = d1 + (ok ? row : 0) * (int64_t)k;
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.a0[i] = 0.f;
        if (SINGLE_TILE && ok && c.col_ok) load_vec<float, VEC>(c.a_row + c.c0, c.a0);
    }
    __device__ __forceinline__ void init_zero(State &s) const { s.my_out = 0.f; }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { s.my_out = 0.f; }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &, int64_t) const {}
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &, int,
                                          int) const {
        b.c[u] = col;
        if constexpr (SINGLE_TILE) load_vec<float, VEC>(d2 + (int64_t)col * k + c.c0, b.b[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool, int64_t,
                                          int jpos) const {
        float p = 0.f;
        if constexpr (SINGLE_TILE) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) p = fmaf(c.a0[i], c.col_ok ? b.b[u][i] : 0.f, p);
        } else {  // wide rows: walk the column tiles (k > LPR*VEC)
            for (int c0 = c.l * VEC; c0 < k; c0 += LPR * VEC) {
                float a[VEC], v[VEC];
                load_vec<float, VEC>(c.a_row + c0, a);
                load_vec<float, VEC>(d2 + (int64_t)b.c[u] * k + c0, v);
#pragma unroll
                for (int i = 0; i < VEC; ++i) p = fmaf(a[i], v[i], p);
            }
        }
        const float r = group_sum<LPR>(p);
        if (c.l == jpos) s.my_out = r;
    }
    __device__ __forceinline__ void chunk_begin(Ctx &, State &, int, int, int, int, int, float *, const LaneVals &) const {}
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &c, State &s, int base, int cnt) const {
        if (c.l < cnt) out[base + c.l] = s.my_out;
    }
    __device__ __forceinline__ void row_end(const Ctx &, const State &, int64_t, bool) const {}
    __device__ __forceinline__ void pack(const State &, float (&)[kRec]) const {}
    __device__ __forceinline__ void unpack(State &, const float (&)[kRec]) const {}
    __device__ __forceinline__ void merge(const Ctx &, State &, const State &) const {}
};

struct SddmmArgs {
    const int32_t *rowptr, *colind;
    const float *d1, *d2;
    float *out;
    int64_t m, k, nnz;
};

template <int VEC, int LPR>
static int launch_sddmm(const SddmmArgs &a, void *ws, size_t wsb, hipStream_t s) {
    constexpr int UNROLL = 4;
    if (a.k <= (int64_t)LPR * VEC) {
        SddmmOp<VEC, LPR, UNROLL, true> op{a.d1, a.d2, a.out, (int)a.k};
        return launch_rowreduce(op, a.rowptr, a.colind, a.m, a.nnz, 1, ws, wsb, s);
    }
    SddmmOp<VEC, LPR, UNROLL, false> op{a.d1, a.d2, a.out, (int)a.k};
    return launch_rowreduce(op, a.rowptr, a.colind, a.m, a.nnz, 1, ws, wsb, s);
}

template <int VEC>
static int dispatch_sddmm(const SddmmArgs &a, void *ws, size_t wsb, hipStream_t s) {
    const int64_t need = (a.k + VEC - 1) / VEC;
    if (need <= 8) return launch_sddmm<VEC, 8>(a, ws, wsb, s);
    if (need <= 16) return launch_sddmm<VEC, 16>(a, ws, wsb, s);
    if (need <= 32) return launch_sddmm<VEC, 32>(a, ws, wsb, s);
    return launch_sddmm<VEC, 64>(a, ws, wsb, s);
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_csr_sddmm(const int32_t *rowptr, const int32_t *colind, const float *d1,
                                   const float *d2, float *out, int64_t m, int64_t k, int64_t nnz, void *stream) {
    if (m < 0 || k < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (m == 0) return COGDL_HIP_OK;
    if (!rowptr || !d1 || !d2 || !out) return COGDL_HIP_EINVAL;
    if (k > 0x7fffffff || nnz > COGDL_HIP_SEGMENT_MAX_EDGES) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (k == 0) return COGDL_HIP_EINVAL;
    SddmmArgs a{rowptr, colind, d1, d2, out, m, k, nnz};
    if (k % 4 == 0 && aligned_to(d1, 16) && aligned_to(d2, 16)) return dispatch_sddmm<4>(a, nullptr, 0, s);
    if (k % 2 == 0 && aligned_to(d1, 8) && aligned_to(d2, 8)) return dispatch_sddmm<2>(a, nullptr, 0, s);
    return dispatch_sddmm<1>(a, nullptr, 0, s);
}

extern "C" int cogdl_hip_csr_sddmm_i64(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                       const float *d1, const float *d2, float *out, int64_t k, void *stream) {
    int rc = segments_valid(seg);
    if (rc != COGDL_HIP_OK) return rc;
    for (int s = 0; s < seg->n; ++s) {
        const int64_t r0 = seg->row[s], e0 = seg->edge[s];
        if (seg->edge[s + 1] == e0) continue;
        rc = cogdl_hip_csr_sddmm(rowptr32 + r0 + s, colind + e0, d1 + r0 * k, d2, out + e0, seg->row[s + 1] - r0, k,
                                 seg->edge[s + 1] - e0, stream);
        if (rc != COGDL_HIP_OK) return rc;
    }
    return COGDL_HIP_OK;
}
