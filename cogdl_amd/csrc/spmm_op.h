// spmm_op.h -- the csr_spmm / mhspmm functor over the row-reduce engine and its dispatch (shared by spmm.hip and
// spmm_epilogue.hip, which instantiate it without / with the fused normalisation epilogue).
#pragma once
#include "rowreduce.h"

namespace cogdl {

constexpr int kDefaultUnroll = 8;
// (Round 5, tried: batches of 16 gathers in flight for 16-bit features -- they move half the bytes per gathered row, so at the
//  fp32 depth a lane group keeps half the bytes in flight -- with the batch held as RAW vectors.  Slower on every graph and
//  width, 5-35 %: arxiv-sized R-MAT bf16 F = 64 125 -> 147 us, F = 128 190 -> 224, F = 256 207 -> 281; uniform F = 64 57 -> 66;
//  Reddit-shaped F = 128 2.93 -> 3.11 ms (profiles/r05_unroll16_ab.txt): the compiler unpacks early, the kernels grow from
//  78-82 to 84-119 VGPRs (one to two waves per SIMD fewer), and a chunk of LPR <= 32 edges has only one or two batches to
//  begin with.  The batch keeps the raw vectors (free for fp32, never worse for 16-bit); the depth stays 8.)

template <bool EXACT>
__device__ __forceinline__ float mul_add(float acc, float w, float v) {
    // The library is compiled with -ffp-contract=off: `acc + w * v` is a rounded multiply
    // followed by a rounded add, exactly what the reference's x86-64 build executes.
    if constexpr (EXACT) return acc + w * v;
    else return fmaf(w, v, acc);
}

// WMODE: 0 = unweighted (csr_spmm_no_edge_value), 1 = one weight per edge (val, dtype T),
//        2 = multi-head (mhspmm): x is [N, H, fdim], weight att[e, head] in fp32, head = column / fdim.
// Fused epilogue (EPI): out[i,:] = act( dst_scale[i] * sum_e val[e] * (src_scale[col[e]] * x[col[e],:]) + bias ) --
// the two broadcast multiplies CogDL's dispatcher wraps around the kernel (out_norm * x before, in_norm * x after:
// cogdl/utils/spmm_utils.py:99-109) and the layer's activation, applied where the operands already are.  Every
// product is a separately rounded fp32 multiply, in the reference's order: bit-identical to the unfused composition.
struct SpmmEpilogue {
    const float *src_scale;  // [n_src] or NULL
    const float *dst_scale;  // [m] or NULL
    const float *bias;       // [k] or NULL
    int act;                 // 0 = none, 1 = relu
};

template <typename T, int VEC_, int LPR_, int UNROLL_, int WMODE, bool EXACT, bool EPI = false, bool A24 = false>
struct SpmmOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_;
    static constexpr bool kReduce = true;
    static constexpr int kLds = 0;
    // (kMinWaves = 8, i.e. 64 VGPRs, was tried: arxiv-sized R-MAT F=64 132 -> 122 us but 44 bytes of scratch per lane make
    //  every other shape 3-10 % slower -- 71 VGPRs / 7 waves per SIMD stay)
    const T *val;      // WMODE 1
    const float *att;  // WMODE 2: [E, heads]
    const T *x;
    T *out;
    int k;       // feature width (heads * fdim for WMODE 2)
    int fdim;
    int acc_mode;  // != 0: out += A x
    const int32_t *eid;  // WMODE 2: attention row of edge e is att[eid[e]] (a transposed view of A; NULL: att[e])
    SpmmEpilogue epi;    // EPI only

    struct Ctx {
        int col0, heads, hd, cc;
        bool col_ok;
    };
    struct State { float acc[VEC]; };
    struct LaneVals {
        float w;
        int id;
    };
    using Raw = typename RawVec<sizeof(T) * VEC>::type;
    struct Batch {
        Raw v[UNROLL];  // as loaded; unpacked in apply()
        float w[UNROLL];
        float s[EPI ? UNROLL : 1];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int tile) const {
        Ctx c;
        c.col0 = (tile * LPR + l) * VEC;
        c.col_ok = c.col0 < k;
        const int cc = c.col_ok ? c.col0 : 0;  // lanes past the last column read column 0 and never store
        c.heads = (WMODE == 2) ? k / fdim : 1;
        c.hd = (WMODE == 2) ? cc / fdim : 0;
        c.cc = cc;
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &, int64_t, bool) const {}
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &c, State &s, int64_t row, bool ok) const {
        init_zero(s);
        if (acc_mode && ok && c.col_ok) load_vec<T, VEC>(out + row * (int64_t)k + c.col0, s.acc);
    }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &lv, int64_t e) const {
        if constexpr (WMODE == 1) lv.w = to_f32<T>(val[e]);
        if constexpr (WMODE == 2) lv.id = eid ? eid[e] : 0;
    }
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t e, const LaneVals &lv,
                                          int sub, int jj) const {
        if constexpr (WMODE == 1) b.w[u] = group_bcast<LPR>(lv.w, sub, jj);
        else if constexpr (WMODE == 2) {  // 4*H-byte run per edge
            const int64_t arow = eid ? (int64_t)group_bcast<LPR>(lv.id, sub, jj) : e;
            b.w[u] = att[arow * c.heads + c.hd];
        }
        else b.w[u] = 1.f;
        if constexpr (EPI) b.s[u] = epi.src_scale ? epi.src_scale[col] : 1.f;
        b.v[u] = load_raw<T, VEC>(gather_row<A24>(x, c.cc, col, (uint32_t)k * (uint32_t)sizeof(T)));
    }
    // Strictly in CSR order.  acc + 0*0 == acc exactly (acc is never -0), so masked slots are no-ops;
    // selecting v (not only w) to zero keeps inf/nan out.
    __device__ __forceinline__ void apply(const Ctx &, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int) const {
        float xv[VEC];
        if constexpr (sizeof(T) == 2 && !EPI && sizeof(Raw) % 4 == 0) {
            // 16-bit features: no bit-exact contract (products and sums are fp32, rounded once on store; parity is a
            // tolerance).  The masked slot is zeroed on its RAW words (VEC / 2 selects instead of VEC) and the update is a
            // fused multiply-add (one packed instruction per two columns instead of two): on hub-heavy graphs this kernel
            // is bound by VALU issue, not by bytes (round 6, profiles/r06_sq_reddit.txt).
            union { Raw raw; uint32_t w[sizeof(Raw) / 4]; } m;
            m.raw = b.v[u];
#pragma unroll
            for (int i = 0; i < (int)(sizeof(Raw) / 4); ++i) m.w[i] = valid ? m.w[i] : 0u;
            unpack_raw<T, VEC>(m.raw, xv);
            const float w = (WMODE != 0) ? (valid ? b.w[u] : 0.f) : 1.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if constexpr (WMODE != 0) s.acc[i] = __builtin_fmaf(w, xv[i], s.acc[i]);
                else s.acc[i] = s.acc[i] + xv[i];
            }
            return;
        }
        unpack_raw<T, VEC>(b.v[u], xv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float vv = valid ? xv[i] : 0.f;
            if constexpr (EPI) vv = valid ? b.s[u] * xv[i] : 0.f;  // (out_norm * x)[col, :], rounded like the reference's product
            if constexpr (WMODE != 0) s.acc[i] = mul_add<EXACT>(s.acc[i], valid ? b.w[u] : 0.f, vv);
            else s.acc[i] = s.acc[i] + vv;
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &, State &, int, int, int, int, int, float *, const LaneVals &) const {}
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t row, bool ok) const {
        if (!(ok && c.col_ok)) return;
        if constexpr (EPI) {
            float o[VEC];
            const float d = epi.dst_scale ? epi.dst_scale[row] : 1.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float v = epi.dst_scale ? d * s.acc[i] : s.acc[i];
                if (epi.bias) v = v + epi.bias[c.col0 + i];
                o[i] = (epi.act == 1) ? fmaxf(v, 0.f) : v;
            }
            store_vec<T, VEC>(out + row * (int64_t)k + c.col0, o);
        } else {
            store_vec<T, VEC>(out + row * (int64_t)k + c.col0, s.acc);
        }
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

template <typename T>
struct SpmmArgs {
    const int32_t *rowptr;
    const int32_t *colind;
    const T *val;
    const float *att;
    const T *x;
    T *out;
    int64_t m, nnz;
    int k, fdim, acc_mode;
    const int32_t *eid;
    SpmmEpilogue epi;
    const cogdl_hip_vrows *vr = nullptr;  // XCD-partitioned plan (rowreduce.h): rowptr / colind are then unused
    const int32_t *row_order = nullptr;   // ordinary launch: the row blocks' schedule (rowreduce.h: RowSched::order)
};

template <typename T, int VEC, int LPR, int UNROLL, int WMODE, bool EXACT, bool EPI = false>
static int launch_spmm(const SpmmArgs<T> &a, void *ws, size_t wsb, hipStream_t s) {
    const int64_t tiles = ((int64_t)a.k + (int64_t)LPR * VEC - 1) / ((int64_t)LPR * VEC);
    if constexpr (VEC == 8) {  // 16-byte lanes of 16-bit elements: XCD-partitioned plans only (spmm_geometry: wide16)
        if (!a.vr) return COGDL_HIP_EUNSUPPORTED;
        if constexpr (kWave % LPR == 0 && !EPI && WMODE != 2 && sizeof(T) == 2) {
            SpmmOp<T, VEC, LPR, UNROLL, WMODE, EXACT, EPI, true> op24{a.val, a.att, a.x, a.out, a.k, a.fdim, a.acc_mode, a.eid, a.epi};
            return launch_rowreduce_vrows(op24, a.vr, tiles, ws, wsb, s);
        } else return COGDL_HIP_EUNSUPPORTED;
    } else {
        if (a.vr) {  // XCD-partitioned plan: the same functor with 24-bit table offsets (common.h: gather_row)
            if constexpr (kWave % LPR == 0 && !EPI && WMODE != 2) {
                SpmmOp<T, VEC, LPR, UNROLL, WMODE, EXACT, EPI, true> op24{a.val, a.att, a.x, a.out, a.k, a.fdim, a.acc_mode, a.eid, a.epi};
                return launch_rowreduce_vrows(op24, a.vr, tiles, ws, wsb, s);
            } else return COGDL_HIP_EUNSUPPORTED;
        }
        SpmmOp<T, VEC, LPR, UNROLL, WMODE, EXACT, EPI> op{a.val, a.att, a.x, a.out, a.k, a.fdim, a.acc_mode, a.eid, a.epi};
        return launch_rowreduce(op, a.rowptr, a.colind, a.m, a.nnz, tiles, ws, wsb, s, a.row_order);
    }
}

// (VEC, LPR) choice: as many lanes per row as the row has VEC-wide columns (whole-wave rows are the
// fastest: scalar column broadcast, no inter-row divergence inside a wave), VEC as small as that allows.
template <typename T, int VEC, int WMODE, bool EPI, int UNROLL>
static int dispatch_lpr_u(const SpmmArgs<T> &a, int lpr, void *ws, size_t wsb, hipStream_t s) {
    if constexpr (WMODE != 2 && !EPI) {  // lane groups that do not divide the wave (spmm_geometry: narrow_groups)
        switch (lpr) {
            case 20: return launch_spmm<T, VEC, 20, UNROLL, WMODE, true, EPI>(a, ws, wsb, s);
            default: break;
        }
    }
    switch (lpr) {
        case 4: return launch_spmm<T, VEC, 4, UNROLL, WMODE, true, EPI>(a, ws, wsb, s);
        case 8: return launch_spmm<T, VEC, 8, UNROLL, WMODE, true, EPI>(a, ws, wsb, s);
        case 16: return launch_spmm<T, VEC, 16, UNROLL, WMODE, true, EPI>(a, ws, wsb, s);
        case 32: return launch_spmm<T, VEC, 32, UNROLL, WMODE, true, EPI>(a, ws, wsb, s);
        default: return launch_spmm<T, VEC, 64, UNROLL, WMODE, true, EPI>(a, ws, wsb, s);
    }
}

template <typename T, int VEC, int WMODE, bool EPI = false>
static int dispatch_lpr(const SpmmArgs<T> &a, int lpr, void *ws, size_t wsb, hipStream_t s) {
    return dispatch_lpr_u<T, VEC, WMODE, EPI, kDefaultUnroll>(a, lpr, ws, wsb, s);
}

static int pointer_alignment(const void *a, const void *b) {
    const uintptr_t v = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
    return (v % 16 == 0) ? 16 : (v % 8 == 0) ? 8 : (v % 4 == 0) ? 4 : 2;
}

template <typename T, int WMODE, bool EPI = false>
static int spmm_auto(const SpmmArgs<T> &a, void *ws, size_t wsb, hipStream_t s) {
    const RowGeometry g = spmm_geometry(a.k, (WMODE == 2) ? a.fdim : a.k, (int)sizeof(T), pointer_alignment(a.x, a.out),
                                        WMODE != 2 && !EPI && !a.vr, WMODE != 2 && !EPI && a.vr != nullptr);
    switch (g.vec) {
        case 8:
            if constexpr (sizeof(T) == 2 && WMODE != 2 && !EPI) return dispatch_lpr<T, 8, WMODE, EPI>(a, g.lpr, ws, wsb, s);
            else return COGDL_HIP_EUNSUPPORTED;
        case 4: return dispatch_lpr<T, 4, WMODE, EPI>(a, g.lpr, ws, wsb, s);
        case 2: return dispatch_lpr<T, 2, WMODE, EPI>(a, g.lpr, ws, wsb, s);
        default: return dispatch_lpr<T, 1, WMODE, EPI>(a, g.lpr, ws, wsb, s);
    }
}

}  // namespace cogdl
