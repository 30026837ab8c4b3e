// gat.hip -- fused GAT attention + aggregation for gfx950 (fused_gat_func of cogdl/operators/fused_gat.py:14-41;
// the reference binds dgNN's fused_gatconv, whose source is absent from the tree, so the semantics are those
// of the unfused layer path cogdl/layers/gat_layer.py:73-77):
//     s[e,h] = LeakyReLU(attn_row[row(e),h] + attn_col[col[e],h])
//     a[e,h] = softmax over the edges of row(e)
//     out[v,h,:] = sum_e a[e,h] * feat[col[e],h,:]
// Forward: ONE pass over the edges with an online softmax (running max / rescaled sum, flash-attention style),
// so the [E,H] score/attention tensors are never materialised: algorithmic bytes per edge =
// 4 (colind) + 4H (attn_col row) + H*F*s (feature row); per node 2*4H + H*F*s.  Row-group decomposition and
// long-row (chunk-parallel) path of rowreduce.h: a group of LPR lanes owns one destination row, each lane VEC
// columns inside one head; the state (max, sum, acc[VEC]) of two edge ranges merges like two flash-attention
// blocks, so hub rows are split over whole workgroups.
// Backward (fp32): with D[v,h] = <g[v,h,:], out[v,h,:]>,
//     dL/ds[e,h]      = a[e,h] * (<g[row,h,:], feat[col,h,:]> - D[row,h]) * LeakyReLU'(.)
//     grad_attn_row[v,h] = sum over the row's edges            (row pass over the CSR, gathers feat)
//     grad_attn_col[u,h] = sum over the column's edges         (column pass over the CSC, gathers g)
//     grad_feat[u,h,:]   = sum over the column's edges a * g[row,h,:]        (same column pass)
// a[e,h] is recomputed from the saved row max / row sum (edge_max, edge_sum), exactly what the reference's
// FusedGATFunction keeps in ctx (operators/fused_gat.py:20).  No atomics anywhere: deterministic.
#include "rowreduce.h"

namespace cogdl {

// Per-edge exponentials (argument = score - row max <= 0): the hardware exponential, as in edge_softmax.hip -- relative
// error |x| * 2^-24 (< 2e-6 wherever the weight is not negligible); the rare state merges keep libm's expf.
__device__ __forceinline__ float gat_exp(float x) { return __expf(x); }


__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }
static bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// Sum over the `lph` lanes (power of two, aligned) that hold one head.
template <int LPR>
__device__ __forceinline__ float head_sum(float v, int lph) {
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1)
        if (s < lph) v += __shfl_xor(v, s, kWave);
    return v;
}

// ------------------------------------------------------------------------------------------ forward
template <typename T, int VEC_, int LPR_, int UNROLL_>
struct GatFwdOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_ + 2;
    static constexpr bool kReduce = true;
    static constexpr int kLds = 0;
    const float *attn_row, *attn_col;
    const T *feat;
    T *out;
    float *edge_max, *edge_sum;
    float slope;
    int heads, fdim;
    struct Ctx {
        int col0, cc, hd;
        bool col_ok;
        float ar;
    };
    struct State {
        float acc[VEC];
        float mx, lsum;
    };
    struct LaneVals {};
    struct Batch {
        float v[UNROLL][VEC];
        float ac[UNROLL];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int tile) const {
        Ctx c;
        c.col0 = (tile * LPR + l) * VEC;
        c.col_ok = c.col0 < heads * fdim;
        c.cc = c.col_ok ? c.col0 : 0;
        c.hd = c.cc / fdim;
        c.ar = 0.f;
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t row, bool ok) const {
        c.ar = ok ? attn_row[row * heads + c.hd] : 0.f;
    }
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = 0.f;
        s.mx = -INFINITY;
        s.lsum = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { init_zero(s); }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &, int64_t) const {}
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &, int,
                                          int) const {
        b.ac[u] = attn_col[(int64_t)col * heads + c.hd];
        load_vec<T, VEC>(feat + (int64_t)col * (heads * fdim) + c.cc, b.v[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int) const {
        if (valid) {
            const float sc = leaky(c.ar + b.ac[u], slope);
            const float mn = fmaxf(s.mx, sc);
            const float scale = (s.lsum == 0.f) ? 0.f : expf(s.mx - mn);
            const float p = gat_exp(sc - mn);
            s.lsum = s.lsum * scale + p;
#pragma unroll
            for (int i = 0; i < VEC; ++i) s.acc[i] = fmaf(p, b.v[u][i], s.acc[i] * scale);
            s.mx = mn;
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &, State &, int, int, int, int, int, float *) const {}
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t row, bool ok) const {
        if (ok && c.col_ok) {
            const float inv = (s.lsum > 0.f) ? 1.f / s.lsum : 0.f;  // empty row -> zeros
            float r[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) r[i] = s.acc[i] * inv;
            store_vec<T, VEC>(out + row * (int64_t)(heads * fdim) + c.col0, r);
            if (c.col0 % fdim == 0) {
                edge_max[row * heads + c.hd] = s.mx;
                edge_sum[row * heads + c.hd] = s.lsum;
            }
        }
    }
    __device__ __forceinline__ void pack(const State &s, float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) rec[i] = s.acc[i];
        rec[VEC] = s.mx;
        rec[VEC + 1] = s.lsum;
    }
    __device__ __forceinline__ void unpack(State &s, const float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = rec[i];
        s.mx = rec[VEC];
        s.lsum = rec[VEC + 1];
    }
    // Two flash-attention blocks: rescale both to the common maximum (an empty block has lsum == 0, mx == -inf).
    __device__ __forceinline__ void merge(const Ctx &, State &a, const State &b) const {
        const float mn = fmaxf(a.mx, b.mx);
        const float sa = (a.lsum == 0.f) ? 0.f : expf(a.mx - mn);
        const float sb = (b.lsum == 0.f) ? 0.f : expf(b.mx - mn);
#pragma unroll
        for (int i = 0; i < VEC; ++i) a.acc[i] = a.acc[i] * sa + b.acc[i] * sb;
        a.lsum = a.lsum * sa + b.lsum * sb;
        a.mx = mn;
    }
};

// Chunk-wise softmax variant (H a power of two <= min(LPR, 16)): the flash-attention block structure.  For every
// chunk of LPR edges the group first computes the chunk's softmax weights ONCE per (edge, head) -- lane l takes head
// l % H of the edges l / H, l / H + LPR/H, ... (coalesced attn_col reads), the per-head chunk maximum and sum are
// wave64 butterflies over the lanes of equal head, the weights go to LDS -- rescales its accumulator once, and only
// then gathers the feature rows, folding each with the weight read back from LDS.  Versus the edge-wise online
// softmax above: one exp per (edge, head) instead of two per (edge, lane), no dependent max/rescale chain per edge.
template <typename T, int VEC_, int LPR_, int UNROLL_>
struct GatFwdChunkOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_ + 2;
    static constexpr bool kReduce = true;
    static constexpr int kMaxHeads = 16;
    static constexpr int kLds = kMaxHeads;  // LPR edges x H heads weights per group
    const float *attn_row, *attn_col;
    const T *feat;
    T *out;
    float *edge_max, *edge_sum;
    float slope;
    int heads, fdim;

    struct Ctx {
        int col0, cc, hd, hs, lane0, tile;  // hd: head of this lane's columns; hs = l % H: head of its softmax duty
        bool col_ok;
        float ar;                            // attn_row[row, hs]
        const float *w;                      // the chunk's weights in LDS: w[edge_in_chunk * H + head]
    };
    struct State {
        float acc[VEC];
        float mx, lsum;  // running max / sum of head hs (identical in all lanes of equal hs)
    };
    struct LaneVals {};
    struct Batch { float v[UNROLL][VEC]; };

    __device__ __forceinline__ Ctx make_ctx(int l, int tile) const {
        Ctx c;
        c.col0 = (tile * LPR + l) * VEC;
        c.col_ok = c.col0 < heads * fdim;
        c.cc = c.col_ok ? c.col0 : 0;
        c.hd = c.cc / fdim;
        c.hs = l % heads;
        c.lane0 = (int)(threadIdx.x & (kWave - 1)) - l;  // first lane of this group inside the wave
        c.tile = tile;
        c.ar = 0.f;
        c.w = nullptr;
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t row, bool ok) const {
        c.ar = ok ? attn_row[row * heads + c.hs] : 0.f;
    }
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = 0.f;
        s.mx = -INFINITY;
        s.lsum = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { init_zero(s); }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &, int64_t) const {}
    // reductions over the lanes of one group that share a softmax head: lane strides H, 2H, ... LPR/2
    __device__ __forceinline__ float head_max(float v) const {
#pragma unroll
        for (int sft = LPR / 2; sft > 0; sft >>= 1)
            if (sft >= heads) v = fmaxf(v, __shfl_xor(v, sft, kWave));
        return v;
    }
    __device__ __forceinline__ float head_add(float v) const {
#pragma unroll
        for (int sft = LPR / 2; sft > 0; sft >>= 1)
            if (sft >= heads) v += __shfl_xor(v, sft, kWave);
        return v;
    }
    __device__ __forceinline__ void chunk_begin(Ctx &c, State &s, int, int cnt, int my_c, int, int l,
                                                float *lds) const {
        const int epr = LPR / heads;  // edges per round
        const int eo = l / heads;
        const int rounds = (cnt + epr - 1) / epr;
        float lmax = -INFINITY;
        for (int r = 0; r < rounds; ++r) {
            const int je = r * epr + eo;
            const int col = __shfl(my_c, c.lane0 + min(je, cnt - 1), kWave);
            const float sc = (je < cnt) ? leaky(c.ar + attn_col[(int64_t)col * heads + c.hs], slope) : -INFINITY;
            lds[r * LPR + l] = sc;  // == lds[je * H + hs]
            lmax = fmaxf(lmax, sc);
        }
        lmax = head_max(lmax);
        const float mn = fmaxf(s.mx, lmax);
        const float scale = (s.lsum == 0.f) ? 0.f : expf(s.mx - mn);
        float psum = 0.f;
        for (int r = 0; r < rounds; ++r) {
            const float p = gat_exp(lds[r * LPR + l] - mn);  // exp(-inf) == 0 for the masked tail
            lds[r * LPR + l] = p;
            psum += p;
        }
        s.lsum = s.lsum * scale + head_add(psum);
        s.mx = mn;
        const float scale_col = __shfl(scale, c.lane0 + c.hd, kWave);  // lane hd of the group serves head hd
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] *= scale_col;
        c.w = lds;
    }
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &, int,
                                          int) const {
        load_vec<T, VEC>(feat + (int64_t)col * (heads * fdim) + c.cc, b.v[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int jpos) const {
        if (valid) {
            const float p = c.w[jpos * heads + c.hd];
#pragma unroll
            for (int i = 0; i < VEC; ++i) s.acc[i] = fmaf(p, b.v[u][i], s.acc[i]);
        }
    }
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t row, bool ok) const {
        const float lsum_col = __shfl(s.lsum, c.lane0 + c.hd, kWave);  // (all lanes of the group take part)
        if (!ok) return;
        if (c.col_ok) {
            const float inv = (lsum_col > 0.f) ? 1.f / lsum_col : 0.f;  // empty row -> zeros
            float r[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) r[i] = s.acc[i] * inv;
            store_vec<T, VEC>(out + row * (int64_t)(heads * fdim) + c.col0, r);
        }
        const int l = (int)(threadIdx.x & (kWave - 1)) - c.lane0;
        if (c.tile == 0 && l < heads) {
            edge_max[row * heads + l] = s.mx;
            edge_sum[row * heads + l] = s.lsum;
        }
    }
    __device__ __forceinline__ void pack(const State &s, float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) rec[i] = s.acc[i];
        rec[VEC] = s.mx;
        rec[VEC + 1] = s.lsum;
    }
    __device__ __forceinline__ void unpack(State &s, const float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = rec[i];
        s.mx = rec[VEC];
        s.lsum = rec[VEC + 1];
    }
    // b's edges follow a's.  The (max, sum) pair belongs to head hs, the accumulator to head hd: its two scale
    // factors are fetched from lane hd of the group (all lanes of the group call merge together).
    __device__ __forceinline__ void merge(const Ctx &c, State &a, const State &b) const {
        const float mn = fmaxf(a.mx, b.mx);
        const float sa = (a.lsum == 0.f) ? 0.f : expf(a.mx - mn);
        const float sb = (b.lsum == 0.f) ? 0.f : expf(b.mx - mn);
        const float sa_c = __shfl(sa, c.lane0 + c.hd, kWave), sb_c = __shfl(sb, c.lane0 + c.hd, kWave);
#pragma unroll
        for (int i = 0; i < VEC; ++i) a.acc[i] = a.acc[i] * sa_c + b.acc[i] * sb_c;
        a.lsum = a.lsum * sa + b.lsum * sb;
        a.mx = mn;
    }
};

// ------------------------------------------------------------------------------------------ backward
// Row pass: D[v,h] and grad_attn_row[v,h].  The whole [H*F] row must fit one group.
template <typename T, int VEC_, int LPR_, int UNROLL_>
struct GatBwdRowOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_ + 1;
    static constexpr bool kReduce = true;
    static constexpr int kLds = 0;
    const float *attn_row, *attn_col;
    const T *feat;  // feat / out / grad_out in the layer's dtype (f32, f16, bf16): read natively, fp32 arithmetic
    const float *edge_max, *edge_sum;
    const T *out, *grad_out;
    float *dvec, *grad_attn_row;
    float slope;
    int heads, fdim, lph;

    struct Ctx {
        int cc, hd;
        bool col_ok, head_lane;
        float g[VEC];
        float d, ar, mx, inv;
    };
    // grad_attn_row[v,h] = sum_e c_e (<g, feat[col_e]> - D) with c_e = a_e * LeakyReLU'(.)
    //                    = < g, sum_e c_e feat[col_e] >  -  D * sum_e c_e :
    // the per-edge dot product (a cross-lane reduction per edge) becomes ONE reduction per row over a weighted
    // accumulation s[] of the gathered rows, exactly the shape of an SpMM.
    struct State {
        float s[VEC];
        float csum;
    };
    struct LaneVals {};
    struct Batch {
        float v[UNROLL][VEC];
        float ac[UNROLL];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int) const {
        Ctx c;
        c.col_ok = l * VEC < heads * fdim;
        c.cc = c.col_ok ? l * VEC : 0;
        c.hd = c.cc / fdim;
        c.head_lane = c.col_ok && (c.cc % fdim == 0);
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t row, bool ok) const {
        const int k = heads * fdim;
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.g[i] = 0.f;
        float d = 0.f;
        c.ar = c.mx = c.inv = 0.f;
        if (ok && c.col_ok) {
            load_vec<T, VEC>(grad_out + row * (int64_t)k + c.cc, c.g);
            float o[VEC];
            load_vec<T, VEC>(out + row * (int64_t)k + c.cc, o);
#pragma unroll
            for (int i = 0; i < VEC; ++i) d = fmaf(c.g[i], o[i], d);
            c.ar = attn_row[row * heads + c.hd];
            c.mx = edge_max[row * heads + c.hd];
            const float ls = edge_sum[row * heads + c.hd];
            c.inv = ls > 0.f ? 1.f / ls : 0.f;
        }
        c.d = head_sum<LPR>(d, lph);
    }
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.s[i] = 0.f;
        s.csum = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { init_zero(s); }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &, int64_t) const {}
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &, int,
                                          int) const {
        b.ac[u] = attn_col[(int64_t)col * heads + c.hd];
        load_vec<T, VEC>(feat + (int64_t)col * (heads * fdim) + c.cc, b.v[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int) const {
        if (valid) {
            const float pre = c.ar + b.ac[u];
            const float ce = gat_exp(leaky(pre, slope) - c.mx) * c.inv * (pre > 0.f ? 1.f : slope);
#pragma unroll
            for (int i = 0; i < VEC; ++i) s.s[i] = fmaf(ce, b.v[u][i], s.s[i]);
            s.csum += ce;
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &, State &, int, int, int, int, int, float *) const {}
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t row, bool ok) const {
        float dot = 0.f;  // (all lanes of the group take part in the reduction)
#pragma unroll
        for (int i = 0; i < VEC; ++i) dot = fmaf(c.g[i], c.col_ok ? s.s[i] : 0.f, dot);
        dot = head_sum<LPR>(dot, lph);
        if (ok && c.head_lane) {
            dvec[row * heads + c.hd] = c.d;
            grad_attn_row[row * heads + c.hd] = dot - c.d * s.csum;
        }
    }
    __device__ __forceinline__ void pack(const State &s, float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) rec[i] = s.s[i];
        rec[VEC] = s.csum;
    }
    __device__ __forceinline__ void unpack(State &s, const float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.s[i] = rec[i];
        s.csum = rec[VEC];
    }
    __device__ __forceinline__ void merge(const Ctx &, State &a, const State &b) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) a.s[i] += b.s[i];
        a.csum += b.csum;
    }
};

// Column pass over the CSC (colptr, rowind): grad_feat[u,h,:] and grad_attn_col[u,h].
template <typename T, int VEC_, int LPR_, int UNROLL_>
struct GatBwdColOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = 2 * VEC_ + 1;
    static constexpr bool kReduce = true;
    static constexpr int kLds = 0;
    const float *attn_row, *attn_col;
    const T *feat;
    const float *edge_max, *edge_sum, *dvec;
    const T *grad_out;
    T *grad_feat;  // rounded once on store (fp32 accumulation)
    float *grad_attn_col;
    float slope;
    int heads, fdim, lph;

    struct Ctx {
        int cc, hd;
        bool col_ok, head_lane;
        float f[VEC];
        float ac;
    };
    // grad_attn_col[u,h] = sum_e c_e (<feat[u], g[row_e]> - D[row_e]) = <feat[u], sum_e c_e g[row_e]> - sum_e c_e D[row_e]:
    // a second weighted accumulation t[] next to grad_feat's (weights c_e = a_e * LeakyReLU' instead of a_e) and one
    // cross-lane reduction per column instead of one per edge.
    struct State {
        float acc[VEC];
        float t[VEC];
        float cd;
    };
    struct LaneVals {};
    struct Batch {
        float g[UNROLL][VEC];
        float ar[UNROLL], mx[UNROLL], ls[UNROLL], dd[UNROLL];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int) const {
        Ctx c;
        c.col_ok = l * VEC < heads * fdim;
        c.cc = c.col_ok ? l * VEC : 0;
        c.hd = c.cc / fdim;
        c.head_lane = c.col_ok && (c.cc % fdim == 0);
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t u_node, bool ok) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.f[i] = 0.f;
        c.ac = 0.f;
        if (ok && c.col_ok) {
            load_vec<T, VEC>(feat + u_node * (int64_t)(heads * fdim) + c.cc, c.f);
            c.ac = attn_col[u_node * heads + c.hd];
        }
    }
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = s.t[i] = 0.f;
        s.cd = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { init_zero(s); }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &, int64_t) const {}
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int r, int64_t, const LaneVals &, int,
                                          int) const {
        const int64_t rh = (int64_t)r * heads + c.hd;
        b.ar[u] = attn_row[rh];
        b.mx[u] = edge_max[rh];
        b.ls[u] = edge_sum[rh];
        b.dd[u] = dvec[rh];
        load_vec<T, VEC>(grad_out + (int64_t)r * (heads * fdim) + c.cc, b.g[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int) const {
        if (valid) {
            const float pre = b.ar[u] + c.ac;
            const float a = gat_exp(leaky(pre, slope) - b.mx[u]) / b.ls[u];
            const float ce = a * (pre > 0.f ? 1.f : slope);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                s.acc[i] = fmaf(a, b.g[u][i], s.acc[i]);
                s.t[i] = fmaf(ce, b.g[u][i], s.t[i]);
            }
            s.cd = fmaf(ce, b.dd[u], s.cd);
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &, State &, int, int, int, int, int, float *) const {}
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t u_node, bool ok) const {
        float dot = 0.f;  // (all lanes of the group take part in the reduction)
#pragma unroll
        for (int i = 0; i < VEC; ++i) dot = fmaf(c.f[i], c.col_ok ? s.t[i] : 0.f, dot);
        dot = head_sum<LPR>(dot, lph);
        if (ok && c.col_ok) {
            store_vec<T, VEC>(grad_feat + u_node * (int64_t)(heads * fdim) + c.cc, s.acc);
            if (c.head_lane) grad_attn_col[u_node * heads + c.hd] = dot - s.cd;
        }
    }
    __device__ __forceinline__ void pack(const State &s, float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            rec[i] = s.acc[i];
            rec[VEC + i] = s.t[i];
        }
        rec[2 * VEC] = s.cd;
    }
    __device__ __forceinline__ void unpack(State &s, const float (&rec)[kRec]) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s.acc[i] = rec[i];
            s.t[i] = rec[VEC + i];
        }
        s.cd = rec[2 * VEC];
    }
    __device__ __forceinline__ void merge(const Ctx &, State &a, const State &b) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            a.acc[i] += b.acc[i];
            a.t[i] += b.t[i];
        }
        a.cd += b.cd;
    }
};

// Forward geometry: the WIDEST legal vector (16-byte lanes) with every lane's columns inside one head.  Unlike
// csr_spmm (which narrows the vector until a row fills 64 lanes) every lane here repeats the per-edge softmax
// arithmetic of its head (leaky-relu, two exps, the rescale), so fewer, fatter lanes win: measured on MI355X,
// reddit-shaped graph, H=8 x F=8: f32 vec 1 -> 4: 8.7 -> 4.7 ms, bf16 vec 2 -> 8: 6.6 -> 3.2 ms.
static RowGeometry gat_fwd_geometry(int64_t h, int64_t f, int elem_bytes, int align) {
    const int maxv = 16 / elem_bytes;
    int vec = 1;
    const int forced = g_tuning[kTuneGatVec];  // experiments: cap the vector width (0 = automatic)
    for (int w = 1; w <= maxv; w <<= 1) {
        if (f % w != 0 || align % (w * elem_bytes) != 0) break;
        vec = w;
        if (forced && w >= forced) break;
    }
    const int64_t need = (h * f + vec - 1) / vec;
    int lpr = 8;
    while (lpr < kWave && lpr < need) lpr <<= 1;
    RowGeometry g;
    g.vec = vec;
    g.lpr = lpr;
    g.tiles = (h * f + (int64_t)lpr * vec - 1) / ((int64_t)lpr * vec);
    return g;
}

struct FwdArgs {
    const int32_t *rowptr, *colind;
    const float *ar, *ac;
    const void *feat;
    float slope;
    void *out;
    float *emax, *esum;
    int64_t v, h, f, nnz;
};

template <typename T, int VEC, int LPR>
static int launch_fwd(const FwdArgs &a, int64_t tiles, void *ws, size_t wsb, hipStream_t s) {
    // Chunk-wise softmax pays once a round of the weight computation covers >= 8 edges (LPR/H) of a >= 16-edge chunk:
    // measured on MI355X (reddit-shaped graph, f32) H=1,F=41: 7.3 -> 3.6 ms, H=1,F=64: 4.1 -> 2.8 ms, H=8,F=8
    // (LPR 16, 2 edges per round): 4.2 -> 4.3 ms, bf16 H=8,F=8 (LPR 8): 2.9 -> 4.6 ms.  tuning key 5: 1 = never, 2 = always.
    const bool can_chunk = pow2(a.h) && a.h <= LPR && a.h <= GatFwdChunkOp<T, VEC, LPR, 8>::kMaxHeads;
    const bool want_chunk = g_tuning[kTuneGatOnline] == 2 || (g_tuning[kTuneGatOnline] == 0 && LPR >= 16 && a.h * 8 <= LPR);
    if (can_chunk && want_chunk) {
        GatFwdChunkOp<T, VEC, LPR, 8> op{a.ar, a.ac, (const T *)a.feat, (T *)a.out, a.emax, a.esum, a.slope, (int)a.h,
                                         (int)a.f};
        return launch_rowreduce(op, a.rowptr, a.colind, a.v, a.nnz, tiles, ws, wsb, s);
    }
    GatFwdOp<T, VEC, LPR, 8> op{a.ar, a.ac, (const T *)a.feat, (T *)a.out, a.emax, a.esum, a.slope, (int)a.h, (int)a.f};
    return launch_rowreduce(op, a.rowptr, a.colind, a.v, a.nnz, tiles, ws, wsb, s);
}

template <typename T, int VEC>
static int dispatch_fwd(const FwdArgs &a, const RowGeometry &g, void *ws, size_t wsb, hipStream_t s) {
    switch (g.lpr) {
        case 8: return launch_fwd<T, VEC, 8>(a, g.tiles, ws, wsb, s);
        case 16: return launch_fwd<T, VEC, 16>(a, g.tiles, ws, wsb, s);
        case 32: return launch_fwd<T, VEC, 32>(a, g.tiles, ws, wsb, s);
        default: return launch_fwd<T, VEC, 64>(a, g.tiles, ws, wsb, s);
    }
}

template <typename T>
static int gat_fwd_typed(const FwdArgs &a, void *ws, size_t wsb, hipStream_t s) {
    constexpr int MAXV = 16 / sizeof(T);
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a.feat) | reinterpret_cast<uintptr_t>(a.out);
    const int align = (bits % 16 == 0) ? 16 : (bits % 8 == 0) ? 8 : (bits % 4 == 0) ? 4 : 2;
    if (align < (int)sizeof(T)) return COGDL_HIP_EALIGN;
    const RowGeometry g = gat_fwd_geometry(a.h, a.f, (int)sizeof(T), align);
    switch (g.vec) {
        case 8:
            if constexpr (MAXV >= 8) return dispatch_fwd<T, 8>(a, g, ws, wsb, s);
        case 4: return dispatch_fwd<T, 4>(a, g, ws, wsb, s);
        case 2: return dispatch_fwd<T, 2>(a, g, ws, wsb, s);
        default: return dispatch_fwd<T, 1>(a, g, ws, wsb, s);
    }
}

struct BwdArgs {
    const int32_t *rowptr, *colind, *colptr, *rowind;
    const float *ar, *ac;
    const void *feat;
    const float *emax, *esum;
    const void *out, *gout;
    float slope;
    void *gfeat;
    float *gar, *gac, *dvec;
    int64_t v, n_src, h, f, nnz;
    void *ws_row, *ws_col;  // long-row scratch of the two passes (either both or none)
    size_t wsb_row, wsb_col;
};

// Backward geometry: the whole [H*F] row in ONE group (the per-head dot products are reduced with shuffles):
// vec in {1,2,4} (and 8 for 2-byte elements) with F % vec == 0, H*F/vec <= 64 lanes and F/vec a power of two
// (unless H == 1); a lane's vector is at most 16 bytes and must be aligned to its size.
static int gat_bwd_vec(int64_t h, int64_t f, int align, int elem_bytes) {
    auto ok = [&](int vec) {
        return vec * elem_bytes <= 16 && align >= vec * elem_bytes && f % vec == 0 && h * f / vec <= kWave &&
               (h == 1 || pow2(f / vec));
    };
    // fat lanes first: the per-edge attention maths is repeated per lane
    for (int vec = 8; vec > 1; vec >>= 1)
        if (ok(vec)) return vec;
    return (h * f <= kWave && (h == 1 || pow2(f))) ? 1 : 0;
}
static int gat_bwd_lpr(int64_t h, int64_t f, int vec) {
    const int64_t need = (h * f + vec - 1) / vec;
    int lpr = 8;
    while (lpr < kWave && lpr < need) lpr <<= 1;
    return lpr;
}

template <typename T, int VEC, int LPR>
static int launch_bwd(const BwdArgs &b, hipStream_t s) {
    const int lph = (b.h == 1) ? LPR : (int)(b.f / VEC);  // one head: reduce over the whole (zero-padded) group
    GatBwdRowOp<T, VEC, LPR, 4> row_op{b.ar, b.ac, (const T *)b.feat, b.emax, b.esum, (const T *)b.out,
                                        (const T *)b.gout, b.dvec, b.gar, b.slope, (int)b.h, (int)b.f, lph};
    int rc = launch_rowreduce(row_op, b.rowptr, b.colind, b.v, b.nnz, 1, b.ws_row, b.wsb_row, s);
    if (rc != COGDL_HIP_OK) return rc;
    GatBwdColOp<T, VEC, LPR, 4> col_op{b.ar, b.ac, (const T *)b.feat, b.emax, b.esum, b.dvec, (const T *)b.gout,
                                        (T *)b.gfeat, b.gac, b.slope, (int)b.h, (int)b.f, lph};
    return launch_rowreduce(col_op, b.colptr, b.rowind, b.n_src, b.nnz, 1, b.ws_col, b.wsb_col, s);
}

template <typename T, int VEC>
static int dispatch_bwd(const BwdArgs &b, hipStream_t s) {
    switch (gat_bwd_lpr(b.h, b.f, VEC)) {
        case 8: return launch_bwd<T, VEC, 8>(b, s);
        case 16: return launch_bwd<T, VEC, 16>(b, s);
        case 32: return launch_bwd<T, VEC, 32>(b, s);
        default: return launch_bwd<T, VEC, 64>(b, s);
    }
}

template <typename T>
static int gat_bwd_typed(const BwdArgs &b, int vec, hipStream_t s) {
    switch (vec) {
        case 8:
            if constexpr (sizeof(T) == 2) return dispatch_bwd<T, 8>(b, s);
            return COGDL_HIP_EINVAL;
        case 4: return dispatch_bwd<T, 4>(b, s);
        case 2: return dispatch_bwd<T, 2>(b, s);
        default: return dispatch_bwd<T, 1>(b, s);
    }
}

static size_t dvec_bytes(int64_t v, int64_t h) { return ((size_t)(v > 0 ? v : 0) * (size_t)(h > 0 ? h : 0) * sizeof(float) + 255) / 256 * 256; }

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_gat_fwd_workspace_bytes(int64_t nnz, int64_t h, int64_t f, int dtype) {
    if (nnz <= 0 || h <= 0 || f <= 0) return 0;
    const RowGeometry g = gat_fwd_geometry(h, f, dtype == COGDL_HIP_F32 ? 4 : 2, 16);
    return rowreduce_workspace_bytes(nnz, g.tiles * (g.vec + 2) * g.lpr);
}

extern "C" int cogdl_hip_gat_fwd(const int32_t *rowptr, const int32_t *colind, const float *attn_row,
                                 const float *attn_col, const void *feat, float negative_slope, void *out,
                                 float *edge_max, float *edge_sum, int64_t v, int64_t h, int64_t f, int64_t nnz,
                                 int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (v < 0 || h < 0 || f < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (v == 0 || h == 0 || f == 0) return COGDL_HIP_OK;
    if (!rowptr || !attn_row || !attn_col || !feat || !out || !edge_max || !edge_sum) return COGDL_HIP_EINVAL;
    if (h * f > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    FwdArgs a{rowptr, colind, attn_row, attn_col, feat, negative_slope, out, edge_max, edge_sum, v, h, f, nnz};
    switch (dtype) {
        case COGDL_HIP_F32: return gat_fwd_typed<float>(a, workspace, workspace_bytes, s);
        case COGDL_HIP_F16: return gat_fwd_typed<__half>(a, workspace, workspace_bytes, s);
        case COGDL_HIP_BF16: return gat_fwd_typed<__hip_bfloat16>(a, workspace, workspace_bytes, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

// Layout of the backward workspace: [D: v*h floats][long-row scratch of the row pass][... of the column pass].
extern "C" size_t cogdl_hip_gat_bwd_workspace_bytes(int64_t v, int64_t h, int64_t f, int64_t nnz, int dtype) {
    size_t total = dvec_bytes(v, h);
    const int vec = gat_bwd_vec(h, f, 16, dtype == COGDL_HIP_F32 ? 4 : 2);
    if (vec > 0 && nnz > 0) {
        const int lpr = gat_bwd_lpr(h, f, vec);
        total += rowreduce_workspace_bytes(nnz, (int64_t)(vec + 1) * lpr);
        total += rowreduce_workspace_bytes(nnz, (int64_t)(2 * vec + 1) * lpr);
    }
    return total + 256;
}

// Returns COGDL_HIP_EUNSUPPORTED for shapes the fused backward does not cover (the [H*F] row must fit one group of
// 64 lanes * 4 columns, and F/VEC must be a power of two unless H == 1): callers then use the unfused
// edge_softmax + mhspmm operators instead.
extern "C" int cogdl_hip_gat_bwd(const int32_t *rowptr, const int32_t *colind, const int32_t *colptr,
                                 const int32_t *rowind, const float *attn_row, const float *attn_col,
                                 const void *feat, float negative_slope, const float *edge_max,
                                 const float *edge_sum, const void *out, const void *grad_out, void *grad_feat,
                                 float *grad_attn_row, float *grad_attn_col, void *workspace, size_t workspace_bytes,
                                 int64_t v, int64_t n_src, int64_t h, int64_t f, int64_t nnz, int dtype,
                                 void *stream) {
    if (v < 0 || n_src < 0 || h <= 0 || f <= 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (dtype != COGDL_HIP_F32 && dtype != COGDL_HIP_F16 && dtype != COGDL_HIP_BF16) return COGDL_HIP_EDTYPE;
    const int elem_bytes = dtype == COGDL_HIP_F32 ? 4 : 2;
    if (!rowptr || !colptr || !attn_row || !attn_col || !feat || !edge_max || !edge_sum || !out || !grad_out ||
        !grad_feat || !grad_attn_row || !grad_attn_col || !workspace)
        return COGDL_HIP_EINVAL;
    if (workspace_bytes < dvec_bytes(v, h) || !aligned_to(workspace, 256)) return COGDL_HIP_EWORKSPACE;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) |
                           reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_feat);
    const int align = (bits % 16 == 0) ? 16 : (bits % 8 == 0) ? 8 : (bits % 4 == 0) ? 4 : 2;
    if (align < elem_bytes) return COGDL_HIP_EALIGN;
    const int vec = gat_bwd_vec(h, f, align, elem_bytes);
    if (vec == 0) return COGDL_HIP_EUNSUPPORTED;
    BwdArgs b{rowptr, colind, colptr, rowind, attn_row, attn_col, feat, edge_max, edge_sum, out, grad_out,
              negative_slope, grad_feat, grad_attn_row, grad_attn_col, (float *)workspace, v, n_src, h, f, nnz,
              nullptr, nullptr, 0, 0};
    // the long-row scratch is used only when the caller's workspace covers all of it
    const int lpr = gat_bwd_lpr(h, f, vec);
    const size_t need_row = rowreduce_workspace_bytes(nnz, (int64_t)(vec + 1) * lpr);
    const size_t need_col = rowreduce_workspace_bytes(nnz, (int64_t)(2 * vec + 1) * lpr);
    if (nnz > 0 && workspace_bytes >= dvec_bytes(v, h) + need_row + need_col) {
        b.ws_row = (char *)workspace + dvec_bytes(v, h);
        b.wsb_row = need_row;
        b.ws_col = (char *)b.ws_row + need_row;
        b.wsb_col = need_col;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32: return gat_bwd_typed<float>(b, vec, s);
        case COGDL_HIP_F16: return gat_bwd_typed<__half>(b, vec, s);
        default: return gat_bwd_typed<__hip_bfloat16>(b, vec, s);
    }
}
