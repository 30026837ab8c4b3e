// gat_op.h -- the functors of the fused GAT attention + aggregation over the row-reduce engine (rowreduce.h), shared by
// gat.hip (plain), gat_drop.hip (attention dropout, DROP) and gat_tiled.hip (backward for rows wider than one lane
// group or heads that are not a power-of-two number of lanes, TILED).
//
// fused_gat_func of cogdl/operators/fused_gat.py:14-41; the reference binds dgNN's fused_gatconv, whose source is absent
// from the tree, so the semantics are those of the unfused layer path cogdl/layers/gat_layer.py:72-77:
//     s[e,h] = LeakyReLU(attn_row[row(e),h] + attn_col[col[e],h])
//     a[e,h] = softmax over the edges of row(e)
//     d[e,h] = 1                                       (plain)
//            = keep(e,h) / (1 - p)                     (DROP: nn.Dropout(attn_drop) on the attention, gat_layer.py:75 --
//                                                       the branch CogDL's gat model takes by default, models/nn/gat.py:30;
//                                                       the mask is a pure function of (seed, e, h), philox.h)
//     out[v,h,:] = sum_e d[e,h] * a[e,h] * feat[col[e],h,:]
// Forward: ONE pass over the edges with an online softmax (running max / rescaled sum, flash-attention style), so the
// [E,H] score / attention / mask tensors are never materialised: algorithmic bytes per edge =
// 4 (colind) + 4H (attn_col row) + H*F*s (feature row); per node 2*4H + H*F*s.  Row-group decomposition and long-row
// (chunk-parallel) path of rowreduce.h: a group of LPR lanes owns one destination row, each lane VEC columns inside one
// head; the state (max, sum, acc[VEC]) of two edge ranges merges like two flash-attention blocks, so hub rows are
// split over whole workgroups.  The softmax statistics run over ALL edges of the row, dropped or not.
// Backward (fp32 arithmetic): with D[v,h] = <g[v,h,:], out[v,h,:]>  (= sum_e a_e d_e <g, feat_e>, dropout included),
//     dL/ds[e,h]         = a[e,h] * (d[e,h] * <g[row,h,:], feat[col,h,:]> - D[row,h]) * LeakyReLU'(.)
//     grad_attn_row[v,h] = sum over the row's edges            (row pass over the CSR, gathers feat)
//     grad_attn_col[u,h] = sum over the column's edges         (column pass over the CSC, gathers g)
//     grad_feat[u,h,:]   = sum over the column's edges a * d * g[row,h,:]        (same column pass)
// a[e,h] is recomputed from the saved row max / row sum (edge_max, edge_sum), exactly what the reference's
// FusedGATFunction keeps in ctx (operators/fused_gat.py:20); d[e,h] is regenerated from the seed (the column pass reads
// the CSR position of its slot from the plan's permutation).  No atomics anywhere: deterministic.
#pragma once
#include "philox.h"
#include "rowreduce.h"

namespace cogdl {

// Per-edge exponentials (argument = score - row max <= 0): the hardware exponential, as in edge_softmax.hip -- relative
// error |x| * 2^-24 (< 2e-6 wherever the weight is not negligible); the rare state merges keep libm's expf.
__device__ __forceinline__ float gat_exp(float x) { return __expf(x); }
__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }
// (Round 6, tried: max(v, slope * v) for slopes in [0, 1] -- two instructions on paper; the uniform test on the slope stays
//  inside the batch loop as two scalar branches per edge and fmaxf brings a canonicalising v_max of its own: no gain.)
// Backward: a[e,h] = exp(s - max) / sum as ONE fused multiply-add and ONE v_exp: exp2(s * log2(e) + nml) with
// nml = -(max * log2(e) + log2(sum)) per (row, head) -- where subtract / scale / v_exp / multiply by 1 / sum were four.  The
// argument carries one rounding instead of two plus the reciprocal's (a moves by ~1e-7 relative; parity of the backward is a
// tolerance).  A row that has edges has sum >= 1 (its maximum contributes exp(0)); sum == 0 gives nml = -inf, a = 0.
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float gat_neg_max_log(float mx, float ls) {
    return ls > 0.f ? -(mx * kLog2e + __log2f(ls)) : -INFINITY;
}
__device__ __forceinline__ float gat_softmax_weight(float s, float nml) {
    return __builtin_amdgcn_exp2f(__builtin_fmaf(s, kLog2e, nml));
}
// A kernel argument that the compiler must keep in a vector register (v_cndmask with VCC takes no scalar second source,
// and the compiler would rather re-materialise the v_mov per edge than spend the register).
__device__ __forceinline__ float pinned_vgpr(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
// LeakyReLU and its derivative from ONE compare: f = 1 or slope, value = v * f (v * 1 is exact).
__device__ __forceinline__ float leaky_with_grad(float v, float slope, float &grad) {
    grad = v > 0.f ? 1.f : slope;
    return v * grad;
}
static inline bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// Attention dropout of the DROP functors.
struct GatDrop {
    uint32_t seed_lo, seed_hi;
    uint32_t thresh;     // keep  <=>  u16(e,h) >= thresh
    float scale;         // value of a kept d[e,h]
    const int32_t *eid;  // CSC pass: CSR position of slot k (the plan's perm); NULL: the slot index itself
};
template <bool DROP>
struct DropLane {};
template <>
struct DropLane<true> {
    uint32_t klo, khi;  // keep bits (heads 0..31 / 32..63) of the edge this lane loaded
};

template <bool DROP>
__device__ __forceinline__ void drop_lane_load(const GatDrop &dr, DropLane<DROP> &lv, int64_t e, int heads) {
    if constexpr (DROP) {
        const uint32_t id = dr.eid ? (uint32_t)dr.eid[e] : (uint32_t)e;
        const uint64_t bits = drop_keep_bits(dr.seed_lo, dr.seed_hi, id, heads, dr.thresh);
        lv.klo = (uint32_t)bits;
        lv.khi = (uint32_t)(bits >> 32);
    }
}
// d[e, hd] of the edge that lane jj of the group loaded.
template <int LPR>
__device__ __forceinline__ float drop_factor(const GatDrop &dr, const DropLane<true> &lv, int sub, int jj, int hd,
                                             int heads) {
    uint32_t w = (uint32_t)group_bcast<LPR>((int)lv.klo, sub, jj);
    if (heads > 32) {  // (uniform)
        const uint32_t w2 = (uint32_t)group_bcast<LPR>((int)lv.khi, sub, jj);
        if (hd >= 32) w = w2;
    }
    // bit hd as a mask of 32 copies (ONE signed bit-field extract), and-ed on to the scale's bits: two instructions where
    // shift / and / compare / select were four plus the condition register's wait states
    return __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)w, (uint32_t)(hd & 31), 1u) & __float_as_uint(dr.scale));
}

// Sum over the `lph` lanes (power of two, aligned) that hold one head.
template <int LPR>
__device__ __forceinline__ float head_sum(float v, int lph) {
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1)
        if (s < lph) v += __shfl_xor(v, s, kWave);
    return v;
}
// TILED: the lanes of one head are a contiguous run [.., seg_last] of the group (any length); on return the FIRST lane
// of the run holds the run's total (the others a suffix sum).  Executed by all lanes of the wave.
template <int LPR>
__device__ __forceinline__ float seg_sum(float v, int l, int seg_last) {
#pragma unroll
    for (int s = 1; s < LPR; s <<= 1) {
        const float o = __shfl_down(v, s, kWave);
        if (l + s <= seg_last) v += o;
    }
    return v;
}

// TILED backward: per-(tile, row, head) partial sums, finished by gat_finish_kernel (gat_tiled.hip).
struct GatTiles {
    float *pdot;   // [tiles][rows][heads]
    float *pd;     // [tiles][rows][heads]   (row pass: partial <g, out>)
    float *hsum;   // [rows][heads]          (row pass: sum_e c_e;  column pass: sum_e c_e D[row_e])
    int64_t rows;
};

// Per-SOURCE scalars of a chunk through LDS (ACH / ACS template arguments of the functors below; round 6, second half).
// The fused kernels over an XCD-partitioned plan are bound by the CU's memory pipeline as much as by the vector ALUs: one
// wave64 memory instruction per ~16 cycles whatever it loads (Reddit-shaped graph, bf16 H = 8 x F = 8 forward: 4.2e7 vector
// memory instructions in 1449 us = one per 19.7 cycles and CU, profiles/r06_mem_pipeline.txt), and every edge costs two:
// its 128-byte feature row and the 4 bytes per lane of attn_col (or the 16-byte stats record of the column pass) -- the
// second one moves a sixteenth of the bytes for the same slot.  With 8-lane groups a chunk is 8 edges and lane l already
// holds the column id of edge l: it loads the WHOLE scalar row of ITS edge once per chunk (H = 1: one dword / one 16-byte
// record; H = 8 -- two 16-byte loads -- is implemented by the same code and measured slower, see gat_launch_fwd), parks it in the group's LDS scratch when the first slot is folded (the loads were issued
// BEFORE the chunk's gathers, so waiting for them does not wait for the gathers), and every slot reads its value back:
// 10 (9) memory instructions per chunk instead of 16, the rest is LDS traffic (which has its own pipeline).  Rows are
// rotated by the group's index inside the wave so that the eight groups of a wave hit different banks.
template <int N>
struct ChunkScalars {
    float v[N > 0 ? N : 1];
    float *lds;
    int rot;
};
template <int N, bool A24>
__device__ __forceinline__ void chunk_scalars_load(ChunkScalars<N> &cs, const float *table, int my_c, int sub, float *lds) {
    if constexpr (N > 0) {
        const float *src = gather_row<A24>(table, 0, my_c, (uint32_t)N * 4u);
        if constexpr (N % 4 == 0) {
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const float4 t = reinterpret_cast<const float4 *>(src)[q];
                cs.v[4 * q] = t.x, cs.v[4 * q + 1] = t.y, cs.v[4 * q + 2] = t.z, cs.v[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < N; ++q) cs.v[q] = src[q];
        }
        cs.lds = lds;
        cs.rot = sub;
    }
}
// called by every lane of the group when the chunk's first slot is folded
template <int N, int LPR>
__device__ __forceinline__ void chunk_scalars_park(const ChunkScalars<N> &cs, int l) {
    if constexpr (N > 0) {
        float *dst = cs.lds + ((l + cs.rot) & (LPR - 1)) * N;
        if constexpr (N % 4 == 0) {
#pragma unroll
            for (int q = 0; q < N / 4; ++q)
                reinterpret_cast<float4 *>(dst)[q] = make_float4(cs.v[4 * q], cs.v[4 * q + 1], cs.v[4 * q + 2], cs.v[4 * q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < N; ++q) dst[q] = cs.v[q];
        }
    }
}
template <int N, int LPR>
__device__ __forceinline__ const float *chunk_scalars_row(const ChunkScalars<N> &cs, int jpos) {
    return cs.lds + ((jpos + cs.rot) & (LPR - 1)) * N;
}

// ------------------------------------------------------------------------------------------ forward
// ACH: 0 = attn_col[col, head] gathered per slot; H (8 or 1, groups of 8 lanes, heads == ACH) = through ChunkScalars.
template <typename T, int VEC_, int LPR_, int UNROLL_, bool DROP, bool A24 = false, int ACH = 0>
struct GatFwdOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_ + 2;
    static constexpr bool kReduce = true;
    static constexpr int kLds = ACH;
    static_assert(ACH == 0 || LPR_ == 8, "ChunkScalars: groups of 8 lanes");
    static constexpr bool kExpensiveLaneLoad = DROP;  // (the keep bits of an edge: one Philox call)
    const float *attn_row, *attn_col;
    const T *feat;
    T *out;
    float *edge_max, *edge_sum;
    float slope;
    int heads, fdim;
    GatDrop drop;
    struct Ctx {
        int col0, cc, hd, l, sub;
        bool col_ok;
        float ar;
        ChunkScalars<ACH> cs;
        DropLane<DROP> lv;
    };
    struct State {
        float acc[VEC];
        float mx, lsum;
    };
    using LaneVals = DropLane<DROP>;
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
        c.l = l;
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
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &lv, int64_t e) const {
        drop_lane_load<DROP>(drop, lv, e, heads);
    }
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &lv, int sub,
                                          int jj) const {
        if constexpr (ACH == 0) b.ac[u] = *gather_row<A24>(attn_col, c.hd, col, (uint32_t)heads * 4u);
        load_vec<T, VEC>(gather_row<A24>(feat, c.cc, col, (uint32_t)(heads * fdim) * (uint32_t)sizeof(T)), b.v[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int jpos) const {
        if constexpr (ACH > 0) {
            if (jpos == 0) chunk_scalars_park<ACH, LPR>(c.cs, c.l);
        }
        float d_u = 1.f;
        if constexpr (DROP) d_u = drop_factor<LPR>(drop, c.lv, c.sub, jpos & (LPR - 1), c.hd, heads);
        if (valid) {
            float ac_u;
            if constexpr (ACH > 0) ac_u = chunk_scalars_row<ACH, LPR>(c.cs, jpos)[ACH == 1 ? 0 : c.hd];
            else ac_u = b.ac[u];
            // Online softmax with ONE exponential per (edge, lane): of exp(mx - mn) and exp(sc - mn), mn = max(mx, sc), one
            // is exp(0) = 1 and the other t = exp(-|sc - mx|) (mx = -inf at the start of a row: t = 0 = the rescale of
            // the empty state).  Round 6: these kernels are bound by instruction issue on hub-heavy graphs (SQ counters,
            // profiles/r06_sq_reddit.txt; with both gathers folded to L2 hits the launch only drops from 1888 to 1474 us: DESIGN
            // section 5); the second exponential was libm's expf,
            // ~25 instructions per edge and lane.  The accumulator is rescaled only when some lane of the wave saw a new
            // maximum (wave-uniform branch; multiplying by exactly 1 is what is skipped): rare beyond the first edges of a row.
            const float sc = leaky(c.ar + ac_u, slope);
            const float dlt = sc - s.mx;
            const bool up = dlt > 0.f;
            const float t = gat_exp(-fabsf(dlt));  // (|.| and the sign are operand modifiers of the multiply in front of v_exp)
            const float p = up ? 1.f : t;
            const float pw = DROP ? p * d_u : p;
            if (__ballot(up) != 0ull) {
                const float scale = up ? t : 1.f;
                s.lsum *= scale;
#pragma unroll
                for (int i = 0; i < VEC; ++i) s.acc[i] *= scale;
                s.mx = up ? sc : s.mx;
            }
            s.lsum += p;
#pragma unroll
            for (int i = 0; i < VEC; ++i) s.acc[i] = fmaf(pw, b.v[u][i], s.acc[i]);
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &c, State &, int, int, int my_c, int sub, int, float *lds,
                                                const LaneVals &lv) const {
        chunk_scalars_load<ACH, A24>(c.cs, attn_col, my_c, sub, lds);
        c.lv = lv;
        c.sub = sub;
    }
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
// DROP: the weight that goes to LDS is p * d[e,h]; the running sum takes p.
template <typename T, int VEC_, int LPR_, int UNROLL_, bool DROP>
struct GatFwdChunkOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_ + 2;
    static constexpr bool kReduce = true;
    static constexpr int kMaxHeads = 16;
    static constexpr int kLds = kMaxHeads;  // LPR edges x H heads weights per group
    static constexpr bool kExpensiveLaneLoad = DROP;
    const float *attn_row, *attn_col;
    const T *feat;
    T *out;
    float *edge_max, *edge_sum;
    float slope;
    int heads, fdim;
    GatDrop drop;

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
    using LaneVals = DropLane<DROP>;
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
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &lv, int64_t e) const {
        drop_lane_load<DROP>(drop, lv, e, heads);
    }
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
    __device__ __forceinline__ void chunk_begin(Ctx &c, State &s, int, int cnt, int my_c, int, int l, float *lds,
                                                const LaneVals &lv) const {
        const int epr = LPR / heads;  // edges per round
        const int eo = l / heads;
        const int rounds = (cnt + epr - 1) / epr;  // <= heads <= 16
        float lmax = -INFINITY;
        uint32_t kept = 0;  // DROP: bit r = (edge of round r, head hs) is kept
        for (int r = 0; r < rounds; ++r) {
            const int je = r * epr + eo;
            const int src = c.lane0 + min(je, cnt - 1);
            const int col = __shfl(my_c, src, kWave);
            const float sc = (je < cnt) ? leaky(c.ar + attn_col[(int64_t)col * heads + c.hs], slope) : -INFINITY;
            if constexpr (DROP) kept |= (((uint32_t)__shfl((int)lv.klo, src, kWave) >> c.hs) & 1u) << r;
            lds[r * LPR + l] = sc;  // == lds[je * H + hs]
            lmax = fmaxf(lmax, sc);
        }
        lmax = head_max(lmax);
        const float mn = fmaxf(s.mx, lmax);
        const float scale = (s.lsum == 0.f) ? 0.f : expf(s.mx - mn);
        float psum = 0.f;
        for (int r = 0; r < rounds; ++r) {
            const float p = gat_exp(lds[r * LPR + l] - mn);  // exp(-inf) == 0 for the masked tail
            if constexpr (DROP) lds[r * LPR + l] = ((kept >> r) & 1u) ? p * drop.scale : 0.f;
            else lds[r * LPR + l] = p;
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
// Column mapping shared by the two backward functors.  !TILED: the whole [H*F] row in ONE group, a head = `lph`
// aligned lanes (power of two; H == 1: the whole zero-padded group).  TILED: column tiles of LPR * VEC columns
// (blockIdx.y), a head = any contiguous run of lanes of a tile; per-tile partial head sums go to GatTiles.
struct GatBwdLane {
    int cc, hd, tile, seg_last;
    bool col_ok, head_lane, owner;  // head_lane: first lane of its head (in this tile); owner: first lane of its head overall
};
template <int VEC, int LPR, bool TILED>
__device__ __forceinline__ GatBwdLane gat_bwd_lane(int l, int tile, int heads, int fdim) {
    GatBwdLane c;
    const int col0 = ((TILED ? tile : 0) * LPR + l) * VEC;
    c.tile = TILED ? tile : 0;
    c.col_ok = col0 < heads * fdim;
    c.cc = c.col_ok ? col0 : 0;
    c.hd = c.cc / fdim;
    c.owner = c.col_ok && (c.cc % fdim == 0);
    c.head_lane = c.owner || (TILED && c.col_ok && l == 0);
    c.seg_last = TILED ? min(LPR - 1, ((c.hd + 1) * fdim - 1) / VEC - c.tile * LPR) : 0;
    return c;
}

// Row pass: D[v,h] and grad_attn_row[v,h].
template <typename T, int VEC_, int LPR_, int UNROLL_, bool DROP, bool TILED, bool A24 = false, int ACH = 0>
struct GatBwdRowOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = VEC_ + 1;
    static constexpr bool kReduce = true;
    static constexpr int kLds = ACH;  // (ChunkScalars: attn_col of a chunk's edges, see the forward functor)
    static_assert(ACH == 0 || (LPR_ == 8 && !TILED), "ChunkScalars: groups of 8 lanes");
    static constexpr int kMinWaves = (VEC_ == 8) ? 5 : 1;  // 8-element lanes: 90-105 VGPRs, five waves per SIMD fit in 102
    static constexpr bool kExpensiveLaneLoad = DROP;  // (the keep bits of an edge: one Philox call)
    const float *attn_row, *attn_col;
    const T *feat;  // feat / out / grad_out in the layer's dtype (f32, f16, bf16): read natively, fp32 arithmetic
    const float *edge_max, *edge_sum;
    const T *out, *grad_out;
    float4 *stats;  // [V, H] {attn_row, nml (gat_neg_max_log), D, 1 / edge_sum}: everything the column pass needs per (row, head), as
                    // ONE 16-byte gather per edge and lane instead of four 4-byte gathers from four arrays (each of
                    // which drags its own 64-byte sector in: the four scalars cost as much traffic as the feature row)
    float *grad_attn_row;
    float slope;
    int heads, fdim, lph;
    GatDrop drop;
    GatTiles tiles;

    struct Ctx {
        GatBwdLane m;
        float g[VEC];
        float d, ar, nml, inv, slope_v;
        int l, sub;
        ChunkScalars<ACH> cs;
        DropLane<DROP> lv;
    };
    // grad_attn_row[v,h] = sum_e c_e (d_e <g, feat[col_e]> - D) with c_e = a_e * LeakyReLU'(.)
    //                    = < g, sum_e c_e d_e feat[col_e] >  -  D * sum_e c_e :
    // the per-edge dot product (a cross-lane reduction per edge) becomes ONE reduction per row over a weighted
    // accumulation s[] of the gathered rows, exactly the shape of an SpMM.
    struct State {
        float s[VEC];
        float csum;
    };
    using LaneVals = DropLane<DROP>;
    struct Batch {
        float v[UNROLL][VEC];
        float ac[UNROLL];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int tile) const {
        Ctx c;
        c.m = gat_bwd_lane<VEC, LPR, TILED>(l, tile, heads, fdim);
        c.slope_v = pinned_vgpr(slope);
        c.l = l;
        return c;
    }
    __device__ __forceinline__ int lane_of(const Ctx &) const { return (int)(threadIdx.x & (kWave - 1)) % LPR; }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t row, bool ok) const {
        const int k = heads * fdim;
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.g[i] = 0.f;
        float d = 0.f;
        c.ar = c.inv = 0.f;
        c.nml = -INFINITY;
        if (ok && c.m.col_ok) {
            load_vec<T, VEC>(grad_out + row * (int64_t)k + c.m.cc, c.g);
            float o[VEC];
            load_vec<T, VEC>(out + row * (int64_t)k + c.m.cc, o);
#pragma unroll
            for (int i = 0; i < VEC; ++i) d = fmaf(c.g[i], o[i], d);
            c.ar = attn_row[row * heads + c.m.hd];
            const float ls = edge_sum[row * heads + c.m.hd];
            c.nml = gat_neg_max_log(edge_max[row * heads + c.m.hd], ls);
            c.inv = ls > 0.f ? 1.f / ls : 0.f;
        }
        if constexpr (TILED) c.d = seg_sum<LPR>(d, lane_of(c), c.m.seg_last);
        else c.d = head_sum<LPR>(d, lph);
    }
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.s[i] = 0.f;
        s.csum = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { init_zero(s); }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &lv, int64_t e) const {
        drop_lane_load<DROP>(drop, lv, e, heads);
    }
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &lv, int sub,
                                          int jj) const {
        if constexpr (ACH == 0) b.ac[u] = *gather_row<A24>(attn_col, c.m.hd, col, (uint32_t)heads * 4u);
        load_vec<T, VEC>(gather_row<A24>(feat, c.m.cc, col, (uint32_t)(heads * fdim) * (uint32_t)sizeof(T)), b.v[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int jpos) const {
        if constexpr (ACH > 0) {
            if (jpos == 0) chunk_scalars_park<ACH, LPR>(c.cs, c.l);
        }
        float d_u = 1.f;
        if constexpr (DROP) d_u = drop_factor<LPR>(drop, c.lv, c.sub, jpos & (LPR - 1), c.m.hd, heads);
        if (valid) {
            float ac_u;
            if constexpr (ACH > 0) ac_u = chunk_scalars_row<ACH, LPR>(c.cs, jpos)[ACH == 1 ? 0 : c.m.hd];
            else ac_u = b.ac[u];
            const float pre = c.ar + ac_u;
            float lg;
            const float lk = leaky_with_grad(pre, c.slope_v, lg);
            const float ce = gat_softmax_weight(lk, c.nml) * lg;
            const float cw = DROP ? ce * d_u : ce;
#pragma unroll
            for (int i = 0; i < VEC; ++i) s.s[i] = fmaf(cw, b.v[u][i], s.s[i]);
            s.csum += ce;
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &c, State &, int, int, int my_c, int sub, int, float *lds,
                                                const LaneVals &lv) const {
        chunk_scalars_load<ACH, A24>(c.cs, attn_col, my_c, sub, lds);
        c.lv = lv;
        c.sub = sub;
    }
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t row, bool ok) const {
        float dot = 0.f;  // (all lanes of the group take part in the reduction)
#pragma unroll
        for (int i = 0; i < VEC; ++i) dot = fmaf(c.g[i], c.m.col_ok ? s.s[i] : 0.f, dot);
        if constexpr (TILED) {
            dot = seg_sum<LPR>(dot, lane_of(c), c.m.seg_last);
            if (ok && c.m.head_lane) {
                const int64_t at = ((int64_t)c.m.tile * tiles.rows + row) * heads + c.m.hd;
                tiles.pdot[at] = dot;
                tiles.pd[at] = c.d;
                if (c.m.owner) tiles.hsum[row * heads + c.m.hd] = s.csum;
            }
        } else {
            dot = head_sum<LPR>(dot, lph);
            if (ok && c.m.head_lane) {
                stats[row * heads + c.m.hd] = make_float4(c.ar, c.nml, c.d, c.inv);
                grad_attn_row[row * heads + c.m.hd] = dot - c.d * s.csum;
            }
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
// ACS: 4 = one head: the 16-byte stats record of a chunk's edges through ChunkScalars (every lane of a group reads the same).
template <typename T, int VEC_, int LPR_, int UNROLL_, bool DROP, bool TILED, bool A24 = false, int ACS = 0>
struct GatBwdColOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = 2 * VEC_ + 1;
    static constexpr bool kReduce = true;
    static constexpr int kLds = ACS;
    static_assert(ACS == 0 || (ACS == 4 && LPR_ == 8 && !TILED), "ChunkScalars: groups of 8 lanes, one head");
    // 8-element lanes sit at 133-138 VGPRs: just past the 128 of four waves per SIMD
    static constexpr int kMinWaves = (VEC_ == 8) ? 4 : 1;
    static constexpr bool kExpensiveLaneLoad = DROP;  // (the keep bits of an edge: one Philox call)
    const float *attn_row, *attn_col;
    const T *feat;
    const float4 *stats;  // [V, H] {attn_row, nml, D, 1 / edge_sum} (written by the row pass; the first three are ONE 12-byte gather)
    const T *grad_out;
    T *grad_feat;  // rounded once on store (fp32 accumulation)
    float *grad_attn_col;
    float slope;
    int heads, fdim, lph;
    GatDrop drop;
    GatTiles tiles;

    struct Ctx {
        GatBwdLane m;
        float f[VEC];
        float ac, slope_v;
        int l, sub;
        ChunkScalars<ACS> cs;
        DropLane<DROP> lv;
    };
    // grad_attn_col[u,h] = sum_e c_e (d_e <feat[u], g[row_e]> - D[row_e])
    //                    = <feat[u], sum_e c_e d_e g[row_e]> - sum_e c_e D[row_e]:
    // a second weighted accumulation t[] next to grad_feat's (weights c_e d_e = a_e d_e LeakyReLU' instead of a_e d_e) and
    // one cross-lane reduction per column instead of one per edge.
    struct State {
        float acc[VEC];
        float t[VEC];
        float cd;
    };
    using LaneVals = DropLane<DROP>;
    struct Batch {
        float g[UNROLL][VEC];
        float4 st[UNROLL];
    };

    __device__ __forceinline__ Ctx make_ctx(int l, int tile) const {
        Ctx c;
        c.m = gat_bwd_lane<VEC, LPR, TILED>(l, tile, heads, fdim);
        c.slope_v = pinned_vgpr(slope);
        c.l = l;
        return c;
    }
    __device__ __forceinline__ int lane_of(const Ctx &) const { return (int)(threadIdx.x & (kWave - 1)) % LPR; }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t u_node, bool ok) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.f[i] = 0.f;
        c.ac = 0.f;
        if (ok && c.m.col_ok) {
            load_vec<T, VEC>(feat + u_node * (int64_t)(heads * fdim) + c.m.cc, c.f);
            c.ac = attn_col[u_node * heads + c.m.hd];
        }
    }
    __device__ __forceinline__ void init_zero(State &s) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.acc[i] = s.t[i] = 0.f;
        s.cd = 0.f;
    }
    __device__ __forceinline__ void init(const Ctx &, State &s, int64_t, bool) const { init_zero(s); }
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &lv, int64_t e) const {
        drop_lane_load<DROP>(drop, lv, e, heads);
    }
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int r, int64_t, const LaneVals &lv, int sub,
                                          int jj) const {
        if constexpr (ACS == 0) b.st[u] = *gather_row<A24>(stats, c.m.hd, r, (uint32_t)heads * 16u);
        load_vec<T, VEC>(gather_row<A24>(grad_out, c.m.cc, r, (uint32_t)(heads * fdim) * (uint32_t)sizeof(T)), b.g[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s, const Batch &b, int u, bool valid, int64_t,
                                          int jpos) const {
        if constexpr (ACS > 0) {
            if (jpos == 0) chunk_scalars_park<ACS, LPR>(c.cs, c.l);
        }
        float d_u = 1.f;
        if constexpr (DROP) d_u = drop_factor<LPR>(drop, c.lv, c.sub, jpos & (LPR - 1), c.m.hd, heads);
        if (valid) {
            float4 st_u;
            if constexpr (ACS > 0) st_u = *reinterpret_cast<const float4 *>(chunk_scalars_row<ACS, LPR>(c.cs, jpos));
            else st_u = b.st[u];
            const float pre = st_u.x + c.ac;
            float lg;
            const float lk = leaky_with_grad(pre, c.slope_v, lg);
            const float a = gat_softmax_weight(lk, st_u.y);
            const float ce = a * lg;
            const float aw = DROP ? a * d_u : a;
            const float cw = DROP ? ce * d_u : ce;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                s.acc[i] = fmaf(aw, b.g[u][i], s.acc[i]);
                s.t[i] = fmaf(cw, b.g[u][i], s.t[i]);
            }
            s.cd = fmaf(ce, st_u.z, s.cd);
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &c, State &, int, int, int my_c, int sub, int, float *lds,
                                                const LaneVals &lv) const {
        chunk_scalars_load<ACS, A24>(c.cs, reinterpret_cast<const float *>(stats), my_c, sub, lds);
        c.lv = lv;
        c.sub = sub;
    }
    __device__ __forceinline__ void batch_end(const Ctx &, State &, int, int, int) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &c, const State &s, int64_t u_node, bool ok) const {
        float dot = 0.f;  // (all lanes of the group take part in the reduction)
#pragma unroll
        for (int i = 0; i < VEC; ++i) dot = fmaf(c.f[i], c.m.col_ok ? s.t[i] : 0.f, dot);
        if constexpr (TILED) dot = seg_sum<LPR>(dot, lane_of(c), c.m.seg_last);
        else dot = head_sum<LPR>(dot, lph);
        if (ok && c.m.col_ok) {
            store_vec<T, VEC>(grad_feat + u_node * (int64_t)(heads * fdim) + c.m.cc, s.acc);
            if constexpr (TILED) {
                if (c.m.head_lane) {
                    tiles.pdot[((int64_t)c.m.tile * tiles.rows + u_node) * heads + c.m.hd] = dot;
                    if (c.m.owner) tiles.hsum[u_node * heads + c.m.hd] = s.cd;
                }
            } else {
                if (c.m.head_lane) grad_attn_col[u_node * heads + c.m.hd] = dot - s.cd;
            }
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

// --------------------------------------------------------------------------------------------- host side
// Forward geometry: the WIDEST legal vector (16-byte lanes) with every lane's columns inside one head.  Unlike
// csr_spmm (which narrows the vector until a row fills 64 lanes) every lane here repeats the per-edge softmax
// arithmetic of its head (leaky-relu, two exps, the rescale), so fewer, fatter lanes win: measured on MI355X,
// reddit-shaped graph, H=8 x F=8: f32 vec 1 -> 4: 8.7 -> 4.7 ms, bf16 vec 2 -> 8: 6.6 -> 3.2 ms.
inline RowGeometry gat_fwd_geometry(int64_t h, int64_t f, int elem_bytes, int align) {
    const int maxv = 16 / elem_bytes;
    int vec = 1;
    const int forced = g_tuning[kTuneGatVec];  // experiments: cap the vector width (0 = automatic)
    for (int w = 1; w <= maxv; w <<= 1) {
        if (f % w != 0 || align % (w * elem_bytes) != 0) break;
        vec = w;
        if (forced && w >= forced) break;
    }
    // (Round 4, first half: one or two heads whose row fits 8 fat lanes ran in 16-lane groups -- chunk-wise softmax -- by
    //  halving the vector once: H=1 x F=48 bf16 2.89 -> 2.39 ms.  With the next chunk's ids prefetched in reduce_edges the
    //  8 fat lanes win again: 2.35 vs 2.49 ms, with dropout 2.66 vs 2.85 ms, H=2 x F=32 2.33 vs 2.70 ms
    //  (tools/gat_h1_probe.py) -- the rule is gone.)
    // (Round 6: one head of 16-bit features in FOUR lanes of 16 elements -- two 16-byte loads per lane, a third fewer
    //  lane-instructions per edge, 16 rows per wave -- measured slower: Reddit-shaped graph, H = 1 x F = 48 bf16 forward 1920 ->
    //  2226 us, F = 64 1587 -> 1995 us (profiles/r06_gat_h1.txt); removed again.  What the same run showed instead: F = 64 is
    //  FASTER than F = 48 -- 128-byte rows are one L2 line each, 96-byte rows straddle -- see _padded_width in
    //  cogdl_amd/operators/fused_gat.py.)
    const int64_t need = (h * f + vec - 1) / vec;
    int lpr = 8;
    while (lpr < kWave && lpr < need) lpr <<= 1;
    RowGeometry g;
    g.vec = vec;
    g.lpr = lpr;
    g.tiles = (h * f + (int64_t)lpr * vec - 1) / ((int64_t)lpr * vec);
    return g;
}

struct GatFwdArgs {
    const int32_t *rowptr, *colind;
    const float *ar, *ac;
    const void *feat;
    float slope;
    void *out;
    float *emax, *esum;
    int64_t v, h, f, nnz;
    GatDrop drop;
    const cogdl_hip_vrows *vr = nullptr;  // XCD-partitioned plan (rowreduce.h): rowptr / colind are then unused
};

template <typename T, int VEC, int LPR, bool DROP>
static int gat_launch_fwd(const GatFwdArgs &a, int64_t tiles, void *ws, size_t wsb, hipStream_t s) {
    // Chunk-wise softmax pays once a round of the weight computation covers >= 8 edges (LPR/H) of a >= 16-edge chunk:
    // measured on MI355X (reddit-shaped graph, f32) H=1,F=41: 7.3 -> 3.6 ms, H=1,F=64: 4.1 -> 2.8 ms, H=8,F=8
    // (LPR 16, 2 edges per round): 4.2 -> 4.3 ms, bf16 H=8,F=8 (LPR 8): 2.9 -> 4.6 ms.  tuning key 5: 1 = never, 2 = always.
    using Chunk = GatFwdChunkOp<T, VEC, LPR, 8, DROP>;
    const bool can_chunk = pow2(a.h) && a.h <= LPR && a.h <= Chunk::kMaxHeads;
    const bool want_chunk = g_tuning[kTuneGatOnline] == 2 || (g_tuning[kTuneGatOnline] == 0 && LPR >= 16 && a.h * 8 <= LPR);
    if (a.vr) {  // virtual rows: the edge-wise functor, also for the shapes the ordinary entry runs chunk-wise -- with an XCD's
        // virtual rows in order of length the plan wins there too (Reddit-shaped graph, fp32 H = 1 x F = 41 layer of the gat
        // model: training step 20.7 -> 16.9 ms, profiles/r06_gat_bench.txt)
        GatDrop dr = a.drop;
        dr.eid = a.vr->eid;
        if constexpr (LPR == 8) {  // ChunkScalars (attn_col through LDS): one head in groups of 8 lanes
            // (8 heads -- two 16-byte loads per lane and chunk instead of eight dword loads per lane group -- measured SLOWER:
            //  bf16 H = 8 x F = 8 forward 1456 -> 1526 us, backward 2909 -> 2993 us: the instantiation is gone)
            if (tiles == 1 && a.h == 1 && g_tuning[kTuneGatOnline] != 3) {
                GatFwdOp<T, VEC, LPR, 8, DROP, true, 1> op{a.ar, a.ac, (const T *)a.feat, (T *)a.out, a.emax, a.esum, a.slope,
                                                           (int)a.h, (int)a.f, dr};
                return launch_rowreduce_vrows(op, a.vr, tiles, ws, wsb, s);
            }
        }
        GatFwdOp<T, VEC, LPR, 8, DROP, true> op{a.ar, a.ac, (const T *)a.feat, (T *)a.out, a.emax, a.esum, a.slope, (int)a.h,
                                                (int)a.f, dr};
        return launch_rowreduce_vrows(op, a.vr, tiles, ws, wsb, s);
    }
    if (can_chunk && want_chunk) {
        Chunk op{a.ar, a.ac, (const T *)a.feat, (T *)a.out, a.emax, a.esum, a.slope, (int)a.h, (int)a.f, a.drop};
        return launch_rowreduce(op, a.rowptr, a.colind, a.v, a.nnz, tiles, ws, wsb, s);
    }
    GatFwdOp<T, VEC, LPR, 8, DROP> op{a.ar, a.ac, (const T *)a.feat, (T *)a.out, a.emax, a.esum, a.slope, (int)a.h,
                                      (int)a.f, a.drop};
    return launch_rowreduce(op, a.rowptr, a.colind, a.v, a.nnz, tiles, ws, wsb, s);
}

template <typename T, int VEC, bool DROP>
static int gat_dispatch_fwd(const GatFwdArgs &a, const RowGeometry &g, void *ws, size_t wsb, hipStream_t s) {
    switch (g.lpr) {
        case 8: return gat_launch_fwd<T, VEC, 8, DROP>(a, g.tiles, ws, wsb, s);
        case 16: return gat_launch_fwd<T, VEC, 16, DROP>(a, g.tiles, ws, wsb, s);
        case 32: return gat_launch_fwd<T, VEC, 32, DROP>(a, g.tiles, ws, wsb, s);
        default: return gat_launch_fwd<T, VEC, 64, DROP>(a, g.tiles, ws, wsb, s);
    }
}

template <typename T, bool DROP>
static int gat_fwd_typed(const GatFwdArgs &a, void *ws, size_t wsb, hipStream_t s) {
    constexpr int MAXV = 16 / sizeof(T);
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a.feat) | reinterpret_cast<uintptr_t>(a.out);
    const int align = (bits % 16 == 0) ? 16 : (bits % 8 == 0) ? 8 : (bits % 4 == 0) ? 4 : 2;
    if (align < (int)sizeof(T)) return COGDL_HIP_EALIGN;
    const RowGeometry g = gat_fwd_geometry(a.h, a.f, (int)sizeof(T), align);
    switch (g.vec) {
        case 8:
            if constexpr (MAXV >= 8) return gat_dispatch_fwd<T, 8, DROP>(a, g, ws, wsb, s);
        case 4: return gat_dispatch_fwd<T, 4, DROP>(a, g, ws, wsb, s);
        case 2: return gat_dispatch_fwd<T, 2, DROP>(a, g, ws, wsb, s);
        default: return gat_dispatch_fwd<T, 1, DROP>(a, g, ws, wsb, s);
    }
}

template <bool DROP>
static int gat_fwd_any(const GatFwdArgs &a, int dtype, void *ws, size_t wsb, hipStream_t s) {
    switch (dtype) {
        case COGDL_HIP_F32: return gat_fwd_typed<float, DROP>(a, ws, wsb, s);
        case COGDL_HIP_F16: return gat_fwd_typed<__half, DROP>(a, ws, wsb, s);
        case COGDL_HIP_BF16: return gat_fwd_typed<__hip_bfloat16, DROP>(a, ws, wsb, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

struct GatBwdArgs {
    const int32_t *rowptr, *colind, *colptr, *rowind;
    const float *ar, *ac;
    const void *feat;
    const float *emax, *esum;
    const void *out, *gout;
    float slope;
    void *gfeat;
    float *gar, *gac;
    float4 *stats;
    int64_t v, n_src, h, f, nnz;
    void *ws_row, *ws_col;  // long-row scratch of the two passes (either both or none)
    size_t wsb_row, wsb_col;
    GatDrop drop;           // (eid = the plan's perm: used by the column pass only)
    float *t_pdot, *t_pd, *t_hsum;  // TILED scratch
    const cogdl_hip_vrows *vr_row = nullptr, *vr_col = nullptr;  // XCD-partitioned plans of the CSR / of the CSC (both or none)
};

// Backward geometry.  One group (fast path): vec in {1,2,4} (and 8 for 2-byte elements) with F % vec == 0, H*F/vec <= 64
// lanes and F/vec a power of two (unless H == 1); a lane's vector is at most 16 bytes and aligned to its size.
// Everything else is TILED: the widest legal vector, groups of 16 or 64 lanes, ceil(H*F / (lanes*vec)) column tiles.
struct GatBwdGeometry {
    int vec, lpr;
    int64_t tiles;
    bool tiled;
};
inline int gat_bwd_vec(int64_t h, int64_t f, int align, int elem_bytes) {
    auto ok = [&](int vec) {
        return vec * elem_bytes <= 16 && align >= vec * elem_bytes && f % vec == 0 && h * f / vec <= kWave &&
               (h == 1 || pow2(f / vec));
    };
    // fat lanes first: the per-edge attention maths is repeated per lane  (tuning key 4 caps the width: experiments)
    const int cap = g_tuning[kTuneGatVec] > 0 ? g_tuning[kTuneGatVec] : 8;
    for (int vec = 8; vec > 1; vec >>= 1)
        if (vec <= cap && ok(vec)) return vec;
    return (h * f <= kWave && (h == 1 || pow2(f))) ? 1 : 0;
}
inline GatBwdGeometry gat_bwd_geometry(int64_t h, int64_t f, int align, int elem_bytes) {
    GatBwdGeometry g;
    g.vec = gat_bwd_vec(h, f, align, elem_bytes);
    g.tiled = g.vec == 0;
    if (g.tiled) {
        g.vec = 1;
        for (int w = 2; w * elem_bytes <= 16; w <<= 1) {
            if (f % w != 0 || align < w * elem_bytes) break;
            g.vec = w;
        }
    }
    const int64_t need = (h * f + g.vec - 1) / g.vec;
    if (g.tiled) {
        g.lpr = need <= 16 ? 16 : 64;
    } else {
        g.lpr = 8;
        while (g.lpr < kWave && g.lpr < need) g.lpr <<= 1;
    }
    g.tiles = (need + g.lpr - 1) / g.lpr;
    return g;
}
inline size_t gat_stats_bytes(int64_t v, int64_t h) {  // [V, H] float4 records of the row pass
    return ((size_t)(v > 0 ? v : 0) * (size_t)(h > 0 ? h : 0) * sizeof(float4) + 255) / 256 * 256;
}

template <typename T, int VEC, int LPR, bool DROP>
static int gat_launch_bwd(const GatBwdArgs &b, hipStream_t s) {
    const int lph = (b.h == 1) ? LPR : (int)(b.f / VEC);  // one head: reduce over the whole (zero-padded) group
    GatDrop row_drop = b.drop;
    row_drop.eid = nullptr;  // the row pass walks the CSR itself
    GatBwdRowOp<T, VEC, LPR, 4, DROP, false> row_op{b.ar, b.ac, (const T *)b.feat, b.emax, b.esum, (const T *)b.out,
                                                    (const T *)b.gout, b.stats, b.gar, b.slope, (int)b.h, (int)b.f,
                                                    lph, row_drop, GatTiles{}};
    if (b.vr_row) {  // XCD-partitioned plans (both passes): the same functors with 24-bit table offsets
        row_drop.eid = b.vr_row->eid;  // (plan position -> CSR position)
        GatDrop col_drop = b.drop;
        col_drop.eid = b.vr_col->eid;  // (plan position of the CSC -> CSR position: the caller composed it with the transpose's perm)
        int rc = COGDL_HIP_EUNSUPPORTED;
        const bool scalars = LPR == 8 && g_tuning[kTuneGatOnline] != 3;  // ChunkScalars (see GatFwdOp)
        if constexpr (LPR == 8) {
            if (scalars && b.h == 1) {
                GatBwdRowOp<T, VEC, LPR, 4, DROP, false, true, 1> row24{b.ar, b.ac, (const T *)b.feat, b.emax, b.esum,
                                                                        (const T *)b.out, (const T *)b.gout, b.stats, b.gar, b.slope,
                                                                        (int)b.h, (int)b.f, lph, row_drop, GatTiles{}};
                rc = launch_rowreduce_vrows(row24, b.vr_row, 1, b.ws_row, b.wsb_row, s);
            }
        }
        if (rc == COGDL_HIP_EUNSUPPORTED) {
            GatBwdRowOp<T, VEC, LPR, 4, DROP, false, true> row24{b.ar, b.ac, (const T *)b.feat, b.emax, b.esum, (const T *)b.out,
                                                                 (const T *)b.gout, b.stats, b.gar, b.slope, (int)b.h, (int)b.f,
                                                                 lph, row_drop, GatTiles{}};
            rc = launch_rowreduce_vrows(row24, b.vr_row, 1, b.ws_row, b.wsb_row, s);
        }
        if (rc != COGDL_HIP_OK) return rc;
        if constexpr (LPR == 8) {
            if (scalars && b.h == 1) {
                GatBwdColOp<T, VEC, LPR, 4, DROP, false, true, 4> col24{b.ar, b.ac, (const T *)b.feat, b.stats, (const T *)b.gout,
                                                                        (T *)b.gfeat, b.gac, b.slope, (int)b.h, (int)b.f, lph,
                                                                        col_drop, GatTiles{}};
                return launch_rowreduce_vrows(col24, b.vr_col, 1, b.ws_col, b.wsb_col, s);
            }
        }
        GatBwdColOp<T, VEC, LPR, 4, DROP, false, true> col24{b.ar, b.ac, (const T *)b.feat, b.stats, (const T *)b.gout,
                                                             (T *)b.gfeat, b.gac, b.slope, (int)b.h, (int)b.f, lph, col_drop,
                                                             GatTiles{}};
        return launch_rowreduce_vrows(col24, b.vr_col, 1, b.ws_col, b.wsb_col, s);
    }
    int rc = launch_rowreduce(row_op, b.rowptr, b.colind, b.v, b.nnz, 1, b.ws_row, b.wsb_row, s);
    if (rc != COGDL_HIP_OK) return rc;
    GatBwdColOp<T, VEC, LPR, 4, DROP, false> col_op{b.ar, b.ac, (const T *)b.feat, b.stats,
                                                    (const T *)b.gout, (T *)b.gfeat, b.gac, b.slope, (int)b.h,
                                                    (int)b.f, lph, b.drop, GatTiles{}};
    return launch_rowreduce(col_op, b.colptr, b.rowind, b.n_src, b.nnz, 1, b.ws_col, b.wsb_col, s);
}

template <typename T, int VEC, bool DROP>
static int gat_dispatch_bwd(const GatBwdArgs &b, int lpr, hipStream_t s) {
    switch (lpr) {
        case 8: return gat_launch_bwd<T, VEC, 8, DROP>(b, s);
        case 16: return gat_launch_bwd<T, VEC, 16, DROP>(b, s);
        case 32: return gat_launch_bwd<T, VEC, 32, DROP>(b, s);
        default: return gat_launch_bwd<T, VEC, 64, DROP>(b, s);
    }
}

template <typename T, bool DROP>
static int gat_bwd_typed(const GatBwdArgs &b, const GatBwdGeometry &g, hipStream_t s) {
    switch (g.vec) {
        case 8:
            if constexpr (sizeof(T) == 2) return gat_dispatch_bwd<T, 8, DROP>(b, g.lpr, s);
            return COGDL_HIP_EINVAL;
        case 4: return gat_dispatch_bwd<T, 4, DROP>(b, g.lpr, s);
        case 2: return gat_dispatch_bwd<T, 2, DROP>(b, g.lpr, s);
        default: return gat_dispatch_bwd<T, 1, DROP>(b, g.lpr, s);
    }
}

template <bool DROP>
static int gat_bwd_any(const GatBwdArgs &b, const GatBwdGeometry &g, int dtype, hipStream_t s) {
    switch (dtype) {
        case COGDL_HIP_F32: return gat_bwd_typed<float, DROP>(b, g, s);
        case COGDL_HIP_F16: return gat_bwd_typed<__half, DROP>(b, g, s);
        default: return gat_bwd_typed<__hip_bfloat16, DROP>(b, g, s);
    }
}

// Implemented in gat_drop.hip / gat_tiled.hip (one translation unit per family keeps the build parallel).
int gat_fwd_drop(const GatFwdArgs &a, int dtype, void *ws, size_t wsb, hipStream_t s);
int gat_bwd_drop(const GatBwdArgs &b, const GatBwdGeometry &g, int dtype, hipStream_t s);
int gat_bwd_tiled(const GatBwdArgs &b, const GatBwdGeometry &g, int dtype, bool drop, hipStream_t s);

}  // namespace cogdl
