// rowreduce.h -- the shared execution engine of the row-wise CSR operators (csr_spmm, mhspmm, fused GAT forward
// and both backward passes, scatter_max, sddmm, mhsddmm) on gfx950.
//
// Every one of those operators walks the edges of a CSR row, gathers one dense row per edge and folds it into a
// small per-row state (or emits one value per edge).  The engine owns the traversal; an `Op` functor owns the
// arithmetic.  Two launches (one for per-edge operators):
//
//   rowreduce_main_kernel      grid = [long-row workgroups | row-block workgroups] x column tiles.
//     row blocks               one group of Op::LPR lanes per row ("row group"), 64/LPR rows per wave, 4 waves per
//                              workgroup; colind (and per-edge scalars) are read in coalesced chunks of LPR edges and
//                              broadcast lane-to-lane; gathers are issued Op::UNROLL at a time; the state is updated
//                              strictly in CSR edge order.  Rows longer than `thresh` edges are skipped.  When several
//                              rows share a wave, a skewed workgroup deals its rows to the lane groups by decreasing
//                              length (deal_rows_by_length): same per-row arithmetic, better-filled waves.
//     long-row workgroups      power-law graphs: a row of 10^4..10^5 edges would serialise one lane group for longer
//                              than the rest of the launch takes.  Edges are cut into aligned chunks of `thresh` edges;
//                              every piece (long row  x  chunk) is processed by a whole workgroup: its 256/LPR groups
//                              take contiguous slices, group states are merged in group order through LDS and the
//                              piece's state is written to a workspace record.  Ops without a per-row state (sddmm)
//                              stop here.  Each workgroup owns a run of chunks and FINDS its pieces itself by binary
//                              searching rowptr (build_chunk_table) -- no pre-pass, no table in memory, no dependency
//                              on the row blocks, hence the same launch.
//   rowreduce_combine_kernel   per long row: merge the piece records in chunk order, finish the row.
//
// No atomics anywhere, every merge order is fixed => results are deterministic, and rows of at most `thresh` edges
// are reduced in exactly the reference's (sequential CSR) order.
//
// Op concept (all methods __device__, const; see spmm.hip for the canonical example):
//   constants  LPR, UNROLL, kRec (floats per lane of a piece record), kReduce (has a per-row state to merge)
//   types      Ctx (per-thread constants: column mapping, per-row operands), State, LaneVals, Batch
//   Ctx  make_ctx(l, tile)                       column mapping of lane l for column tile `tile` (blockIdx.y)
//   void row_load(Ctx&, row, ok)                 per-row operands; called by ALL lanes of the group (may shuffle)
//   void init(Ctx, State&, row, ok)              start state of a whole row (may read the existing output)
//   void init_zero(State&)                       start state of a piece
//   void lane_load(Ctx, LaneVals&, e)            lane-parallel per-edge scalar of edge e (coalesced), e.g. the weight
//   void fetch(Ctx, Batch&, u, col, e, LaneVals, sub, jj)   issue the loads of edge e (column col) into slot u
//   void apply(Ctx, State&, Batch, u, valid, e, jpos)       fold slot u (edge e, position jpos in its LPR-chunk)
//   constant   kLds: floats of LDS scratch per lane (0: none); the group's LPR*kLds floats are passed to chunk_begin
//   void chunk_begin(Ctx&, State&, base, cnt, my_c, sub, l, lds, lv)   before the gathers of each LPR-chunk of edges
//                                                 (fused GAT: the chunk's softmax weights, once per (edge, head))
//   void batch_end(Ctx, State&, base, j, cnt)    after the UNROLL applies of one batch (per-edge outputs: one joint
//                                                 cross-lane reduction of the batch's partial dot products)
//   void chunk_end(Ctx, State&, base, cnt)       after each LPR-chunk of edges (per-edge outputs: coalesced store)
//   void row_end(Ctx, State, row, ok)            write the finished row
//   void pack(State, float(&)[kRec]) / unpack / merge(Ctx, State& a, State b)   (kReduce only; b's edges follow a's;
//                                                 called by all lanes of a group together: may shuffle)
#pragma once
#include <algorithm>
#include <climits>
#include <type_traits>
#include <utility>

#include "common.h"

namespace cogdl {

struct LongRows {
    int thresh;              // rows with more edges take the chunk-parallel path; also the chunk size (INT_MAX: off)
    float *partial;          // [n_chunks][2][rec_stride] fp32 piece records (kReduce ops)
    int32_t *found;          // [n_long_blocks]: did that long-row workgroup find a piece?  (written by every one of
                             //   them on every launch; lets the combine kernel leave without searching)
    int64_t n_chunks;
    int64_t rec_stride;      // tiles * kRec * LPR
    int chunks_per_block;    // a long-row workgroup owns a contiguous run of <= kMaxChunksPerBlock chunks
    unsigned n_long_blocks;  // leading workgroups of the main grid that process long-row pieces (multiple of 8)
    int64_t nnz;
};
constexpr int kMaxChunksPerBlock = 255;

// Lanes per row / vector width / column tiles chosen for a row of k columns (see spmm.hip: spmm_geometry).
struct RowGeometry {
    int vec, lpr;
    int64_t tiles;
};
RowGeometry spmm_geometry(int64_t k, int64_t unit, int elem_bytes, int align, bool narrow_groups = false, bool wide16 = false);

struct RowSched {
    const int32_t *rowptr;
    const int32_t *colind;
    int64_t m;
    XcdMap rowblocks;
    LongRows lr;
    int sort_rows;  // deal a workgroup's rows to its lane groups by decreasing length (see rowreduce_main_kernel)
    int debug;      // timing experiments (tuning key 13; WRONG results): 1 = row blocks exit, 2 = long-row workgroups exit
    const int32_t *order;  // NULL, or a permutation of the rows: row block b walks rows order[b * GPB ..] (a plan-time schedule,
                           // e.g. rows by degree inside windows: the lane groups of a wave then carry rows of one length; every
                           // row is still reduced by one group in CSR order and written to its own place -- same results)
};

// Operators may ask for a register budget: `static constexpr int kMinWaves = W` compiles their main kernel for at least W
// waves per SIMD (512 / W VGPRs per lane).
template <class Op, class = void>
struct MinWaves {
    static constexpr int value = 1;
};
template <class Op>
struct MinWaves<Op, std::void_t<decltype(Op::kMinWaves)>> {
    static constexpr int value = Op::kMinWaves;
};

// Threshold above which a row is split.  The sequential time of a row of T edges (~T/UNROLL gather round trips of
// ~1 us) must stay a small fraction of the launch; small graphs need a low threshold, large ones amortise more.
inline int pick_long_thresh(int64_t nnz) {
    if (g_tuning[kTuneLongThresh] > 0) return g_tuning[kTuneLongThresh];
    int t = 128;
    while (t < 1024 && (int64_t)t * 32768 < nnz) t <<= 1;
    // Launches of 2^28 edges and more (the row segments of a 64-bit CSR, csrc/bigcsr.hip; a papers100M shard): 512.  Round 5,
    // symmetrised papers100M-shaped graph (3.2e9 edges, 7 segments, F = 128 fp32) on the MI355X, forward ms by threshold:
    // 256: 347, 512: 345, 1024: 384, 2048: 474, 4096: 460, 16384: 571 (profiles/r05_papers_threshold.txt) -- a row of
    // ~1000 edges is 125 dependent gather batches of ONE wave; by the time such rows are common (hub-heavy graphs at this
    // scale) four waves on contiguous slices finish them sooner than the wave's neighbours can hide them.
    if (nnz >= ((int64_t)1 << 28)) t = 512;
    return t;
}
inline int64_t n_chunks_for(int64_t nnz, int thresh) { return (nnz + thresh - 1) / thresh; }
// Workspace of an operator whose piece record holds `rec_stride` floats.
constexpr size_t kFoundBytes = 8192;  // found[] of <= 2048 long-row workgroups, in front of the records
inline size_t rowreduce_workspace_bytes(int64_t nnz, int64_t rec_stride, int thresh_scale = 1) {
    if (nnz <= 0 || rec_stride <= 0) return 0;
    const int64_t n_chunks = n_chunks_for(nnz, pick_long_thresh(nnz) * thresh_scale);
    return kFoundBytes + ((size_t)n_chunks * 2 * (size_t)rec_stride * sizeof(float) + 255) / 256 * 256;
}
// Fill the scheduling fields of the long-row path (everything but `partial`/`rec_stride`).
inline void plan_long_rows(LongRows &lr, int64_t nnz, int thresh_scale = 1) {
    lr.thresh = pick_long_thresh(nnz) * thresh_scale;
    lr.n_chunks = n_chunks_for(nnz, lr.thresh);
    lr.nnz = nnz;
    const int64_t max_wg = std::min(2040, std::max(64, g_tuning[kTuneLongGrid]));
    lr.chunks_per_block = (int)std::min<int64_t>(kMaxChunksPerBlock, (lr.n_chunks + max_wg - 1) / max_wg);
    const int64_t blocks = (lr.n_chunks + lr.chunks_per_block - 1) / lr.chunks_per_block;
    lr.n_long_blocks = (unsigned)((blocks + kXcds - 1) / kXcds * kXcds);  // keeps block % 8 == XCD for the row blocks
}

// The row that contains edge e (0 <= e < nnz): rowptr[r] <= e < rowptr[r+1].
__device__ __forceinline__ int row_of_edge(const int32_t *__restrict__ rowptr, int64_t m, int e) {
    int lo = 0, hi = (int)m;  // invariant: rowptr[lo] <= e < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = (int)(((int64_t)lo + hi) >> 1);
        if (rowptr[mid] <= e) lo = mid;
        else hi = mid;
    }
    return lo;
}

// Bookkeeping without a pre-pass, a table in memory or a second launch: chunk c = edges [c*T, (c+1)*T).  A long row
// (> T edges) intersecting chunk c is either the row that owns the chunk's FIRST edge ("slot 0") or a row that
// starts inside the chunk -- which, being longer than T, then owns the first edge of chunk c+1 ("slot 1").  So a
// workgroup that owns the chunks [c_begin, c_begin + n) finds all its pieces from n+1 binary searches of rowptr
// (one per thread, concurrently): tbl[t] = the row containing edge (c_begin+t)*T if that row is long, else -1.
// Returns (block-wide) whether any entry is a long row.
__device__ __forceinline__ bool build_chunk_table(const LongRows &lr, const int32_t *__restrict__ rowptr, int64_t m,
                                                  int64_t c_begin, int n_entries, int32_t *tbl) {
    int32_t mine = -1;
    if ((int)threadIdx.x < n_entries) {
        const int64_t e = (c_begin + threadIdx.x) * lr.thresh;
        if (e < lr.nnz) {
            const int r = row_of_edge(rowptr, m, (int)e);
            if (rowptr[r + 1] - rowptr[r] > lr.thresh) mine = r;
        }
        tbl[threadIdx.x] = mine;
    }
    return __syncthreads_or(mine >= 0) != 0;
}

// Piece (t-th chunk of the block's run, slot) -> the long row it belongs to and its edge range [lo, hi).
__device__ __forceinline__ bool decode_piece(const LongRows &lr, const int32_t *__restrict__ rowptr,
                                             const int32_t *tbl, int64_t c, int t, int slot, int32_t &row, int &lo,
                                             int &hi) {
    row = (slot == 0) ? tbl[t] : ((tbl[t + 1] != tbl[t]) ? tbl[t + 1] : -1);
    if (row < 0) return false;
    lo = max(rowptr[row], (int)(c * lr.thresh));
    hi = (int)min((int64_t)rowptr[row + 1], (c + 1) * lr.thresh);
    return lo < hi;
}

// Lane jj of the caller's group of LPR lanes, broadcast to the group.  Round 6: on hub-heavy graphs the row-reduce kernels are
// bound by VALU issue (profiles/r06_sq_reddit.txt), and a ds_bpermute costs four vector instructions for its address
// (clamp, add the group's base, wrap, scale) per broadcast.  reduce_chunk unrolls a chunk completely, so jj is a compile-time
// constant wherever it matters, and the broadcast is an instruction with an IMMEDIATE pattern: v_readlane (LPR = 64), a DPP
// row_share move (16: a DPP row IS 16 lanes) or quad_perm move (4), a ds_swizzle in bit-mask mode (8, 32: lane' = (lane &
// ~(LPR - 1)) | jj inside each half wave) -- no address arithmetic, and no LDS traffic for 4 / 16 / 64.  The `switch` folds
// away under constant propagation; a run-time jj still works (through a jump table: slow, not wrong).  Lane groups that do
// not tile a wave (10 / 12 / 20 lanes) keep ds_bpermute.
template <int LPR, int JJ>
__device__ __forceinline__ int group_bcast_imm(int v) {
    static_assert(JJ >= 0 && JJ < LPR, "lane inside the group");
    // (bound_ctrl: the source lane of a row_share / quad_perm is always a lane of the caller's own group, i.e. active whenever
    //  the destination is -- the DPP move needs no "old" value, and with bound_ctrl the compiler does not materialise one: a
    //  v_mov 0 in front of EVERY broadcast otherwise, two per edge in csr_spmm's batches.)
    if constexpr (LPR == 16) return __builtin_amdgcn_update_dpp(0, v, 0x150 + JJ, 0xf, 0xf, true);  // row_share:JJ
    else if constexpr (LPR == 4) return __builtin_amdgcn_update_dpp(0, v, JJ * 0x55, 0xf, 0xf, true);  // quad_perm:[JJ,JJ,JJ,JJ]
    else return __builtin_amdgcn_ds_swizzle(v, (JJ << 5) | (0x1f & ~(LPR - 1)));  // bit mode: and_mask | or_mask << 5
}
template <int LPR, int... JJ>
__device__ __forceinline__ int group_bcast_switch(int v, int jj, std::integer_sequence<int, JJ...>) {
    int r = v;
    (void)((jj == JJ ? (r = group_bcast_imm<LPR, JJ>(v), true) : false) || ...);
    return r;
}
template <int LPR>
__device__ __forceinline__ int group_bcast(int v, int sub, int jj) {
    if constexpr (LPR == kWave) return __builtin_amdgcn_readlane(v, jj);  // -> SGPR: scalar address arithmetic
    else if constexpr (LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32) {
        if (__builtin_constant_p(jj)) return group_bcast_switch<LPR>(v, jj, std::make_integer_sequence<int, LPR>{});
        return __shfl(v, sub * LPR + jj, kWave);
    } else return __shfl(v, sub * LPR + jj, kWave);
}
template <int LPR>
__device__ __forceinline__ float group_bcast(float v, int sub, int jj) {
    return __int_as_float(group_bcast<LPR>(__float_as_int(v), sub, jj));
}

// Joint reduction of U per-lane partials over an aligned group of W lanes ("transpose reduce"): the first log2(U)
// butterfly stages exchange HALF of the values each (a lane keeps the upper or the lower half by its stride bit), the
// remaining stages reduce the single survivor: U - 1 + log2(W / U) shuffles instead of U * log2(W).  On return the lane
// l (relative to its group) with (l % (W / U)) == 0 holds the total of p[l / (W / U)].
template <int U, int W>
__device__ __forceinline__ float transpose_reduce(const float (&p)[U], int l) {
    static_assert(U <= W && (U & (U - 1)) == 0, "U must be a power of two <= W");
    float v[U];
#pragma unroll
    for (int i = 0; i < U; ++i) v[i] = p[i];
    int n = U;
#pragma unroll
    for (int s = W / 2; s >= W / U && s > 0; s >>= 1) {
        const bool upper = (l & s) != 0;
        n >>= 1;
#pragma unroll
        for (int i = 0; i < U / 2; ++i) {
            if (i < n) {
                const float send = upper ? v[i] : v[i + n];
                const float keep = upper ? v[i + n] : v[i];
                v[i] = keep + __shfl_xor(send, s, kWave);
            }
        }
    }
#pragma unroll
    for (int s = W / U / 2; s > 0; s >>= 1) v[0] += __shfl_xor(v[0], s, kWave);
    return v[0];
}

// Operators whose lane_load is arithmetic, not a load (the Philox mask of the dropout GAT functors), declare
// `static constexpr bool kExpensiveLaneLoad = true`: reduce_edges then prefetches only the column ids of the next chunk.
template <class Op, class = void>
struct LaneLoadCost {
    static constexpr bool expensive = false;
};
template <class Op>
struct LaneLoadCost<Op, std::void_t<decltype(Op::kExpensiveLaneLoad)>> {
    static constexpr bool expensive = Op::kExpensiveLaneLoad;
};

// Fold ONE chunk of cnt <= LPR edges (base .. base + cnt) whose column ids / per-edge scalars the lanes already hold.
template <class Op>
__device__ __forceinline__ void reduce_chunk(const Op &op, typename Op::Ctx &ctx, typename Op::State &st, int base, int cnt,
                                             int my_c, const typename Op::LaneVals &lv, int sub, int l, float *lds) {
    constexpr int LPR = Op::LPR, UNROLL = Op::UNROLL;
    op.chunk_begin(ctx, st, base, cnt, my_c, sub, l, lds, lv);
    // Lane groups that tile a wave and are narrower than it: the chunk is unrolled completely (LPR / UNROLL <= 4 batches, the
    // later ones guarded), so that the position jj of a slot inside the chunk is a compile-time constant -- which is what
    // lets group_bcast use immediate lane patterns.  Whole-wave groups (v_readlane takes a scalar lane anyway; eight
    // unrolled batches only cost instruction-cache misses: fused GAT forward H = 1 x F = 41 in 2-byte lanes 4.4 -> 6.8 ms)
    // and the groups of 10 / 12 / 20 lanes (ds_bpermute) keep the loop.
    constexpr bool kUnrolled = LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32;
    if constexpr (kUnrolled) {
        constexpr int NB = (LPR + UNROLL - 1) / UNROLL;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int j = jb * UNROLL;
            if (j < cnt) {  // (group-uniform)
                typename Op::Batch b;
                // Issue all UNROLL gathers back to back (no branches: a slot past the chunk's end re-reads the row's last
                // valid neighbour -- the lanes past the end hold ITS column id and per-edge scalars, their loads were
                // clamped into the row -- an L1 hit, and is ignored by apply()).
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int jj = (j + u < LPR) ? j + u : LPR - 1;  // (compile-time)
                    op.fetch(ctx, b, u, group_bcast<LPR>(my_c, sub, jj), base + min(jj, cnt - 1), lv, sub, jj);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) op.apply(ctx, st, b, u, (j + u) < cnt, base + j + u, j + u);
                op.batch_end(ctx, st, base, j, cnt);
            }
        }
    } else {
        for (int j = 0; j < cnt; j += UNROLL) {
            typename Op::Batch b;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int jj = min(j + u, cnt - 1);
                op.fetch(ctx, b, u, group_bcast<LPR>(my_c, sub, jj), base + jj, lv, sub, jj);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) op.apply(ctx, st, b, u, (j + u) < cnt, base + j + u, j + u);
            op.batch_end(ctx, st, base, j, cnt);
        }
    }
    op.chunk_end(ctx, st, base, cnt);
}

// Fold edges [start, end) of one row into st, in order.  All lanes of the group execute it.
// The column ids / per-edge scalars of chunk k+1 are requested BEFORE the gathers of chunk k are issued (round 4): a row of
// more than LPR edges used to pay three dependent round trips per chunk (ids, then two gather batches), i.e. a third of
// a medium row's life was the bubble in front of its gathers -- on the R-MAT graphs 80 % of the edges sit in such rows.
// The loads are unconditional (index clamped into the row): a load inside a per-lane branch makes the compiler park the
// wave at the join (lesson 1), and a lane past the chunk's end is never read by fetch().
template <class Op>
__device__ __forceinline__ void reduce_edges(const Op &op, typename Op::Ctx &ctx, typename Op::State &st,
                                             const int32_t *__restrict__ colind, int start, int end, int sub, int l,
                                             float *lds) {
    constexpr int LPR = Op::LPR;
    constexpr bool kPrefetchVals = !LaneLoadCost<Op>::expensive;
    if (start >= end) return;
    const int last = end - 1;
    int my_c = colind[min(start + l, last)];
    typename Op::LaneVals lv{};
    if constexpr (kPrefetchVals) op.lane_load(ctx, lv, min(start + l, last));
    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        const int nidx = min(base + LPR + l, last);  // (past the row's end: its last edge again, an L1 hit, never used)
        const int next_c = colind[nidx];
        typename Op::LaneVals next_lv{};
        if constexpr (kPrefetchVals) op.lane_load(ctx, next_lv, nidx);
        else op.lane_load(ctx, lv, min(base + l, last));
        reduce_chunk<Op>(op, ctx, st, base, cnt, my_c, lv, sub, l, lds);
        my_c = next_c;
        if constexpr (kPrefetchVals) lv = next_lv;
    }
}

// One workgroup per run of `chunks_per_block` chunks: the leading blocks of the main grid.
//
// Round 4: the run is walked as ONE stream per lane group.  A hub row of 10^4 edges spans ~80 chunks and a workgroup's run
// of ~19 chunks mostly belongs to one row, so a group keeps ONE state across all pieces of the same row (group g takes
// slice g of every piece) and the 16 group states are merged -- LDS, two barriers, a sequential merge by group 0 -- once
// per (row, run) instead of once per 128-edge piece; the pieces of the run that carry no state of their own get the
// operator's identity record, so the combine kernel is unchanged.  And the column ids / per-edge scalars of the NEXT
// piece's slice are loaded before the gathers of the current one are issued: a piece costs one memory round trip, not
// two plus two barriers (measured: the long-row workgroups alone took 99 us for the 1.2 M hub edges of the arxiv-sized
// R-MAT graph -- 19 pieces x ~5 us per workgroup, a serial chain -- against 51 us at the gather rate of the row blocks).
// Deterministic as before: a long row's summation order is fixed by (chunk size, group count), not by timing.
template <class Op>
__device__ __forceinline__ void rowreduce_long_block(const Op &op, const RowSched &s, float *op_lds) {
    constexpr int LPR = Op::LPR;
    constexpr int RPW = kWave / LPR;   // lane groups per wave; LPR need not divide 64 (e.g. 10 lanes for a 160-byte
    constexpr int G = RPW * 4;         // row): the last 64 % LPR lanes of every wave then idle ("lane_on" below)
    constexpr int NREC = Op::kReduce ? Op::kRec : 1;
    __shared__ float red[Op::kReduce ? G : 1][NREC][LPR];
    __shared__ int32_t tbl[kMaxChunksPerBlock + 1];
    const LongRows &lr = s.lr;
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    if (c_begin >= lr.n_chunks) return;
    const int n = (int)min((int64_t)lr.chunks_per_block, lr.n_chunks - c_begin);
    const bool any = build_chunk_table(lr, s.rowptr, s.m, c_begin, n + 1, tbl);
    if (lr.found && threadIdx.x == 0 && blockIdx.y == 0) lr.found[blockIdx.x] = any ? 1 : 0;
    if (!any) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const bool lane_on = (kWave % LPR == 0) || sub < RPW;
    const int g = (threadIdx.x >> 6) * RPW + sub;
    float *const my_lds = op_lds + (threadIdx.x / LPR) * LPR * Op::kLds;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);

    // piece index p = 2 * t + slot (t-th chunk of the run); this group's slice of it
    struct Piece {
        int p, sb, se;
        int32_t row;
    };
    auto find_piece = [&](int from) {  // the first valid piece at or after index `from` (workgroup-uniform)
        Piece q{2 * n, 0, 0, -1};
        for (int p = from; p < 2 * n; ++p) {
            int32_t row;
            int lo, hi;
            if (!decode_piece(lr, s.rowptr, tbl, c_begin + (p >> 1), p >> 1, p & 1, row, lo, hi)) continue;
            const int per = (hi - lo + G - 1) / G;  // contiguous slices
            int sb = min(lo + g * per, hi), se = min(sb + per, hi);
            if (!lane_on) sb = se = hi;
            if constexpr (LPR == kWave) {  // one wave per slice: keep the loop bounds scalar
                sb = __builtin_amdgcn_readfirstlane(sb);
                se = __builtin_amdgcn_readfirstlane(se);
            }
            q = Piece{p, sb, se, row};
            break;
        }
        return q;
    };
    auto load_first_chunk = [&](const Piece &q, int &c, typename Op::LaneVals &lv) {
        c = 0;
        lv = typename Op::LaneVals{};
        if (q.sb < q.se) {  // (group-uniform; the loads themselves are unconditional: index clamped into the slice)
            const int idx = min(q.sb + l, q.se - 1);
            c = s.colind[idx];
            op.lane_load(ctx, lv, idx);
        }
    };
    auto record_of = [&](int p) {
        return lr.partial + (2 * (c_begin + (p >> 1)) + (p & 1)) * lr.rec_stride + (int64_t)blockIdx.y * NREC * LPR;
    };

    typename Op::State st;
    op.init_zero(st);
    int32_t cur_row = -1;
    int cur_p = -1;
    auto flush = [&]() {  // merge the G group states of the row in hand (group order) into the record of its first piece here
        if constexpr (Op::kReduce) {
            if (cur_row < 0) return;
            float rec[NREC];
            op.pack(st, rec);
            if (lane_on) {
#pragma unroll
                for (int i = 0; i < NREC; ++i) red[g][i][l] = rec[i];
            }
            __syncthreads();
            if (g == 0) {
                for (int q = 1; q < G; ++q) {  // fixed order: group 0, 1, 2, ...
#pragma unroll
                    for (int i = 0; i < NREC; ++i) rec[i] = red[q][i][l];
                    typename Op::State other;
                    op.unpack(other, rec);
                    op.merge(ctx, st, other);
                }
                op.pack(st, rec);
                float *dst = record_of(cur_p);
#pragma unroll
                for (int i = 0; i < NREC; ++i) dst[i * LPR + l] = rec[i];
            }
            __syncthreads();
        }
    };

    Piece cur = find_piece(0);
    int pc;
    typename Op::LaneVals plv;
    if (cur.row >= 0) load_first_chunk(cur, pc, plv);
    while (cur.row >= 0) {
        const Piece nxt = find_piece(cur.p + 1);
        const int my_c = pc;
        const typename Op::LaneVals lv = plv;
        if (nxt.row >= 0) load_first_chunk(nxt, pc, plv);  // in flight while this piece gathers
        if (cur.row != cur_row) {
            flush();
            cur_row = cur.row;
            cur_p = cur.p;
            op.row_load(ctx, cur.row, true);
            op.init_zero(st);
        } else if constexpr (Op::kReduce) {
            if (g == 0) {  // this piece's record: the operator's identity (its edges are in the state carried on)
                typename Op::State zero;
                op.init_zero(zero);
                float rec[NREC];
                op.pack(zero, rec);
                float *dst = record_of(cur.p);
#pragma unroll
                for (int i = 0; i < NREC; ++i) dst[i * LPR + l] = rec[i];
            }
        }
        if (cur.sb < cur.se) {
            reduce_chunk<Op>(op, ctx, st, cur.sb, min(LPR, cur.se - cur.sb), my_c, lv, sub, l, my_lds);
            if (cur.sb + LPR < cur.se) reduce_edges<Op>(op, ctx, st, s.colind, cur.sb + LPR, cur.se, sub, l, my_lds);
        }
        cur = nxt;
    }
    flush();
}

// Deal the GPB rows of a workgroup to its GPB lane groups by decreasing length -- if the workgroup is skewed (longest
// row well above the mean; the test is workgroup-uniform, uniform graphs pay one barrier and one pass over GPB LDS
// words).  `slot` = this group's natural position, `leader` = one lane per group.  On return (true) `mine` is the
// natural position of the row this group now owns and (ok, start, end) describe that row.  Called by ALL threads.
// A group takes the rank of its own slot: wave 0 gets the longest rows (a wave takes as long as its longest row).
template <int GPB>
__device__ __forceinline__ bool deal_rows_by_length(int slot, bool leader, bool &ok, int &start, int &end, int &mine) {
    __shared__ int sort_deg[GPB], sort_start[GPB], sort_slot[GPB];
    const int deg = ok ? end - start : -1;
    if (leader) {
        sort_deg[slot] = deg;
        sort_start[slot] = start;
    }
    __syncthreads();
    int dmax = 0, dsum = 0;
    for (int q = 0; q < GPB; ++q) {
        const int dq = max(sort_deg[q], 0);
        dmax = max(dmax, dq);
        dsum += dq;
    }
    if (dmax * GPB < 2 * dsum + 8 * GPB) return false;
    int rank = 0;  // by (length descending, natural position ascending)
    for (int q = 0; q < GPB; ++q) {
        const int dq = sort_deg[q];
        rank += (dq > deg || (dq == deg && q < slot)) ? 1 : 0;
    }
    if (leader) sort_slot[rank] = slot;
    __syncthreads();
    mine = sort_slot[slot];
    ok = sort_deg[mine] >= 0;
    start = sort_start[mine];
    end = start + max(sort_deg[mine], 0);
    return true;
}

// tuning key 2: 1 = keep the natural row -> lane-group assignment inside a workgroup (experiments)
__device__ __forceinline__ bool g_sort_rows(const RowSched &s) { return s.sort_rows != 0; }

// Grid: [ n_long_blocks long-row workgroups | row-block workgroups ]  x  column tiles.  The long-row workgroups come
// first so that the (critical-path) hub rows start at once; they cost a graph without hub rows ~17 dependent
// L2-resident loads in <= 1024 workgroups, overlapped with the row blocks.
template <class Op>
__global__ __launch_bounds__(256, MinWaves<Op>::value) void rowreduce_main_kernel(const Op op, const RowSched s) {
    __shared__ float op_lds[Op::kLds > 0 ? 256 * Op::kLds : 1];  // one buffer for both kinds of workgroup
    if (blockIdx.x < s.lr.n_long_blocks) {
        if (s.debug != 2) rowreduce_long_block<Op>(op, s, op_lds);
        return;
    }
    if (s.debug == 1) return;
    constexpr int LPR = Op::LPR;
    constexpr int RPW = kWave / LPR;  // row groups per wave
    constexpr int GPB = RPW * 4;      // row groups per 256-thread workgroup
    const int64_t rb = xcd_remap(blockIdx.x - s.lr.n_long_blocks, s.rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int l = lane % LPR;
    const bool lane_on = (kWave % LPR == 0) || lane / LPR < RPW;  // LPR not dividing 64: the trailing lanes idle
    const int sub = lane_on ? lane / LPR : 0;
    int64_t row = rb * GPB + wave * RPW + sub;
    bool ok = lane_on && row < s.m;
    int start = 0, end = 0;
    if (ok) {
        if (s.order) row = s.order[row];
        start = s.rowptr[row];
        end = s.rowptr[row + 1];
    }
    if constexpr (RPW > 1) {
        // Several rows share a wave, and a wave takes as long as its LONGEST row: on a skewed graph a wave with one
        // 100-edge row and three 5-edge rows idles 70 % of its lanes.  The workgroup's GPB rows are therefore dealt to
        // the row groups in order of decreasing length (ranks by counting through LDS, ties by row id): wave 0 gets the
        // RPW longest rows, wave 3 the shortest, the sum over waves of their longest row -- the lane time spent --
        // drops by 2-3x on R-MAT graphs.  Only the row -> lane-group assignment changes: every row is still reduced
        // sequentially by one group in CSR order (bit-identical results).
        if (g_sort_rows(s)) {
            int mine;
            if (deal_rows_by_length<GPB>(wave * RPW + sub, lane_on && l == 0, ok, start, end, mine)) {
                row = rb * GPB + mine;
                if (s.order && ok) row = s.order[row];
            }
            if (!lane_on) {
                ok = false;
                start = end = 0;
            }
        }
    }
    if constexpr (LPR == kWave) {  // whole wave on one row: make the loop bounds scalar
        start = __builtin_amdgcn_readfirstlane(start);
        end = __builtin_amdgcn_readfirstlane(end);
    }
    const bool is_long = end - start > s.lr.thresh;  // long row: the long-row workgroups compute it
    float *const my_lds = op_lds + (threadIdx.x / LPR) * LPR * Op::kLds;
    if (is_long) return;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    op.row_load(ctx, row, ok);
    typename Op::State st;
    op.init(ctx, st, row, ok);
    reduce_edges<Op>(op, ctx, st, s.colind, start, end, sub, l, my_lds);
    op.row_end(ctx, st, row, ok);
}

// For every long row merge its piece records in chunk order and finish the row.  The row is combined by the lane
// group that finds it at its FIRST full chunk (the row's head piece, if any, sits in slot 1 of the chunk before).
template <class Op>
__global__ __launch_bounds__(256) void rowreduce_combine_kernel(const Op op, const RowSched s) {
    constexpr int LPR = Op::LPR;
    constexpr int RPW = kWave / LPR;
    constexpr int G = RPW * 4;
    constexpr int NREC = Op::kRec;
    __shared__ int32_t tbl[kMaxChunksPerBlock + 1];
    const LongRows &lr = s.lr;
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    if (c_begin >= lr.n_chunks) return;
    // A row combined here owns the first edge of one of this run's chunks, so the long-row workgroup of the same run
    // met it (slot 0): found[] == 0 means nothing to do -- one load instead of a search.
    if (lr.found && lr.found[blockIdx.x] == 0) return;
    const int n = (int)min((int64_t)lr.chunks_per_block, lr.n_chunks - c_begin);
    if (!build_chunk_table(lr, s.rowptr, s.m, c_begin, n, tbl)) return;
    const int ch = lr.thresh;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const bool lane_on = (kWave % LPR == 0) || sub < RPW;
    const int g = (threadIdx.x >> 6) * RPW + sub;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    for (int t0 = 0; t0 < n; t0 += G) {
        const int t = t0 + g;
        const int64_t c = c_begin + t;
        int32_t row = (lane_on && t < n) ? tbl[t] : -1;
        int start = 0, end = 0;
        if (row >= 0) {
            start = s.rowptr[row];
            end = s.rowptr[row + 1];
            if ((int64_t)(start + ch - 1) / ch != c) row = -1;  // not the row's first owned chunk
        }
        if (row < 0) continue;  // (group-uniform; the shuffles of row_load are group-local)
        op.row_load(ctx, row, true);
        typename Op::State st;
        op.init(ctx, st, row, true);
        const int64_t c_last = (end - 1) / ch;
        const int64_t tile_off = (int64_t)blockIdx.y * NREC * LPR + l;
        float rec[NREC];
        typename Op::State piece;
        if ((int64_t)start != c * ch) {  // a head piece lives in chunk c-1, slot 1
            const float *src = lr.partial + (2 * (c - 1) + 1) * lr.rec_stride + tile_off;
#pragma unroll
            for (int i = 0; i < NREC; ++i) rec[i] = src[i * LPR];
            op.unpack(piece, rec);
            op.merge(ctx, st, piece);
        }
        for (int64_t q = c; q <= c_last; ++q) {
            const float *src = lr.partial + (2 * q) * lr.rec_stride + tile_off;
#pragma unroll
            for (int i = 0; i < NREC; ++i) rec[i] = src[i * LPR];
            op.unpack(piece, rec);
            op.merge(ctx, st, piece);
        }
        op.row_end(ctx, st, row, true);
    }
}

// ---- XCD-partitioned columns (round 6) -------------------------------------------------------------------------------
// Every XCD of the MI355X has a private 4 MiB L2.  A gathered table of tens of MB (Reddit-shaped graph: 233 k rows of 128 /
// 256 bytes) fits none of them, so a hub-heavy launch re-fetches it from the Infinity Cache at the fabric rate (PMC: 11-16 GB
// per launch for 0.5 GB compulsory) -- but it does fit the eight L2s TOGETHER.  A plan (cogdl_hip_vrows, built once per
// structure: cogdl_amd/xcdplan.py) therefore gives every column an owner XCD by a hash of its id, cuts each long row into the
// eight sub-rows of its edges by owner (CSR order inside a sub-row, pieces of at most T edges) and lays these VIRTUAL rows out
// so that a workgroup on XCD x only ever holds virtual rows whose columns XCD x owns: an XCD's L2 then sees one eighth of the
// table.  Short rows stay whole.  A virtual row of a multi-part row leaves its state as a record; rowreduce_vcombine_kernel
// merges a row's records in a fixed order (owner XCD, then piece) -- deterministic, re-associated like the long-row path.
// Measured on the Reddit-shaped graph before the kernels existed (tools/xcdpart_probe.py, the layout expressed as a permuted
// CSR on the unchanged csr_spmm): F = 64 fp32 3220 -> 1613 us, bf16 1501 -> 1290 us; the same virtual rows dealt to the XCDs
// regardless of their columns: 2594 / 1379 us.
// Layout: slots in UNITS of kVUnit; unit u belongs to XCD u % 8; workgroup w runs on XCD w % 8 (hardware round-robin) and is
// that XCD's (w / 8)-th workgroup.
constexpr int kVUnit = 64;
struct VRows {
    const int32_t *vrowptr;  // [n_slots + 1] edge offsets of the slots in the permuted edge order
    const int32_t *vcol;     // [nnz] column ids in that order
    const int2 *vdesc;       // [n_slots] {row (-1: padding), record index (-1: the row's only part: finished here)}
    const int32_t *mrow;     // [n_multi] rows with several parts ...
    const int32_t *mptr;     // [n_multi + 1] ... and their records [mptr[i], mptr[i + 1]) in merge order
    const int32_t *big;      // [n_big] indices into mrow of the rows with more than kVBigParts parts
    float *partial;          // [n_parts][rec_stride]
    int64_t n_slots, n_multi, n_parts, n_big, rec_stride;
    int sort_rows;
};

template <class Op>
__global__ __launch_bounds__(256, MinWaves<Op>::value) void rowreduce_vrow_kernel(const Op op, const VRows v) {
    static_assert(Op::kReduce, "virtual rows: operators with a per-row state");
    __shared__ float op_lds[Op::kLds > 0 ? 256 * Op::kLds : 1];
    constexpr int LPR = Op::LPR;
    constexpr int RPW = kWave / LPR;
    constexpr int GPB = RPW * 4;
    constexpr int WPU = kVUnit / GPB;  // workgroups per unit
    static_assert(kWave % LPR == 0 && kVUnit % GPB == 0, "virtual rows: lane groups that tile a wave");
    const unsigned x = blockIdx.x % kXcds, i = blockIdx.x / kXcds;
    const int64_t slot0 = ((int64_t)(i / WPU) * kXcds + x) * kVUnit + (int64_t)(i % WPU) * GPB;
    if (slot0 >= v.n_slots) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int l = lane % LPR, sub = lane / LPR;
    int64_t slot = slot0 + wave * RPW + sub;
    int2 d = make_int2(-1, -1);
    int start = 0, end = 0;
    if (slot < v.n_slots) {
        d = v.vdesc[slot];
        start = v.vrowptr[slot];
        end = v.vrowptr[slot + 1];
    }
    bool ok = d.x >= 0;
    if constexpr (RPW > 1) {
        if (v.sort_rows) {  // (same dealing as the row blocks: a wave takes as long as its longest virtual row)
            int mine;
            if (deal_rows_by_length<GPB>(wave * RPW + sub, l == 0, ok, start, end, mine)) {
                slot = slot0 + mine;
                d = ok ? v.vdesc[slot] : make_int2(-1, -1);
            }
        }
    }
    if constexpr (LPR == kWave) {
        start = __builtin_amdgcn_readfirstlane(start);
        end = __builtin_amdgcn_readfirstlane(end);
    }
    float *const my_lds = op_lds + (threadIdx.x / LPR) * LPR * Op::kLds;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    const int64_t row = d.x;
    const bool whole = d.y < 0;
    op.row_load(ctx, row, ok);
    typename Op::State st;
    if (whole) op.init(ctx, st, row, ok);
    else op.init_zero(st);
    reduce_edges<Op>(op, ctx, st, v.vcol, start, end, sub, l, my_lds);
    if (whole) {
        op.row_end(ctx, st, row, ok);
    } else if (ok) {
        float rec[Op::kRec];
        op.pack(st, rec);
        float *dst = v.partial + (int64_t)d.y * v.rec_stride + (int64_t)blockIdx.y * Op::kRec * LPR;
#pragma unroll
        for (int q = 0; q < Op::kRec; ++q) dst[q * LPR + l] = rec[q];
    }
}

// Merge the part records of every multi-part row in plan order (owner XCD, then piece) and finish the row.  One lane group per
// row -- except the rows of more than kVBigParts parts (hubs: a row of 10^5 edges has ~400), which get a whole workgroup each
// (the leading n_big blocks): its G lane groups merge contiguous slices of the row's records, the G states are merged in group
// order through LDS.  Measured before that split (Reddit-shaped graph, fused GAT forward, 1.2 M records): 266 us for 0.35 GB --
// the launch WAS the serial chain of the longest row (profiles/r06_pmc_gat.json).  The next record is requested before the
// current one is merged.
constexpr int kVBigParts = 32;
template <class Op>
__global__ __launch_bounds__(256) void rowreduce_vcombine_kernel(const Op op, const VRows v) {
    constexpr int LPR = Op::LPR;
    constexpr int RPW = kWave / LPR;
    constexpr int GPB = RPW * 4;
    constexpr int NREC = Op::kRec;
    __shared__ float red[GPB][NREC][LPR];
    const int lane = threadIdx.x & (kWave - 1);
    const int l = lane % LPR, sub = lane / LPR;
    const int g = (threadIdx.x >> 6) * RPW + sub;
    const bool big = (int64_t)blockIdx.x < v.n_big;
    const int64_t idx = big ? (int64_t)v.big[blockIdx.x] : ((int64_t)blockIdx.x - v.n_big) * GPB + g;
    if (idx >= v.n_multi) return;  // (group-uniform; the shuffles of the operator hooks are group-local)
    const int64_t row = v.mrow[idx];
    int p0 = v.mptr[idx], p1 = v.mptr[idx + 1];
    if (!big && p1 - p0 > kVBigParts) return;  // (a whole workgroup has it)
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    op.row_load(ctx, row, true);
    const int64_t tile_off = (int64_t)blockIdx.y * NREC * LPR + l;
    auto load = [&](float(&rec)[NREC], int p) {
        const float *src = v.partial + (int64_t)p * v.rec_stride + tile_off;
#pragma unroll
        for (int q = 0; q < NREC; ++q) rec[q] = src[q * LPR];
    };
    auto merge_range = [&](typename Op::State &st, int a, int b) {
        if (a >= b) return;
        float cur[NREC], nxt[NREC];
        load(cur, a);
#pragma unroll 1
        for (int p = a; p < b; ++p) {
            load(nxt, min(p + 1, b - 1));
            typename Op::State piece;
            op.unpack(piece, cur);
            op.merge(ctx, st, piece);
#pragma unroll
            for (int q = 0; q < NREC; ++q) cur[q] = nxt[q];
        }
    };
    typename Op::State st;
    if (!big) {
        op.init(ctx, st, row, true);
        merge_range(st, p0, p1);
        op.row_end(ctx, st, row, true);
        return;
    }
    const int per = (p1 - p0 + GPB - 1) / GPB;
    const int a = min(p0 + g * per, p1), b = min(a + per, p1);
    op.init_zero(st);
    merge_range(st, a, b);
    float rec[NREC];
    op.pack(st, rec);
#pragma unroll
    for (int q = 0; q < NREC; ++q) red[g][q][l] = rec[q];
    __syncthreads();
    if (g != 0) return;
    op.init(ctx, st, row, true);
#pragma unroll 1  // (unrolled by the compiler this loop cost the 17-float records of the GAT column pass 256 VGPRs + 84 AGPRs)
    for (int q = 0; q < GPB; ++q) {  // fixed order: group 0, 1, 2, ...
#pragma unroll
        for (int i = 0; i < NREC; ++i) rec[i] = red[q][i][l];
        typename Op::State other;
        op.unpack(other, rec);
        op.merge(ctx, st, other);
    }
    op.row_end(ctx, st, row, true);
}

inline size_t vrows_workspace_bytes(int64_t n_parts, int64_t rec_stride) {
    if (n_parts <= 0 || rec_stride <= 0) return 256;
    return ((size_t)n_parts * (size_t)rec_stride * sizeof(float) + 255) / 256 * 256;
}
inline int vrows_valid(const cogdl_hip_vrows *p) {
    if (!p || p->n_slots < 0 || p->n_multi < 0 || p->n_parts < 0 || p->nnz < 0) return COGDL_HIP_EINVAL;
    if (p->n_slots % (kVUnit * kXcds) != 0 || p->n_slots > 0x7fffffff || p->nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (p->n_slots > 0 && (!p->vrowptr || !p->vdesc)) return COGDL_HIP_EINVAL;
    if (p->nnz > 0 && !p->vcol) return COGDL_HIP_EINVAL;
    if (p->n_multi > 0 && (!p->mrow || !p->mptr)) return COGDL_HIP_EINVAL;
    if (p->n_big < 0 || p->n_big > p->n_multi || (p->n_big > 0 && !p->big)) return COGDL_HIP_EINVAL;
    return COGDL_HIP_OK;
}

// The two launches of one operator call over a virtual-row plan.  workspace: vrows_workspace_bytes(n_parts, tiles * kRec * LPR).
template <class Op>
static int launch_rowreduce_vrows(const Op &op, const cogdl_hip_vrows *p, int64_t tiles, void *workspace,
                                  size_t workspace_bytes, hipStream_t stream) {
    int rc = vrows_valid(p);
    if (rc != COGDL_HIP_OK) return rc;
    if (p->n_slots == 0) return COGDL_HIP_OK;
    if (tiles > 65535 || tiles < 1) return COGDL_HIP_ERANGE;
    constexpr int GPB = (kWave / Op::LPR) * 4;
    VRows v{};
    v.vrowptr = p->vrowptr, v.vcol = p->vcol, v.vdesc = (const int2 *)p->vdesc, v.mrow = p->mrow, v.mptr = p->mptr;
    v.n_slots = p->n_slots, v.n_multi = p->n_multi, v.n_parts = p->n_parts;
    v.big = p->big, v.n_big = p->n_big;
    v.rec_stride = tiles * Op::kRec * Op::LPR;
    v.sort_rows = g_tuning[kTuneRowSort] == 0 ? 1 : 0;
    if (p->n_parts > 0) {
        if (!workspace || workspace_bytes < vrows_workspace_bytes(p->n_parts, v.rec_stride)) return COGDL_HIP_EWORKSPACE;
        if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
        v.partial = (float *)workspace;
    }
    const int64_t wgs = p->n_slots / GPB;  // (n_slots is a multiple of 8 units)
    if (wgs > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL((rowreduce_vrow_kernel<Op>), dim3((unsigned)wgs, (unsigned)tiles), dim3(256), 0, stream, op, v);
    if (p->n_multi > 0)
        hipLaunchKernelGGL((rowreduce_vcombine_kernel<Op>),
                           dim3((unsigned)(p->n_big + (p->n_multi + GPB - 1) / GPB), (unsigned)tiles), dim3(256), 0, stream, op, v);
    return launch_status();
}

// Launch the 1..2 kernels of one operator call.  `tiles` = column tiles (gridDim.y).  Operators with a per-row state
// (kReduce) use the long-row path only when given a workspace; with workspace == nullptr every row is reduced
// sequentially by its lane group (exact reference order for any length).  Per-edge operators need no scratch.
template <class Op>
static int launch_rowreduce(const Op &op, const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t nnz,
                            int64_t tiles, void *workspace, size_t workspace_bytes, hipStream_t stream,
                            const int32_t *row_order = nullptr) {
    const int64_t RPB = (int64_t)(kWave / Op::LPR) * 4;
    const int64_t n_rowblocks = (m + RPB - 1) / RPB;
    if (n_rowblocks == 0) return COGDL_HIP_OK;
    if (tiles > 65535 || tiles < 1) return COGDL_HIP_ERANGE;
    RowSched s{};
    s.rowptr = rowptr;
    s.colind = colind;
    s.m = m;
    s.rowblocks = make_xcd_map(n_rowblocks);
    s.sort_rows = g_tuning[kTuneRowSort] == 0 ? 1 : 0;
    s.debug = g_tuning[kTuneRowDebug];
    s.order = row_order;
    s.lr.thresh = INT_MAX;
    if (nnz > 0 && (!Op::kReduce || workspace)) {
        plan_long_rows(s.lr, nnz);
        if constexpr (Op::kReduce) {
            s.lr.rec_stride = tiles * Op::kRec * Op::LPR;
            if (workspace_bytes < rowreduce_workspace_bytes(nnz, s.lr.rec_stride)) return COGDL_HIP_EWORKSPACE;
            if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
            if (s.lr.n_long_blocks * sizeof(int32_t) <= kFoundBytes) s.lr.found = (int32_t *)workspace;
            s.lr.partial = (float *)((char *)workspace + kFoundBytes);
        }
    }
    if (!grid_fits(s.rowblocks, s.lr.n_long_blocks)) return COGDL_HIP_ERANGE;
    dim3 grid(s.lr.n_long_blocks + xcd_grid(s.rowblocks), (unsigned)tiles);
    hipLaunchKernelGGL((rowreduce_main_kernel<Op>), grid, dim3(256), 0, stream, op, s);
    if constexpr (Op::kReduce) {
        if (s.lr.n_long_blocks > 0)
            hipLaunchKernelGGL((rowreduce_combine_kernel<Op>), dim3(s.lr.n_long_blocks, (unsigned)tiles), dim3(256), 0,
                               stream, op, s);
    }
    return launch_status();
}

}  // namespace cogdl
