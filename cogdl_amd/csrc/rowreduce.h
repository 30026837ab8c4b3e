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
RowGeometry spmm_geometry(int64_t k, int64_t unit, int elem_bytes, int align, bool narrow_groups = false);

struct RowSched {
    const int32_t *rowptr;
    const int32_t *colind;
    int64_t m;
    XcdMap rowblocks;
    LongRows lr;
    int sort_rows;  // deal a workgroup's rows to its lane groups by decreasing length (see rowreduce_main_kernel)
    int wave_split; // > 0: in a skewed workgroup, rows of more than this many edges are reduced by ALL lane groups of their wave
    int debug;      // timing experiments (tuning key 13; WRONG results): 1 = row blocks exit, 2 = long-row workgroups exit
    int64_t nnz;    // (the row-tile kernel needs a valid edge index for the rows that have none)
    int queue;      // rowreduce_queue_kernel: wave-units of rows per wave handed out by the workgroup's queue (0: not used)
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

// Operators that can run in ROW TILES (rowreduce_tile_kernel) declare `static constexpr int kRowTile = R`.
template <class Op, class = void>
struct RowTile {
    static constexpr int value = 1;
};
template <class Op>
struct RowTile<Op, std::void_t<decltype(Op::kRowTile)>> {
    static constexpr int value = Op::kRowTile;
};

// Operators that may run through the per-workgroup ROW QUEUE (rowreduce_queue_kernel) declare `static constexpr bool
// kRowQueue = true` (their hooks must not contain workgroup barriers: the waves of a workgroup walk different rows).
template <class Op, class = void>
struct RowQueue {
    static constexpr bool value = false;
};
template <class Op>
struct RowQueue<Op, std::void_t<decltype(Op::kRowQueue)>> {
    static constexpr bool value = Op::kRowQueue;
};

// Rows of at most this many edges are always reduced sequentially by one lane group, in CSR order (the reference's
// summation order: bit-identical fp32 results); longer rows MAY be cut into contiguous pieces whose partial states are
// merged in order (wave-scope split in skewed workgroups, workgroup-scope chunks beyond the long-row threshold).
// OFF by default (tuning key 12: <= 0 = off, n > 0 = split rows of more than n edges): measured on the MI355X it buys
// nothing -- arxiv-sized R-MAT graph, csr_spmm F=64: 133.3 us off / 136.3 (n = 64) / 132.1 (n = 32), F=40: 118.9 / 116.2 /
// 114.8, fused GAT forward 239 / 238 / 233 us (profiles/r04_wavesplit_ab.txt) -- because only 2.8 % of that graph's edges
// sit in rows of 65..128 edges: half of its edges are in rows ABOVE the long-row threshold (the workgroup-scope path),
// and the rest of the imbalance is between waves, not inside them.  Kept as an option (and tested) because graphs with
// a heavier middle of the degree distribution are where it would pay.
inline int wave_split_edges() {
    const int t = g_tuning[kTuneWaveSplit];
    return t > 0 ? t : 0;
}

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

template <int LPR>
__device__ __forceinline__ int group_bcast(int v, int sub, int jj) {
    if constexpr (LPR == kWave) return __builtin_amdgcn_readlane(v, jj);  // -> SGPR: scalar address arithmetic
    else return __shfl(v, sub * LPR + jj, kWave);
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
    for (int j = 0; j < cnt; j += UNROLL) {
        typename Op::Batch b;
        // Issue all UNROLL gathers back to back (no branches: a masked tail slot re-reads the row's last valid
        // neighbour, an L1 hit, and is ignored by apply()).
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int jj = min(j + u, cnt - 1);
            op.fetch(ctx, b, u, group_bcast<LPR>(my_c, sub, jj), base + jj, lv, sub, jj);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) op.apply(ctx, st, b, u, (j + u) < cnt, base + j + u, j + u);
        op.batch_end(ctx, st, base, j, cnt);
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
// Which rank a group takes: its own slot (wave 0 gets the longest rows: best when a wave takes as long as its longest
// row) -- or, when the workgroup holds a MEDIUM row (more than split_min edges, at most `thresh`) that its wave will
// reduce with all its groups together (wave time = sum / groups), rank `rr_rank` = sub * 4 + wave: round-robin over
// the waves, equal sums.
template <int GPB>
__device__ __forceinline__ bool deal_rows_by_length(int slot, bool leader, bool &ok, int &start, int &end, int &mine,
                                                    int rr_rank, int split_min, int thresh) {
    __shared__ int sort_deg[GPB], sort_start[GPB], sort_slot[GPB];
    const int deg = ok ? end - start : -1;
    if (leader) {
        sort_deg[slot] = deg;
        sort_start[slot] = start;
    }
    __syncthreads();
    int dmax = 0, dsum = 0, dmed = 0;
    for (int q = 0; q < GPB; ++q) {
        const int dq = max(sort_deg[q], 0);
        dmax = max(dmax, dq);
        dsum += dq;
        if (dq <= thresh) dmed = max(dmed, dq);
    }
    if (dmax * GPB < 2 * dsum + 8 * GPB) return false;
    int rank = 0;  // by (length descending, natural position ascending)
    for (int q = 0; q < GPB; ++q) {
        const int dq = sort_deg[q];
        rank += (dq > deg || (dq == deg && q < slot)) ? 1 : 0;
    }
    if (leader) sort_slot[rank] = slot;
    __syncthreads();
    mine = sort_slot[(split_min > 0 && dmed > split_min) ? rr_rank : slot];
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
        start = s.rowptr[row];
        end = s.rowptr[row + 1];
    }
    bool skewed = false;
    if constexpr (RPW > 1) {
        // Several rows share a wave, and a wave takes as long as its LONGEST row: on a skewed graph a wave with one
        // 100-edge row and three 5-edge rows idles 70 % of its lanes.  The workgroup's GPB rows are therefore dealt to
        // the row groups in order of decreasing length (ranks by counting through LDS, ties by row id): wave 0 gets the
        // RPW longest rows, wave 3 the shortest, the sum over waves of their longest row -- the lane time spent --
        // drops by 2-3x on R-MAT graphs.  Only the row -> lane-group assignment changes: every row is still reduced
        // sequentially by one group in CSR order (bit-identical results).
        // With the wave-scope split (below) a wave's time is the SUM of its rows' lengths / RPW instead of their maximum,
        // so the ranks are dealt ROUND-ROBIN over the four waves (wave w takes ranks w, w + 4, ...: equal sums).
        if (g_sort_rows(s)) {
            int mine;
            skewed = deal_rows_by_length<GPB>(wave * RPW + sub, lane_on && l == 0, ok, start, end, mine, sub * 4 + wave,
                                              kWave % LPR == 0 ? s.wave_split : 0, s.lr.thresh);
            if (skewed) row = rb * GPB + mine;
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
    if constexpr (RPW > 1 && kWave % LPR == 0) {
        // Wave-scope split of MEDIUM rows (skewed workgroups only).  A row of a few hundred edges, below the long-row
        // threshold, is one lane group walking ~len / UNROLL dependent gather batches while the other groups of its wave
        // finished their 5-edge rows long ago.  In a wave that holds such a row (> wave_split edges) the groups first
        // reduce their own SHORT rows as always, then every medium row of the wave is reduced by all RPW groups together:
        // contiguous slices, the partial states merged in slice order through shuffles (no LDS, no barrier) and finished
        // by group 0.  Rows of at most wave_split edges keep the reference's sequential order (bit-identical); a split
        // row is re-associated at its slice borders only (deterministic).
        const bool medium = ok && !is_long && s.wave_split > 0 && end - start > s.wave_split;
        if (skewed && __ballot(medium) != 0ull) {
            typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
            if (ok && !is_long && !medium) {  // (group-uniform)
                op.row_load(ctx, row, true);
                typename Op::State st;
                op.init(ctx, st, row, true);
                reduce_edges<Op>(op, ctx, st, s.colind, start, end, sub, l, my_lds);
                op.row_end(ctx, st, row, true);
            }
            unsigned long long todo = __ballot(medium && l == 0);
            while (todo) {  // (wave-uniform)
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int64_t r2 = rb * GPB + (int64_t)__shfl((int)(row - rb * GPB), src, kWave);
                const int s2 = __shfl(start, src, kWave), e2 = __shfl(end, src, kWave);
                const int per = ((e2 - s2 + RPW - 1) / RPW + Op::UNROLL - 1) / Op::UNROLL * Op::UNROLL;
                const int sb = min(s2 + sub * per, e2), se = min(sb + per, e2);
                op.row_load(ctx, r2, true);
                typename Op::State st;
                if (sub == 0) op.init(ctx, st, r2, true);  // (an accumulating operator reads the existing output once)
                else op.init_zero(st);
                reduce_edges<Op>(op, ctx, st, s.colind, sb, se, sub, l, my_lds);
                if constexpr (Op::kReduce) {
                    float rec[Op::kRec];
                    op.pack(st, rec);
                    for (int q = 1; q < RPW; ++q) {  // slice order: group 0, 1, 2, ...
                        float other_rec[Op::kRec];
#pragma unroll
                        for (int i = 0; i < Op::kRec; ++i) other_rec[i] = __shfl(rec[i], q * LPR + l, kWave);
                        typename Op::State other;
                        op.unpack(other, other_rec);
                        op.merge(ctx, st, other);  // (all groups execute it: merge may shuffle inside a group)
                    }
                }
                op.row_end(ctx, st, r2, sub == 0);
            }
            return;
        }
    }
    if (is_long) return;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    op.row_load(ctx, row, ok);
    typename Op::State st;
    op.init(ctx, st, row, ok);
    reduce_edges<Op>(op, ctx, st, s.colind, start, end, sub, l, my_lds);
    op.row_end(ctx, st, row, ok);
}

// ---- row tiles (round 4) --------------------------------------------------------------------------------------------
// What bounds the row blocks on graphs with many tiny rows is the dependent chain of a row -- row pointer, column ids,
// gather, store: three round trips even for ONE edge (169 k one-edge rows take 38 us; half the rows of an R-MAT graph
// have at most two edges, tools/rowcost_probe.py).  Here a lane group owns R CONSECUTIVE rows and walks their chains
// together: one load brings the R + 1 row pointers, the R first id chunks are requested at once, and the first
// UNROLL / R edges of every row are gathered in ONE batch -- three round trips per R rows.  Whatever a row has beyond
// that is folded afterwards, row by row, in CSR order: per row the arithmetic and its order are those of the plain
// kernel (bit-identical results); R rows per group also average the row lengths a wave sees.  Operators whose context
// and hooks do not depend on the row (SpmmOp without attention / epilogue) opt in with kRowTile.
template <class Op, int R>
__global__ __launch_bounds__(256) void rowreduce_tile_kernel(const Op op, const RowSched s) {
    constexpr int LPR = Op::LPR, UNROLL = Op::UNROLL;
    static_assert(Op::kLds == 0 && UNROLL % R == 0 && kWave % LPR == 0 && R < LPR, "row tiles: plain operators only");
    constexpr int RPW = kWave / LPR, GPB = RPW * 4, U1 = UNROLL / R;
    __shared__ float op_lds[1];
    if (blockIdx.x < s.lr.n_long_blocks) {
        if (s.debug != 2) rowreduce_long_block<Op>(op, s, op_lds);
        return;
    }
    if (s.debug == 1) return;
    const int64_t rb = xcd_remap(blockIdx.x - s.lr.n_long_blocks, s.rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int l = lane % LPR, sub = lane / LPR;
    const int64_t row0 = (rb * GPB + wave * RPW + sub) * R;
    // the R + 1 row pointers of the group: one coalesced load (rows past the end read rowptr[m]: empty)
    const int pv = s.rowptr[min(row0 + min(l, R), s.m)];
    int st_[R], cnt_[R], en_[R];
    bool live[R];  // false: a row past the end, or a long row (the long-row workgroups and the combine kernel own it)
#pragma unroll
    for (int i = 0; i < R; ++i) {
        st_[i] = group_bcast<LPR>(pv, sub, i);
        en_[i] = group_bcast<LPR>(pv, sub, i + 1);
        live[i] = row0 + i < s.m && en_[i] - st_[i] <= s.lr.thresh;
        if (!live[i]) en_[i] = st_[i];
        cnt_[i] = min(LPR, en_[i] - st_[i]);
    }
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    // phase 0: the first id chunk of every row (unconditional loads; a row without edges reads edge 0 and ignores it)
    int c_[R];
    typename Op::LaneVals lv_[R];
    if (s.nnz > 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int idx = cnt_[i] > 0 ? min(st_[i] + l, en_[i] - 1) : 0;
            c_[i] = s.colind[idx];
            lv_[i] = typename Op::LaneVals{};
            op.lane_load(ctx, lv_[i], idx);
        }
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            c_[i] = 0;
            lv_[i] = typename Op::LaneVals{};
        }
    }
    typename Op::State acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        op.row_load(ctx, row0 + i, live[i]);
        op.init(ctx, acc[i], row0 + i, live[i]);
    }
    // phase 1: the first U1 edges of all R rows in one batch of gathers
    if (s.nnz > 0) {
        typename Op::Batch b;
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int k = 0; k < U1; ++k) {
                const int jj = max(min(k, cnt_[i] - 1), 0);
                op.fetch(ctx, b, i * U1 + k, group_bcast<LPR>(c_[i], sub, jj), st_[i] + jj, lv_[i], sub, jj);
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int k = 0; k < U1; ++k) op.apply(ctx, acc[i], b, i * U1 + k, k < cnt_[i], st_[i] + k, k);
        }
    }
    // phase 2: what a row has beyond its first U1 edges, row by row, in order
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (cnt_[i] > U1) {  // (group-uniform)
            for (int j = U1; j < cnt_[i]; j += UNROLL) {
                typename Op::Batch b;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int jj = min(j + u, cnt_[i] - 1);
                    op.fetch(ctx, b, u, group_bcast<LPR>(c_[i], sub, jj), st_[i] + jj, lv_[i], sub, jj);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) op.apply(ctx, acc[i], b, u, (j + u) < cnt_[i], st_[i] + j + u, j + u);
            }
            if (st_[i] + LPR < en_[i]) reduce_edges<Op>(op, ctx, acc[i], s.colind, st_[i] + LPR, en_[i], sub, l, op_lds);
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (live[i]) op.row_end(ctx, acc[i], row0 + i, true);
}

// ---- row queue (round 5) ---------------------------------------------------------------------------------------------
// What the degree-sorted dealing of rowreduce_main_kernel cannot fix on skewed graphs is the imbalance BETWEEN waves and
// between workgroups: a workgroup lives as long as its longest row while its other waves finished their 5-edge rows long
// ago (arxiv-sized R-MAT graph, F = 64: row blocks alone 107 us, 77 us with rows and columns relabelled by degree,
// profiles/r04_longpath_probe.txt).  Here a workgroup owns `queue` times as many consecutive rows and its four waves PULL
// wave-units of RPW rows from a counter in LDS until the pool is empty: a wave that drew a 100-edge row simply takes fewer
// units.  Per row nothing changes -- one lane group, CSR order, the operator's own hooks (bit-identical results); long rows
// are skipped as before.  One LDS atomic per unit; no barrier after the first.
template <class Op>
__global__ __launch_bounds__(256, MinWaves<Op>::value) void rowreduce_queue_kernel(const Op op, const RowSched s) {
    __shared__ float op_lds[Op::kLds > 0 ? 256 * Op::kLds : 1];
    __shared__ int q_next;
    if (blockIdx.x < s.lr.n_long_blocks) {
        if (s.debug != 2) rowreduce_long_block<Op>(op, s, op_lds);
        return;
    }
    if (s.debug == 1) return;
    constexpr int LPR = Op::LPR;
    constexpr int RPW = kWave / LPR;
    const int64_t rb = xcd_remap(blockIdx.x - s.lr.n_long_blocks, s.rowblocks);
    if (rb < 0) return;
    if (threadIdx.x == 0) q_next = 0;
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const int l = lane % LPR;
    const bool lane_on = (kWave % LPR == 0) || lane / LPR < RPW;
    const int sub = lane_on ? lane / LPR : 0;
    const int n_units = s.queue * 4;
    const int64_t row0 = rb * (int64_t)n_units * RPW;
    float *const my_lds = op_lds + (threadIdx.x / LPR) * LPR * Op::kLds;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    // The row pointers of the NEXT unit are requested before the gathers of the current one: a unit's dependent chain is
    // (row pointers ->) column ids -> gathers instead of three round trips -- on graphs where half the rows have one or two
    // edges the chain, not the bytes, is what a row costs.
    auto claim = [&]() {
        int u = 0;
        if (lane == 0) u = atomicAdd(&q_next, 1);
        return __builtin_amdgcn_readfirstlane(u);  // (lane 0 is the first active lane)
    };
    auto bounds = [&](int u, int64_t &row, bool &ok, int &start, int &end) {
        row = row0 + (int64_t)u * RPW + sub;
        ok = lane_on && u < n_units && row < s.m;
        start = end = 0;
        if (ok) {
            start = s.rowptr[row];
            end = s.rowptr[row + 1];
        }
    };
    int u = claim();
    int64_t row;
    bool ok;
    int start, end;
    bounds(u, row, ok, start, end);
    while (u < n_units && row0 + (int64_t)u * RPW < s.m) {  // (wave-uniform)
        const int un = claim();
        int64_t row_n;
        bool ok_n;
        int start_n, end_n;
        bounds(un, row_n, ok_n, start_n, end_n);  // in flight while this unit's rows are reduced
        if constexpr (LPR == kWave) {
            start = __builtin_amdgcn_readfirstlane(start);
            end = __builtin_amdgcn_readfirstlane(end);
        }
        if (end - start <= s.lr.thresh) {  // (group-uniform) a long row is the long-row workgroups' and the combine kernel's
            op.row_load(ctx, row, ok);
            typename Op::State st;
            op.init(ctx, st, row, ok);
            reduce_edges<Op>(op, ctx, st, s.colind, start, end, sub, l, my_lds);
            op.row_end(ctx, st, row, ok);
        }
        u = un;
        row = row_n;
        ok = ok_n;
        start = start_n;
        end = end_n;
    }
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

// Launch the 1..2 kernels of one operator call.  `tiles` = column tiles (gridDim.y).  Operators with a per-row state
// (kReduce) use the long-row path only when given a workspace; with workspace == nullptr every row is reduced
// sequentially by its lane group (exact reference order for any length).  Per-edge operators need no scratch.
template <class Op>
static int launch_rowreduce(const Op &op, const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t nnz,
                            int64_t tiles, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    // Row tiles (rowreduce_tile_kernel): operators that allow it, lane groups that tile a wave, several rows per wave.
    // tuning key 14: 2 = on; OFF by default -- measured (profiles/r04_rowtile_ab.txt): 169 k one-edge rows 40.7 -> 31.3 us
    // (half of what is left above the 19 us of an empty launch), two-edge rows 41 -> 37, but rows of >= 4 edges 4-8 %
    // slower, arxiv-sized R-MAT F=64 132 -> 145 us, uniform 105 -> 111 us: four rows' remainders walked one after the
    // other cost more than the shared first round trip saves.
    constexpr int kTile = (RowTile<Op>::value > 1 && kWave % Op::LPR == 0 && Op::LPR < kWave && Op::LPR > RowTile<Op>::value)
                              ? RowTile<Op>::value : 1;
    const bool tile_rows = kTile > 1 && g_tuning[kTuneRowTile] == 2;
    // Row queue (rowreduce_queue_kernel): tuning key 16 = wave-units per wave (0 = off)
    const int queue = (RowQueue<Op>::value && !tile_rows) ? std::min(std::max(g_tuning[kTuneRowQueue], 0), 64) : 0;
    const int64_t RPB = (int64_t)(kWave / Op::LPR) * 4 * (tile_rows ? kTile : 1) * (queue > 0 ? queue : 1);
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
    s.nnz = nnz;
    s.queue = queue;
    s.wave_split = wave_split_edges();  // (set to 0 below when the caller asked for sequential rows: no workspace)
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
    if (s.lr.thresh == INT_MAX) s.wave_split = 0;  // no workspace = every row sequentially, in the reference's order
    if (!grid_fits(s.rowblocks, s.lr.n_long_blocks)) return COGDL_HIP_ERANGE;
    dim3 grid(s.lr.n_long_blocks + xcd_grid(s.rowblocks), (unsigned)tiles);
    bool launched = false;
    if constexpr (RowQueue<Op>::value) {
        if (queue > 0) {
            hipLaunchKernelGGL((rowreduce_queue_kernel<Op>), grid, dim3(256), 0, stream, op, s);
            launched = true;
        }
    }
    if constexpr (kTile > 1) {
        if (!launched && tile_rows) {
            hipLaunchKernelGGL((rowreduce_tile_kernel<Op, kTile>), grid, dim3(256), 0, stream, op, s);
            launched = true;
        }
    }
    if (!launched) hipLaunchKernelGGL((rowreduce_main_kernel<Op>), grid, dim3(256), 0, stream, op, s);
    if constexpr (Op::kReduce) {
        if (s.lr.n_long_blocks > 0)
            hipLaunchKernelGGL((rowreduce_combine_kernel<Op>), dim3(s.lr.n_long_blocks, (unsigned)tiles), dim3(256), 0,
                               stream, op, s);
    }
    return launch_status();
}

}  // namespace cogdl
