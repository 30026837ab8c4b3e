// rowreduce.h -- the shared execution engine of the row-wise CSR operators (csr_spmm, mhspmm, fused GAT forward
// and both backward passes, scatter_max, sddmm, mhsddmm) on gfx950.
//
// Every one of those operators walks the edges of a CSR row, gathers one dense row per edge and folds it into a
// small per-row state (or emits one value per edge).  The engine owns the traversal; an `Op` functor owns the
// arithmetic.  Three kernels:
//
//   rowreduce_main_kernel      one group of Op::LPR lanes per row ("row group"), 64/LPR rows per wave, 4 waves per
//                              workgroup; colind (and per-edge scalars) are read in coalesced chunks of LPR edges and
//                              broadcast lane-to-lane; gathers are issued Op::UNROLL at a time; the state is updated
//                              strictly in CSR edge order.  Rows longer than `thresh` edges are skipped and recorded.
//   rowreduce_long_kernel      power-law graphs: a row of 10^4..10^5 edges would serialise one lane group for longer
//                              than the rest of the launch takes.  Edges are cut into aligned chunks of `thresh` edges;
//                              every piece (long row  x  chunk) is processed by a whole workgroup: its 256/LPR groups
//                              take contiguous slices, group states are merged in group order through LDS and the
//                              piece's state is written to a workspace record.  Ops without a per-row state (sddmm)
//                              stop here.
//   rowreduce_combine_kernel   per long row: merge the piece records in chunk order, finish the row.
//
// No atomics anywhere, every merge order is fixed => results are deterministic, and rows of at most `thresh` edges
// are reduced in exactly the reference's (sequential CSR) order.
//
// Bookkeeping without a pre-pass or a memset: chunk c = edges [c*T, (c+1)*T).  A long row intersecting chunk c is
// either the row that owns the chunk's FIRST edge (chunk_row[c], "slot 0") or a row that starts inside the chunk --
// which, being longer than T, then owns the first edge of chunk c+1 ("slot 1" of c == chunk_row[c+1]).  The main
// kernel writes chunk_row[c] (the long row's id, or -1) for every chunk whose first edge lies in one of its rows, so
// every entry is rewritten on every launch.
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
//   void chunk_end(Ctx, State&, base, cnt)       after each LPR-chunk of edges (per-edge outputs: coalesced store)
//   void row_end(Ctx, State, row, ok)            write the finished row
//   void pack(State, float(&)[kRec]) / unpack / merge(State& a, State b)   (kReduce only; b's edges follow a's)
#pragma once
#include <algorithm>
#include <climits>

#include "common.h"

namespace cogdl {

struct LongRows {
    int thresh;           // rows with more edges take the chunk-parallel path; also the chunk size (INT_MAX: off)
    int32_t *chunk_row;   // [n_chunks + 1]
    float *partial;       // [n_chunks][2][rec_stride] fp32 piece records (kReduce ops)
    int64_t n_chunks;
    int64_t rec_stride;   // tiles * kRec * LPR
    int chunks_per_block; // the long/combine kernels give each workgroup a contiguous run of chunks
};

// Lanes per row / vector width / column tiles chosen for a row of k columns (see spmm.hip: spmm_geometry).
struct RowGeometry {
    int vec, lpr;
    int64_t tiles;
};
RowGeometry spmm_geometry(int64_t k, int64_t unit, int elem_bytes, int align);

struct RowSched {
    const int32_t *rowptr;
    const int32_t *colind;
    int64_t m;
    XcdMap rowblocks;
    LongRows lr;
};

// Threshold above which a row is split.  The sequential time of a row of T edges (~T/UNROLL gather round trips of
// ~1 us) must stay a small fraction of the launch; small graphs need a low threshold, large ones amortise more.
inline int pick_long_thresh(int64_t nnz) {
    if (g_tuning[kTuneLongThresh] > 0) return g_tuning[kTuneLongThresh];
    int t = 128;
    while (t < 1024 && (int64_t)t * 32768 < nnz) t <<= 1;
    return t;
}
inline int64_t n_chunks_for(int64_t nnz, int thresh) { return (nnz + thresh - 1) / thresh; }
inline size_t chunk_row_bytes(int64_t n_chunks) { return ((size_t)(n_chunks + 1) * sizeof(int32_t) + 255) / 256 * 256; }
// Workspace of an operator whose piece record holds `rec_stride` floats (0: no per-row state).
inline size_t rowreduce_workspace_bytes(int64_t nnz, int64_t rec_stride) {
    if (nnz <= 0) return 0;
    const int64_t n_chunks = n_chunks_for(nnz, pick_long_thresh(nnz));
    const size_t b = chunk_row_bytes(n_chunks) + (size_t)n_chunks * 2 * (size_t)rec_stride * sizeof(float);
    return (b + 255) / 256 * 256;
}

// Main-kernel side of the bookkeeping: the group (lanes l = 0..LPR-1) that owns row `row` = edges [start, end)
// writes chunk_row[c] for every chunk whose first edge lies in the row.  Returns true when the row is long.
template <int LPR>
__device__ __forceinline__ bool mark_long_chunks(const LongRows &lr, int64_t row, int start, int end, int l) {
    const int ch = lr.thresh;
    const bool is_long = end - start > ch;
    if (lr.chunk_row && end > start) {
        const int32_t mark = is_long ? (int32_t)row : -1;
        for (int c = (start + ch - 1) / ch + l; (int64_t)c * ch < end; c += LPR) lr.chunk_row[c] = mark;
    }
    return is_long;
}

// Piece (chunk c, slot) -> the long row it belongs to and its edge range [lo, hi); false if there is none.
__device__ __forceinline__ bool decode_piece(const LongRows &lr, const int32_t *rowptr, int64_t c, int slot,
                                             int32_t &row, int &lo, int &hi) {
    const int ch = lr.thresh;
    row = -1;
    if (slot == 0) row = lr.chunk_row[c];
    else if (c + 1 < lr.n_chunks) {
        row = lr.chunk_row[c + 1];
        if (row >= 0 && (int64_t)rowptr[row] <= c * ch) row = -1;  // same row as slot 0, not a new head
    }
    if (row < 0) return false;
    lo = max(rowptr[row], (int)(c * ch));
    hi = (int)min((int64_t)rowptr[row + 1], (c + 1) * ch);
    return true;
}

// Does the run of chunks [c_begin, c_end] (inclusive: slot 1 of the last chunk) contain any long row?  Block-wide.
__device__ __forceinline__ bool run_has_long_rows(const LongRows &lr, int64_t c_begin, int64_t c_end_incl, int *flag) {
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    for (int64_t c = c_begin + threadIdx.x; c <= c_end_incl && c < lr.n_chunks; c += blockDim.x)
        if (lr.chunk_row[c] >= 0) *flag = 1;
    __syncthreads();
    return *flag != 0;
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

// Fold edges [start, end) of one row into st, in order.  All lanes of the group execute it.
template <class Op>
__device__ __forceinline__ void reduce_edges(const Op &op, const typename Op::Ctx &ctx, typename Op::State &st,
                                             const int32_t *__restrict__ colind, int start, int end, int sub, int l) {
    constexpr int LPR = Op::LPR, UNROLL = Op::UNROLL;
    for (int base = start; base < end; base += LPR) {
        const int cnt = min(LPR, end - base);
        int my_c = 0;
        typename Op::LaneVals lv{};
        if (l < cnt) {
            my_c = colind[base + l];
            op.lane_load(ctx, lv, base + l);
        }
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
        }
        op.chunk_end(ctx, st, base, cnt);
    }
}

template <class Op>
__global__ __launch_bounds__(256) void rowreduce_main_kernel(const Op op, const RowSched s) {
    constexpr int LPR = Op::LPR;
    constexpr int RPW = kWave / LPR;  // row groups per wave
    constexpr int GPB = RPW * 4;      // row groups per 256-thread workgroup
    const int64_t rb = xcd_remap(blockIdx.x, s.rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int64_t row = rb * GPB + wave * RPW + sub;
    const bool ok = row < s.m;
    int start = 0, end = 0;
    if (ok) {
        start = s.rowptr[row];
        end = s.rowptr[row + 1];
    }
    if constexpr (LPR == kWave) {  // whole wave on one row: make the loop bounds scalar
        start = __builtin_amdgcn_readfirstlane(start);
        end = __builtin_amdgcn_readfirstlane(end);
    }
    if (blockIdx.y == 0) mark_long_chunks<LPR>(s.lr, row, start, end, l);
    if (end - start > s.lr.thresh) return;  // long row: the chunk-parallel kernels compute it
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    op.row_load(ctx, row, ok);
    typename Op::State st;
    op.init(ctx, st, row, ok);
    reduce_edges<Op>(op, ctx, st, s.colind, start, end, sub, l);
    op.row_end(ctx, st, row, ok);
}

// One workgroup per run of `chunks_per_block` chunks; runs without long rows are skipped after one coalesced look.
template <class Op>
__global__ __launch_bounds__(256) void rowreduce_long_kernel(const Op op, const RowSched s) {
    constexpr int LPR = Op::LPR;
    constexpr int G = 256 / LPR;  // groups per workgroup
    constexpr int NREC = Op::kReduce ? Op::kRec : 1;
    __shared__ float red[Op::kReduce ? G : 1][NREC][LPR];
    __shared__ int any_long;
    const LongRows &lr = s.lr;
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    const int64_t c_end = min(c_begin + lr.chunks_per_block, lr.n_chunks);
    if (!run_has_long_rows(lr, c_begin, c_end, &any_long)) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int g = (threadIdx.x >> 6) * (kWave / LPR) + sub;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    for (int64_t c = c_begin; c < c_end; ++c) {
        for (int slot = 0; slot < 2; ++slot) {
            int32_t row;  // workgroup-uniform
            int lo, hi;
            if (!decode_piece(lr, s.rowptr, c, slot, row, lo, hi)) continue;
            const int per = (hi - lo + G - 1) / G;  // contiguous slices
            int sb = min(lo + g * per, hi), se = min(sb + per, hi);
            if constexpr (LPR == kWave) {  // one wave per slice: keep the loop bounds scalar
                sb = __builtin_amdgcn_readfirstlane(sb);
                se = __builtin_amdgcn_readfirstlane(se);
            }
            op.row_load(ctx, row, true);
            typename Op::State st;
            op.init_zero(st);
            reduce_edges<Op>(op, ctx, st, s.colind, sb, se, sub, l);
            if constexpr (Op::kReduce) {
                float rec[NREC];
                op.pack(st, rec);
#pragma unroll
                for (int i = 0; i < NREC; ++i) red[g][i][l] = rec[i];
                __syncthreads();
                if (g == 0) {
                    for (int q = 1; q < G; ++q) {  // fixed order: group 0, 1, 2, ...
#pragma unroll
                        for (int i = 0; i < NREC; ++i) rec[i] = red[q][i][l];
                        typename Op::State other;
                        op.unpack(other, rec);
                        op.merge(st, other);
                    }
                    op.pack(st, rec);
                    float *dst = lr.partial + (2 * c + slot) * lr.rec_stride + (int64_t)blockIdx.y * NREC * LPR;
#pragma unroll
                    for (int i = 0; i < NREC; ++i) dst[i * LPR + l] = rec[i];
                }
                __syncthreads();
            }
        }
    }
}

// For every long row merge its piece records in chunk order and finish the row.  The row is combined by the lane
// group that finds it at its FIRST full chunk (the row's head piece, if any, sits in slot 1 of the chunk before).
template <class Op>
__global__ __launch_bounds__(256) void rowreduce_combine_kernel(const Op op, const RowSched s) {
    constexpr int LPR = Op::LPR;
    constexpr int G = 256 / LPR;
    constexpr int NREC = Op::kRec;
    __shared__ int any_long;
    const LongRows &lr = s.lr;
    const int64_t c_begin = (int64_t)blockIdx.x * lr.chunks_per_block;
    const int64_t c_end = min(c_begin + lr.chunks_per_block, lr.n_chunks);
    if (!run_has_long_rows(lr, c_begin, c_end - 1, &any_long)) return;
    const int ch = lr.thresh;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int g = (threadIdx.x >> 6) * (kWave / LPR) + sub;
    typename Op::Ctx ctx = op.make_ctx(l, blockIdx.y);
    // every group of a wave runs the same number of iterations (row_load may shuffle across the whole wave)
    for (int64_t c0 = c_begin; c0 < c_end; c0 += G) {
        const int64_t c = c0 + g;
        int32_t row = (c < c_end) ? lr.chunk_row[c] : -1;
        int start = 0, end = 0;
        if (row >= 0) {
            start = s.rowptr[row];
            end = s.rowptr[row + 1];
            if ((int64_t)(start + ch - 1) / ch != c) row = -1;  // not the row's first owned chunk
        }
        const bool ok = row >= 0;
        op.row_load(ctx, ok ? row : 0, ok);
        if (!ok) continue;
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
            op.merge(st, piece);
        }
        for (int64_t q = c; q <= c_last; ++q) {
            const float *src = lr.partial + (2 * q) * lr.rec_stride + tile_off;
#pragma unroll
            for (int i = 0; i < NREC; ++i) rec[i] = src[i * LPR];
            op.unpack(piece, rec);
            op.merge(st, piece);
        }
        op.row_end(ctx, st, row, true);
    }
}

// Launch the 1..3 kernels of one operator call.  `tiles` = column tiles (gridDim.y).  workspace == nullptr: every
// row is reduced sequentially by its lane group (exact reference order for any length).
template <class Op>
static int launch_rowreduce(const Op &op, const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t nnz,
                            int64_t tiles, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    constexpr int RPB = (kWave / Op::LPR) * 4;
    const int64_t n_rowblocks = (m + RPB - 1) / RPB;
    if (n_rowblocks == 0) return COGDL_HIP_OK;
    if (n_rowblocks > 0x7fffffff / (kXcds * 64) || tiles > 65535 || tiles < 1) return COGDL_HIP_ERANGE;
    RowSched s{};
    s.rowptr = rowptr;
    s.colind = colind;
    s.m = m;
    s.rowblocks = make_xcd_map(n_rowblocks);
    s.lr.thresh = INT_MAX;
    if (workspace && nnz > 0) {
        const int t = pick_long_thresh(nnz);
        const int64_t n_chunks = n_chunks_for(nnz, t);
        const int64_t rec_stride = Op::kReduce ? tiles * Op::kRec * Op::LPR : 0;
        const size_t need = chunk_row_bytes(n_chunks) + (size_t)n_chunks * 2 * (size_t)rec_stride * sizeof(float);
        if (workspace_bytes < need) return COGDL_HIP_EWORKSPACE;
        if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
        s.lr.thresh = t;
        s.lr.chunk_row = (int32_t *)workspace;
        s.lr.partial = (float *)((char *)workspace + chunk_row_bytes(n_chunks));
        s.lr.n_chunks = n_chunks;
        s.lr.rec_stride = rec_stride;
        // a bounded number of workgroups, each owning a contiguous run of chunks (cheap to skip when no row is long)
        const int64_t max_wg = std::max(64, g_tuning[kTuneLongGrid]);
        s.lr.chunks_per_block = (int)((n_chunks + max_wg - 1) / max_wg);
    }
    dim3 grid(xcd_grid(s.rowblocks), (unsigned)tiles);
    hipLaunchKernelGGL((rowreduce_main_kernel<Op>), grid, dim3(256), 0, stream, op, s);
    if (s.lr.n_chunks > 0) {
        const unsigned gx = (unsigned)((s.lr.n_chunks + s.lr.chunks_per_block - 1) / s.lr.chunks_per_block);
        hipLaunchKernelGGL((rowreduce_long_kernel<Op>), dim3(gx, (unsigned)tiles), dim3(256), 0, stream, op, s);
        if constexpr (Op::kReduce)
            hipLaunchKernelGGL((rowreduce_combine_kernel<Op>), dim3(gx, (unsigned)tiles), dim3(256), 0, stream, op, s);
    }
    return launch_status();
}

}  // namespace cogdl
