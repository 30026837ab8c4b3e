// radix_transpose.hip -- csr2csc as a hand-written LSD radix sort for gfx950 (replaces the
// reference's cusparseCsr2cscEx2, cogdl/operators/spmm/spmm_kernel.cu:514-532, and this library's earlier rocPRIM
// pipeline, removed in round 5).
//
// What is sorted: the nnz edge slots by column id, stably (a column's edges keep their CSR = row order), carrying TWO
// payloads per edge -- its CSR position e (-> perm) and its row (-> rowind).  The row is found once, in the first pass,
// where e is still sequential: every row that starts inside a tile drops its id at its first slot in LDS and a running
// maximum spreads it -- no search per edge; the earlier pipeline looked the row of every perm[j] up afterwards, 0.6 ms of
// random reads on the Reddit-shaped graph.
//
// Digits of up to 9 bits (512 bins): column ids of up to 18 bits -- every graph of BASELINE.json's configs[0..3] -- take
// TWO passes where rocPRIM's onesweep takes three 8-bit ones.  One pass = three kernels, no atomics on global memory, no
// memset, no look-back spinning (=> deterministic, and capturable in a hipGraph at any size):
//   upsweep    per tile of 8192 slots: digit histogram -> table[digit][tile]
//   scan       exclusive scan of the table in digit-major order (scan.h): where tile t's run of digit d starts
//   downsweep  per tile: stable rank of every slot inside the tile (per wave: lanes with the same digit found with
//              ballots, a running per-wave counter per digit in LDS; waves own contiguous quarters of the tile), the tile
//              is reordered by digit through LDS, then written out as runs of equal digits (64 B on average at 512
//              bins / 8192 slots; hub columns give long runs).  Keys and the two payloads take turns in one 32 KB LDS
//              buffer: 45 KB per workgroup, 3 workgroups per CU.
// Traffic per slot: 4 (upsweep) + 4 + 12 (first downsweep) + 4 + 12 + 12 (second) = 48 B against 16 B algorithmic
// (36 B with the packed intermediate records of the two-pass case, see rt_downsweep_kernel).
// Measured on the Reddit-shaped graph (114.8 M slots, 2 passes, before the XCD-contiguous tile order): 2.38 ms, of which the ballot matching is ~0.1 ms and
// the two payload arrays ~0.9 ms -- the scatter of 64-byte runs over the whole output (every run lands in a different
// DRAM page) is what bounds a pass, not the instruction count: 4096-slot tiles (more workgroups in flight, 32-byte runs)
// are slower (2.89 ms), 16384-slot tiles (128-byte runs, one workgroup per CU) the same (2.34 ms).  Dropping one staged
// array AND two global stores (packed records) gains only 3 %: a staging round costs its two 8-wave barriers, not its
// bytes -- with 2 workgroups per CU little else runs while a workgroup waits.
#include "common.h"
#include "scan.h"

namespace cogdl {
namespace rt {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / kWave;  // 8
constexpr int kTile = 8192;
constexpr int kPerWave = kTile / kWaves;  // contiguous slots of one wave
constexpr int kRows = kPerWave / kWave;   // 16 wave-rows of 64 slots
constexpr int kMaxBits = 9;
constexpr int kMaxBins = 1 << kMaxBits;
static_assert(kMaxBins <= kThreads, "one thread per digit in the downsweep's run layout");

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct Params {
    // first pass: keys = colind (slots >= valid get pad_key), payloads are generated; later passes: the three arrays
    const int32_t *colind;
    const int32_t *rowptr;
    const uint32_t *keys_in;
    const int32_t *e_in, *row_in;
    uint32_t *keys_out;
    int32_t *e_out, *row_out;
    uint32_t *table;        // upsweep: raw counts out; downsweep: scanned offsets in
    int32_t *tile_rows;     // [2 * n_tiles]: the rows of every tile's first and last slot (written by the first upsweep)
    int64_t m, nnz, n_tiles;
    uint32_t pad_key;
    int shift, bits, padded;
    int row_bits;           // packed records: word = (remaining key digits << row_bits) | row
};

__device__ __forceinline__ int64_t rt_row_search(const int32_t *__restrict__ rowptr, int64_t lo, int64_t hi, int64_t e) {
    while (hi - lo > 1) {  // invariant: rowptr[lo] <= e (< rowptr[hi] where hi < m + 1); empty rows share offsets
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)rowptr[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// lanes (among `valid` ones) whose digit equals this lane's.  Per bit: x = 0 / ~0 (signed 1-bit extract), the ballot of
// the bit, and peers &= ~(ballot ^ x) on both halves (v_xnor + v_and): 6 vector instructions per bit.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, int bits, bool valid) {
    const uint64_t v = __ballot(valid);
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    for (int b = 0; b < bits; ++b) {
        const int32_t x = __builtin_amdgcn_sbfe((int32_t)d, (uint32_t)b, 1u);
        const uint64_t mset = __ballot(x != 0);
        lo &= ~((uint32_t)mset ^ (uint32_t)x);
        hi &= ~((uint32_t)(mset >> 32) ^ (uint32_t)x);
    }
    return ((uint64_t)hi << 32) | lo;
}

template <bool FIRST>
__device__ __forceinline__ uint32_t load_key(const Params &p, int64_t i, int64_t valid) {
    if constexpr (FIRST) return i < valid ? (uint32_t)p.colind[i] : p.pad_key;
    else return p.keys_in[i];
}

template <bool FIRST, bool IN_PACKED = false>
__global__ __launch_bounds__(kThreads) void rt_upsweep_kernel(const Params p) {
    __shared__ uint32_t hist[kWaves][kMaxBins];
    const int t = threadIdx.x, lane = t & (kWave - 1), w = t >> 6;
    const int nbins = 1 << p.bits;
    const int64_t tile = blockIdx.x;
    const int64_t tile0 = tile * kTile, lim = p.nnz, tidx0 = tile, tstride = p.n_tiles;  // table index = tidx0 + d * tstride
    for (int i = t; i < kWaves * kMaxBins; i += kThreads) (&hist[0][0])[i] = 0;
    const int64_t base = tile0 + (int64_t)w * kPerWave;
    const int64_t valid = (FIRST && p.padded) ? (int64_t)p.rowptr[p.m] : p.nnz;
    uint32_t key[kRows];
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const int64_t i = base + j * kWave + lane;
        if constexpr (IN_PACKED) key[j] = i < lim ? ((uint32_t)p.row_in[i] >> p.row_bits) : 0u;  // (shift = 0)
        else key[j] = i < lim ? load_key<FIRST>(p, i, valid) : 0u;
    }
    __syncthreads();
    const uint32_t mask = (uint32_t)nbins - 1u;
    // counting only: order does not matter, so plain LDS atomics on the wave's private histogram (integer adds: the
    // result is deterministic); the ballot-based matching of the downsweep costs ~75 vector instructions per 64 slots
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const int64_t i = base + j * kWave + lane;
        if (i < lim) atomicAdd(&hist[w][(key[j] >> p.shift) & mask], 1u);
    }
    if constexpr (FIRST) {  // two lanes of two different waves look the tile's boundary rows up for the downsweep
        if (t == 0 || t == kWave) {
            const int64_t e = t == 0 ? tile * kTile : min(p.nnz, (tile + 1) * kTile) - 1;
            p.tile_rows[2 * tile + (t == 0 ? 0 : 1)] = (int32_t)rt_row_search(p.rowptr, 0, p.m, e);
        }
    }
    __syncthreads();
    for (int d = t; d < nbins; d += kThreads) {
        uint32_t c = 0;
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) c += hist[ww][d];
        p.table[(int64_t)d * tstride + tidx0] = c;
    }
}

// PACK = 0: records are three arrays (key, e, row) on both sides.  Two-pass sorts (column ids of <= 18 bits) use the
// PACKED intermediate record instead -- after the first pass only the key's second digit is still needed, and it fits
// one word together with the row: (digit << row_bits) | row --
//   PACK = 1 (first pass):  writes (e, word); the keys go through LDS only to place the slots, not to memory;
//   PACK = 2 (second pass): reads (e, word); one LDS round carries digit and row together, so the pass stages and
//                           writes two arrays (rowind, perm) and no sorted keys (colptr comes from rt_colptr_perm).
template <bool FIRST, int PACK = 0>
__global__ __launch_bounds__(kThreads, 4) void rt_downsweep_kernel(const Params p) {
    static_assert(PACK == 0 || (FIRST ? PACK == 1 : PACK == 2), "packed out in the first pass, packed in in the second");
    constexpr bool PACK_IN = PACK == 2;
    __shared__ __attribute__((aligned(16))) uint32_t buf[kTile];
    __shared__ uint32_t cnt[kWaves][kMaxBins];  // per wave: running count, then the wave's start inside the digit's run
    __shared__ uint32_t dstart[kMaxBins];       // where the digit's run starts in the reordered tile
    __shared__ uint32_t delta[kMaxBins];        // output position = delta[digit] + position in the reordered tile
    __shared__ uint32_t wsum[kWaves];
    const int t = threadIdx.x, lane = t & (kWave - 1), w = t >> 6;
    const int nbins = 1 << p.bits;
    const uint32_t mask = (uint32_t)nbins - 1u;
    // Workgroup -> tile: consecutive workgroup ids go round-robin over the 8 XCDs (private L2 each), and consecutive
    // TILES write adjacent pieces of every digit's run.  Each XCD therefore takes a contiguous eighth of the tiles: the
    // ~64 tiles in flight on one XCD are neighbours, and their 64-byte pieces merge into full lines in that XCD's L2
    // before they are written back (PMC: 1.8x write amplification + the read-modify-write fills with the natural order).
    const int64_t n_tiles_here = p.n_tiles;
    const int64_t per_xcd = (n_tiles_here + kXcds - 1) / kXcds;
    const int64_t tile = (int64_t)(blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds;
    if (tile >= n_tiles_here) return;
    const int64_t tile0 = tile * kTile, tidx0 = tile, tstride = p.n_tiles;  // table index = tidx0 + digit * tstride
    const int n_here = (int)min((int64_t)kTile, p.nnz - tile0);
    for (int i = t; i < kWaves * kMaxBins; i += kThreads) (&cnt[0][0])[i] = 0;
    const int l0 = w * kPerWave + lane;  // local slot of wave-row 0; wave-row j: l0 + 64 * j

    // ---- everything this tile reads from global memory is requested up front: keys, and (later passes) both payloads
    uint32_t key[kRows];
    int32_t pe[FIRST ? 1 : kRows], pr[kRows];
    if constexpr (FIRST) {
        const int32_t *__restrict__ src = p.colind + tile0;
        int valid = n_here;
        if (p.padded) valid = (int)max((int64_t)0, min((int64_t)n_here, (int64_t)p.rowptr[p.m] - tile0));
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int l = l0 + j * kWave;
            key[j] = l < valid ? (uint32_t)src[l] : p.pad_key;
        }
    } else if constexpr (PACK_IN) {
        const int32_t *__restrict__ esrc = p.e_in + tile0, *__restrict__ wsrc = p.row_in + tile0;
        const uint32_t row_mask = (1u << p.row_bits) - 1u;
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int l = l0 + j * kWave;
            const bool ok = l < n_here;
            const uint32_t wd = ok ? (uint32_t)wsrc[l] : 0u;
            key[j] = wd >> p.row_bits;  // (the digits still to be sorted by, this pass's at the bottom; p.shift = 0)
            pr[j] = (int32_t)(wd & row_mask);
            pe[j] = ok ? esrc[l] : 0;
        }
    } else {
        const uint32_t *__restrict__ ksrc = p.keys_in + tile0;
        const int32_t *__restrict__ esrc = p.e_in + tile0, *__restrict__ rsrc = p.row_in + tile0;
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int l = l0 + j * kWave;
            const bool ok = l < n_here;
            key[j] = ok ? ksrc[l] : 0u;
            pe[j] = ok ? esrc[l] : 0;
            pr[j] = ok ? rsrc[l] : 0;
        }
    }
    // ---- first pass: the row of every slot.  The slots are consecutive CSR positions: every row that STARTS inside
    // the tile drops its (relative) id at its first slot, a running maximum spreads it over the row -- no search.
    if constexpr (FIRST) {
        const int r_first = p.tile_rows[2 * tile], r_last = p.tile_rows[2 * tile + 1];
        for (int i = t; i < kTile; i += kThreads) buf[i] = 0;
        __syncthreads();
        for (int r = r_first + 1 + t; r <= r_last; r += kThreads) {
            const int64_t pos = (int64_t)p.rowptr[r] - tile0;  // (> 0: r_first is the LAST row that starts at or before tile0)
            if (pos < n_here) atomicMax(&buf[pos], (uint32_t)(r - r_first));  // empty rows share a slot: the last one owns it
        }
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            uint32_t v = buf[l0 + j * kWave];
#pragma unroll
            for (int sft = 1; sft < kWave; sft <<= 1) {
                const uint32_t u = __shfl_up(v, sft, kWave);
                if (lane >= sft) v = max(v, u);
            }
            v = max(v, carry);
            carry = __shfl(v, kWave - 1, kWave);
            pr[j] = (int32_t)v;
        }
        if (lane == 0) wsum[w] = carry;
        __syncthreads();
        uint32_t prev = 0;
        for (int ww = 0; ww < w; ++ww) prev = max(prev, wsum[ww]);
#pragma unroll
        for (int j = 0; j < kRows; ++j) pr[j] = r_first + (int32_t)max((uint32_t)pr[j], prev);
    }
    __syncthreads();
    // ---- stable rank inside the wave's part: (same digit in earlier wave-rows) + (same digit in lower lanes)
    uint16_t rank[kRows];
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const bool ok = l0 + j * kWave < n_here;
        const uint32_t d = (key[j] >> p.shift) & mask;
        const uint64_t peers = match_digit(d, p.bits, ok);
        const int lower = __popcll(peers & lt);
        const uint32_t before = cnt[w][d];      // (LDS operations of a wave complete in order: read, then the leader's write)
        rank[j] = (uint16_t)(before + lower);
        if (ok && lower == 0) cnt[w][d] = before + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    // ---- per digit: total of the tile, each wave's start inside the run; runs laid out in digit order
    uint32_t tot = 0;  // thread t owns digit t (kThreads == kMaxBins)
    uint32_t gb = 0;
    if (t < nbins) {
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) {
            const uint32_t c = cnt[ww][t];
            cnt[ww][t] = tot;
            tot += c;
        }
        gb = p.table[(int64_t)t * tstride + tidx0];
    }
    uint32_t incl = tot;  // exclusive scan of the totals over the digits
#pragma unroll
    for (int sft = 1; sft < kWave; sft <<= 1) {
        const uint32_t v = __shfl_up(incl, sft, kWave);
        if (lane >= sft) incl += v;
    }
    __syncthreads();  // (wsum is reused)
    if (lane == kWave - 1) wsum[w] = incl;
    __syncthreads();
    uint32_t off = incl - tot;
    for (int ww = 0; ww < w; ++ww) off += wsum[ww];
    if (t < nbins) {
        dstart[t] = off;
        delta[t] = gb - off;  // (mod 2^32)
    }
    __syncthreads();
    // ---- position of every slot in the reordered tile (in place of its rank)
    uint16_t (&spos)[kRows] = rank;
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const uint32_t d = (key[j] >> p.shift) & mask;
        spos[j] = (uint16_t)(dstart[d] + cnt[w][d] + rank[j]);
    }
    // ---- keys through LDS -> runs of equal digits in the output; remember where each reordered slot goes
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        if (l0 + j * kWave < n_here) {
            if constexpr (PACK_IN) buf[spos[j]] = (key[j] << p.row_bits) | (uint32_t)pr[j];  // digit(s) and row in one round
            else buf[spos[j]] = key[j];
        }
    }
    __syncthreads();
    uint32_t gpos[kRows];  // (nnz < 2^31)
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const int sidx = q * kThreads + t;
        gpos[q] = 0;
        if (sidx < n_here) {
            const uint32_t k = buf[sidx];
            if constexpr (PACK == 2) {
                gpos[q] = delta[(k >> p.row_bits) & mask] + (uint32_t)sidx;
                p.row_out[gpos[q]] = (int32_t)(k & ((1u << p.row_bits) - 1u));
            } else {
                gpos[q] = delta[(k >> p.shift) & mask] + (uint32_t)sidx;
                if constexpr (PACK == 0) p.keys_out[gpos[q]] = k;
            }
        }
    }
    __syncthreads();
    // ---- payload 1: the CSR position
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const int l = l0 + j * kWave;
        if (l < n_here) buf[spos[j]] = FIRST ? (uint32_t)(tile0 + l) : (uint32_t)pe[j];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRows; ++q)
        if (q * kThreads + t < n_here) p.e_out[gpos[q]] = (int32_t)buf[q * kThreads + t];
    if constexpr (PACK_IN) return;  // (the row went out with the digit)
    __syncthreads();
    // ---- payload 2: the row (packed: with the key's remaining digit above it)
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        if (l0 + j * kWave < n_here) {
            if constexpr (PACK == 1)  // the digit the second pass still needs, above the row
                buf[spos[j]] = ((key[j] >> p.bits) << p.row_bits) | (uint32_t)pr[j];
            else buf[spos[j]] = (uint32_t)pr[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRows; ++q)
        if (q * kThreads + t < n_here) p.row_out[gpos[q]] = (int32_t)buf[q * kThreads + t];
}

// colptr without the sorted keys (packed sorts do not write them): colptr[c] = the first position j whose column
// colind[perm[j]] is >= c -- one binary search per column through perm (<= 2^18 columns x ~27 steps x 2 loads).
__global__ void rt_colptr_perm_kernel(const int32_t *__restrict__ colind, const int32_t *__restrict__ perm,
                                      const int32_t *__restrict__ rowptr, int64_t m, int padded, int64_t nnz,
                                      int64_t n_cols, int32_t *__restrict__ colptr) {
    const int64_t valid = padded ? (int64_t)rowptr[m] : nnz;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= n_cols; c += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const int64_t e = perm[mid];
            const int64_t key = e < valid ? (int64_t)colind[e] : n_cols;
            if (key < c) lo = mid + 1; else hi = mid;
        }
        colptr[c] = (int32_t)lo;
    }
}

__global__ void rt_colptr_sorted_kernel(const uint32_t *__restrict__ keys, int32_t *__restrict__ colptr, int64_t nnz,
                                        int64_t n_cols) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= n_cols; c += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = nnz;  // first position whose key >= c
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < c) lo = mid + 1; else hi = mid;
        }
        colptr[c] = (int32_t)lo;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small structures (sampled blocks of a mini-batch step: at most 16 k edge slots, 16 k columns): the WHOLE stable
// transpose by ONE workgroup in ONE launch, everything in LDS -- where the rocPRIM pipeline takes 9 launches (key padding,
// a block sort, four merges, three look-up kernels) and a captured mini-batch step pays ~4.5 us per dependent launch.
//   1. column histogram (LDS atomics: integer counts, deterministic) -> exclusive scan -> colptr
//   2. row of every slot: each non-empty row drops its id at its first slot, a running maximum spreads it
//   3. two stable 7-bit LSD passes over the packed words (column << 14 | slot): per-wave counters + ballot matching for
//      the ranks (as rt_downsweep_kernel), the wave totals scanned per digit, the tile reordered through LDS
//   4. perm = the slot, rowind = its row, straight from the sorted words
// Layout: wave w owns the slots [1024 w, 1024 (w + 1)), item j of lane l is slot 1024 w + 64 j + l.
constexpr int kSmThreads = 1024, kSmWaves = 16, kSmItems = 16, kSmTile = kSmThreads * kSmItems;  // 16384 slots
constexpr int kSmSlotBits = 14, kSmDigit = 7, kSmBins = 1 << kSmDigit;

// ITEMS slots per thread (the tile = 1024 ITEMS slots must hold the edge slots AND the column keys 0 .. n_cols + 1), DB =
// digit bits of the two LSD passes (2 DB >= the key bits): the 128-seed block of a step (1280 slots) takes 2 items and
// 6-bit digits instead of 16 and 7 -- the kernel is one workgroup's instruction stream, its time is its length.
template <int ITEMS, int DB>
__global__ __launch_bounds__(kSmThreads) void small_transpose_kernel(const int32_t *__restrict__ rowptr,
                                                                   const int32_t *__restrict__ colind, int m, int n_cols,
                                                                   int cap, int padded, int32_t *__restrict__ colptr,
                                                                   int32_t *__restrict__ rowind, int32_t *__restrict__ perm) {
    constexpr int TILE = kSmThreads * ITEMS, BINS = 1 << DB;
    static_assert(ITEMS <= kSmItems && DB <= kSmDigit && 2 * DB <= 32 - kSmSlotBits, "tile / digit geometry");
    __shared__ __attribute__((aligned(16))) uint32_t buf[TILE];  // (16 items: 64 KB) first the histogram, then the reorder buffer
    __shared__ uint16_t rowof[TILE];                             // (32 KB)
    __shared__ uint32_t cnt[kSmWaves][BINS];                     // (7-bit digits: 8 KB)
    __shared__ uint32_t dstart[kSmBins];
    __shared__ uint32_t wtot[kSmWaves + 1];
    const int t = threadIdx.x, lane = t & (kWave - 1), w = t >> 6;
    const int valid = padded ? min(max(rowptr[m], 0), cap) : cap;
    const uint32_t pad_key = (uint32_t)n_cols, none_key = (uint32_t)n_cols + 1u;  // (n_cols + 2 <= 16384: 14 key bits)
    // ---- the slots' words; histogram of the columns
    uint32_t word[ITEMS];
    for (int i = t; i < TILE; i += kSmThreads) {
        buf[i] = 0;
        rowof[i] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = w * (ITEMS * kWave) + j * kWave + lane;
        uint32_t key = none_key;
        if (e < cap) key = e < valid ? (uint32_t)colind[e] : pad_key;
        if (key > none_key) key = none_key;  // (a column id outside [0, n_cols): sorted behind everything, never written)
        word[j] = (key << kSmSlotBits) | (uint32_t)e;
        if (e < cap && key <= pad_key) atomicAdd(&buf[key], 1u);
    }
    for (int r = t; r < m; r += kSmThreads) {  // a non-empty row marks its first slot (distinct slots: no conflicts)
        const int s0 = rowptr[r], s1 = rowptr[r + 1];
        if (s0 < s1 && s0 >= 0 && s0 < TILE) rowof[s0] = (uint16_t)r;
    }
    __syncthreads();
    // ---- colptr = exclusive scan of the histogram over the columns 0 .. n_cols (thread t: 16 consecutive columns)
    {
        uint32_t h[ITEMS], sum = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            h[i] = buf[t * ITEMS + i];
            sum += h[i];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int sft = 1; sft < kWave; sft <<= 1) {
            const uint32_t u = __shfl_up(incl, sft, kWave);
            if (lane >= sft) incl += u;
        }
        if (lane == kWave - 1) wtot[w] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int ww = 0; ww < w; ++ww) run += wtot[ww];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int c = t * ITEMS + i;
            if (c <= n_cols) colptr[c] = (int32_t)run;
            run += h[i];
        }
    }
    // ---- rows: running maximum of the marks in slot order (thread t: the 16 consecutive slots 16 t ..)
    {
        uint32_t v[ITEMS], mx = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            mx = max(mx, (uint32_t)rowof[t * ITEMS + i]);
            v[i] = mx;
        }
        uint32_t incl = mx;
#pragma unroll
        for (int sft = 1; sft < kWave; sft <<= 1) {
            const uint32_t u = __shfl_up(incl, sft, kWave);
            if (lane >= sft) incl = max(incl, u);
        }
        __syncthreads();  // (wtot is reused; every thread has read its marks)
        if (lane == kWave - 1) wtot[w] = incl;
        __syncthreads();
        uint32_t before = __shfl_up(incl, 1, kWave);
        if (lane == 0) before = 0;
        for (int ww = 0; ww < w; ++ww) before = max(before, wtot[ww]);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) rowof[t * ITEMS + i] = (uint16_t)max(v[i], before);
    }
    __syncthreads();  // (buf is free again: the histogram has been read)
    // ---- two stable LSD passes over the 14 key bits
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = kSmSlotBits + pass * DB;
        for (int i = t; i < kSmWaves * BINS; i += kSmThreads) (&cnt[0][0])[i] = 0;
        __syncthreads();
        uint16_t rank[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t d = (word[j] >> shift) & (BINS - 1);
            const uint64_t peers = match_digit(d, DB, true);
            const int lower = __popcll(peers & lt);
            const uint32_t before = cnt[w][d];  // (LDS operations of a wave complete in order: read, then the leader's write)
            rank[j] = (uint16_t)(before + lower);
            if (lower == 0) cnt[w][d] = before + (uint32_t)__popcll(peers);
        }
        __syncthreads();
        uint32_t tot = 0;
        if (t < BINS) {  // thread t owns digit t: every wave's start inside the digit's run
#pragma unroll
            for (int ww = 0; ww < kSmWaves; ++ww) {
                const uint32_t c = cnt[ww][t];
                cnt[ww][t] = tot;
                tot += c;
            }
        }
        uint32_t incl = tot;  // exclusive scan of the totals over the 128 digits (waves 0 and 1)
#pragma unroll
        for (int sft = 1; sft < kWave; sft <<= 1) {
            const uint32_t u = __shfl_up(incl, sft, kWave);
            if (lane >= sft) incl += u;
        }
        if (t < BINS && lane == kWave - 1) wtot[w] = incl;
        __syncthreads();
        if (t < BINS) dstart[t] = incl - tot + (w == 1 ? wtot[0] : 0u);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t d = (word[j] >> shift) & (BINS - 1);
            buf[dstart[d] + cnt[w][d] + rank[j]] = word[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) word[j] = buf[w * (ITEMS * kWave) + j * kWave + lane];
        __syncthreads();
    }
    // ---- outputs: position p = 1024 w + 64 j + l of the sorted order
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int pos = w * (ITEMS * kWave) + j * kWave + lane;
        const uint32_t e = word[j] & ((1u << kSmSlotBits) - 1u);
        if (pos < cap && (word[j] >> kSmSlotBits) <= pad_key) {
            perm[pos] = (int32_t)e;
            rowind[pos] = (int32_t)rowof[e];
        }
    }
}

struct Geometry {
    int bits, n_pass, dbits;
    int64_t n_tiles, table_len;
    size_t scan_temp, off_table, off_scanned, off_temp, off_sets, set_bytes, off_keys_final, off_tile_rows, total;
    int n_sets;
};

static unsigned bits_for(int64_t n_keys) {  // enough bits for key values 0 .. n_keys - 1
    unsigned b = 1;
    while (b < 32 && (int64_t(1) << b) < n_keys) ++b;
    return b;
}

static Geometry geometry(int64_t n_cols, int64_t nnz, bool padded) {
    Geometry g{};
    g.bits = (int)bits_for(padded ? n_cols + 1 : n_cols);
    g.n_pass = (g.bits + kMaxBits - 1) / kMaxBits;
    g.dbits = (g.bits + g.n_pass - 1) / g.n_pass;
    g.n_tiles = (nnz + kTile - 1) / kTile;
    g.table_len = g.n_tiles << g.dbits;
    g.scan_temp = device_scan_temp_bytes(g.table_len, sizeof(uint32_t));
    g.n_sets = std::min(g.n_pass - 1, 2);
    g.set_bytes = 3 * up256((size_t)nnz * 4);
    size_t o = 0;
    g.off_table = o;
    o += up256((size_t)g.table_len * 4);
    g.off_scanned = o;
    o += up256((size_t)g.table_len * 4);
    g.off_temp = o;
    o += up256(g.scan_temp);
    g.off_keys_final = o;
    o += up256((size_t)nnz * 4);
    g.off_tile_rows = o;
    o += up256((size_t)g.n_tiles * 8);
    g.off_sets = o;
    o += (size_t)g.n_sets * g.set_bytes;
    g.total = o + 256;
    return g;
}

}  // namespace rt

size_t radix_transpose_workspace_bytes(int64_t n_cols, int64_t nnz, bool padded) {
    if (nnz <= 0) return 256;
    return rt::geometry(n_cols, nnz, padded).total;
}

// -> colptr, rowind, perm.  nnz > 0, m > 0.
int radix_transpose(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols, int64_t nnz, bool padded,
                    int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace, hipStream_t s) {
    using namespace rt;
    const Geometry g = geometry(n_cols, nnz, padded);
    const int row_bits = (int)bits_for(m);
    const int tune = g_tuning[kTuneCsr2csc];
    // Two passes (column ids of up to 18 bits) with a digit and the row fitting one word: PACKED intermediate records
    // (LSD: packed records from 16 M slots on: below that the per-column searches of rt_colptr_perm cost more than the
    //  one staged array they save -- arxiv-shaped, 2.5 M slots: 125 vs 117 us; Reddit-shaped, 115 M: 2.00 vs 2.06 ms)
    const bool packed = g.n_pass == 2 && row_bits + g.dbits <= 32 && (nnz >= (int64_t(1) << 24) || tune == 3);
    if (g.n_tiles > 0x7fffffff || g.table_len > 0x7fffffff || m > 0x7fff0000) return COGDL_HIP_ERANGE;  // (int row loops)
    char *ws = (char *)workspace;
    uint32_t *table = (uint32_t *)(ws + g.off_table), *scanned = (uint32_t *)(ws + g.off_scanned);
    void *temp = ws + g.off_temp;
    uint32_t *keys_final = (uint32_t *)(ws + g.off_keys_final);
    auto set_ptr = [&](int idx, int which) { return ws + g.off_sets + (size_t)idx * g.set_bytes + (size_t)which * up256((size_t)nnz * 4); };
    auto scan = [&](const uint32_t *in, uint32_t *out, int64_t len) -> bool {
        return device_exclusive_sum(in, out, len, temp, s) == COGDL_HIP_OK;
    };
    Params p{};
    p.colind = colind;
    p.rowptr = rowptr;
    p.m = m;
    p.nnz = nnz;
    p.n_tiles = g.n_tiles;
    p.pad_key = (uint32_t)n_cols;
    p.tile_rows = (int32_t *)(ws + g.off_tile_rows);
    p.padded = padded ? 1 : 0;
    p.bits = g.dbits;
    p.row_bits = row_bits;
    const unsigned dgrid = (unsigned)(((g.n_tiles + kXcds - 1) / kXcds) * kXcds);  // (XCD-contiguous tile ranges)
    for (int pass = 0; pass < g.n_pass; ++pass) {
        const bool first = pass == 0, last = pass == g.n_pass - 1;
        p.shift = pass * g.dbits;
        if (!first) {
            const int src = (pass - 1) % 2;
            p.keys_in = (const uint32_t *)set_ptr(src, 0);
            p.e_in = (const int32_t *)set_ptr(src, 1);
            p.row_in = (const int32_t *)set_ptr(src, 2);
        }
        if (last) {
            p.keys_out = keys_final;
            p.e_out = perm;
            p.row_out = rowind;
        } else {
            const int dst = pass % 2;
            p.keys_out = (uint32_t *)set_ptr(dst, 0);
            p.e_out = (int32_t *)set_ptr(dst, 1);
            p.row_out = (int32_t *)set_ptr(dst, 2);
        }
        p.table = table;
        if (packed && !first) p.shift = 0;  // (the packed word holds the remaining digit at its bottom)
        if (first) hipLaunchKernelGGL((rt_upsweep_kernel<true, false>), dim3((unsigned)g.n_tiles), dim3(kThreads), 0, s, p);
        else if (packed) hipLaunchKernelGGL((rt_upsweep_kernel<false, true>), dim3((unsigned)g.n_tiles), dim3(kThreads), 0, s, p);
        else hipLaunchKernelGGL((rt_upsweep_kernel<false, false>), dim3((unsigned)g.n_tiles), dim3(kThreads), 0, s, p);
        if (!scan(table, scanned, g.n_tiles << g.dbits)) return COGDL_HIP_ELAUNCH;
        p.table = scanned;
        if (packed) {
            if (first) hipLaunchKernelGGL((rt_downsweep_kernel<true, 1>), dim3(dgrid), dim3(kThreads), 0, s, p);
            else hipLaunchKernelGGL((rt_downsweep_kernel<false, 2>), dim3(dgrid), dim3(kThreads), 0, s, p);
        } else {
            if (first) hipLaunchKernelGGL((rt_downsweep_kernel<true, 0>), dim3(dgrid), dim3(kThreads), 0, s, p);
            else hipLaunchKernelGGL((rt_downsweep_kernel<false, 0>), dim3(dgrid), dim3(kThreads), 0, s, p);
        }
    }
    const unsigned cb = (unsigned)std::min<int64_t>((n_cols + 256) / 256, 1 << 20);
    if (!colptr) return launch_status();  // (a caller that only wants the stable order: the sampler's position sort)
    if (packed)
        hipLaunchKernelGGL(rt_colptr_perm_kernel, dim3(cb), dim3(256), 0, s, colind, perm, rowptr, m, padded ? 1 : 0, nnz,
                           n_cols, colptr);
    else
        hipLaunchKernelGGL(rt_colptr_sorted_kernel, dim3(cb), dim3(256), 0, s, keys_final, colptr, nnz, n_cols);
    return launch_status();
}

bool small_transpose_covers(int64_t m, int64_t n_cols, int64_t nnz) {
    return nnz <= rt::kSmTile && n_cols + 2 <= (int64_t(1) << rt::kSmSlotBits) && m >= 1 && m <= 65535;
}

// One launch, no workspace.  Preconditions: small_transpose_covers(m, n_cols, nnz).
int small_transpose(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols, int64_t nnz, bool padded,
                    int32_t *colptr, int32_t *rowind, int32_t *perm, hipStream_t s) {
    const int64_t need = std::max(nnz, n_cols + 2);  // slots and keys one tile must hold
#define COGDL_SMALL_TRANSPOSE(ITEMS, DB)                                                                                   \
    hipLaunchKernelGGL((rt::small_transpose_kernel<ITEMS, DB>), dim3(1), dim3(rt::kSmThreads), 0, s, rowptr, colind, (int)m, \
                       (int)n_cols, (int)nnz, padded ? 1 : 0, colptr, rowind, perm)
    if (need <= 2048) COGDL_SMALL_TRANSPOSE(2, 6);
    else if (need <= 4096) COGDL_SMALL_TRANSPOSE(4, 6);
    else if (need <= 8192) COGDL_SMALL_TRANSPOSE(8, 7);
    else if (need <= 12288) COGDL_SMALL_TRANSPOSE(12, 7);
    else COGDL_SMALL_TRANSPOSE(16, 7);
#undef COGDL_SMALL_TRANSPOSE
    return launch_status();
}

}  // namespace cogdl
