// edge_softmax_flat.hip -- edge_softmax forward / backward as a SEGMENTED softmax over the flat [E*H] value array,
// gfx950.  Replaces cogdl/operators/edge_softmax/edge_softmax.cu:7-98 (one 32 x H block per row, three passes over the
// row) for H a power of two <= 64 and f32 / f16 / bf16 values; every other shape keeps the row kernels of
// edge_softmax.hip.
//
// Why flat: a CSR row owns a contiguous run of deg*H values, so the whole operator is a streaming pass over one
// array with segment boundaries from rowptr.  Work is cut into TILES of a fixed number of elements (not rows):
//   * every workgroup moves the same bytes, whatever the degree distribution (Reddit-shaped graphs: 82 % of the
//     edges sit in rows of > 1024 edges, the longest has 10^5);
//   * global loads and stores are full 16-byte-per-lane vectors, 8 of them in flight per lane;
//   * the values are read from HBM exactly ONCE and written once (algorithmic bytes E*H*2s forward, E*H*3s backward):
//     the tile lives in LDS between the statistics and the apply step.
// Inside a tile (LDS resident): rows that lie completely in the tile are reduced by lane groups (short rows) or by
// the whole workgroup (long rows).  A row that crosses tile borders is reduced piecewise: every tile publishes the
// (max, sum) -- backward: the partial <softmax, grad> -- of its piece in a workspace record (one self-flagging 8-byte
// write-through store per head, MI355X_MICROARCH.md "inter-workgroup visibility" recipe R2), reads the records of the
// other pieces of the row until all are there, merges them in tile order (deterministic) and applies the row totals
// to the piece it still holds on chip.
//
// Two launches per call.  The INIT kernel zeroes the piece records, computes per tile the rows of its first and last edge
// (one search per tile, all tiles in parallel -- the main kernel then starts with a 32-byte record instead of two
// dependent searches) and produces the piece records of "super-long" rows (below).  The MAIN kernel runs one tile per
// workgroup, tile = blockIdx.  A tile that lies inside ONE row (the common case on hub-heavy graphs) never touches
// LDS with its values: statistics, exchange and apply work on the registers the loads landed in.
//
// Progress: a piece is published BEFORE its workgroup waits, so a wait only needs the other tiles of the row to START.
// Workgroups start in index order on this hardware (observed, MI355X_MICROARCH.md; not a contract), rows of at most
// K_max = 512 tiles take part in the exchange, and more workgroups than that are resident at once (4 per CU x 256 CUs)
// -- so in practice no wait outlives the start-up skew of neighbouring workgroups.  CORRECTNESS does not depend on any of this: every
// wait is bounded, and on time-out the workgroup recomputes the row statistics from global memory itself (right under
// any scheduling, merely slower; the count of such escapes is kept in the workspace header for the tests).  Rows
// longer than K_max tiles never wait at all: their piece records come from the init kernel, at the price of one extra
// read of those rows only.
#include "rowreduce.h"

#include <type_traits>

namespace cogdl {
namespace esf {

constexpr int kThreads = 256;
constexpr int kRowChunk = 512;    // rowptr entries staged in LDS per pass over a tile's rows
constexpr int kKMax = 512;        // longest row (in tiles) that uses the in-launch exchange
constexpr unsigned kSpinLimit = 1u << 12;

// Elements per tile: 32 KB of LDS and 32 KB (forward: read) / 2 x 16 KB (backward) of loads in flight per workgroup
// whatever the element size.  Round 3: 16-bit values are staged in LDS and held in registers in their OWN type
// (rounds 1-2: widened to fp32, hence 8192-element tiles for every type: a bf16 tile moved half the bytes of an fp32
// one at the same fixed per-tile cost -- publish, poll, merge -- and ran no faster, 1.66 vs 1.78 ms on the Reddit-shaped
// graph); what makes that possible is that nothing is stored back between the statistics and the result any more: the
// forward output is exp(v - max_row) / sum_row computed from the RAW value at the end, not p * exp(max_piece - max_row).
template <bool BWD, int ELEM_BYTES = 4> struct TileSize { static constexpr int value = (BWD ? 16384 : 32768) / ELEM_BYTES; };

__device__ __forceinline__ float es_exp(float x) { return __expf(x); }

// What the main kernel needs to know about its tile, computed once by the init kernel.
struct __attribute__((aligned(16))) TileInfo {
    int32_t r_first, r_last;  // rows of the tile's first and last edge
    int32_t hs, he;           // edge range of row r_first
    int32_t ts, te;           // edge range of row r_last
    int32_t pad0, pad1;
};

struct Params {
    const int32_t *rowptr;
    const void *a;       // values (forward) | softmax output (backward)
    const void *g;       // upstream gradient (backward)
    void *out;
    int64_t m, nnz;
    int h;               // power of two <= 64
    int tile_e;          // edges per tile = tile elements / h
    int64_t n_tiles;
    unsigned *stats;     // workspace header: [0] = number of timed-out waits (zeroed by the init kernel)
    TileInfo *tinfo;     // [n_tiles], written by the init kernel
    float2 *rec;         // [n_tiles * 2][h]: (max, sum) | (dot, 1); zeroed by the init kernel, .y != 0 = published
    int64_t long_edges;  // rows with more edges are "super-long": records come from the init kernel
    unsigned spin_limit; // polls of one flag before the wait gives up (tuning key 8; tests force the escape path)
    int debug;           // bit 0 (tuning key 9, timing experiments, results are WRONG): no cross-tile exchange;
                         // bit 1 (tuning key 8 < 0, tests): every cross-tile wait takes the time-out escape at once
    int n_seg;           // init kernel: leading workgroups that own a segment of K_max tiles
    int info_per_wave;   // init kernel: 1 = one wave per tile (64-ary search), 0 = one thread per tile (binary search)
};


// ---- agent-scope accesses (write-through stores / L1-bypassing loads; see the file header) ---------------------
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_rec(float2 *p, float x, float y) {
    const unsigned long long bits = ((unsigned long long)__float_as_uint(y) << 32) | __float_as_uint(x);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_rec(const float2 *p) {
    const unsigned long long bits =
        __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)bits), __uint_as_float((unsigned)(bits >> 32)));
}

// (max, sum) pairs: b's elements follow a's.
__device__ __forceinline__ float2 ms_combine(float2 a, float2 b) {
    const float mx = fmaxf(a.x, b.x);
    const float sa = (a.y == 0.f) ? 0.f : a.y * es_exp(a.x - mx);
    const float sb = (b.y == 0.f) ? 0.f : b.y * es_exp(b.x - mx);
    return make_float2(mx, sa + sb);
}

// 256-ary search, whole workgroup: the row that contains edge e (rowptr[r] <= e < rowptr[r+1]), 0 <= e < nnz.
// Every round is one L2 load per thread and one barrier (3 rounds up to 16 M rows) instead of ~24 dependent loads.
__device__ __forceinline__ int64_t wg_row_of_edge(const int32_t *__restrict__ rowptr, int64_t m, int64_t e) {
    int64_t lo = 0, hi = m;  // invariant: rowptr[lo] <= e < rowptr[hi]
    while (hi - lo > 1) {
        const int64_t step = (hi - lo + kThreads - 1) / kThreads;
        const int64_t idx = lo + (int64_t)(threadIdx.x + 1) * step;
        const int below = (idx < hi && (int64_t)rowptr[idx] <= e) ? 1 : 0;
        const int cnt = __syncthreads_count(below);  // probes are monotone: the first `cnt` are <= e
        lo = lo + (int64_t)cnt * step;
        hi = min(lo + step, hi);
    }
    return lo;
}

// ---- whole-workgroup reduction of one value per thread over the threads that share a head (t % h) ------------------
// In-wave butterfly over the lane strides >= h, the four wave partials through LDS in wave order.  Result: every
// thread gets the total of its head.  `red` = 4 * 64 floats.
template <bool MAX>
__device__ __forceinline__ float wg_head_reduce(float v, int h, float *red) {
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1)
        if (s >= h) v = MAX ? fmaxf(v, __shfl_xor(v, s, kWave)) : v + __shfl_xor(v, s, kWave);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    __syncthreads();  // the previous use of `red` is over
    if (lane < h) red[wave * kWave + lane] = v;
    __syncthreads();
    const int hd = threadIdx.x & (h - 1);
    float r = red[hd];
#pragma unroll
    for (int w = 1; w < 4; ++w) r = MAX ? fmaxf(r, red[w * kWave + hd]) : r + red[w * kWave + hd];
    return r;
}

// LDS staging type: fp32 values as float, 16-bit values as themselves.
template <typename S> __device__ __forceinline__ float lds_ld(const S *p, int i) { return to_f32<S>(p[i]); }
template <typename S> __device__ __forceinline__ void lds_st(S *p, int i, float v) { p[i] = from_f32<S>(v); }

// Statistics of the LDS span [off, off + cnt) (one row piece; off and cnt multiples of h): thread t walks the
// elements t, t + 256, ... (head t % h).  Forward: (max, sum of exp(v - max)); backward: <a, g> in .x.  The span is NOT
// modified.  All threads return the value of head t % h.
template <bool BWD, typename S>
__device__ __forceinline__ float2 wg_piece_lds(const S *tile, const S *tile_g, int off, int cnt, int h, float *red) {
    const int t = threadIdx.x;
    if constexpr (BWD) {
        float dot = 0.f;
        for (int i = t; i < cnt; i += kThreads) dot = fmaf(lds_ld(tile, off + i), lds_ld(tile_g, off + i), dot);
        return make_float2(wg_head_reduce<false>(dot, h, red), 0.f);
    } else {
        float mx = -INFINITY;
        for (int i = t; i < cnt; i += kThreads) mx = fmaxf(mx, lds_ld(tile, off + i));
        mx = wg_head_reduce<true>(mx, h, red);
        float sum = 0.f;
        for (int i = t; i < cnt; i += kThreads) sum += es_exp(lds_ld(tile, off + i) - mx);
        return make_float2(mx, wg_head_reduce<false>(sum, h, red));
    }
}

// The same statistics straight from global memory, elements [lo, hi) of the flat arrays (time-out escape of the
// exchange and the init kernel's pass over super-long rows).  Nothing is modified.
template <typename T, bool BWD>
__device__ __forceinline__ float2 wg_piece_global(const T *__restrict__ a, const T *__restrict__ g, int64_t lo,
                                                  int64_t hi, int h, float *red) {
    const int t = threadIdx.x;
    constexpr int U = 8;
    if constexpr (BWD) {
        float dot = 0.f;
        for (int64_t i = lo + t; i < hi; i += (int64_t)kThreads * U) {
            float va[U], vg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t idx = i + (int64_t)u * kThreads;
                const bool ok = idx < hi;
                va[u] = ok ? to_f32<T>(a[idx]) : 0.f;
                vg[u] = ok ? to_f32<T>(g[idx]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) dot = fmaf(va[u], vg[u], dot);
        }
        return make_float2(wg_head_reduce<false>(dot, h, red), 0.f);
    } else {
        float mx = -INFINITY, sum = 0.f;  // online (max, sum), one rescale per batch of U
        for (int64_t i = lo + t; i < hi; i += (int64_t)kThreads * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t idx = i + (int64_t)u * kThreads;
                v[u] = idx < hi ? to_f32<T>(a[idx]) : -INFINITY;
            }
            float bm = v[0];
#pragma unroll
            for (int u = 1; u < U; ++u) bm = fmaxf(bm, v[u]);
            const float mn = fmaxf(mx, bm);
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) s += es_exp(v[u] - mn);
            sum = ((sum == 0.f) ? 0.f : sum * es_exp(mx - mn)) + s;
            mx = mn;
        }
        const float gm = wg_head_reduce<true>(mx, h, red);
        const float mine = (sum == 0.f) ? 0.f : sum * es_exp(mx - gm);
        return make_float2(gm, wg_head_reduce<false>(mine, h, red));
    }
}

// Publish the record of (tile, slot): threads 0..h-1 store their head's pair as ONE aligned 8-byte write-through
// store.  The record is its own flag ("granule", MI355X_MICROARCH.md recipe R2): the init kernel zeroes every record
// and a published one never has .y == 0 (forward: .y = sum of exp(v - max) >= 1; backward: .y = 1), so a reader that
// sees .y != 0 sees the whole record -- no drain of the store queue, no separate flag, no counter.
__device__ __forceinline__ int slot_of(int64_t start, int64_t c, int tile_e) { return start <= c * tile_e ? 0 : 1; }
template <bool BWD>
__device__ __forceinline__ void publish(const Params &p, int64_t tile, int slot, float2 mine) {
    if (p.debug & 1) return;
    if ((int)threadIdx.x < p.h) st_rec(p.rec + (tile * 2 + slot) * p.h + threadIdx.x, mine.x, BWD ? 1.f : mine.y);
}

// (slot_of: record slot of a row in tile c -- the row is known to intersect c and to extend beyond it.)

// Totals of the row [start, end) that crosses tile borders: wait for the records of all its pieces, merge them in
// tile order.  Returns, for the head t % h of every thread, (max, sum) | (dot, -).  Returns false on time-out.
// `mrg` = 256 float2 of LDS.
template <bool BWD>
__device__ __forceinline__ bool row_totals(const Params &p, int64_t start, int64_t end, float2 *mrg, float2 &tot) {
    const int64_t c_a = start / p.tile_e, c_b = (end - 1) / p.tile_e;
    const int k = (int)(c_b - c_a + 1);
    const int t = threadIdx.x;
    if (p.debug & 1) {
        tot = BWD ? make_float2(0.f, 0.f) : make_float2(0.f, 1.f);
        return true;
    }
    if (p.debug & 2) {  // tests: take the time-out escape on every wait
        if (t == 0) atomicAdd(p.stats, 1u);
        return false;
    }
    const int slot_a = slot_of(start, c_a, p.tile_e);
    // thread (q0 = t / h, hd = t % h) fetches the records q0, q0 + 256/h, ... of its head, 8 loads in flight, and
    // re-reads only while one of ITS records is still unpublished (.y == 0); then the 256/h partials are merged in
    // order.  The grouping is a fixed function of (k, h): every tile of the row computes the identical totals.
    const int h = p.h, hd = t & (h - 1), q0 = t / h, nq = kThreads / h;
    float2 acc = BWD ? make_float2(0.f, 0.f) : make_float2(-INFINITY, 0.f);
    constexpr int U = 8;
    bool ok = true;
    for (int qb = q0; qb < k && ok; qb += nq * U) {
        float2 r[U];
        unsigned spins = 0;
        for (;;) {
            bool all = true;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = min(qb + u * nq, k - 1);  // clamped: branch-free issue, the surplus is ignored below
                r[u] = ld_rec(p.rec + ((c_a + q) * 2 + ((q == 0) ? slot_a : 0)) * h + hd);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) all = all && (r[u].y != 0.f);
            if (all) break;
            if (++spins > p.spin_limit) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (qb + u * nq < k) {
                if constexpr (BWD) acc.x += r[u].x;
                else acc = ms_combine(acc, r[u]);
            }
        }
    }
    if (!__syncthreads_and(ok ? 1 : 0)) {
        if (t == 0) atomicAdd(p.stats, 1u);
        return false;
    }
    __syncthreads();
    mrg[t] = acc;
    __syncthreads();
    float2 r = mrg[hd];
    for (int q = 1; q < nq && q < k; ++q) {
        if constexpr (BWD) r.x += mrg[q * h + hd].x;
        else r = ms_combine(r, mrg[q * h + hd]);
    }
    tot = r;
    return true;
}

// ---- one complete row in LDS, reduced by an aligned group of `lpr` lanes (a power of two, a multiple of h: lane l sees
// head l % h).  The group size is a run-time value: one copy of this code serves every row length (seven unrolled
// instantiations had grown the forward kernel to ~80 KB of instructions, beyond the instruction cache).
template <bool BWD, typename S>
__device__ __forceinline__ void row_in_lds(S *tile, const S *tile_g, int base, int cnt, int l, int lpr, int h) {
    if constexpr (BWD) {
        float dot = 0.f;
        for (int j = l; j < cnt; j += lpr) dot = fmaf(lds_ld(tile, base + j), lds_ld(tile_g, base + j), dot);
        for (int s = lpr >> 1; s >= h; s >>= 1) dot += __shfl_xor(dot, s, kWave);
        for (int j = l; j < cnt; j += lpr) lds_st(tile, base + j, lds_ld(tile, base + j) * (lds_ld(tile_g, base + j) - dot));
    } else {
        float mx = -INFINITY;
        for (int j = l; j < cnt; j += lpr) mx = fmaxf(mx, lds_ld(tile, base + j));
        for (int s = lpr >> 1; s >= h; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, kWave));
        float sum = 0.f;
        for (int j = l; j < cnt; j += lpr) sum += es_exp(lds_ld(tile, base + j) - mx);
        for (int s = lpr >> 1; s >= h; s >>= 1) sum += __shfl_xor(sum, s, kWave);
        const float inv = 1.f / sum;
        // the FINAL value replaces the raw one (rounded to the output type here, once): the store step copies it
        for (int j = l; j < cnt; j += lpr) lds_st(tile, base + j, es_exp(lds_ld(tile, base + j) - mx) * inv);
    }
}

// Rows of at most `lthr` edges: one lane group per row.  rp = LDS copy of rowptr[r0 .. r0 + nrows].
template <bool BWD, typename S>
__device__ __forceinline__ void rows_small(S *tile, const S *tile_g, const int32_t *rp, int nrows, int64_t e0,
                                           int h, int lpr, int lthr) {
    const int ng = kThreads / lpr;
    const int grp = threadIdx.x / lpr, l = threadIdx.x & (lpr - 1);
    for (int i = grp; i < nrows; i += ng) {
        const int len = rp[i + 1] - rp[i];
        if (len == 0 || len > lthr) continue;  // (group-uniform)
        row_in_lds<BWD, S>(tile, tile_g, (int)(rp[i] - e0) * h, len * h, l, lpr, h);
    }
}

// ---- 16-bit values, h >= 2: two elements per LDS access --------------------------------------------------------------
// A 32-bit LDS word holds two consecutive elements = the heads (2q % h, 2q % h + 1) of ONE edge (rows start at
// multiples of h, h even).  A lane that walks words keeps two statistics and touches LDS half as often; lanes whose
// index agrees modulo h / 2 hold the same pair of heads.  (With one element per lane and access the LDS passes of the
// multi-row tiles were what bounded the bf16 kernels once a tile held 16 k elements.)
template <typename S>
__device__ __forceinline__ void lds_ld2(const S *p, int word, float &a, float &b) {
    union { uint32_t raw; S e[2]; } u;
    u.raw = reinterpret_cast<const uint32_t *>(p)[word];
    a = to_f32<S>(u.e[0]);
    b = to_f32<S>(u.e[1]);
}
template <typename S>
__device__ __forceinline__ void lds_st2(S *p, int word, float a, float b) {
    union { uint32_t raw; S e[2]; } u;
    u.e[0] = from_f32<S>(a);
    u.e[1] = from_f32<S>(b);
    reinterpret_cast<uint32_t *>(p)[word] = u.raw;
}

// wg_piece_lds with two elements per access.  Same contract: every thread returns the statistics of head t % h.
template <bool BWD, typename S>
__device__ __forceinline__ float2 wg_piece_lds2(const S *tile, const S *tile_g, int off, int cnt, int h, float *red) {
    const int t = threadIdx.x, hp = h >> 1;
    const int w0 = off >> 1, nw = cnt >> 1;  // thread t: words t, t + 256, ... = heads (2t % h, 2t % h + 1)
    float r0, r1, q0 = 0.f, q1 = 0.f;
    if constexpr (BWD) {
        float d0 = 0.f, d1 = 0.f;
        for (int i = t; i < nw; i += kThreads) {
            float a0, a1, g0, g1;
            lds_ld2(tile, w0 + i, a0, a1);
            lds_ld2(tile_g, w0 + i, g0, g1);
            d0 = fmaf(a0, g0, d0);
            d1 = fmaf(a1, g1, d1);
        }
        r0 = wg_head_reduce<false>(d0, hp, red);
        r1 = wg_head_reduce<false>(d1, hp, red);
    } else {
        float m0 = -INFINITY, m1 = -INFINITY;
        for (int i = t; i < nw; i += kThreads) {
            float a0, a1;
            lds_ld2(tile, w0 + i, a0, a1);
            m0 = fmaxf(m0, a0);
            m1 = fmaxf(m1, a1);
        }
        m0 = wg_head_reduce<true>(m0, hp, red);
        m1 = wg_head_reduce<true>(m1, hp, red);
        float s0 = 0.f, s1 = 0.f;
        for (int i = t; i < nw; i += kThreads) {
            float a0, a1;
            lds_ld2(tile, w0 + i, a0, a1);
            s0 += es_exp(a0 - m0);
            s1 += es_exp(a1 - m1);
        }
        r0 = m0;
        r1 = m1;
        q0 = wg_head_reduce<false>(s0, hp, red);
        q1 = wg_head_reduce<false>(s1, hp, red);
    }
    // thread t holds the heads (2 (t % hp), 2 (t % hp) + 1): hand them over so that every thread returns head t % h
    __syncthreads();  // (the reads of `red` by the last reduction are over)
    if (t < hp) {
        red[2 * t] = r0;
        red[2 * t + 1] = r1;
        red[kWave + 2 * t] = q0;
        red[kWave + 2 * t + 1] = q1;
    }
    __syncthreads();
    const int hd = t & (h - 1);
    return make_float2(red[hd], red[kWave + hd]);
}

// row_in_lds with two elements per access: `lpr` lanes (a power of two, a multiple of h / 2) per row, lane l walks the
// words l, l + lpr, ... of the row = heads (2l % h, 2l % h + 1).
template <bool BWD, typename S>
__device__ __forceinline__ void row_in_lds2(S *tile, const S *tile_g, int base, int cnt, int l, int lpr, int h) {
    const int hp = h >> 1, w0 = base >> 1, nw = cnt >> 1;
    if constexpr (BWD) {
        float d0 = 0.f, d1 = 0.f;
        for (int j = l; j < nw; j += lpr) {
            float a0, a1, g0, g1;
            lds_ld2(tile, w0 + j, a0, a1);
            lds_ld2(tile_g, w0 + j, g0, g1);
            d0 = fmaf(a0, g0, d0);
            d1 = fmaf(a1, g1, d1);
        }
        for (int s = lpr >> 1; s >= hp && s > 0; s >>= 1) {
            d0 += __shfl_xor(d0, s, kWave);
            d1 += __shfl_xor(d1, s, kWave);
        }
        for (int j = l; j < nw; j += lpr) {
            float a0, a1, g0, g1;
            lds_ld2(tile, w0 + j, a0, a1);
            lds_ld2(tile_g, w0 + j, g0, g1);
            lds_st2(tile, w0 + j, a0 * (g0 - d0), a1 * (g1 - d1));
        }
    } else {
        float m0 = -INFINITY, m1 = -INFINITY;
        for (int j = l; j < nw; j += lpr) {
            float a0, a1;
            lds_ld2(tile, w0 + j, a0, a1);
            m0 = fmaxf(m0, a0);
            m1 = fmaxf(m1, a1);
        }
        for (int s = lpr >> 1; s >= hp && s > 0; s >>= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, s, kWave));
            m1 = fmaxf(m1, __shfl_xor(m1, s, kWave));
        }
        float s0 = 0.f, s1 = 0.f;
        for (int j = l; j < nw; j += lpr) {
            float a0, a1;
            lds_ld2(tile, w0 + j, a0, a1);
            s0 += es_exp(a0 - m0);
            s1 += es_exp(a1 - m1);
        }
        for (int s = lpr >> 1; s >= hp && s > 0; s >>= 1) {
            s0 += __shfl_xor(s0, s, kWave);
            s1 += __shfl_xor(s1, s, kWave);
        }
        const float i0 = 1.f / s0, i1 = 1.f / s1;
        for (int j = l; j < nw; j += lpr) {
            float a0, a1;
            lds_ld2(tile, w0 + j, a0, a1);
            lds_st2(tile, w0 + j, es_exp(a0 - m0) * i0, es_exp(a1 - m1) * i1);
        }
    }
}

template <bool BWD, typename S>
__device__ __forceinline__ void rows_small2(S *tile, const S *tile_g, const int32_t *rp, int nrows, int64_t e0, int h,
                                            int lpr, int lthr) {
    const int ng = kThreads / lpr;
    const int grp = threadIdx.x / lpr, l = threadIdx.x & (lpr - 1);
    for (int i = grp; i < nrows; i += ng) {
        const int len = rp[i + 1] - rp[i];
        if (len == 0 || len > lthr) continue;  // (group-uniform)
        row_in_lds2<BWD, S>(tile, tile_g, (int)(rp[i] - e0) * h, len * h, l, lpr, h);
    }
}

// ---- 16-byte global vector <-> V floats ------------------------------------------------------------------------
template <typename T> struct VecOf { static constexpr int V = 16 / sizeof(T); };

// A thread's piece of a tile stays in registers as RAW 16-byte vectors (V elements of T each) and is unpacked where it
// is used: a 16 k-element bf16 tile costs the 32 VGPRs per array an 8 k-element fp32 tile does.
template <typename T>
__device__ __forceinline__ void unpack16(const uint4 &q, float (&v)[VecOf<T>::V]) {
    union { uint4 raw; T e[VecOf<T>::V]; } u;
    u.raw = q;
#pragma unroll
    for (int i = 0; i < VecOf<T>::V; ++i) v[i] = to_f32<T>(u.e[i]);
}
template <typename T>
__device__ __forceinline__ uint4 pack16(const float (&v)[VecOf<T>::V]) {
    union { uint4 raw; T e[VecOf<T>::V]; } u;
#pragma unroll
    for (int i = 0; i < VecOf<T>::V; ++i) u.e[i] = from_f32<T>(v[i]);
    return u.raw;
}
// the last (partial) tile: element-wise guarded, masked elements 0
template <typename T>
__device__ __forceinline__ uint4 load16_guarded(const T *p, int i0, int count) {
    union { uint4 raw; T e[VecOf<T>::V]; } u;
#pragma unroll
    for (int k = 0; k < VecOf<T>::V; ++k) u.e[k] = (i0 + k < count) ? p[i0 + k] : from_f32<T>(0.f);
    return u.raw;
}

// Reduction of V per-thread slot values over the workgroup, slot k of thread t belonging to head (t*V + k) % h
// (the register layout of a tile: thread t holds the elements (j*256 + t)*V + k).  On return every slot holds the
// total of its head.  `red` = 4 * 64 floats.
template <int V, bool MAX>
__device__ __forceinline__ void wg_slot_reduce(float (&x)[V], int h, float *red) {
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    if (h < V) {  // several slots of one thread share a head: fold them first (slots >= h become copies)
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (k >= h) x[k & (h - 1)] = MAX ? fmaxf(x[k & (h - 1)], x[k]) : x[k & (h - 1)] + x[k];
    }
    const int period = h > V ? h / V : 1;  // lanes with equal (lane % period) hold the same heads
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) {
        if (s >= period) {
#pragma unroll
            for (int k = 0; k < V; ++k) x[k] = MAX ? fmaxf(x[k], __shfl_xor(x[k], s, kWave)) : x[k] + __shfl_xor(x[k], s, kWave);
        }
    }
    __syncthreads();  // the previous use of `red` is over
    if (lane < period) {
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (k < h) red[wave * kWave + ((lane * V + k) & (h - 1))] = x[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int hd = (t * V + k) & (h - 1);
        float r = red[hd];
#pragma unroll
        for (int w = 1; w < 4; ++w) r = MAX ? fmaxf(r, red[w * kWave + hd]) : r + red[w * kWave + hd];
        x[k] = r;
    }
}

// =====================================================================================================================
// main kernel: one tile per workgroup
// TILE elements per workgroup, WPE waves per SIMD the register allocation aims at (= workgroups per CU):
//   default      32 KB of values per tile and array, 4 workgroups per CU (<= 128 VGPRs, <= 40 KB LDS)
//   16-bit alt   half of that per tile, 6 workgroups per CU (<= 84 VGPRs, <= 26 KB LDS): tuning key 9 bit 2, A/B runs
//   KEEP_P       forward, tiles inside ONE row: exp(v - max_piece) of the thread's elements stays in fp32 registers and the
//                output is p * exp(max_piece - max_row) / sum_row -- one exponential per element instead of two, at the
//                price of TILE / 256 more live registers (the default forward; tuning key 9 bit 3 = off, A/B runs)
//   SPLIT        (round 5, opt-in experiment: tuning key 9 bit 8, measured slower -- see es_flat_launch) the tiles that lie inside ONE row -- and the partial pieces of rows that
//                cross tile borders -- are left to es_stream_kernel (below): this kernel returns at once on a one-row tile
//                (its tile record is read BEFORE the value loads are issued), publishes the statistics of its partial
//                pieces without waiting for anybody, and does not store those pieces.  No wait is left in this kernel.
template <typename T, bool BWD, int TILE, int WPE, bool KEEP_P = false, bool SPLIT = false>
__global__ __launch_bounds__(kThreads, WPE) void es_flat_kernel(const Params p) {
    constexpr int V = VecOf<T>::V;
    constexpr int NV = TILE / (kThreads * V);  // 16-byte vectors per thread (per array): 8 forward, 4 + 4 backward
    constexpr int kMaxLong = 512;
    using S = T;                               // LDS staging type = the value type (fp32 as float, 16-bit as itself)
    __shared__ __attribute__((aligned(16))) S tile[TILE];
    __shared__ __attribute__((aligned(16))) S tile_gs[BWD ? TILE : 16 / sizeof(S)];
    __shared__ int32_t rp[kRowChunk + 1];
    __shared__ float red[4 * kWave];
    __shared__ float2 mrg[kThreads];
    __shared__ float2 pstat[kWave];   // piece statistics per head (register path)
    __shared__ float fac[2][kWave];   // per partial piece (head, tail) and head id: forward 1 / sum_row, backward dot_row
    __shared__ float facm[2][kWave];  // forward: the row maximum
    __shared__ int s_nlong;
    __shared__ uint16_t s_long[kMaxLong];  // (indices into rp: < kRowChunk)
    const int t = threadIdx.x;
    const T *__restrict__ a = (const T *)p.a;
    const T *__restrict__ g = (const T *)p.g;
    T *__restrict__ out = (T *)p.out;
    const int h = p.h;
    const int64_t c = blockIdx.x;
    const int64_t e0 = c * p.tile_e, e1 = min(p.nnz, e0 + p.tile_e);
    const int64_t b0 = e0 * h;
    const int count = (int)((e1 - e0) * h);
    const bool full = count == TILE;

    // Phase stamps (tuning key 9 bit 7, measurement only): thread 0 of every 61st one-row tile adds the 10 ns ticks it spent
    // between the marks to stats[8 + k], stats[15] counts the tiles.  The waits are forced at the marks.
    const bool stamp = (p.debug & 128) && t == 0 && (c % 61) == 17;
    long long tk = stamp ? wall_clock64() : 0;
    auto mark = [&](int k) {
        if (stamp) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const long long now = wall_clock64();
            atomicAdd(p.stats + 8 + k, (unsigned)(now - tk));
            tk = now;
        }
    };
    TileInfo ti_early{};
    if constexpr (SPLIT) {
        ti_early = p.tinfo[c];
        if (ti_early.r_first == ti_early.r_last) return;  // a one-row tile: es_stream_kernel's
    }
    // ---- 1. the tile's values, 16 bytes per lane per load, all loads in flight before anything waits ------------
    uint4 ra[NV];
    uint4 rg[BWD ? NV : 1];
    if (full) {
#pragma unroll
        for (int j = 0; j < NV; ++j) ra[j] = *reinterpret_cast<const uint4 *>(a + b0 + (int64_t)(j * kThreads + t) * V);
        if constexpr (BWD) {
#pragma unroll
            for (int j = 0; j < NV; ++j) rg[j] = *reinterpret_cast<const uint4 *>(g + b0 + (int64_t)(j * kThreads + t) * V);
        }
    } else {  // the last tile: element-wise guarded (masked elements: 0)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            ra[j] = load16_guarded<T>(a + b0, (j * kThreads + t) * V, count);
            if constexpr (BWD) rg[j] = load16_guarded<T>(g + b0, (j * kThreads + t) * V, count);
        }
    }
    // ---- 2. the rows of the tile (precomputed by the init kernel) ----------------------------------------------
    const TileInfo ti = SPLIT ? ti_early : p.tinfo[c];
    const int64_t r_first = ti.r_first, r_last = ti.r_last;
    const int64_t hs = ti.hs, he = ti.he, ts = ti.ts, te = ti.te;

    if (r_first == r_last) {
        // (the guards of the last, partial tile cost two VALU operations per element and pass: compiled out for full tiles)
        auto one_row_tile = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        // ================= the whole tile is one piece of one row: everything stays in registers =================
        const bool partial = hs < e0 || he > e1;
        mark(0);  // loads + tile record landed (thread 0's)
        float x[V];
        float pv[(KEEP_P && !BWD) ? NV : 1][V];
        float2 piece = make_float2(0.f, 0.f);  // this thread's view: statistics of head t % h (threads < h publish)
        if constexpr (BWD) {
#pragma unroll
            for (int k = 0; k < V; ++k) x[k] = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float va[V], vg[V];
                unpack16<T>(ra[j], va);
                unpack16<T>(rg[j], vg);
#pragma unroll
                for (int k = 0; k < V; ++k) x[k] = fmaf(va[k], vg[k], x[k]);  // masked elements are 0 * 0
            }
            wg_slot_reduce<V, false>(x, h, red);  // x[k] = <a, g> of head (t*V + k) % h over the tile
            if (t * V < h || (h < V && t == 0)) {
#pragma unroll
                for (int k = 0; k < V; ++k)
                    if (t * V + k < h) pstat[t * V + k] = make_float2(x[k], 0.f);
            }
        } else {
            float mxs[V];
#pragma unroll
            for (int k = 0; k < V; ++k) mxs[k] = -INFINITY;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float va[V];
                unpack16<T>(ra[j], va);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const bool valid = FULL || ((j * kThreads + t) * V + k) < count;
                    mxs[k] = fmaxf(mxs[k], valid ? va[k] : -INFINITY);
                }
            }
            wg_slot_reduce<V, true>(mxs, h, red);
            mark(1);  // maximum pass + workgroup reduction
#pragma unroll
            for (int k = 0; k < V; ++k) x[k] = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float va[V];
                unpack16<T>(ra[j], va);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const bool valid = FULL || ((j * kThreads + t) * V + k) < count;
                    const float pe = valid ? es_exp(va[k] - mxs[k]) : 0.f;
                    if constexpr (KEEP_P && !BWD) pv[j][k] = pe;
                    x[k] += pe;
                }
            }
            wg_slot_reduce<V, false>(x, h, red);  // x[k] = sum of exp(v - max) of its head
            // hand (max, sum) per head to the threads that publish / finish: slot k of thread t is head (t*V+k) % h
            if (t * V < h || (h < V && t == 0)) {
#pragma unroll
                for (int k = 0; k < V; ++k)
                    if (t * V + k < h) pstat[t * V + k] = make_float2(mxs[k], x[k]);
            }
        }
        __syncthreads();
        mark(2);  // exp pass + workgroup reduction
        if (t < h) piece = pstat[t];
        float2 tot = piece;
        if (partial) {
            if (he - hs <= p.long_edges) publish<BWD>(p, c, 0, piece);  // super-long rows: published by the init kernel
            if (!row_totals<BWD>(p, hs, he, mrg, tot)) tot = wg_piece_global<T, BWD>(a, g, hs * h, he * h, h, red);
        }
        if (t < h) {
            if constexpr (KEEP_P && !BWD) fac[0][t] = es_exp(piece.x - tot.x) / tot.y;  // what p = exp(v - max_piece) is scaled by
            else fac[0][t] = BWD ? tot.x : 1.f / tot.y;
            facm[0][t] = tot.x;
        }
        __syncthreads();
        mark(3);  // publish + wait for the other pieces + merge
        float f[V], fm[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            f[k] = fac[0][(t * V + k) & (h - 1)];
            fm[k] = facm[0][(t * V + k) & (h - 1)];
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i0 = (j * kThreads + t) * V;
            float va[V], o[V];
            unpack16<T>(ra[j], va);
            if constexpr (BWD) {
                float vg[V];
                unpack16<T>(rg[j], vg);
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = va[k] * (vg[k] - f[k]);
            } else if constexpr (KEEP_P) {
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = pv[j][k] * f[k];
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = es_exp(va[k] - fm[k]) * f[k];
            }
            if (FULL) {
                store_vec<T, V>(out + b0 + i0, o);
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k)
                    if (i0 + k < count) out[b0 + i0 + k] = from_f32<T>(o[k]);
            }
        }
        mark(4);  // scale + stores (issued and acknowledged)
        if (stamp) atomicAdd(p.stats + 15, 1u);
                };
        if (full) one_row_tile(std::true_type{});
        else one_row_tile(std::false_type{});
        return;
    }

    // ================= several rows in the tile: LDS resident (raw values, in their own type) ====================
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        reinterpret_cast<uint4 *>(tile)[j * kThreads + t] = ra[j];
        if constexpr (BWD) reinterpret_cast<uint4 *>(tile_gs)[j * kThreads + t] = rg[j];
    }
    __syncthreads();
    const S *tile_g = tile_gs;

    // ---- partial pieces: the row of the first edge if it began before the tile, of the last edge if it goes on ----
    const bool head_partial = hs < e0;
    const bool tail_partial = te > e1;
    const int head_end = head_partial ? (int)(he - e0) * h : 0;        // LDS span [0, head_end)
    const int tail_begin = tail_partial ? (int)(ts - e0) * h : count;  // LDS span [tail_begin, count)
    // A SHORT row that straddles a tile border (at most a quarter of a tile: it touches exactly two tiles) is not
    // exchanged at all: both tiles read the whole row from global memory (a few hundred bytes beyond their own tile,
    // L2-hot because the neighbour loads them at the same time) and compute the identical statistics independently --
    // one read instead of publish + poll + merge, and no waiting on another workgroup.  On low-degree graphs
    // (arxiv-shaped: 15 edges per row) every tile border is of this kind.
    const int64_t halo_max = p.tile_e / 4;
    const bool head_halo = head_partial && (he - hs) <= halo_max;
    const bool tail_halo = tail_partial && (te - ts) <= halo_max;
    float2 head_tot = make_float2(0.f, 1.f), tail_tot = make_float2(0.f, 1.f);
    const bool two = sizeof(S) == 2 && h >= 2 && !(p.debug & 16);  // 16-bit values: two elements per LDS access
    auto piece_lds = [&](int off, int cnt) {
        if constexpr (sizeof(S) == 2) {
            if (two) return wg_piece_lds2<BWD, S>(tile, tile_g, off, cnt, h, red);
        }
        return wg_piece_lds<BWD, S>(tile, tile_g, off, cnt, h, red);
    };
    if (head_halo) {
        head_tot = wg_piece_global<T, BWD>(a, g, hs * h, he * h, h, red);
    } else if (head_partial) {
        const float2 piece = piece_lds(0, head_end);
        if (he - hs <= p.long_edges) publish<BWD>(p, c, 0, piece);
    }
    if (tail_halo) {
        tail_tot = wg_piece_global<T, BWD>(a, g, ts * h, te * h, h, red);
    } else if (tail_partial) {
        const float2 piece = piece_lds(tail_begin, count - tail_begin);
        if (te - ts <= p.long_edges) publish<BWD>(p, c, 1, piece);
    }

    // ---- rows completely inside the tile ---------------------------------------------------------------------------
    const int64_t rc0 = r_first + (head_partial ? 1 : 0), rc1 = r_last - (tail_partial ? 1 : 0);  // inclusive range
    if (rc1 >= rc0) {
        // lanes per row from the mean row length of this tile: ~8 sequential steps per lane, at least h lanes (one
        // per head: short rows then need no cross-lane reduction at all, and a workgroup keeps 256 / h rows in flight
        // -- with 8+ lanes per 15-element row the dependent shuffle chains of a few rows were all a wave had to do)
        const int64_t nrows_all = rc1 - rc0 + 1;
        const int mean_elems = (int)min((int64_t)TILE, (int64_t)(tail_begin - head_end) / nrows_all);
        // (two elements per access: half the lanes cover a row, a lane's step is two elements)
        int lpr = two ? max(h >> 1, 1) : h;
        while (lpr < kWave && lpr * (two ? 16 : 8) < mean_elems) lpr <<= 1;
        const int lthr = two ? 64 * lpr / h : 32 * (lpr / h);  // (<= 32 steps per lane) longer rows: one wave each
        for (int64_t r0 = rc0; r0 <= rc1; r0 += kRowChunk) {
            const int nrows = (int)min((int64_t)kRowChunk, rc1 - r0 + 1);
            __syncthreads();
            if (t == 0) s_nlong = 0;
            for (int i = t; i <= nrows; i += kThreads) rp[i] = p.rowptr[r0 + i];
            __syncthreads();
            if constexpr (sizeof(S) == 2) {
                if (two) rows_small2<BWD, S>(tile, tile_g, rp, nrows, e0, h, lpr, lthr);
                else rows_small<BWD, S>(tile, tile_g, rp, nrows, e0, h, lpr, lthr);
            } else {
                rows_small<BWD, S>(tile, tile_g, rp, nrows, e0, h, lpr, lthr);
            }
            // Long complete rows of this chunk: one WAVE per row, four at a time, no workgroup barrier.  A tile holds at
            // most TILE / h / 33 <= 496 of them (more than lthr >= 32 edges each): the list cannot overflow.
            for (int i = t; i < nrows; i += kThreads) {
                if (rp[i + 1] - rp[i] > lthr) {
                    const int pos = atomicAdd(&s_nlong, 1);
                    if (pos < kMaxLong) s_long[pos] = (uint16_t)i;
                }
            }
            __syncthreads();
            const int n_long = min(s_nlong, kMaxLong);
            for (int q = t >> 6; q < n_long; q += 4) {
                const int i2 = s_long[q];
                const int rb = (int)(rp[i2] - e0) * h, rc = (rp[i2 + 1] - rp[i2]) * h;
                if constexpr (sizeof(S) == 2) {
                    if (two) row_in_lds2<BWD, S>(tile, tile_g, rb, rc, t & (kWave - 1), kWave, h);
                    else row_in_lds<BWD, S>(tile, tile_g, rb, rc, t & (kWave - 1), kWave, h);
                } else {
                    row_in_lds<BWD, S>(tile, tile_g, rb, rc, t & (kWave - 1), kWave, h);
                }
            }
        }
    }

    // ---- row totals of the partial pieces -> what the store step applies to their (still raw) values ---------------
    //   forward : out = exp(v - max_row) / sum_row          (facm = max_row, fac = 1 / sum_row)
    //   backward: out = a * (g - dot_row)                   (fac = dot_row)
    // (SPLIT: the exchanged pieces are finished by es_stream_kernel -- nothing to wait for here, and nothing of them is stored)
    const bool skip_head = SPLIT && head_partial && !head_halo, skip_tail = SPLIT && tail_partial && !tail_halo;
    if (!SPLIT && head_partial && !head_halo) {
        if (!row_totals<BWD>(p, hs, he, mrg, head_tot)) head_tot = wg_piece_global<T, BWD>(a, g, hs * h, he * h, h, red);
    }
    if (!SPLIT && tail_partial && !tail_halo) {
        if (!row_totals<BWD>(p, ts, te, mrg, tail_tot)) tail_tot = wg_piece_global<T, BWD>(a, g, ts * h, te * h, h, red);
    }
    if (t < h) {
        fac[0][t] = BWD ? head_tot.x : 1.f / head_tot.y;
        facm[0][t] = head_tot.x;
        fac[1][t] = BWD ? tail_tot.x : 1.f / tail_tot.y;
        facm[1][t] = tail_tot.x;
    }
    __syncthreads();

    // ---- LDS -> global, 16 bytes per lane ---------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i0 = (j * kThreads + t) * V;
        float o[V];
        unpack16<T>(reinterpret_cast<const uint4 *>(tile)[j * kThreads + t], o);
        if (i0 < head_end || i0 + V > tail_begin) {  // (some element of the vector lies in a partial piece)
            float vgs[V];
            if constexpr (BWD) unpack16<T>(reinterpret_cast<const uint4 *>(tile_gs)[j * kThreads + t], vgs);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int i = i0 + k;
                if (i < head_end || i >= tail_begin) {
                    const int which = i < head_end ? 0 : 1;
                    const float f = fac[which][i & (h - 1)];
                    if constexpr (BWD) o[k] = o[k] * (vgs[k] - f);
                    else o[k] = es_exp(o[k] - facm[which][i & (h - 1)]) * f;
                }
            }
        }
        if constexpr (SPLIT) {
            const bool in_head = skip_head && i0 < head_end, in_tail = skip_tail && i0 + V > tail_begin;
            if (in_head || in_tail) {  // (part of) the vector belongs to a piece es_stream_kernel writes
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const int i = i0 + k;
                    const bool skipped = (skip_head && i < head_end) || (skip_tail && i >= tail_begin);
                    if (!skipped && i < count) out[b0 + i] = from_f32<T>(o[k]);
                }
                continue;
            }
        }
        if (full) {
            store_vec<T, V>(out + b0 + i0, o);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (i0 + k < count) out[b0 + i0 + k] = from_f32<T>(o[k]);
        }
    }
}

// =====================================================================================================================
// stream kernel (round 5, forward; OPT-IN, measured slower than the one-kernel form: see es_flat_launch): the one-row tiles
// and the exchanged partial pieces of a SPLIT launch.
// What bounded the one-row tiles inside es_flat_kernel was not memory but concurrency: a tile held its 32 KB in registers
// across the cross-tile exchange (publish, two or three polling rounds, merge: half of an fp32 tile's life, DESIGN
// section 8) next to a 32 KB LDS buffer it never used, four workgroups per CU.  Here a one-row tile
//   1. streams its values ONCE for the online (max, sum) of its piece -- two batches of four 16-byte vectors, ~40 VGPRs,
//      nothing kept --, publishes the piece record,
//   2. waits for the row's other pieces (row_totals, as before),
//   3. reads the tile AGAIN -- 32 KB it read microseconds ago: L2 / Infinity Cache, not HBM -- and stores
//      exp(v - max_row) / sum_row.
// 4 KB of LDS and <= 64 VGPRs: eight workgroups per CU hide the exchange.  A multi-row tile's workgroup finishes the
// tile's exchanged partial pieces (head / tail of rows that cross the tile border; es_flat_kernel<SPLIT> published their
// statistics and left them unwritten): totals, then the piece re-read element-wise (row pieces start at multiples of h
// elements, not of 16 bytes) -- a small fraction of the array.  Waits only concern records of THIS kernel's one-row tiles
// (published before their workgroup waits) or of the previous kernel: bounded as before, same time-out escape.
template <typename T>
__global__ __launch_bounds__(kThreads, 7) void es_stream_kernel(const Params p) {
    constexpr int V = VecOf<T>::V;
    constexpr int kBatch = 16 / V;            // 16-byte vectors in flight per thread and batch (16 unpacked values: 64 VGPRs hold)
    __shared__ float red[4 * kWave];
    __shared__ float2 mrg[kThreads];
    __shared__ float2 pstat[kWave];
    __shared__ float fac[kWave], facm[kWave];
    const int t = threadIdx.x;
    const T *__restrict__ a = (const T *)p.a;
    T *__restrict__ out = (T *)p.out;
    const int h = p.h;
    const int64_t c = blockIdx.x;
    const TileInfo ti = p.tinfo[c];
    const int64_t e0 = c * p.tile_e, e1 = min(p.nnz, e0 + p.tile_e);
    const int64_t b0 = e0 * h;
    const int count = (int)((e1 - e0) * h);
    const int tile_elems = p.tile_e * h;
    const int nv = tile_elems / (kThreads * V);  // vectors per thread of a full tile (8; 2 for quarter tiles)
    const bool full = count == tile_elems;
    if (ti.r_first != ti.r_last) {
        // ---- exchanged partial pieces of a multi-row tile -------------------------------------------------------------
        const int64_t halo_max = p.tile_e / 4;
        for (int which = 0; which < 2; ++which) {
            const int64_t rs = which == 0 ? ti.hs : ti.ts, re = which == 0 ? ti.he : ti.te;
            const bool partial = which == 0 ? rs < e0 : re > e1;
            if (!partial || (re - rs) <= halo_max) continue;  // (workgroup-uniform)
            float2 tot;
            if (!row_totals<false>(p, rs, re, mrg, tot)) tot = wg_piece_global<T, false>(a, nullptr, rs * h, re * h, h, red);
            __syncthreads();  // (fac / facm of the previous piece have been read)
            if (t < h) {
                fac[t] = 1.f / tot.y;
                facm[t] = tot.x;
            }
            __syncthreads();
            const int64_t lo = max(rs, e0) * h, hi = min(re, e1) * h;
            for (int64_t i = lo + t; i < hi; i += kThreads) {
                const int hd = (int)(i & (h - 1));
                out[i] = from_f32<T>(es_exp(to_f32<T>(a[i]) - facm[hd]) * fac[hd]);
            }
        }
        return;
    }
    // ---- a tile inside ONE row ------------------------------------------------------------------------------------------
    const bool partial = ti.hs < e0 || ti.he > e1;
    float mx[V], sm[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        mx[k] = -INFINITY;
        sm[k] = 0.f;
    }
    for (int j0 = 0; j0 < nv; j0 += kBatch) {  // pass 1: online (max, sum) per slot (slot k of thread t = head (t*V + k) % h)
        uint4 r[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int j = min(j0 + u, nv - 1);
            if (full) r[u] = *reinterpret_cast<const uint4 *>(a + b0 + (int64_t)(j * kThreads + t) * V);
            else r[u] = load16_guarded<T>(a + b0, (j * kThreads + t) * V, count);
        }
        float v[kBatch][V];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) unpack16<T>(r[u], v[u]);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float bm = -INFINITY;
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const bool valid = (j0 + u < nv) && (full || ((j0 + u) * kThreads + t) * V + k < count);
                v[u][k] = valid ? v[u][k] : -INFINITY;
                bm = fmaxf(bm, v[u][k]);
            }
            const float mn = fmaxf(mx[k], bm);
            float acc = (sm[k] == 0.f) ? 0.f : sm[k] * es_exp(mx[k] - mn);
#pragma unroll
            for (int u = 0; u < kBatch; ++u) acc += (v[u][k] == -INFINITY) ? 0.f : es_exp(v[u][k] - mn);
            sm[k] = acc;
            mx[k] = mn;
        }
    }
    float gm[V];
#pragma unroll
    for (int k = 0; k < V; ++k) gm[k] = mx[k];
    wg_slot_reduce<V, true>(gm, h, red);
#pragma unroll
    for (int k = 0; k < V; ++k) sm[k] = (sm[k] == 0.f) ? 0.f : sm[k] * es_exp(mx[k] - gm[k]);
    wg_slot_reduce<V, false>(sm, h, red);
    if (t * V < h || (h < V && t == 0)) {
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (t * V + k < h) pstat[t * V + k] = make_float2(gm[k], sm[k]);
    }
    __syncthreads();
    float2 piece = make_float2(0.f, 0.f);
    if (t < h) piece = pstat[t];
    float2 tot = piece;
    if (partial) {
        if (ti.he - ti.hs <= p.long_edges) publish<false>(p, c, 0, piece);  // super-long rows: published by the init kernel
        if (!row_totals<false>(p, ti.hs, ti.he, mrg, tot)) tot = wg_piece_global<T, false>(a, nullptr, ti.hs * h, ti.he * h, h, red);
    }
    if (t < h) {
        fac[t] = 1.f / tot.y;
        facm[t] = tot.x;
    }
    __syncthreads();
    float f[V], fm[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        f[k] = fac[(t * V + k) & (h - 1)];
        fm[k] = facm[(t * V + k) & (h - 1)];
    }
    for (int j0 = 0; j0 < nv; j0 += kBatch) {  // pass 2: the tile again (L2 / Infinity Cache), scaled, stored
        uint4 r[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int j = min(j0 + u, nv - 1);
            if (full) r[u] = *reinterpret_cast<const uint4 *>(a + b0 + (int64_t)(j * kThreads + t) * V);
            else r[u] = load16_guarded<T>(a + b0, (j * kThreads + t) * V, count);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (j0 + u >= nv) continue;
            const int i0 = ((j0 + u) * kThreads + t) * V;
            float va[V], o[V];
            unpack16<T>(r[u], va);
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = es_exp(va[k] - fm[k]) * f[k];
            if (full) {
                store_vec<T, V>(out + b0 + i0, o);
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k)
                    if (i0 + k < count) out[b0 + i0 + k] = from_f32<T>(o[k]);
            }
        }
    }
}

// =====================================================================================================================
// init kernel.  Workgroups [0, n_seg): segment k owns the tiles [k * K_max, (k+1) * K_max), i.e. the edges
// [k * L, (k+1) * L) with L = K_max * tile_e: it zeroes their piece records and produces the piece records of super-long rows.
// A row of more than L edges contains a multiple of L, so it is the row of the first or of the last edge of some
// segment: each segment checks those two rows and, for a super-long one, computes the statistics of its pieces inside
// the segment, tile by tile, publishing them exactly where the main kernel looks for them.
// Workgroups [n_seg, ...): the TileInfo records -- one search per tile edge, a thread (binary search) or a wave
// (64-ary search: 3 dependent rounds instead of 18, for graphs with few tiles) per tile.
__device__ __forceinline__ int row_of_edge_wave(const int32_t *__restrict__ rowptr, int64_t m, int64_t e, int lane) {
    int64_t lo = 0, hi = m;
    while (hi - lo > 1) {
        const int64_t step = (hi - lo + kWave - 1) / kWave;
        const int64_t idx = lo + (int64_t)(lane + 1) * step;
        const bool below = idx < hi && (int64_t)rowptr[idx] <= e;
        const int cnt = __popcll(__ballot(below));
        lo = lo + (int64_t)cnt * step;
        hi = min(lo + step, hi);
    }
    return (int)lo;
}

template <typename T, bool BWD>
__global__ __launch_bounds__(kThreads) void es_flat_init_kernel(const Params p) {
    __shared__ float red[4 * kWave];
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= p.n_seg) {  // ---- tile records
        const int64_t b = blockIdx.x - p.n_seg;
        const int64_t c = p.info_per_wave ? b * 4 + (t >> 6) : b * kThreads + t;
        if (c >= p.n_tiles) return;
        const int64_t e0 = c * p.tile_e, e1 = min(p.nnz, e0 + p.tile_e);
        int r_first, r_last;
        if (p.info_per_wave) {
            r_first = row_of_edge_wave(p.rowptr, p.m, e0, t & (kWave - 1));
            r_last = (int64_t)p.rowptr[r_first + 1] >= e1 ? r_first : row_of_edge_wave(p.rowptr, p.m, e1 - 1, t & (kWave - 1));
            if ((t & (kWave - 1)) != 0) return;
        } else {
            r_first = row_of_edge(p.rowptr, p.m, (int)e0);
            r_last = (int64_t)p.rowptr[r_first + 1] >= e1 ? r_first : row_of_edge(p.rowptr, p.m, (int)(e1 - 1));
        }
        TileInfo ti;
        ti.r_first = r_first;
        ti.r_last = r_last;
        ti.hs = p.rowptr[r_first];
        ti.he = p.rowptr[r_first + 1];
        ti.ts = p.rowptr[r_last];
        ti.te = p.rowptr[r_last + 1];
        ti.pad0 = ti.pad1 = 0;
        p.tinfo[c] = ti;
        return;
    }
    const int64_t seg = blockIdx.x;
    const int64_t c_lo = seg * kKMax, c_hi = min(p.n_tiles, c_lo + kKMax);
    if (seg == 0 && t == 0) st_agent(p.stats, 0u);
    if (seg == 0 && t < 16 && (p.debug & 128)) st_agent(p.stats + 8 + t, 0u);  // phase stamps (tuning key 9 bit 7)
    for (int64_t i = c_lo * 2 * p.h + t; i < c_hi * 2 * p.h; i += kThreads) st_rec(p.rec + i, 0.f, 0.f);
    const int64_t L = (int64_t)kKMax * p.tile_e;
    if (p.nnz <= L) return;  // no row can be super-long (uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int64_t s_lo = c_lo * p.tile_e, s_hi = min(p.nnz, c_hi * p.tile_e);
    const int64_t ra = wg_row_of_edge(p.rowptr, p.m, s_lo);
    const int64_t rb = wg_row_of_edge(p.rowptr, p.m, s_hi - 1);
    const T *__restrict__ a = (const T *)p.a;
    const T *__restrict__ g = (const T *)p.g;
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && rb == ra) break;
        const int64_t r = which == 0 ? ra : rb;
        const int64_t start = p.rowptr[r], end = p.rowptr[r + 1];
        if (end - start <= p.long_edges) continue;
        const int64_t lo = max(start, s_lo), hi = min(end, s_hi);
        for (int64_t c = lo / p.tile_e; c * p.tile_e < hi; ++c) {
            const int64_t pl = max(lo, c * p.tile_e), ph = min(hi, (c + 1) * p.tile_e);
            const float2 st = wg_piece_global<T, BWD>(a, g, pl * p.h, ph * p.h, p.h, red);
            publish<BWD>(p, c, slot_of(start, c, p.tile_e), st);
        }
    }
}

template <typename T, bool BWD, int TILE, int WPE, bool KEEP_P = false>
static int launch_typed(Params &p, hipStream_t s, bool split = false) {
    const int64_t n_seg = (p.n_tiles + kKMax - 1) / kKMax;
    p.info_per_wave = p.n_tiles < 16384 ? 1 : 0;
    const int64_t n_info = p.info_per_wave ? (p.n_tiles + 3) / 4 : (p.n_tiles + kThreads - 1) / kThreads;
    if (n_seg + n_info > 0x7fffffff || p.n_tiles > 0x7fffffff) return COGDL_HIP_ERANGE;
    p.n_seg = (int)n_seg;
    hipLaunchKernelGGL((es_flat_init_kernel<T, BWD>), dim3((unsigned)(n_seg + n_info)), dim3(kThreads), 0, s, p);
    if constexpr (!BWD) {
        if (split) {  // three launches: multi-row tiles through LDS (no waits), then one-row tiles + exchanged pieces streamed
            hipLaunchKernelGGL((es_flat_kernel<T, false, TILE, WPE, false, true>), dim3((unsigned)p.n_tiles), dim3(kThreads), 0, s, p);
            hipLaunchKernelGGL((es_stream_kernel<T>), dim3((unsigned)p.n_tiles), dim3(kThreads), 0, s, p);
            return launch_status();
        }
    }
    hipLaunchKernelGGL((es_flat_kernel<T, BWD, TILE, WPE, KEEP_P>), dim3((unsigned)p.n_tiles), dim3(kThreads), 0, s, p);
    return launch_status();
}

}  // namespace esf

// ---- interface to edge_softmax.hip ------------------------------------------------------------------------------------
bool es_flat_covers(int64_t h, int dtype, const void *a, const void *g, const void *out) {
    if (h < 1 || h > 64 || (h & (h - 1)) != 0) return false;
    if (dtype != COGDL_HIP_F32 && dtype != COGDL_HIP_F16 && dtype != COGDL_HIP_BF16) return false;
    return aligned_to(a, 16) && aligned_to(out, 16) && (g == nullptr || aligned_to(g, 16));
}

static bool es_flat_small16() { return (g_tuning[kTuneEsDebug] & 4) != 0; }  // 16-bit values: the half-size tiles
// Small problems (fewer full-size tiles than the chip has workgroup slots: an arxiv-sized graph with one or two heads)
// run in QUARTER-size tiles: four times the workgroups, a quarter of the serial work in each, 16 KB of LDS instead of 40 --
// with full-size tiles an arxiv-sized H = 1 call is 293 workgroups on 256 CUs, each walking ~600 rows through LDS.
// Measured (tools/es_small_ab.py, arxiv-sized R-MAT graph, forward): H = 1 fp32 65 -> 39 us, bf16 121 -> 48 us, H = 2
// 64 -> 47 / 80 -> 45 us; at H = 8 (2441 full tiles: the chip is full either way) quarter tiles are SLOWER (69 -> 78 us
// forward, 69 -> 98 backward: four times the pieces per row), hence the bound.  A function of (nnz, h) only, so that the
// workspace query and the launch agree.  Tuning key 9 bit 6 = off (A/B runs, tests of the full-size tiles on small graphs).
constexpr int kSmallDiv = 4;
static bool es_flat_small_problem(int64_t nnz, int64_t h) {
    if (g_tuning[kTuneEsDebug] & 64) return false;
    return nnz * h < (int64_t)1024 * esf::TileSize<false, 4>::value;
}
static int64_t es_flat_tile_elems(bool bwd, int elem_bytes, bool small) {
    int64_t t;
    if (elem_bytes == 4 || es_flat_small16()) t = bwd ? esf::TileSize<true, 4>::value : esf::TileSize<false, 4>::value;
    else t = bwd ? esf::TileSize<true, 2>::value : esf::TileSize<false, 2>::value;
    return small ? t / kSmallDiv : t;
}
static int64_t es_flat_tiles(int64_t nnz, int64_t h, bool bwd, int elem_bytes = 4) {
    const int64_t tile_e = es_flat_tile_elems(bwd, elem_bytes, es_flat_small_problem(nnz, h)) / h;
    return (nnz + tile_e - 1) / tile_e;
}

// [header: 256 B][tile records: n_tiles * 32 B, padded to 256 B][piece records: n_tiles * 2 * h float2], sized for the
// backward's smaller tiles (more of them).
static size_t es_flat_info_bytes(int64_t nnz, int64_t h) {
    return ((size_t)es_flat_tiles(nnz, h, true) * sizeof(esf::TileInfo) + 255) / 256 * 256;
}
size_t es_flat_workspace_bytes(int64_t nnz, int64_t h) {
    return 256 + es_flat_info_bytes(nnz, h) + (size_t)es_flat_tiles(nnz, h, true) * 2 * (size_t)h * sizeof(float2);
}

int es_flat_launch(bool bwd, const int32_t *rowptr, const void *a, const void *g, void *out, int64_t m, int64_t nnz,
                   int64_t h, int dtype, void *ws, size_t wsb, hipStream_t s) {
    if (!ws || wsb < es_flat_workspace_bytes(nnz, h)) return COGDL_HIP_EWORKSPACE;
    if (!aligned_to(ws, 256)) return COGDL_HIP_EALIGN;
    esf::Params p{};
    p.rowptr = rowptr;
    p.a = a;
    p.g = g;
    p.out = out;
    p.m = m;
    p.nnz = nnz;
    p.h = (int)h;
    const int elem_bytes = dtype == COGDL_HIP_F32 ? 4 : 2;
    const bool small = es_flat_small_problem(nnz, h);
    p.tile_e = (int)(es_flat_tile_elems(bwd, elem_bytes, small) / h);
    p.n_tiles = es_flat_tiles(nnz, h, bwd, elem_bytes);
    p.stats = (unsigned *)ws;
    p.tinfo = (esf::TileInfo *)((char *)ws + 256);
    p.rec = (float2 *)((char *)ws + 256 + es_flat_info_bytes(nnz, h));
    p.long_edges = (int64_t)esf::kKMax * p.tile_e;
    p.debug = (g_tuning[kTuneEsDebug] & (1 | 16 | 128)) | (g_tuning[kTuneEsSpin] < 0 ? 2 : 0);
    p.spin_limit = g_tuning[kTuneEsSpin] > 0 ? (unsigned)g_tuning[kTuneEsSpin] : esf::kSpinLimit;
    constexpr int F32F = esf::TileSize<false, 4>::value, F32B = esf::TileSize<true, 4>::value;  // 8192 / 4096 elements
    constexpr int B16F = esf::TileSize<false, 2>::value, B16B = esf::TileSize<true, 2>::value;  // 16384 / 8192
    const bool small16 = es_flat_small16();
    if (small && !small16) {  // quarter-size tiles, six to eight workgroups per CU
        constexpr int D = kSmallDiv;
        switch (dtype) {
            case COGDL_HIP_F32:
                return bwd ? esf::launch_typed<float, true, F32B / D, 6>(p, s) : esf::launch_typed<float, false, F32F / D, 6, true>(p, s);
            case COGDL_HIP_F16:
                return bwd ? esf::launch_typed<__half, true, B16B / D, 6>(p, s) : esf::launch_typed<__half, false, B16F / D, 6, true>(p, s);
            case COGDL_HIP_BF16:
                return bwd ? esf::launch_typed<__hip_bfloat16, true, B16B / D, 6>(p, s)
                           : esf::launch_typed<__hip_bfloat16, false, B16F / D, 6, true>(p, s);
            default: return COGDL_HIP_EDTYPE;
        }
    }
    // forward: exp(v - max_piece) of single-row tiles kept in registers (one exponential per element; measured on the
    // Reddit-shaped graph: fp32 1.68 -> 1.62 ms, bf16 1.41 -> 1.37 ms, the latter despite 65 spilled registers per lane);
    // tuning key 9 bit 3 switches it off for A/B runs
    if (!bwd && !(g_tuning[kTuneEsDebug] & 8) && !small16) {
        // The two-kernel form (round 5: one-row tiles and exchanged pieces streamed twice by es_stream_kernel at 7 workgroups
        // per CU instead of held in registers across the exchange) is OPT-IN, tuning key 9 bit 8: measured on the MI355X it
        // loses -- Reddit-shaped graph, H = 8, forward: fp32 1674 -> 2512 us, bf16 1474 -> 1899 us; H = 1: 331 -> 441 us
        // (profiles/r05_es_split_ab.txt).  The second read of a tile does not come out of L2: seven resident tiles of 32 KB
        // per CU are 7 MB per XCD against 4 MB of L2, and a hit in the Infinity Cache crosses the same fabric as an HBM
        // read -- three passes over the array instead of two is the 1.5x that was measured.  Fitting L2 means four tiles per
        // CU or smaller tiles, i.e. the occupancy of the one-kernel form or more pieces per row (measured slower in round 3).
        const bool split = (g_tuning[kTuneEsDebug] & 256) != 0;
        switch (dtype) {
            case COGDL_HIP_F32: return esf::launch_typed<float, false, F32F, 4, true>(p, s, split);
            case COGDL_HIP_F16: return esf::launch_typed<__half, false, B16F, 4, true>(p, s, split);
            case COGDL_HIP_BF16: return esf::launch_typed<__hip_bfloat16, false, B16F, 4, true>(p, s, split);
            default: return COGDL_HIP_EDTYPE;
        }
    }
    switch (dtype) {
        case COGDL_HIP_F32:
            return bwd ? esf::launch_typed<float, true, F32B, 4>(p, s) : esf::launch_typed<float, false, F32F, 4>(p, s);
        case COGDL_HIP_F16:
            if (small16) return bwd ? esf::launch_typed<__half, true, F32B, 6>(p, s) : esf::launch_typed<__half, false, F32F, 6>(p, s);
            return bwd ? esf::launch_typed<__half, true, B16B, 4>(p, s) : esf::launch_typed<__half, false, B16F, 4>(p, s);
        case COGDL_HIP_BF16:
            if (small16)
                return bwd ? esf::launch_typed<__hip_bfloat16, true, F32B, 6>(p, s)
                           : esf::launch_typed<__hip_bfloat16, false, F32F, 6>(p, s);
            return bwd ? esf::launch_typed<__hip_bfloat16, true, B16B, 4>(p, s)
                       : esf::launch_typed<__hip_bfloat16, false, B16F, 4>(p, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

}  // namespace cogdl
