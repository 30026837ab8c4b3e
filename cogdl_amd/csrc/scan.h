// scan.h -- device-wide prefix scans for the integer bookkeeping of this library (digit tables of the radix transpose, row
// counts of the sampler / subgraph / shard kernels, keep flags of add_remaining_self_loops).  Hand-written for gfx950; rounds
// 1-5 called rocprim::exclusive_scan / inclusive_scan here.
//
//   n <= kScanSmall     ONE launch: one 1024-thread workgroup walks the array in steps of 4096 entries with a running carry
//                       (sampled blocks and mini-batch frontiers: a captured step is bound by its number of dependent launches)
//   larger              THREE launches, no atomics, no look-back spinning (deterministic; capturable in a hipGraph at any size):
//                         scan_tile_totals_kernel   per tile of 4096 entries: the tile's total -> totals[tile]
//                         scan_single_kernel        exclusive scan of the totals, in place (one workgroup)
//                         scan_tiles_kernel         per tile: local scan + the tile's offset
// In-place operation (out == in) is allowed: a thread reads its own entries before it writes them, a tile touches nothing else
// (the single-workgroup kernel: a step's loads are all issued before its first barrier, its stores behind the second).
// `Op` is associative and commutative over Out (sums, maxima); `identity` is its neutral element.
#pragma once
#include "common.h"

namespace cogdl {

struct ScanPlus {
    template <typename T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
struct ScanMax {
    template <typename T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
};

constexpr int kScanThreads1 = 1024;                       // the single-workgroup kernel
constexpr int kScanItems = 4;                             // consecutive entries per thread and step
constexpr int kScanStep = kScanThreads1 * kScanItems;     // 4096
constexpr int64_t kScanSmall = 1 << 16;
constexpr int kScanTileThreads = 256, kScanTileItems = 16;
constexpr int kScanTile = kScanTileThreads * kScanTileItems;  // 4096

template <typename T>
__device__ __forceinline__ T scan_shfl_up(T v, int delta) {
    if constexpr (sizeof(T) == 8) {
        const unsigned long long u = (unsigned long long)v;
        const unsigned lo = __shfl_up((unsigned)u, delta, kWave), hi = __shfl_up((unsigned)(u >> 32), delta, kWave);
        return (T)(((unsigned long long)hi << 32) | lo);
    } else {
        return (T)__shfl_up(v, delta, kWave);
    }
}

// Inclusive scan of one value per lane over the wave.
template <typename T, class Op>
__device__ __forceinline__ T wave_inclusive(T v, int lane, Op op) {
#pragma unroll
    for (int sft = 1; sft < kWave; sft <<= 1) {
        const T u = scan_shfl_up(v, sft);
        if (lane >= sft) v = op(v, u);
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T scan_bcast_last(T v) {
    if constexpr (sizeof(T) == 8) {
        const unsigned long long u = (unsigned long long)v;
        const unsigned lo = __shfl((unsigned)u, kWave - 1, kWave), hi = __shfl((unsigned)(u >> 32), kWave - 1, kWave);
        return (T)(((unsigned long long)hi << 32) | lo);
    } else {
        return (T)__shfl(v, kWave - 1, kWave);
    }
}

// A wave owns ROWS * 64 CONSECUTIVE entries as ROWS rows of 64 (entry j * 64 + lane: every load and store of a row is one
// coalesced access).  In: v[j] = the entries; out: their scan in that order (exclusive or inclusive, without any offset), and
// the wave's total.
template <typename T, class Op, int ROWS, bool INCLUSIVE>
__device__ __forceinline__ void wave_rows_scan(T (&v)[ROWS], int lane, Op op, T identity, T &wave_total) {
    T carry = identity;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const T incl = wave_inclusive(v[j], lane, op);
        const T total = scan_bcast_last(incl);
        if constexpr (INCLUSIVE) {
            v[j] = op(carry, incl);
        } else {
            T ex = scan_shfl_up(incl, 1);
            if (lane == 0) ex = identity;
            v[j] = op(carry, ex);
        }
        carry = op(carry, total);
    }
    wave_total = carry;
}

// The combination of the totals of all LOWER waves of the workgroup (identity for wave 0) and the workgroup's total.
// Two barriers; `wsum`: LDS scratch of WAVES entries.
template <typename T, class Op, int WAVES>
__device__ __forceinline__ T waves_exclusive(T wave_total, T identity, Op op, T *wsum, T &total) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    __syncthreads();  // (wsum may still be read from a previous step)
    if (lane == 0) wsum[w] = wave_total;
    __syncthreads();
    T before = identity, tot = identity;
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww) {
        const T s = wsum[ww];
        if (ww < w) before = op(before, s);
        tot = op(tot, s);
    }
    total = tot;
    return before;
}

// One workgroup, any n: steps of 4096 entries (wave w: the 256 entries from 256 w on, 4 rows of 64) with a running carry.
// out[i] = init (+) in[0] (+) .. (+) in[i - 1] (exclusive) or .. (+) in[i] (inclusive).
template <typename In, typename Out, class Op, bool INCLUSIVE>
__global__ __launch_bounds__(kScanThreads1) void scan_single_kernel(const In *__restrict__ in, Out *__restrict__ out, int64_t n,
                                                                    Out init, Out identity, Op op) {
    constexpr int WAVES = kScanThreads1 / kWave;
    __shared__ Out wsum[WAVES];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    Out carry = init;
    for (int64_t base = 0; base < n; base += kScanStep) {
        const int64_t i0 = base + (int64_t)w * (kScanItems * kWave) + lane;
        Out v[kScanItems];
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {  // (unconditional loads, index clamped: no load behind a per-lane branch)
            const int64_t i = i0 + j * kWave;
            const In x = in[min(i, n - 1)];
            v[j] = i < n ? (Out)x : identity;
        }
        Out wave_total, total;
        wave_rows_scan<Out, Op, kScanItems, INCLUSIVE>(v, lane, op, identity, wave_total);
        const Out off = op(carry, waves_exclusive<Out, Op, WAVES>(wave_total, identity, op, wsum, total));
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            const int64_t i = i0 + j * kWave;
            if (i < n) out[i] = op(off, v[j]);
        }
        carry = op(carry, total);
    }
}

template <typename In, typename Out, class Op>
__global__ __launch_bounds__(kScanTileThreads) void scan_tile_totals_kernel(const In *__restrict__ in, Out *__restrict__ totals,
                                                                          int64_t n, Out identity, Op op) {
    constexpr int WAVES = kScanTileThreads / kWave;
    __shared__ Out wsum[WAVES];
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
    Out sum = identity;  // (the order inside a tile does not matter for a total)
#pragma unroll
    for (int k = 0; k < kScanTileItems; ++k) {
        const int64_t i = tile0 + k * kScanTileThreads + threadIdx.x;
        const In x = in[min(i, n - 1)];
        sum = op(sum, i < n ? (Out)x : identity);
    }
    const Out wave_total = scan_bcast_last(wave_inclusive(sum, lane, op));
    Out total;
    (void)waves_exclusive<Out, Op, WAVES>(wave_total, identity, op, wsum, total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// Wave w of a tile: the 1024 entries from 1024 w on, 16 rows of 64.
template <typename In, typename Out, class Op, bool INCLUSIVE>
__global__ __launch_bounds__(kScanTileThreads) void scan_tiles_kernel(const In *__restrict__ in, Out *__restrict__ out,
                                                                    const Out *__restrict__ offsets, int64_t n, Out init,
                                                                    Out identity, Op op) {
    constexpr int WAVES = kScanTileThreads / kWave;
    __shared__ Out wsum[WAVES];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)w * (kScanTileItems * kWave) + lane;
    const Out tile_off = op(init, offsets[blockIdx.x]);
    Out v[kScanTileItems];
#pragma unroll
    for (int j = 0; j < kScanTileItems; ++j) {
        const int64_t i = i0 + j * kWave;
        const In x = in[min(i, n - 1)];
        v[j] = i < n ? (Out)x : identity;
    }
    Out wave_total, total;
    wave_rows_scan<Out, Op, kScanTileItems, INCLUSIVE>(v, lane, op, identity, wave_total);
    const Out off = op(tile_off, waves_exclusive<Out, Op, WAVES>(wave_total, identity, op, wsum, total));
#pragma unroll
    for (int j = 0; j < kScanTileItems; ++j) {
        const int64_t i = i0 + j * kWave;
        if (i < n) out[i] = op(off, v[j]);
    }
}

// Scratch of device_scan for n entries whose Out type has `out_bytes` bytes (0 for the one-launch sizes).
inline size_t device_scan_temp_bytes(int64_t n, size_t out_bytes) {
    if (n <= kScanSmall) return 0;
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    return ((size_t)tiles * out_bytes + 255) / 256 * 256;
}

// Enqueue the scan on `s`.  temp: >= device_scan_temp_bytes(n, sizeof(Out)) bytes, aligned to sizeof(Out) (may be null for
// the one-launch sizes).  Returns a COGDL_HIP_* status (launch errors are picked up by the caller's launch_status()).
template <bool INCLUSIVE, typename In, typename Out, class Op>
inline int device_scan(const In *in, Out *out, int64_t n, Out init, Out identity, Op op, void *temp, hipStream_t s) {
    if (n <= 0) return COGDL_HIP_OK;
    if (n <= kScanSmall) {
        hipLaunchKernelGGL((scan_single_kernel<In, Out, Op, INCLUSIVE>), dim3(1), dim3(kScanThreads1), 0, s, in, out, n, init,
                           identity, op);
        return COGDL_HIP_OK;
    }
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles > 0x7fffffff || !temp) return COGDL_HIP_ERANGE;
    Out *totals = (Out *)temp;
    hipLaunchKernelGGL((scan_tile_totals_kernel<In, Out, Op>), dim3((unsigned)tiles), dim3(kScanTileThreads), 0, s, in, totals, n,
                       identity, op);
    hipLaunchKernelGGL((scan_single_kernel<Out, Out, Op, false>), dim3(1), dim3(kScanThreads1), 0, s, (const Out *)totals, totals,
                       tiles, identity, identity, op);
    hipLaunchKernelGGL((scan_tiles_kernel<In, Out, Op, INCLUSIVE>), dim3((unsigned)tiles), dim3(kScanTileThreads), 0, s, in, out,
                       (const Out *)totals, n, init, identity, op);
    return COGDL_HIP_OK;
}

template <typename In, typename Out>
inline int device_exclusive_sum(const In *in, Out *out, int64_t n, void *temp, hipStream_t s) {
    return device_scan<false>(in, out, n, Out(0), Out(0), ScanPlus(), temp, s);
}

}  // namespace cogdl
