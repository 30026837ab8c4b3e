// common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels of libcogdl_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/cogdl_hip.h"

#include <algorithm>

namespace cogdl {

constexpr int kWave = 64;   // CDNA wavefront width, hard-coded on purpose (gfx950 only)
constexpr int kXcds = 8;    // MI355X: 8 XCDs, each with a private 4 MiB L2

extern thread_local int g_last_hip_error;

inline int launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    }
    return COGDL_HIP_OK;
}

// bigcsr.hip: a segment table is well-formed and every segment fits the 32-bit kernels
int segments_valid(const cogdl_hip_segments *seg);

inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Stream-ordered fill of 32-bit words.  A kernel, not hipMemsetAsync: inside a captured hipGraph the small memsets of
// this library did not take effect on replay on ROCm 7.2 (measured: a 4-byte flag word kept the bytes of whatever had
// owned the pool block before); a kernel node replays like every other launch.
template <int UNUSED = 0>
__global__ void fill_u32_kernel(uint32_t *__restrict__ p, uint32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
inline hipError_t fill_u32_async(void *p, uint32_t v, size_t n_words, hipStream_t s) {
    if (n_words == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n_words + 255) / 256, 4096);
    hipLaunchKernelGGL(fill_u32_kernel<0>, dim3(blocks), dim3(256), 0, s, (uint32_t *)p, v, n_words);
    return hipGetLastError();
}

// Run-time tuning knobs (cogdl_hip_set_tuning): experiments without recompiling.
enum Tuning { kTuneXcdStripe = 0, kTuneLongThresh = 1, kTuneRowSort = 2, kTuneLongGrid = 3, kTuneGatVec = 4, kTuneGatOnline = 5, kTuneSpmmVec = 6, kTuneEsScalar = 7, kTuneEsSpin = 8, kTuneEsDebug = 9, kTuneCsr2csc = 10, kTuneSampleRelabel = 11, kTuneWaveSplit = 12, kTuneRowDebug = 13, kTuneRowTile = 14, kTuneSegmentEdges = 15, kTuneRowQueue = 16, kTuneCount = 17 };
extern int g_tuning[kTuneCount];

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8; observed, used for
// speed only).  Consecutive row-blocks gather overlapping neighbour rows on graphs with id-locality,
// so they should share an XCD's private L2 -- but handing each XCD one contiguous 1/8th of the rows
// unbalances the XCDs whenever degree correlates with node id (citation graphs: old papers are hubs).
// Compromise: logical blocks are dealt to the XCDs in stripes of `stripe` consecutive blocks.
// The grid is padded to a multiple of 8*stripe; xcd_remap returns -1 for the padding blocks.
struct XcdMap {
    int64_t n_logical;
    int stripe;  // 0: identity (hardware round-robin)
};
inline XcdMap make_xcd_map(int64_t n_logical) {
    int stripe = g_tuning[kTuneXcdStripe];
    if (stripe < 0) stripe = 0;
    return {n_logical, stripe};
}
inline int64_t xcd_grid64(const XcdMap &m) {
    if (m.stripe == 0) return m.n_logical;
    const int64_t per = (int64_t)kXcds * m.stripe;
    return (m.n_logical + per - 1) / per * per;
}
inline unsigned xcd_grid(const XcdMap &m) { return (unsigned)xcd_grid64(m); }
// gridDim.x is a 31-bit quantity: `extra` leading workgroups + the padded row-block grid must fit.
inline bool grid_fits(const XcdMap &m, int64_t extra = 0) { return xcd_grid64(m) + extra <= 0x7fffffff; }
__device__ __forceinline__ int64_t xcd_remap(unsigned bid, const XcdMap &m) {
    if (m.stripe == 0) return bid;
    const int64_t idx = bid / kXcds, xcd = bid % kXcds;
    const int64_t logical = (idx / m.stripe) * ((int64_t)kXcds * m.stripe) + xcd * m.stripe + idx % m.stripe;
    return logical < m.n_logical ? logical : -1;
}

// ---- 16/8/4-byte vector access of T[VEC] -------------------------------------------------
template <int BYTES> struct RawVec;
template <> struct RawVec<16> { using type = uint4; };
template <> struct RawVec<8>  { using type = uint2; };
template <> struct RawVec<4>  { using type = uint32_t; };
template <> struct RawVec<2>  { using type = uint16_t; };

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

// Load VEC contiguous elements of T as one vector memory instruction into fp32 registers.
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *p, float (&dst)[VEC]) {
    using Raw = typename RawVec<sizeof(T) * VEC>::type;
    union { Raw raw; T e[VEC]; } u;
    u.raw = *reinterpret_cast<const Raw *>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) dst[i] = to_f32<T>(u.e[i]);
}

// The same in two steps: the raw vector as loaded (16-bit types: half the registers of their fp32 values while the load is
// in flight), unpacked where it is used.
template <typename T, int VEC>
__device__ __forceinline__ typename RawVec<sizeof(T) * VEC>::type load_raw(const T *p) {
    return *reinterpret_cast<const typename RawVec<sizeof(T) * VEC>::type *>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void unpack_raw(const typename RawVec<sizeof(T) * VEC>::type &raw, float (&dst)[VEC]) {
    union { typename RawVec<sizeof(T) * VEC>::type raw; T e[VEC]; } u;
    u.raw = raw;
#pragma unroll
    for (int i = 0; i < VEC; ++i) dst[i] = to_f32<T>(u.e[i]);
}

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T *p, const float (&src)[VEC]) {
    using Raw = typename RawVec<sizeof(T) * VEC>::type;
    union { Raw raw; T e[VEC]; } u;
#pragma unroll
    for (int i = 0; i < VEC; ++i) u.e[i] = from_f32<T>(src[i]);
    *reinterpret_cast<Raw *>(p) = u.raw;
}

// Address of the lane's piece of row `col` of a row-major table: table + col * row_bytes + lane_elems elements.
//   A24 = false   ONE v_mad_u64_u32 (unsigned 32 x 32 -> 64 multiply with the lane's 64-bit base as addend) where
//                 `base + (int64_t)col * k` in ELEMENTS compiles to a signed 64-bit multiply-add plus a 64-bit shift-add: two
//                 multi-cycle vector instructions per gather in kernels bound by VALU issue on hub-heavy graphs (round 6).
//                 col >= 0, row_bytes < 2^32.
//   A24 = true    tables of < 2^24 rows of < 2^24 bytes, < 4 GiB in all (what an XCD-partitioned plan is built for,
//                 cogdl_amd/xcdplan.py: wanted()): a 32-bit byte offset from ONE full-rate v_mad_u32_u24 (col * row_bytes +
//                 the lane's own offset), added to the table's UNIFORM base by the load itself (global_load ... v_off, s[base]).
//                 v_mad_u64_u32 is a quarter-rate instruction; the fused GAT kernels issue two gathers per edge.
template <bool A24, typename T>
__device__ __forceinline__ const T *gather_row(const T *table, int lane_elems, int col, uint32_t row_bytes) {
    if constexpr (A24) {
        const uint32_t off = __umul24((uint32_t)col, row_bytes) + (uint32_t)lane_elems * (uint32_t)sizeof(T);
        return reinterpret_cast<const T *>(reinterpret_cast<const char *>(table) + off);
    } else {
        return reinterpret_cast<const T *>(reinterpret_cast<const char *>(table + lane_elems) +
                                           (uint64_t)(uint32_t)col * (uint64_t)row_bytes);
    }
}

// Butterfly reductions inside an aligned group of WIDTH lanes (WIDTH power of two <= 64).
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int s = WIDTH / 2; s > 0; s >>= 1) v += __shfl_xor(v, s, kWave);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int s = WIDTH / 2; s > 0; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, kWave));
    return v;
}

}  // namespace cogdl
