// common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels of libcogdl_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/cogdl_hip.h"

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

inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed,
// speed-only).  Remap so that each XCD walks one contiguous 1/8th of the logical blocks:
// consecutive row-blocks then share an L2, which is where the gathered feature rows of
// graphs with id-locality get re-used.  Grid must be launched with xcd_grid(n) blocks;
// returns -1 for the padding blocks.
__host__ __device__ inline int64_t xcd_chunk(int64_t n_logical) { return (n_logical + kXcds - 1) / kXcds; }
inline unsigned xcd_grid(int64_t n_logical) { return (unsigned)(xcd_chunk(n_logical) * kXcds); }
__device__ __forceinline__ int64_t xcd_remap(unsigned bid, int64_t n_logical) {
    const int64_t chunk = xcd_chunk(n_logical);
    const int64_t logical = (int64_t)(bid % kXcds) * chunk + (bid / kXcds);
    return logical < n_logical ? logical : -1;
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

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T *p, const float (&src)[VEC]) {
    using Raw = typename RawVec<sizeof(T) * VEC>::type;
    union { Raw raw; T e[VEC]; } u;
#pragma unroll
    for (int i = 0; i < VEC; ++i) u.e[i] = from_f32<T>(src[i]);
    *reinterpret_cast<Raw *>(p) = u.raw;
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
