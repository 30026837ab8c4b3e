// core.hip -- status strings / ABI version of libcogdl_hip.
#include "common.h"

namespace cogdl {
thread_local int g_last_hip_error = 0;
int g_tuning[kTuneCount] = {/*0: xcd stripe*/ 32, /*1: long-row threshold override (0 = automatic)*/ 0, /*2: 1 = no degree-sorted row assignment inside a workgroup*/ 0,
                            /*3: max long-row workgroups*/ 1024, /*4: fused-GAT vector cap*/ 0,
                            /*5: fused-GAT forward kernel*/ 0, /*6: spmm vector cap (<0: force)*/ 0, /*7: edge_softmax scalar lanes only*/ 0, /*8: edge_softmax flat kernel: polls before a wait gives up (0 = default)*/ 0, /*9: edge_softmax flat kernel experiments: bit 0 = no cross-tile exchange (WRONG results, timing only), bit 2 = 16-bit values in half-size tiles at 6 workgroups per CU, bit 3 = forward recomputes the exponential instead of keeping it in registers*/ 0,
                            /*10: csr2csc: 0 = automatic (one single-workgroup launch up to 16 k slots, the radix sort above), 2 = the radix sort at every size, 3 = radix, packed records at any size*/ 0,
                            /*11: sample_adj relabelling: 0 = hash table of first positions (4 launches), 1 = the sort-based form of rounds 1-2 (~25 launches; its sort is this library's radix transpose since round 5)*/ 0,
                            /*12: retired (round 4's wave-scope split of medium rows, removed in round 6)*/ 0,
                            /*13: timing experiments on the row-reduce engine (WRONG results): 1 = the row blocks exit at once, 2 = the long-row workgroups exit at once*/ 0,
                            /*14: retired (round 4's csr_spmm row tiles, removed in round 6)*/ 0,
                            /*15: 64-bit CSR: edges per row segment (0 = default 2^29; tests use tiny values)*/ 0,
                            /*16: retired (round 5's per-workgroup row queue, removed in round 6)*/ 0};
}

namespace cogdl {
// Measurement hooks (bench.py: the box's own read / copy roofs beside the spec peak).  A workgroup
// owns contiguous 32 KiB tiles (8 x 16-byte vectors per lane, all eight requested before the first is used) and strides
// over the buffer; the read kernel folds what it loads into one word per wave so that the loads cannot be elided.
constexpr int kProbeUnroll = 8;
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_stream_kernel(const U32x4 *__restrict__ p, size_t n_vec, uint32_t *__restrict__ sink) {
    const size_t tile = (size_t)256 * kProbeUnroll;
    uint32_t acc = 0;
    for (size_t base = (size_t)blockIdx.x * tile; base < n_vec; base += (size_t)gridDim.x * tile) {
        U32x4 v[kProbeUnroll];
#pragma unroll
        for (int j = 0; j < kProbeUnroll; ++j) {
            const size_t i = base + (size_t)j * 256 + threadIdx.x;
            v[j] = i < n_vec ? __builtin_nontemporal_load(p + i) : U32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < kProbeUnroll; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) acc ^= __shfl_xor(acc, s, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) sink[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}
__global__ __launch_bounds__(256) void copy_stream_kernel(const U32x4 *__restrict__ p, U32x4 *__restrict__ q, size_t n_vec) {
    const size_t tile = (size_t)256 * kProbeUnroll;
    for (size_t base = (size_t)blockIdx.x * tile; base < n_vec; base += (size_t)gridDim.x * tile) {
        U32x4 v[kProbeUnroll];
#pragma unroll
        for (int j = 0; j < kProbeUnroll; ++j) {
            const size_t i = base + (size_t)j * 256 + threadIdx.x;
            v[j] = i < n_vec ? __builtin_nontemporal_load(p + i) : U32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < kProbeUnroll; ++j) {
            const size_t i = base + (size_t)j * 256 + threadIdx.x;
            if (i < n_vec) __builtin_nontemporal_store(v[j], q + i);
        }
    }
}
}  // namespace cogdl

extern "C" int cogdl_hip_probe_read_stream(const void *p, size_t bytes, void *sink, void *stream) {
    if (!p || !sink || !cogdl::aligned_to(p, 16)) return COGDL_HIP_EINVAL;
    hipLaunchKernelGGL(cogdl::read_stream_kernel, dim3(COGDL_HIP_PROBE_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const cogdl::U32x4 *)p,
                       bytes / 16, (uint32_t *)sink);
    return cogdl::launch_status();
}

extern "C" int cogdl_hip_probe_copy_stream(const void *src, void *dst, size_t bytes, void *stream) {
    if (!src || !dst || !cogdl::aligned_to(src, 16) || !cogdl::aligned_to(dst, 16)) return COGDL_HIP_EINVAL;
    hipLaunchKernelGGL(cogdl::copy_stream_kernel, dim3(COGDL_HIP_PROBE_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const cogdl::U32x4 *)src,
                       (cogdl::U32x4 *)dst, bytes / 16);
    return cogdl::launch_status();
}

extern "C" int cogdl_hip_set_tuning(int key, int value) {
    if (key < 0 || key >= cogdl::kTuneCount) return COGDL_HIP_EINVAL;
    cogdl::g_tuning[key] = value;
    return COGDL_HIP_OK;
}

extern "C" int cogdl_hip_abi_version(void) { return COGDL_HIP_ABI_VERSION; }

extern "C" int cogdl_hip_last_hip_error(void) { return cogdl::g_last_hip_error; }

extern "C" const char *cogdl_hip_strerror(int status) {
    switch (status) {
        case COGDL_HIP_OK: return "ok";
        case COGDL_HIP_EINVAL: return "invalid argument";
        case COGDL_HIP_EDTYPE: return "unsupported dtype";
        case COGDL_HIP_EALIGN: return "misaligned pointer";
        case COGDL_HIP_ELAUNCH: return "HIP launch/runtime error";
        case COGDL_HIP_EWORKSPACE: return "workspace too small";
        case COGDL_HIP_ERANGE: return "size out of range for int32 CSR indices";
        case COGDL_HIP_EUNSUPPORTED: return "shape not covered by this entry point";
        default: return "unknown status";
    }
}
