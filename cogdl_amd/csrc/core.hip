// core.hip -- status strings / ABI version of libcogdl_hip.
#include "common.h"

namespace cogdl {
thread_local int g_last_hip_error = 0;
int g_tuning[kTuneCount] = {/*0: xcd stripe*/ 32, /*1: long-row threshold override (0 = automatic)*/ 0, /*2: 1 = no degree-sorted row assignment inside a workgroup*/ 0,
                            /*3: max long-row workgroups*/ 1024, /*4: fused-GAT vector cap*/ 0,
                            /*5: fused-GAT forward kernel*/ 0, /*6: spmm vector cap (<0: force)*/ 0, /*7: edge_softmax scalar lanes only*/ 0, /*8: edge_softmax flat kernel: polls before a wait gives up (0 = default)*/ 0, /*9: edge_softmax flat kernel experiments: bit 0 = no cross-tile exchange (WRONG results, timing only), bit 2 = 16-bit values in half-size tiles at 6 workgroups per CU, bit 3 = forward recomputes the exponential instead of keeping it in registers*/ 0,
                            /*10: csr2csc: 0 = automatic, 1 = the rocPRIM pipeline, 2 = the hand-written radix sort, 3 = radix, packed records at any size, 5 = radix, MSD-first two-pass order*/ 0,
                            /*11: sample_adj relabelling: 0 = hash table of first positions (4 launches), 1 = the rocPRIM sort-based form (~25)*/ 0,
                            /*12: wave-scope split of medium rows: n > 0 = rows of more than n edges in skewed workgroups (off by default)*/ 0,
                            /*13: timing experiments on the row-reduce engine (WRONG results): 1 = the row blocks exit at once, 2 = the long-row workgroups exit at once*/ 0,
                            /*14: csr_spmm row tiles (several rows per lane group, their first gathers in flight together): 2 = on wherever the operator allows; off by default (measured slower except for rows of one or two edges)*/ 0,
                            /*15: 64-bit CSR: edges per row segment (0 = default 2^29; tests use tiny values)*/ 0};
}

namespace cogdl {
// Measurement hook (bench.py: the box's read roof beside the spec peak): every lane streams 16-byte vectors, four in
// flight, and folds them into one word per workgroup -- read-only HBM traffic, nothing else.
__global__ __launch_bounds__(256) void read_stream_kernel(const uint4 *__restrict__ p, size_t n_vec, uint32_t *__restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n_vec; i += stride) {
        const uint4 a = p[i];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) acc ^= __shfl_xor(acc, s, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) sink[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}
}  // namespace cogdl

extern "C" int cogdl_hip_probe_read_stream(const void *p, size_t bytes, void *sink, void *stream) {
    if (!p || !sink || !cogdl::aligned_to(p, 16)) return COGDL_HIP_EINVAL;
    hipLaunchKernelGGL(cogdl::read_stream_kernel, dim3(COGDL_HIP_PROBE_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const uint4 *)p,
                       bytes / 16, (uint32_t *)sink);
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
