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
                            /*14: csr_spmm row tiles (several rows per lane group, their first gathers in flight together): 2 = on wherever the operator allows; off by default (measured slower except for rows of one or two edges)*/ 0};
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
