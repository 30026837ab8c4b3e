// spmm_epilogue.hip -- csr_spmm with the fused normalisation / bias / activation epilogue:
//     out[i,:] = act( dst_scale[i] * sum_e val[e] * (src_scale[col[e]] * x[col[e],:]) + bias )
// What CogDL's dispatcher does around the kernel for CSR-only graphs (cogdl/utils/spmm_utils.py:99-109: `out_norm * x`
// before, `in_norm * x` after -- two extra passes over an [N, F] matrix and two launches) and what a layer does behind
// it (activation, cogdl/layers/gcn_layer.py:51-64), folded into the gather (the source scale is one more 4-byte read
// next to colind) and the row's final store.  Same engine and geometry as csr_spmm (spmm_op.h); every product is a
// separately rounded fp32 multiply in the reference's order, so the result is bit-identical to the unfused
// composition for rows up to the long-row threshold.  fp32 operands only (the norms are fp32, and float32 * float16
// promotes to float32 in the reference's dispatcher).
#include "spmm_op.h"

using namespace cogdl;

extern "C" int cogdl_hip_csr_spmm_epilogue(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                           void *out, int64_t m, int64_t k, int64_t nnz, int dtype,
                                           const float *src_scale, const float *dst_scale, const float *bias, int act,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    if (m < 0 || k < 0 || nnz < 0 || act < 0 || act > 1) return COGDL_HIP_EINVAL;
    if (m == 0 || k == 0) return COGDL_HIP_OK;
    if (!rowptr || !x || !out) return COGDL_HIP_EINVAL;
    if (k > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (dtype != COGDL_HIP_F32) return COGDL_HIP_EUNSUPPORTED;
    if (!aligned_to(x, 4) || !aligned_to(out, 4)) return COGDL_HIP_EALIGN;
    SpmmArgs<float> a{rowptr, colind, (const float *)val, nullptr, (const float *)x, (float *)out, m, nnz, (int)k, (int)k,
                      0, nullptr, SpmmEpilogue{src_scale, dst_scale, bias, act}};
    hipStream_t s = (hipStream_t)stream;
    return val ? spmm_auto<float, 1, true>(a, workspace, workspace_bytes, s)
               : spmm_auto<float, 0, true>(a, workspace, workspace_bytes, s);
}
