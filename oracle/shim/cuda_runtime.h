/* oracle/shim -- CHECKER infrastructure only (never on the product's include path).
 * The reference's spmm/computeUtil.h includes four CUDA toolkit headers by name; torch's hipify translates the .cu
 * files the reference's JIT recipe lists, but not this header.  These shims let the reference's OWN, unmodified
 * sources compile for gfx950 so that their outputs can pin the oracle (tests/golden/make_golden_gpu.py). */
#pragma once
#include <hip/hip_runtime.h>
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetLastError hipGetLastError
#define CUSPARSE_STATUS_SUCCESS HIPSPARSE_STATUS_SUCCESS
