// linear_fwd16.hip -- the tall-skinny dense product of a layer under bf16 autocast on gfx950 MFMA:
//     out[R, N] (bf16) = bf16(x[R, K]) . bf16(B) (+ bias),   R = number of graph nodes, N <= 64, K up to ~1000,
// x in fp32 (the layer's input as the model holds it: the rounding to bf16 that autocast performs with a copy kernel of
// its own happens in registers here, v_cvt_pk_bf16_f32 = round-to-nearest-even like the copy) or in bf16.  The case it is
// written for is the first layer of BASELINE configs[2] (cogdl/layers/gat_layer.py:59: h = torch.matmul(x, self.W),
// 232,965 x 602 -> 64): hipBLASLt needs 525 us for that product (34 TFLOP/s, 0.5 TB/s) plus 260 us for the cast of x,
// while all there is to do is stream x once (561 MB).
//
// v_mfma_f32_32x32x16_bf16: lane l holds A[i = l & 31][k = 8 (l >> 5) .. + 7] and B[k = 8 (l >> 5) .. + 7][j = l & 31] --
// EIGHT CONSECUTIVE k per lane, so a lane's A operand is a contiguous piece of a row of the row-major x: no LDS staging
// of x, no transposition (the fp32 kernel of linear_fwd.hip needs both: its MFMA wants one k per lane).  k is only a
// summation index: a lane reads SIXTEEN consecutive elements per step (a macro-step of 32 k: half-wave h takes columns
// 32 s + 16 h .. + 15 -- the two half-waves of a row read one whole 128-byte line of an fp32 x) and feeds two MFMAs,
// the first with its elements 0..7, the second with 8..15; B is laid out to match.  B is small: converted once per
// (persistent) workgroup into LDS in exactly the operand order, one conflict-free 16-byte read per lane and MFMA.
#include "common.h"

namespace cogdl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// OT: the 16-bit type of the product's operands and result (bf16: torch.autocast(bfloat16); f16: the autocast of the
// reference's Trainer(fp16=True), cogdl/trainer/trainer.py) -- its 8-element MFMA operand and the MFMA itself
template <typename OT> struct Half16;
template <> struct Half16<__hip_bfloat16> {
    using vec8 = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Half16<__half> {
    using vec8 = f16x8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr size_t kFwd16MaxBBytes = 128 * 1024;

template <typename XT> struct Fwd16Elems { uint32_t r[4 * sizeof(XT)]; };  // the 16 elements of a lane and macro-step as loaded

// LB: bytes per load instruction (16 / 8 / 4: what the row stride of x allows)
template <typename XT, int LB>
__device__ __forceinline__ void fwd16_load(Fwd16Elems<XT> &e, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    constexpr int kBytes = 16 * (int)sizeof(XT);
    if constexpr (LB == 16) {
#pragma unroll
        for (int q = 0; q < kBytes / 16; ++q) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16 * q, soff, 0);
            e.r[4 * q] = v[0], e.r[4 * q + 1] = v[1], e.r[4 * q + 2] = v[2], e.r[4 * q + 3] = v[3];
        }
    } else if constexpr (LB == 8) {
#pragma unroll
        for (int q = 0; q < kBytes / 8; ++q) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + 8 * q, soff, 0);
            e.r[2 * q] = v[0], e.r[2 * q + 1] = v[1];
        }
    } else {
#pragma unroll
        for (int q = 0; q < kBytes / 4; ++q) e.r[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 4 * q, soff, 0);
    }
}

// -> the two A operands of the macro-step; n_ok: how many of the lane's 16 elements lie inside the row (>= 16: all) --
// elements past the row's end belong to the NEXT row (or are the buffer's zeros): B is zero there, but 0 * NaN is not
template <typename XT, typename OT, bool MASK>
__device__ __forceinline__ void fwd16_operands(const Fwd16Elems<XT> &e, int n_ok, typename Half16<OT>::vec8 (&a)[2]) {
    using vec8 = typename Half16<OT>::vec8;
    if constexpr (sizeof(XT) == 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x8 v;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                v[q] = __uint_as_float(e.r[8 * s + q]);
                if constexpr (MASK) v[q] = (8 * s + q < n_ok) ? v[q] : 0.f;
            }
            a[s] = __builtin_convertvector(v, vec8);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            union { uint32_t u[4]; vec8 v; } t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t w = e.r[4 * s + q];
                if constexpr (MASK) {
                    const int k = 8 * s + 2 * q;
                    w = (k + 1 < n_ok) ? w : (k < n_ok ? (w & 0xffffu) : 0u);
                }
                t.u[q] = w;
            }
            a[s] = t.v;
        }
    }
}

// B into LDS: entry e = ((ms * 2 + sub) * NT + nt) * 64 + lane holds B[32 ms + 16 h + 8 sub .. + 7][32 nt + j] as 8 bf16.  Every
// load is unconditional (an element outside B reads w[0] and is replaced by zero afterwards): no branch per element, so the
// 32 loads of a thread's four entries are all in flight before the first is used -- with a branch (and its s_waitcnt
// vmcnt(0)) per element the prologue of a 602 x 64 weight took ~100 us, two thirds of the whole kernel.
template <typename WT, typename OT, bool N_BY_K, int NT>
__device__ __forceinline__ void fwd16_fill_b(uint4 *bs, const WT *__restrict__ w, int k_dim, int n_dim, int n_ms) {
    using vec8 = typename Half16<OT>::vec8;
    const int n_entries = n_ms * 2 * NT * 64;
    for (int e0 = threadIdx.x; e0 < n_entries; e0 += 256 * 4) {
        WT raw[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + 256 * u, n_entries - 1);
            const int el = e & 63, nt = (e >> 6) % NT, sub = ((e >> 6) / NT) & 1, ms = (e >> 6) / (2 * NT);
            const int n = 32 * nt + (el & 31), k0 = 32 * ms + 16 * (el >> 5) + 8 * sub;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q;
                const bool ok = k < k_dim && n < n_dim;
                const int idx = N_BY_K ? n * k_dim + k : k * n_dim + n;  // (B of at most 64 x 4096 elements: int)
                raw[u][q] = w[ok ? idx : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            const int ec = min(e, n_entries - 1);
            const int el = ec & 63, nt = (ec >> 6) % NT, sub = ((ec >> 6) / NT) & 1, ms = (ec >> 6) / (2 * NT);
            const int n = 32 * nt + (el & 31), k0 = 32 * ms + 16 * (el >> 5) + 8 * sub;
            f32x8 f;
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = (k0 + q < k_dim && n < n_dim) ? to_f32<WT>(raw[u][q]) : 0.f;
            union { vec8 b; uint4 r; } t;
            t.b = __builtin_convertvector(f, vec8);
            if (e < n_entries) bs[e] = t.r;
        }
    }
}

template <typename XT, typename OT, int NT, int LB>
__global__ __launch_bounds__(256) void linear_fwd16_kernel(const XT *__restrict__ x, const void *__restrict__ w, int w_is_f32,
                                                           const float *__restrict__ bias, OT *__restrict__ out,
                                                           int64_t rows, int k_dim, int n_dim, int w_is_n_by_k, int n_ms) {
    using vec8 = typename Half16<OT>::vec8;
    extern __shared__ uint4 bs[];  // [n_ms][2][NT][64]: lane (h, j)'s operand = B[32 ms + 16 h + 8 sub .. + 7][32 nt + j]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    // ---- prologue: B into LDS, converted and in operand order (fwd16_fill_b: branch-free, 32 loads in flight per thread)
    if (w_is_f32) {
        if (w_is_n_by_k) fwd16_fill_b<float, OT, true, NT>(bs, reinterpret_cast<const float *>(w), k_dim, n_dim, n_ms);
        else fwd16_fill_b<float, OT, false, NT>(bs, reinterpret_cast<const float *>(w), k_dim, n_dim, n_ms);
    } else {
        if (w_is_n_by_k) fwd16_fill_b<OT, OT, true, NT>(bs, reinterpret_cast<const OT *>(w), k_dim, n_dim, n_ms);
        else fwd16_fill_b<OT, OT, false, NT>(bs, reinterpret_cast<const OT *>(w), k_dim, n_dim, n_ms);
    }
    __syncthreads();
    // ---- main loop: every wave walks 32-row tiles on its own (no barrier), the flattened (tile, macro-step) sequence with
    // the loads of the next step in flight while this one's MFMAs run
    const int64_t n_tiles = (rows + 31) / 32;
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    constexpr int kOob = (int)0x80000000;
    const int x_row_bytes = k_dim * (int)sizeof(XT), o_row_bytes = n_dim * 2;
    const int ld_lane = j * x_row_bytes + 16 * h * (int)sizeof(XT);
    auto x_rsrc = [&](int64_t tile, int &valid) {
        const int64_t row0 = tile * 32;
        valid = (int)max((int64_t)0, min((int64_t)32, rows - row0));
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<XT *>(x + (valid ? row0 : 0) * k_dim), 0, valid * x_row_bytes, 0x00020000);
    };
    auto load_step = [&](int64_t tile, int ms, Fwd16Elems<XT> &e) {
        int valid;
        const __amdgpu_buffer_rsrc_t r = x_rsrc(tile, valid);
        fwd16_load<XT, LB>(e, r, j < valid ? ld_lane : kOob, ms * 32 * (int)sizeof(XT));
    };
    float bias_v[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) bias_v[a] = (bias && 32 * a + j < n_dim) ? bias[32 * a + j] : 0.f;
    int st_lane[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) st_lane[a] = (32 * a + j < n_dim) ? 4 * h * o_row_bytes + (32 * a + j) * 2 : kOob;
    const bool ragged = (k_dim & 31) != 0;
    f32x16 acc[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    int ms = 0;
    Fwd16Elems<XT> cur, nxt;
    load_step(tile, 0, cur);
    while (tile < n_tiles) {
        const bool last = ms + 1 == n_ms;
        const int64_t next_tile = last ? tile + tile_step : tile;
        const int next_ms = last ? 0 : ms + 1;
        load_step(next_tile, next_ms, nxt);
        vec8 a_op[2];
        if (last && ragged) fwd16_operands<XT, OT, true>(cur, k_dim - (32 * ms + 16 * h), a_op);
        else fwd16_operands<XT, OT, false>(cur, 16, a_op);
        const uint4 *brow = bs + (size_t)ms * 2 * NT * 64 + lane;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                union { uint4 r; vec8 b; } t;
                t.r = brow[(sub * NT + a) * 64];
                acc[a] = Half16<OT>::mfma(a_op[sub], t.b, acc[a]);
            }
        if (last) {
            const int64_t row0 = tile * 32;
            const int valid = (int)min((int64_t)32, rows - row0);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + row0 * n_dim, 0, valid * o_row_bytes, 0x00020000);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int q = 0; q < 16; ++q) {  // C/D map of the 32x32 MFMA: register q holds row (q & 3) + 8 (q >> 2) + 4 h
                    union { OT b; uint16_t u; } o;
                    o.b = from_f32<OT>(acc[a][q] + bias_v[a]);
                    __builtin_amdgcn_raw_buffer_store_b16(o.u, r, st_lane[a] + ((q & 3) + 8 * (q >> 2)) * o_row_bytes, 0, 0);
                    acc[a][q] = 0.f;
                }
        }
        cur = nxt;  // (the compiler renames the two sets across the unrolled loop: no wait for `nxt` here -- the ISA waits with
        tile = next_tile;  //  vmcnt(6) / vmcnt(4) in front of the operand conversion, i.e. for `cur` only)
        ms = next_ms;
    }
}

template <typename XT, typename OT, int NT, int LB>
static int launch_fwd16(const XT *x, const void *w, int w_is_f32, const float *bias, OT *out, int64_t rows, int64_t k_dim,
                        int64_t n_dim, int w_is_n_by_k, hipStream_t s) {
    const int n_ms = (int)((k_dim + 31) / 32);
    const size_t lds = (size_t)n_ms * 2 * NT * 64 * 16;
    if (lds > kFwd16MaxBBytes) return COGDL_HIP_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {  // more than 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd16_kernel<XT, OT, NT, LB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kFwd16MaxBBytes);
        attr_set = true;
    }
    const int64_t n_tiles = (rows + 31) / 32;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / std::max<size_t>(lds, 1)));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, (int64_t)256 * per_cu));
    hipLaunchKernelGGL((linear_fwd16_kernel<XT, OT, NT, LB>), dim3(grid), dim3(256), lds, s, x, w, w_is_f32, bias, out, rows, (int)k_dim,
                       (int)n_dim, w_is_n_by_k, n_ms);
    return launch_status();
}

template <typename XT, typename OT, int NT>
static int dispatch_fwd16(const XT *x, const void *w, int w_is_f32, const float *bias, OT *out, int64_t rows, int64_t k_dim,
                          int64_t n_dim, int w_is_n_by_k, hipStream_t s) {
    const int64_t row_bytes = k_dim * (int64_t)sizeof(XT);
    if (row_bytes % 16 == 0) return launch_fwd16<XT, OT, NT, 16>(x, w, w_is_f32, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s);
    if (row_bytes % 8 == 0) return launch_fwd16<XT, OT, NT, 8>(x, w, w_is_f32, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s);
    if (row_bytes % 4 == 0) return launch_fwd16<XT, OT, NT, 4>(x, w, w_is_f32, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s);
    return COGDL_HIP_EUNSUPPORTED;  // (bf16 rows of an odd number of elements)
}

}  // namespace cogdl

using namespace cogdl;

namespace cogdl {
template <typename OT>
static int linear_fwd16_entry(const void *x, int x_dtype, const void *w, int w_dtype, int half_code, const float *bias, void *out, int64_t rows,
                              int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream) {
    if (rows < 0 || k_dim <= 0 || n_dim <= 0) return COGDL_HIP_EINVAL;
    if ((x_dtype != COGDL_HIP_F32 && x_dtype != half_code) || (w_dtype != COGDL_HIP_F32 && w_dtype != half_code)) return COGDL_HIP_EDTYPE;
    if (rows == 0) return COGDL_HIP_OK;
    if (!x || !w || !out) return COGDL_HIP_EINVAL;
    if (n_dim > 64 || k_dim > 4096) return COGDL_HIP_EUNSUPPORTED;
    if (!aligned_to(x, 16) || !aligned_to(out, 2) || !aligned_to(w, 4)) return COGDL_HIP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int wf = w_dtype == COGDL_HIP_F32;
    OT *o = (OT *)out;
    if (x_dtype == COGDL_HIP_F32) {
        return n_dim <= 32 ? dispatch_fwd16<float, OT, 1>((const float *)x, w, wf, bias, o, rows, k_dim, n_dim, w_is_n_by_k, s)
                           : dispatch_fwd16<float, OT, 2>((const float *)x, w, wf, bias, o, rows, k_dim, n_dim, w_is_n_by_k, s);
    }
    return n_dim <= 32 ? dispatch_fwd16<OT, OT, 1>((const OT *)x, w, wf, bias, o, rows, k_dim, n_dim, w_is_n_by_k, s)
                       : dispatch_fwd16<OT, OT, 2>((const OT *)x, w, wf, bias, o, rows, k_dim, n_dim, w_is_n_by_k, s);
}
}  // namespace cogdl

// out[rows, n] (bf16 / f16) = h(x[rows, k]) . h(B) (+ bias[n], fp32), h = the rounding to the 16-bit type;  x_dtype / w_dtype:
// COGDL_HIP_F32 or the entry's own 16-bit type;  w_is_n_by_k != 0: B = w^T with w stored [n, k] (nn.Linear); == 0: B = w stored
// [k, n] (torch.matmul(x, W), grad_input).  COGDL_HIP_EUNSUPPORTED: n > 64, B beyond 128 KB of 16-bit operands (k > 1024 at
// n > 32), 16-bit rows of odd length.
extern "C" int cogdl_hip_linear_fwd_bf16(const void *x, int x_dtype, const void *w, int w_dtype, const float *bias, void *out,
                                         int64_t rows, int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream) {
    return linear_fwd16_entry<__hip_bfloat16>(x, x_dtype, w, w_dtype, COGDL_HIP_BF16, bias, out, rows, k_dim, n_dim, w_is_n_by_k, stream);
}

extern "C" int cogdl_hip_linear_fwd_f16(const void *x, int x_dtype, const void *w, int w_dtype, const float *bias, void *out,
                                        int64_t rows, int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream) {
    return linear_fwd16_entry<__half>(x, x_dtype, w, w_dtype, COGDL_HIP_F16, bias, out, rows, k_dim, n_dim, w_is_n_by_k, stream);
}
