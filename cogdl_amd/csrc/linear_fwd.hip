// linear_fwd.hip -- tall-skinny dense product of a layer on gfx950 MFMA:  out[R, N] = x[R, K] . B (+ bias),
// R = number of graph nodes, K and N a few tens to hundreds.  Two uses, both torch.nn.Linear's (cogdl/layers/
// gcn_layer.py:52): the forward  x . W^T + b  (B = W^T, W stored [N, K]) and grad_input = grad_out . W (B = W,
// stored [K', N'] = [out, in]).  hipBLASLt needs 39-90 us for these on the ogbn-arxiv-shaped GCN although they only
// stream x and out once (130 MB -> ~22 us); see profiles/r01_gcn_epoch_breakdown.txt.
//
// v_mfma_f32_32x32x2_f32 wants its A operand as A[i = l & 31][k = l >> 5]: 32 different ROWS of x across the lanes,
// i.e. a transposed read of the row-major x.  So every wave stages its 32-row tile of x through its own LDS region
// (coalesced 16-byte global reads, +2 padded rows: conflict-free operand reads), 64 columns at a time.  B is small: the whole of it sits in LDS ([k][n], n contiguous: B[k = l >> 5][n = l & 31] is a conflict-free
// read), loaded once per workgroup; workgroups are persistent and their waves walk row tiles independently (no
// workgroup barrier after the prologue).  fp32 in, fp32 accumulate (an fma chain over k per output element).
// Shapes with more than 64 output columns or a B beyond 96 KB are declined (COGDL_HIP_ERANGE): the caller keeps its BLAS.
#include "common.h"

namespace cogdl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFwdKC = 64;                        // columns of x staged per step
constexpr int kFwdXsStride = kFwdKC + 2;          // padded row of the staged tile: bank (2 row + k) % 64 -- the operand
                                                  // read of lane (row, k parity) touches 64 distinct banks
constexpr size_t kFwdMaxBBytes = 96 * 1024;

template <int NT>  // 32-wide output tiles per wave (N <= 32 * NT)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ out,
                                                         int64_t rows, int k_dim, int n_dim, int w_is_n_by_k, int k_pad) {
    extern __shared__ float smem[];
    float *bs = smem;                                                   // [k_pad][32 * NT]
    float *xs = smem + (size_t)k_pad * 32 * NT + (threadIdx.x >> 6) * 32 * kFwdXsStride;  // this wave's [32][KC + 2]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    // prologue: B, zero padded: bs[k][n ^ swz(k)] = w_is_n_by_k ? w[n][k] : w[k][n].  w is read in ITS memory order
    // (coalesced), eight independent loads per thread in flight (a load-store loop would pay the L2 latency once per
    // element: ~20 us for a 32 KB weight); the transposition, if any, happens in the scattered LDS writes.  For
    // NT == 2 the columns of odd rows are XOR-ed with 32, so that the two half-waves of an operand read (rows k, k + 1)
    // use complementary banks.
    constexpr int kSwz = (NT == 2) ? 32 : 0;
    for (int e = threadIdx.x; e < k_pad * 32 * NT; e += 256) bs[e] = 0.f;
    __syncthreads();
    const int w_elems = k_dim * n_dim;
    for (int e0 = threadIdx.x; e0 < w_elems; e0 += 256 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (e0 + u * 256 < w_elems) ? w[e0 + u * 256] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            if (e < w_elems) {
                const int k = w_is_n_by_k ? e % k_dim : e / n_dim;
                const int n = w_is_n_by_k ? e / k_dim : e % n_dim;
                bs[k * 32 * NT + (n ^ ((k & 1) * kSwz))] = t[u];
            }
        }
    }
    __syncthreads();
    const bool vec_ok = (k_dim & 3) == 0;
    const int64_t n_tiles = (rows + 31) / 32;
    const int n_chunks = (k_pad + kFwdKC - 1) / kFwdKC;
    // Flat sequence of steps (tile, K-chunk) for this wave.  The global loads of step s+1 are issued BEFORE the MFMAs of
    // step s (register double buffer), so the x stream overlaps the matrix pipe inside one wave as well.
    // Branch-free when rows are whole 16-byte groups (k_dim % 4 == 0): a lane outside the matrix reads a dummy address.
    // Nothing may TOUCH the loaded registers before the next step stages them -- a select on the data (or per-lane
    // branches, at whose joins the compiler parks the wave on vmcnt(0)) puts the wait for the prefetch in front of the
    // MFMAs it was meant to overlap with.  The zeroing of out-of-range lanes therefore happens at staging time.
    auto lane_ok = [&](int64_t tile, int chunk, int i) {
        return tile * 32 + (lane >> 4) + 4 * i < rows && chunk * kFwdKC + 4 * (lane & 15) < k_dim;
    };
    auto load_step = [&](int64_t tile, int chunk, float (&v)[8][4]) {
        const int64_t row0 = tile * 32;
        const int k0 = chunk * kFwdKC;
        const int c = k0 + 4 * (lane & 15);
        if (vec_ok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t row = row0 + (lane >> 4) + 4 * i;
                load_vec<float, 4>(lane_ok(tile, chunk, i) ? x + row * k_dim + c : x, v[i]);  // tile >= n_tiles: not ok
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = row0 + (lane >> 4) + 4 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[i][q] = 0.f;
            if (row < rows) {
                const float *p = x + row * k_dim + c;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c + q < k_dim) v[i][q] = p[q];
            }
        }
    };
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    int chunk = 0;
    float v[8][4];
    load_step(tile, 0, v);
    f32x16 acc[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // A finished tile is written at the START of the following step, before that step's prefetch goes out: on gfx9
    // stores count in vmcnt like loads, so stores issued after the prefetch would make the next `vmcnt(0)` wait for
    // their write acknowledgements as well (one exposed HBM round trip per tile); issued first they drain behind a whole
    // step of MFMAs.
    float res[NT][16];
    int64_t res_row0 = -1;  // tile whose results sit in `res` (-1: none)
    auto flush = [&]() {
        if (res_row0 < 0) return;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int n = 32 * a + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = res_row0 + (r & 3) + 8 * (r >> 2) + 4 * half;  // C/D map of the 32x32 MFMA
                if (row < rows && n < n_dim) out[row * n_dim + n] = res[a][r];
            }
        }
        res_row0 = -1;
    };
    while (tile < n_tiles) {
        // stage the current step's 32 x 64 block of x (the wave's own LDS writes are visible to its own later reads)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = !vec_ok || lane_ok(tile, chunk, i);  // the slow path zeroed its registers itself
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xs[((lane >> 4) + 4 * i) * kFwdXsStride + 4 * (lane & 15) + q] = ok ? v[i][q] : 0.f;
        }
        flush();
        // next step's loads go out now
        const bool last_chunk = chunk + 1 == n_chunks;
        const int64_t next_tile = last_chunk ? tile + tile_step : tile;
        const int next_chunk = last_chunk ? 0 : chunk + 1;
        load_step(next_tile, next_chunk, v);
        const int k0 = chunk * kFwdKC;
        const int kc = min(kFwdKC, k_pad - k0);
        // 8 k-pairs per trip: their LDS operand reads are issued together, ahead of the 8 * NT MFMAs (k_pad % 16 == 0)
        for (int kk = 0; kk < kc; kk += 16) {
            float a_op[8], b_op[8][NT];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                a_op[t] = xs[j * kFwdXsStride + kk + 2 * t + half];
                const float *brow = bs + (size_t)(k0 + kk + 2 * t + half) * 32 * NT;  // row parity == half
#pragma unroll
                for (int a = 0; a < NT; ++a) b_op[t][a] = brow[(j + 32 * a) ^ (half * kSwz)];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int a = 0; a < NT; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op[t], b_op[t][a], acc[a], 0, 0, 0);
        }
        if (last_chunk) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int n = 32 * a + j;
                const float b = (bias && n < n_dim) ? bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    res[a][r] = acc[a][r] + b;
                    acc[a][r] = 0.f;
                }
            }
            res_row0 = tile * 32;
        }
        tile = next_tile;
        chunk = next_chunk;
    }
    flush();
}

template <int NT>
static int launch_fwd_gemm(const float *x, const float *w, const float *bias, float *out, int64_t rows, int64_t k_dim,
                           int64_t n_dim, int w_is_n_by_k, hipStream_t s) {
    const int k_pad = (int)((k_dim + 15) / 16 * 16);  // whole trips of 8 k-pairs; the padding is zeros in B and x
    const size_t b_bytes = (size_t)k_pad * 32 * NT * sizeof(float);
    if (b_bytes > kFwdMaxBBytes) return COGDL_HIP_ERANGE;
    const size_t lds = b_bytes + (size_t)4 * 32 * kFwdXsStride * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {  // more than 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_kernel<NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFwdMaxBBytes + 40 * 1024));
        attr_set = true;
    }
    const int64_t n_tiles = (rows + 31) / 32;
    // persistent workgroups: exactly as many as are resident at once (LDS- or register-limited), or a second round
    // of workgroups would start when the first is done
    static thread_local size_t cached_lds = 0;  // the occupancy query costs a few microseconds of host time
    static thread_local int cached_per_cu = 0;
    if (cached_per_cu == 0 || cached_lds != lds) {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void *>(&linear_fwd_kernel<NT>), 256,
                                                         lds) != hipSuccess || q < 1)
            q = (int)std::max<int64_t>(1, std::min<int64_t>(2, (160 * 1024) / (int64_t)lds));
        cached_lds = lds;
        cached_per_cu = q;
    }
    const int per_cu = cached_per_cu;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, (int64_t)256 * std::min(per_cu, 4)));
    hipLaunchKernelGGL((linear_fwd_kernel<NT>), dim3(grid), dim3(256), lds, s, x, w, bias, out, rows, (int)k_dim, (int)n_dim,
                       w_is_n_by_k, k_pad);
    return launch_status();
}

}  // namespace cogdl

using namespace cogdl;

// out[rows, n] = x[rows, k] . B (+ bias[n]);  w_is_n_by_k != 0: B = w^T with w stored [n, k] (Linear forward);
// w_is_n_by_k == 0: B = w stored [k, n] (grad_input = grad_out . W).  COGDL_HIP_ERANGE: shape not covered.
extern "C" int cogdl_hip_linear_fwd_f32(const float *x, const float *w, const float *bias, float *out, int64_t rows,
                                        int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream) {
    if (rows < 0 || k_dim <= 0 || n_dim <= 0) return COGDL_HIP_EINVAL;
    if (rows == 0) return COGDL_HIP_OK;
    if (!x || !w || !out) return COGDL_HIP_EINVAL;
    // Measured on MI355X (profiles/r01_ops_bench.txt): with <= 64 output columns this kernel beats hipBLASLt on the
    // node-count-tall shapes (169 k x 128 -> 64: 30-44 us vs 83 us; x 64 -> 40: vs 38-65 us); with 128 columns and more
    // hipBLASLt wins (2.4 M x 100 -> 128: 0.90 ms vs 1.6 ms here).  Decline those.
    if (n_dim > 64 || k_dim > 4096) return COGDL_HIP_ERANGE;
    if (!aligned_to(x, 16) || !aligned_to(out, 4)) return COGDL_HIP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (n_dim <= 32) return launch_fwd_gemm<1>(x, w, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s);
    return launch_fwd_gemm<2>(x, w, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s);
}
