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
// Shapes with more than 64 output columns or a B beyond 96 KB are declined (COGDL_HIP_EUNSUPPORTED): the caller keeps its BLAS.
#include "common.h"

namespace cogdl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFwdKC = 64;                        // columns of x staged per step
constexpr int kFwdXsStride = kFwdKC + 2;          // padded row of the staged tile: bank (2 row + k) % 64 -- the operand
                                                  // read of lane (row, k parity) touches 64 distinct banks
constexpr size_t kFwdMaxBBytes = 96 * 1024;

// NT: 32-wide output tiles per wave (N <= 32 * NT).  VEC: rows of x are whole 16-byte groups (k_dim % 4 == 0) -- the
// buffer-addressed fast path; a template parameter so that the element-wise path's pending-load bookkeeping cannot leak
// into the fast loop's s_waitcnt placement.
template <int NT, bool VEC>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ out,
                                                         int64_t rows, int k_dim, int n_dim, int w_is_n_by_k, int k_pad) {
    extern __shared__ float smem[];
    float *bs = smem;                                                   // [k_pad][32 * NT]
    float *xs = smem + (size_t)k_pad * 32 * NT + (threadIdx.x >> 6) * 32 * kFwdXsStride;  // this wave's [32][KC + 2]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: tile ids and descriptors stay in SGPRs
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
    float *tmp = smem + (size_t)k_pad * 32 * NT;  // the (still unused) x staging area of the four waves
    const bool via_tmp = w_is_n_by_k && (int64_t)n_dim * (k_dim + 1) <= 4 * 32 * kFwdXsStride;
    if (via_tmp) {
        // w is [n][k]: a direct scatter bs[k][n] = w[e] has all lanes of a wave on one bank (k-stride 32 * NT floats:
        // 64-way conflicts, ~3 us per workgroup for a 32 KB weight).  Two conflict-free hops instead: w -> tmp[n][k] with
        // an odd row stride (coalesced read, linear write), then tmp -> bs with n running across the lanes.
        const int ts = k_dim + 1;
        for (int e0 = threadIdx.x; e0 < w_elems; e0 += 256 * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = (e0 + u * 256 < w_elems) ? w[e0 + u * 256] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256;
                if (e < w_elems) tmp[(e / k_dim) * ts + e % k_dim] = t[u];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < w_elems; e += 256) {
            const int k = e / n_dim, n = e % n_dim;
            bs[k * 32 * NT + (n ^ ((k & 1) * kSwz))] = tmp[n * ts + k];
        }
    } else {
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
    }
    __syncthreads();
    const int64_t n_tiles = (rows + 31) / 32;
    const int n_chunks = (k_pad + kFwdKC - 1) / kFwdKC;
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    // ---- addressing: bounds-checked BUFFER loads/stores on per-tile descriptors ---------------------------------------
    // A tile's descriptor covers exactly its valid rows, so rows past the matrix read 0 / drop the store; a lane past the
    // last column gets an offset beyond any tile (kOob).  Everything is straight-line code: one add per load or store,
    // no per-lane branches (at whose joins the compiler parks the wave on vmcnt(0)), no selects on loaded data (which
    // would put the wait for a prefetch in front of the MFMAs it overlaps with), exact vmcnt bookkeeping by the compiler.
    constexpr int kOob = (int)0x80000000;
    const int x_row_bytes = k_dim * 4, o_row_bytes = n_dim * 4;
    const int ld_lane = (lane >> 4) * x_row_bytes + 16 * (lane & 15);  // row (lane >> 4), columns 4 (lane & 15) .. + 3
    int st_lane[NT];                                                   // row 4 half, column 32 a + j of the output tile
#pragma unroll
    for (int a = 0; a < NT; ++a) st_lane[a] = (32 * a + j < n_dim) ? 4 * half * o_row_bytes + (32 * a + j) * 4 : kOob;
    auto x_rsrc = [&](int64_t tile) {
        const int64_t row0 = tile * 32;
        const int valid = (int)max((int64_t)0, min((int64_t)32, rows - row0));
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (valid ? row0 : 0) * k_dim), 0, valid * x_row_bytes,
                                                 0x00020000);
    };
    auto load_step = [&](int64_t tile, int chunk, float (&v)[8][4]) {
        const int k0 = chunk * kFwdKC;
        if constexpr (VEC) {
            const __amdgpu_buffer_rsrc_t r = x_rsrc(tile);
            const int col_ok = k0 + 4 * (lane & 15) < k_dim;  // whole 16-byte groups: k_dim % 4 == 0
            const int base = col_ok ? ld_lane : kOob;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(r, base + i * 4 * x_row_bytes, k0 * 4, 0);
                v[i][0] = __uint_as_float(q[0]);
                v[i][1] = __uint_as_float(q[1]);
                v[i][2] = __uint_as_float(q[2]);
                v[i][3] = __uint_as_float(q[3]);
            }
            return;
        }
        const int64_t row0 = tile * 32;
        const int c = k0 + 4 * (lane & 15);
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // rows that are not whole 16-byte groups: element-wise, checked
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
    float bias_v[NT];  // this lane's output columns 32 a + j
#pragma unroll
    for (int a = 0; a < NT; ++a) bias_v[a] = (bias && 32 * a + j < n_dim) ? bias[32 * a + j] : 0.f;
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
    // stores count in vmcnt like loads; issued here they drain behind a whole step of MFMAs.
    float res[NT][16];
    int64_t res_tile = -1;  // tile whose results sit in `res` (-1: none)
    auto flush = [&]() {
        if (res_tile < 0) return;  // wave-uniform
        const int64_t row0 = res_tile * 32;
        const int valid = (int)min((int64_t)32, rows - row0);
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc(out + row0 * n_dim, 0, valid * o_row_bytes, 0x00020000);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int q = 0; q < 16; ++q)  // C/D map of the 32x32 MFMA: register q holds row (q & 3) + 8 (q >> 2) + 4 half
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[a][q]), r,
                                                      st_lane[a] + ((q & 3) + 8 * (q >> 2)) * o_row_bytes, 0, 0);
        res_tile = -1;
    };
    while (tile < n_tiles) {
        // stage the current step's 32 x 64 block of x (the wave's own LDS writes are visible to its own later reads)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[((lane >> 4) + 4 * i) * kFwdXsStride + 4 * (lane & 15) + q] = v[i][q];
        flush();
        // next step's loads go out now
        const bool last_chunk = chunk + 1 == n_chunks;
        const int64_t next_tile = last_chunk ? tile + tile_step : tile;
        const int next_chunk = last_chunk ? 0 : chunk + 1;
        load_step(next_tile, next_chunk, v);
        const int k0 = chunk * kFwdKC;
        const int kc = min(kFwdKC, k_pad - k0);
        // The chunk is a sequence of groups of two k-pairs (4 columns, 2 * NT MFMAs = 128 * NT pipe cycles).  The LDS
        // operand reads of group g + 1 are issued before the MFMAs of group g (two register sets), so that their
        // latency hides behind the matrix pipe instead of opening a bubble after every group.
        const int n_groups = kc / 4;  // a multiple of 4 (k_pad % 16 == 0)
        float a_op[2][2], b_op[2][2][NT];
        auto read_group = [&](int g, float (&ao)[2], float (&bo)[2][NT]) {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                ao[p2] = xs[j * kFwdXsStride + 4 * g + 2 * p2 + half];
                const float *brow = bs + (size_t)(k0 + 4 * g + 2 * p2 + half) * 32 * NT;  // row parity == half
#pragma unroll
                for (int a = 0; a < NT; ++a) bo[p2][a] = brow[(j + 32 * a) ^ (half * kSwz)];
            }
        };
        read_group(0, a_op[0], b_op[0]);
        for (int g4 = 0; g4 < n_groups; g4 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g4 + u;
                // unconditional: the read past the chunk's last group stays inside the LDS allocation (row padding of
                // xs / the region behind bs) and is discarded; a condition here would cost the exact lgkmcnt bookkeeping
                read_group(g + 1, a_op[(u + 1) & 1], b_op[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                    for (int a = 0; a < NT; ++a)
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op[u & 1][p2], b_op[u & 1][p2][a], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (last_chunk) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    res[a][r] = acc[a][r] + bias_v[a];
                    acc[a][r] = 0.f;
                }
            res_tile = tile;
        }
        tile = next_tile;
        chunk = next_chunk;
    }
    flush();
}

template <int NT, bool VEC>
static int launch_fwd_gemm(const float *x, const float *w, const float *bias, float *out, int64_t rows, int64_t k_dim,
                           int64_t n_dim, int w_is_n_by_k, hipStream_t s) {
    const int k_pad = (int)((k_dim + 15) / 16 * 16);  // whole trips of 8 k-pairs; the padding is zeros in B and x
    const size_t b_bytes = (size_t)k_pad * 32 * NT * sizeof(float);
    if (b_bytes > kFwdMaxBBytes) return COGDL_HIP_EUNSUPPORTED;
    const size_t lds = b_bytes + (size_t)4 * 32 * kFwdXsStride * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {  // more than 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_kernel<NT, VEC>),
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
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void *>(&linear_fwd_kernel<NT, VEC>), 256,
                                                         lds) != hipSuccess || q < 1)
            q = (int)std::max<int64_t>(1, std::min<int64_t>(2, (160 * 1024) / (int64_t)lds));
        cached_lds = lds;
        cached_per_cu = q;
    }
    const int per_cu = cached_per_cu;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, (int64_t)256 * std::min(per_cu, 4)));
    hipLaunchKernelGGL((linear_fwd_kernel<NT, VEC>), dim3(grid), dim3(256), lds, s, x, w, bias, out, rows, (int)k_dim, (int)n_dim,
                       w_is_n_by_k, k_pad);
    return launch_status();
}

}  // namespace cogdl

using namespace cogdl;

// out[rows, n] = x[rows, k] . B (+ bias[n]);  w_is_n_by_k != 0: B = w^T with w stored [n, k] (Linear forward);
// w_is_n_by_k == 0: B = w stored [k, n] (grad_input = grad_out . W).  COGDL_HIP_EUNSUPPORTED: shape not covered.
extern "C" int cogdl_hip_linear_fwd_f32(const float *x, const float *w, const float *bias, float *out, int64_t rows,
                                        int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream) {
    if (rows < 0 || k_dim <= 0 || n_dim <= 0) return COGDL_HIP_EINVAL;
    if (rows == 0) return COGDL_HIP_OK;
    if (!x || !w || !out) return COGDL_HIP_EINVAL;
    // Measured on MI355X (profiles/r01_ops_bench.txt): with <= 64 output columns this kernel beats hipBLASLt on the
    // node-count-tall shapes (169 k x 128 -> 64: 30-44 us vs 83 us; x 64 -> 40: vs 38-65 us); with 128 columns and more
    // hipBLASLt wins (2.4 M x 100 -> 128: 0.90 ms vs 1.6 ms here).  Decline those.
    if (n_dim > 64 || k_dim > 4096) return COGDL_HIP_EUNSUPPORTED;
    if (!aligned_to(x, 16) || !aligned_to(out, 4)) return COGDL_HIP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (k_dim & 3) == 0;
    if (n_dim <= 32)
        return vec ? launch_fwd_gemm<1, true>(x, w, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s)
                   : launch_fwd_gemm<1, false>(x, w, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s);
    return vec ? launch_fwd_gemm<2, true>(x, w, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s)
               : launch_fwd_gemm<2, false>(x, w, bias, out, rows, k_dim, n_dim, w_is_n_by_k, s);
}
