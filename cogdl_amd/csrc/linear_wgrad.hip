// linear_wgrad.hip -- weight (and bias) gradient of a dense layer on tall-skinny operands, gfx950 MFMA.
//     grad_w[out, in] = grad_out[K, out]^T . x[K, in]          grad_b[out] = column sums of grad_out
// with K = number of graph nodes (10^5 .. 10^8) and in/out = a few tens to hundreds: the backward of the
// `self.linear(x)` inside every CogDL layer (cogdl/layers/gcn_layer.py:52, gat_layer.py:60, sage_layer.py:72).
// This is the one place of the message-passing path where the dense node-feature x weight product dominates:
// on the ogbn-arxiv-shaped GCN epoch torch hands these two reductions to hipBLASLt kernels that run on 4 workgroups
// (349 + 442 us of a 2.2 ms epoch, profiles/r01_gcn_epoch_breakdown.txt) although they only have to stream
// x and grad_out once (130 MB -> ~20 us at HBM rate; 2.8 GFLOP -> ~18 us at the fp32 MFMA rate).
//
// Kernel: split-K over <= 1024 workgroups.  A workgroup owns TM x TN output tiles of 32x32 (<= 8) and a slab of rows;
// its 4 waves share the TILES (T >= 4: every wave walks all row pairs of the slab for its T/4 tiles; fewer tiles: the
// waves also split the row pairs), so no wave ever has to be combined with another inside the kernel -- no LDS, 16-32
// accumulator registers per wave, 8 waves per SIMD to hide the loads.  Operands go STRAIGHT from global memory into
// v_mfma_f32_32x32x2_f32: the layout of that instruction (lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]) is exactly "32 consecutive floats of row k0 in lanes 0..31, of row k0+1 in lanes 32..63",
// i.e. a coalesced read of two rows; no transpose.  Rows shared by several waves of a workgroup (grad_out is needed by
// every wave) come from the CU's L1.  fp32 in, fp32 accumulate (bit-for-bit an fma chain per output element, k
// ascending inside a wave's share); a second kernel adds the per-slab partials in a fixed order: deterministic.
// Roofline: HBM (K * (in + out) * 4 bytes, each read once per column group) vs fp32 MFMA (2 * K * in * out flop at
// 157 TFLOP/s); for in = 128, out = 64 both give ~20 us.
#include "common.h"

namespace cogdl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWgradUnroll = 4;      // row pairs per batch; two batches in flight per wave
constexpr int kWgradMaxSlabs = 512;

struct WgradGeom {
    int tm, tn;            // 32x32 tiles per workgroup along in / out
    int m_groups, n_groups;
    int ksplit;            // k-split instances per workgroup (4 waves / waves needed for the tiles)
    int64_t slabs, rows_per_slab;
    int64_t block_floats;  // floats of one (slab, k-split, m-group, n-group) partial: tiles * 16 * 64 + 32 * tn (bias)
};

static WgradGeom wgrad_geometry(int64_t k_rows, int64_t in_f, int64_t out_f) {
    auto tiles = [](int64_t cols, int cap) {  // power of two >= ceil(cols / 32), capped
        int t = 1;
        while (t < cap && (int64_t)32 * t < cols) t *= 2;
        return t;
    };
    WgradGeom g;
    g.tm = tiles(in_f, 4);
    g.tn = tiles(out_f, 8 / g.tm);
    g.m_groups = (int)((in_f + 32 * g.tm - 1) / (32 * g.tm));
    g.n_groups = (int)((out_f + 32 * g.tn - 1) / (32 * g.tn));
    const int t = g.tm * g.tn;
    g.ksplit = (t >= 4) ? 1 : 4 / t;
    g.slabs = std::max<int64_t>(1, std::min<int64_t>(kWgradMaxSlabs, (k_rows + 127) / 128));
    g.rows_per_slab = ((k_rows + g.slabs - 1) / g.slabs + 7) / 8 * 8;
    g.slabs = std::max<int64_t>(1, (k_rows + g.rows_per_slab - 1) / g.rows_per_slab);
    g.block_floats = (int64_t)t * 16 * 64 + 32 * g.tn;
    return g;
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                           float *__restrict__ partial, int64_t k_rows, int in_f,
                                                           int out_f, int64_t rows_per_slab, int64_t block_floats) {
    constexpr int T = TM * TN;
    constexpr int KW = (T >= 4) ? 1 : 4 / T;       // k-split instances
    constexpr int WPI = 4 / KW;                      // waves per instance
    constexpr int TPW = T / WPI;                     // tiles per wave: consecutive t = tn * TM + tm
    constexpr int WM = (TPW < TM) ? TPW : TM;        // ... = WM tiles along in  x  WN along out
    constexpr int WN = TPW / WM;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: batch selection = scalar branches
    const int half = lane >> 5, j = lane & 31;
    const int ks = wave / WPI;                       // which share of the row pairs
    const int t0 = (wave % WPI) * TPW;               // first tile of this wave
    const int tm0 = t0 % TM, tn0 = t0 / TM;
    const int m_base = ((int)blockIdx.y * TM + tm0) * 32 + j;  // column of x for this lane's first tile
    const int n_base = ((int)blockIdx.z * TN + tn0) * 32 + j;  // column of grad_out
    const int64_t k_begin = (int64_t)blockIdx.x * rows_per_slab;
    const int64_t k_end = min(k_rows, k_begin + rows_per_slab);

    f32x16 acc[WN][WM];
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bs[WN];
#pragma unroll
    for (int a = 0; a < WN; ++a) bs[a] = 0.f;

    // Loads are bounds-checked BUFFER loads on a per-slab descriptor: rows past the slab (and lanes past the last
    // column, which get an offset beyond any slab) read 0 without a branch, so the 3*kWgradUnroll loads of a batch are
    // straight-line code the compiler can keep in flight behind the MFMAs of the previous batch.
    const int slab_rows = (int)(k_end - k_begin);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + k_begin * in_f), 0, slab_rows * in_f * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(g + k_begin * out_f), 0, slab_rows * out_f * (int)sizeof(float), 0x00020000);
    constexpr int kOob = (int)0x80000000;  // as an unsigned byte offset: beyond every slab
    int xcol[WM], gcol[WN];
#pragma unroll
    for (int b = 0; b < WM; ++b) xcol[b] = (m_base + 32 * b < in_f) ? (m_base + 32 * b) * 4 : kOob;
#pragma unroll
    for (int a = 0; a < WN; ++a) gcol[a] = (n_base + 32 * a < out_f) ? (n_base + 32 * a) * 4 : kOob;
    const int x_row_bytes = in_f * 4, g_row_bytes = out_f * 4;
    constexpr int kStep = 2 * KW * kWgradUnroll;  // rows covered by one batch of this wave
    // Batch i of this wave = rows r0(i) + 2*KW*u, u < kWgradUnroll, r0(i) = 2*ks + half + i*kStep.  When the whole batch
    // lies inside the slab (all but the last one or two) the per-row part of the address is the SCALAR offset of the
    // buffer instruction (u * 2*KW * row_bytes: not bounds-checked, and needs no check) and the lane part is one add per
    // column stream; only the tail batches pay for per-row checks.
    auto load_full = [&](int i, float (&xv)[kWgradUnroll][WM], float (&gv)[kWgradUnroll][WN]) {
        const int r0 = 2 * ks + half + i * kStep;
        int xo[WM], go[WN];
#pragma unroll
        for (int b = 0; b < WM; ++b) xo[b] = r0 * x_row_bytes + xcol[b];  // (top bit of xcol survives: no wrap)
#pragma unroll
        for (int a = 0; a < WN; ++a) go[a] = r0 * g_row_bytes + gcol[a];
#pragma unroll
        for (int u = 0; u < kWgradUnroll; ++u) {
#pragma unroll
            for (int b = 0; b < WM; ++b)
                xv[u][b] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, xo[b], u * 2 * KW * x_row_bytes, 0));
#pragma unroll
            for (int a = 0; a < WN; ++a)
                gv[u][a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gr, go[a], u * 2 * KW * g_row_bytes, 0));
        }
    };
    auto load_tail = [&](int i, float (&xv)[kWgradUnroll][WM], float (&gv)[kWgradUnroll][WN]) {
        const int r0 = 2 * ks + half + i * kStep;
#pragma unroll
        for (int u = 0; u < kWgradUnroll; ++u) {
            const int r = r0 + 2 * KW * u;                       // row inside the slab; past its end -> reads 0
            const int xrow = (r < slab_rows) ? r * x_row_bytes : kOob;
            const int grow = (r < slab_rows) ? r * g_row_bytes : kOob;
            // offset = row + column, with the top bit forced whenever either part is out of bounds (pure bit math:
            // no branches, and the sum of two out-of-bounds markers cannot wrap back into range)
#pragma unroll
            for (int b = 0; b < WM; ++b)
                xv[u][b] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    xr, (xrow + xcol[b]) | (((xrow | xcol[b]) >> 31) & kOob), 0, 0));
#pragma unroll
            for (int a = 0; a < WN; ++a)
                gv[u][a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    gr, (grow + gcol[a]) | (((grow | gcol[a]) >> 31) & kOob), 0, 0));
        }
    };
    auto mfma_batch = [&](const float (&xv)[kWgradUnroll][WM], const float (&gv)[kWgradUnroll][WN]) {
#pragma unroll
        for (int u = 0; u < kWgradUnroll; ++u)
#pragma unroll
            for (int a = 0; a < WN; ++a) {
                bs[a] += gv[u][a];
#pragma unroll
                for (int b = 0; b < WM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[u][a], xv[u][b], acc[a][b], 0, 0, 0);
            }
    };
    // Two register buffers: the loads of batch i+1 are in flight behind the MFMAs of batch i.  The pipelined loop runs
    // over the FULL batches only and its loads are unconditional (an index past the last full batch is clamped: the
    // reload is discarded): with the full/tail choice inside the loop the loads sat in two conditional blocks, the
    // compiler could not count them and made the MFMAs of batch i wait for the loads of batch i+1 (vmcnt(0)).
    const int n_batches = (slab_rows - 2 * ks + kStep - 1) / kStep;  // wave-uniform; batches past it would read zeros
    const int full_span = slab_rows - 2 * ks - 2 - 2 * KW * (kWgradUnroll - 1);  // >= i * kStep  <=>  batch i is full
    const int n_full = full_span >= 0 ? min(n_batches, full_span / kStep + 1) : 0;
    float xa[kWgradUnroll][WM], ga[kWgradUnroll][WN], xb[kWgradUnroll][WM], gb[kWgradUnroll][WN];
    if (n_full > 0) {
        load_full(0, xa, ga);
        int i = 0;
        for (; i + 1 < n_full; i += 2) {  // two batches per trip: A holds batch i on entry
            // Pin the order.  The loads are side-effect-free intrinsics: without the memory clobber LLVM sinks them to
            // their first use (behind the MFMAs, one register buffer, zero overlap); the sched_barrier stops the machine
            // scheduler from doing the same later.
            load_full(i + 1, xb, gb);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_batch(xa, ga);
            __builtin_amdgcn_sched_barrier(0);
            load_full(min(i + 2, n_full - 1), xa, ga);  // (past the end: a discarded reload of the last full batch)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_batch(xb, gb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i < n_full) mfma_batch(xa, ga);  // odd count: the last full batch sits in A
    }
    for (int i = n_full; i < n_batches; ++i) {  // at most two checked tail batches
        load_tail(i, xb, gb);
        mfma_batch(xb, gb);
    }

    // fragment order [tile][register][lane], one partial per (slab, k-split instance)
    float *dst = partial + ((((int64_t)blockIdx.x * KW + ks) * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * block_floats;
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b) {
            const int t = (tn0 + a) * TM + tm0 + b;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * kWave + lane] = acc[a][b][r];
        }
    if (tm0 == 0) {  // the wave holding the first in-tile of an out-tile also owns that tile's bias partial
#pragma unroll
        for (int a = 0; a < WN; ++a) {
            const float v = bs[a] + __shfl_xor(bs[a], 32, kWave);  // rows k and k+1
            if (half == 0) dst[T * 16 * kWave + (tn0 + a) * 32 + j] = v;
        }
    }
}

// Sum the partials and write grad_w / grad_b.  A workgroup owns 64 consecutive fragment elements; its 4 waves take
// the partials p = w, w+4, w+8, ... (8 independent coalesced loads in flight per lane) and are combined in wave order:
// a fixed summation order, hence deterministic.
template <int TM, int TN>
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ partial,
                                                                  float *__restrict__ grad_w, float *__restrict__ grad_b,
                                                                  int in_f, int out_f, int64_t n_partials, int m_groups,
                                                                  int n_groups, int64_t block_floats) {
    constexpr int T = TM * TN;
    __shared__ float part[4][kWave];
    const int mg = blockIdx.y, ng = blockIdx.z;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int e = blockIdx.x * kWave + lane;  // element of the fragment block (+ the bias tail)
    const bool live = e < T * 16 * kWave + 32 * TN;
    const int64_t stride = (int64_t)m_groups * n_groups * block_floats;
    const float *src = partial + ((int64_t)mg * n_groups + ng) * block_floats + (live ? e : 0);
    float sum = 0.f;
    int64_t s = w;
    for (; s + 28 < n_partials; s += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(s + 4 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; s < n_partials; s += 4) sum += src[s * stride];
    part[w][lane] = sum;
    __syncthreads();
    if (w != 0 || !live) return;
    sum = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    if (e < T * 16 * kWave) {
        const int fl = e & (kWave - 1), r = (e >> 6) & 15, t = e >> 10;
        const int tn = t / TM, tm = t % TM;
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (fl >> 5), jj = fl & 31;  // C/D map of the 32x32 MFMA
        const int n = (ng * TN + tn) * 32 + i, m = (mg * TM + tm) * 32 + jj;
        if (n < out_f && m < in_f) grad_w[(int64_t)n * in_f + m] = sum;
    } else if (grad_b && mg == 0) {
        const int n = ng * TN * 32 + (e - T * 16 * kWave);
        if (n < out_f) grad_b[n] = sum;
    }
}

template <int TM, int TN>
static int launch_wgrad(const float *x, const float *g, float *grad_w, float *grad_b, int64_t k_rows, int64_t in_f,
                        int64_t out_f, const WgradGeom &geo, float *ws, hipStream_t s) {
    hipLaunchKernelGGL((linear_wgrad_kernel<TM, TN>), dim3((unsigned)geo.slabs, geo.m_groups, geo.n_groups), dim3(256), 0,
                       s, x, g, ws, k_rows, (int)in_f, (int)out_f, geo.rows_per_slab, geo.block_floats);
    const unsigned rb = (unsigned)((geo.block_floats + kWave - 1) / kWave);
    hipLaunchKernelGGL((linear_wgrad_reduce_kernel<TM, TN>), dim3(rb, geo.m_groups, geo.n_groups), dim3(256), 0, s, ws,
                       grad_w, grad_b, (int)in_f, (int)out_f, geo.slabs * geo.ksplit, geo.m_groups, geo.n_groups,
                       geo.block_floats);
    return launch_status();
}

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_linear_wgrad_workspace_bytes(int64_t k_rows, int64_t in_features, int64_t out_features) {
    if (k_rows <= 0 || in_features <= 0 || out_features <= 0) return 256;
    const WgradGeom g = wgrad_geometry(k_rows, in_features, out_features);
    return ((size_t)g.slabs * g.ksplit * g.m_groups * g.n_groups * (size_t)g.block_floats * sizeof(float) + 255) / 256 * 256;
}

extern "C" int cogdl_hip_linear_wgrad_f32(const float *x, const float *grad_out, float *grad_w, float *grad_b,
                                          int64_t k_rows, int64_t in_features, int64_t out_features, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    if (k_rows < 0 || in_features <= 0 || out_features <= 0 || !grad_w) return COGDL_HIP_EINVAL;
    if (in_features > 0x7fffffff / 64 || out_features > 0x7fffffff / 64) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (k_rows == 0) {  // empty reduction: zeros
        hipError_t e = fill_u32_async(grad_w, 0u, (size_t)in_features * out_features, s);
        if (e == hipSuccess && grad_b) e = fill_u32_async(grad_b, 0u, (size_t)out_features, s);
        if (e != hipSuccess) {
            g_last_hip_error = (int)e;
            return COGDL_HIP_ELAUNCH;
        }
        return COGDL_HIP_OK;
    }
    if (!x || !grad_out || !workspace) return COGDL_HIP_EINVAL;
    if (workspace_bytes < cogdl_hip_linear_wgrad_workspace_bytes(k_rows, in_features, out_features)) return COGDL_HIP_EWORKSPACE;
    if (!aligned_to(x, 4) || !aligned_to(grad_out, 4) || !aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    const WgradGeom geo = wgrad_geometry(k_rows, in_features, out_features);
    float *ws = (float *)workspace;
#define WG(A_, B_) \
    if (geo.tm == A_ && geo.tn == B_) return launch_wgrad<A_, B_>(x, grad_out, grad_w, grad_b, k_rows, in_features, out_features, geo, ws, s);
    WG(4, 2) WG(4, 1) WG(2, 4) WG(2, 2) WG(2, 1) WG(1, 8) WG(1, 4) WG(1, 2) WG(1, 1)
#undef WG
    return COGDL_HIP_EINVAL;
}
