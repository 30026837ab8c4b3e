// mhsddmm.hip -- multi-head SDDMM for gfx950: out[e,h] = < grad[row(e),h,:], feat[colind[e],h,:] >.
// Replaces cogdl/operators/spmm/multiheadSddmm.cu:6-113 (binary-searched rows, warp=32).
// Row-group engine of rowreduce.h (as spmm.hip/sddmm.hip, incl. the long-row path): a group of LPR lanes owns one CSR row and
// the whole [H*F] feature row (VEC columns per lane, all inside one head); grad[row] stays in
// registers; each neighbour row is one coalesced load; the F/VEC lanes of a head reduce their
// partial dot with a butterfly and the head's first lane stores out[e,h].
#include "rowreduce.h"

namespace cogdl {

// Per-edge outputs (no per-row state): long rows are cut into chunks processed by whole workgroups.
template <int VEC_, int LPR_, int UNROLL_, bool ONE_HEAD>
struct MhsddmmOp {
    static constexpr int VEC = VEC_, LPR = LPR_, UNROLL = UNROLL_, kRec = 1;
    static constexpr bool kReduce = false;
    static constexpr int kLds = 0;
    const float *grad, *feat;
    float *out;
    int heads, fdim, lph;  // lph = lanes per head

    struct Ctx {
        int l, c0, hd;
        bool col_ok, head_lane;
        float a[VEC];
    };
    struct State { float p[UNROLL_]; };  // ONE_HEAD: the batch's per-lane partial dot products
    struct LaneVals {};
    struct Batch { float b[UNROLL][VEC]; };

    __device__ __forceinline__ Ctx make_ctx(int l, int) const {
        Ctx c;
        c.l = l;
        c.col_ok = l * VEC < heads * fdim;
        c.c0 = c.col_ok ? l * VEC : 0;
        c.hd = c.c0 / fdim;
        c.head_lane = c.col_ok && (l % lph) == 0;
        return c;
    }
    __device__ __forceinline__ void row_load(Ctx &c, int64_t row, bool ok) const {
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.a[i] = 0.f;
        if (ok && c.col_ok) load_vec<float, VEC>(grad + row * (int64_t)(heads * fdim) + c.c0, c.a);
    }
    __device__ __forceinline__ void init_zero(State &) const {}
    __device__ __forceinline__ void init(const Ctx &, State &, int64_t, bool) const {}
    __device__ __forceinline__ void lane_load(const Ctx &, LaneVals &, int64_t) const {}
    __device__ __forceinline__ void fetch(const Ctx &c, Batch &b, int u, int col, int64_t, const LaneVals &, int,
                                          int) const {
        load_vec<float, VEC>(feat + (int64_t)col * (heads * fdim) + c.c0, b.b[u]);
    }
    __device__ __forceinline__ void apply(const Ctx &c, State &s_, const Batch &b, int u, bool valid, int64_t e,
                                          int) const {
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) p = fmaf(c.a[i], c.col_ok ? b.b[u][i] : 0.f, p);
        if constexpr (ONE_HEAD) {
            s_.p[u] = p;  // reduced jointly in batch_end
        } else {
#pragma unroll
            for (int s = LPR / 2; s > 0; s >>= 1)
                if (s < lph) p += __shfl_xor(p, s, kWave);
            if (valid && c.head_lane) out[e * heads + c.hd] = p;
        }
    }
    // Single head: the whole group reduces, so the UNROLL dot products of a batch share one transpose reduce.
    __device__ __forceinline__ void batch_end(const Ctx &c, State &s_, int base, int j, int cnt) const {
        if constexpr (ONE_HEAD) {
            const float r = transpose_reduce<UNROLL, LPR>(s_.p, c.l);
            constexpr int kSpan = LPR / UNROLL;
            const int idx = c.l / kSpan;
            if (c.l % kSpan == 0 && j + idx < cnt) out[base + j + idx] = r;
        }
    }
    __device__ __forceinline__ void chunk_begin(Ctx &, State &, int, int, int, int, int, float *, const LaneVals &) const {}
    __device__ __forceinline__ void chunk_end(const Ctx &, State &, int, int) const {}
    __device__ __forceinline__ void row_end(const Ctx &, const State &, int64_t, bool) const {}
    __device__ __forceinline__ void pack(const State &, float (&)[kRec]) const {}
    __device__ __forceinline__ void unpack(State &, const float (&)[kRec]) const {}
    __device__ __forceinline__ void merge(const Ctx &, State &, const State &) const {}
};

// Any shape: a group of 64 lanes per row; lane t walks (edge, head) pairs and dots F terms serially.
__global__ __launch_bounds__(256) void mhsddmm_generic_kernel(const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ colind,
                                                              const float *__restrict__ grad,
                                                              const float *__restrict__ feat,
                                                              float *__restrict__ out, int64_t m, int heads,
                                                              int fdim, XcdMap n_rowblocks) {
    const int64_t rb = xcd_remap(blockIdx.x, n_rowblocks);
    if (rb < 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = rb * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int64_t lo = rowptr[row], hi = rowptr[row + 1];
    const int64_t k = (int64_t)heads * fdim;
    for (int64_t p = lo * heads + lane; p < hi * heads; p += kWave) {
        const int64_t e = p / heads;
        const int hd = (int)(p - e * heads);
        const float *a = grad + row * k + (int64_t)hd * fdim;
        const float *b = feat + (int64_t)colind[e] * k + (int64_t)hd * fdim;
        float acc = 0.f;
        for (int t = 0; t < fdim; ++t) acc = fmaf(a[t], b[t], acc);
        out[p] = acc;
    }
}

struct MhsddmmArgs {
    const int32_t *rowptr, *colind;
    const float *grad, *feat;
    float *out;
    int64_t m, h, f, nnz;
};

template <int VEC, int LPR>
static int launch_mhsddmm(const MhsddmmArgs &a, void *ws, size_t wsb, hipStream_t s) {
    // one head: reduce over the whole (zero-padded) group, so F/VEC need not be a power of two
    if (a.h == 1) {
        MhsddmmOp<VEC, LPR, 4, true> op{a.grad, a.feat, a.out, 1, (int)a.f, LPR};
        return launch_rowreduce(op, a.rowptr, a.colind, a.m, a.nnz, 1, ws, wsb, s);
    }
    MhsddmmOp<VEC, LPR, 4, false> op{a.grad, a.feat, a.out, (int)a.h, (int)a.f, (int)(a.f / VEC)};
    return launch_rowreduce(op, a.rowptr, a.colind, a.m, a.nnz, 1, ws, wsb, s);
}

template <int VEC>
static int dispatch_mhsddmm(const MhsddmmArgs &a, void *ws, size_t wsb, hipStream_t s) {
    const int64_t need = (a.h * a.f + VEC - 1) / VEC;
    if (need <= 8) return launch_mhsddmm<VEC, 8>(a, ws, wsb, s);
    if (need <= 16) return launch_mhsddmm<VEC, 16>(a, ws, wsb, s);
    if (need <= 32) return launch_mhsddmm<VEC, 32>(a, ws, wsb, s);
    return launch_mhsddmm<VEC, 64>(a, ws, wsb, s);
}

static bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_mhsddmm(const int32_t *rowptr, const int32_t *colind, const float *grad,
                                 const float *feat, float *out, int64_t v, int64_t h, int64_t f, int64_t nnz,
                                 void *stream) {
    if (v < 0 || h < 0 || f < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (v == 0 || h == 0) return COGDL_HIP_OK;
    if (!rowptr || !grad || !feat || !out || f == 0) return COGDL_HIP_EINVAL;
    if (h * f > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    const bool al16 = aligned_to(grad, 16) && aligned_to(feat, 16);
    const bool al8 = aligned_to(grad, 8) && aligned_to(feat, 8);
    MhsddmmArgs a{rowptr, colind, grad, feat, out, v, h, f, nnz};
    // fast path: the whole [H*F] row fits one group of <= 64 lanes and F/VEC is a power of two (any F if H == 1)
    auto fits = [&](int vec) { return f % vec == 0 && h * f / vec <= kWave && (h == 1 || is_pow2(f / vec)); };
    if (fits(4) && al16) return dispatch_mhsddmm<4>(a, nullptr, 0, s);
    if (fits(2) && al8) return dispatch_mhsddmm<2>(a, nullptr, 0, s);
    if (fits(1)) return dispatch_mhsddmm<1>(a, nullptr, 0, s);
    const int64_t nrb = (v + 3) / 4;
    if (!grid_fits(make_xcd_map(nrb), 4096)) return COGDL_HIP_ERANGE;
    hipLaunchKernelGGL(mhsddmm_generic_kernel, dim3(xcd_grid(make_xcd_map(nrb))), dim3(256), 0, s, rowptr, colind, grad, feat, out,
                       v, (int)h, (int)f, make_xcd_map(nrb));
    return launch_status();
}
