// spmm.hip -- csr_spmm / mhspmm for gfx950 (MI355X).  Replaces the reference's GE-SpMM style CUDA
// kernels (cogdl/operators/spmm/spmm_kernel.cu:7-512 and multiheadSpmm.cu:6-51: warp=32,
// shared-memory colind tiles) with a wave64 design; the arithmetic follows the reference CPU
// operator (cogdl/operators/spmm/spmm_cpu.cpp:24-35) so fp32 results are bit-identical to it.
//
// Work decomposition ("row groups", engine in rowreduce.h):
//   * a group of LPR lanes owns one CSR row and VEC consecutive feature columns per lane, so one
//     gathered neighbour row x[col,:] is ONE fully coalesced vector load of LPR*VEC*sizeof(T)
//     bytes (F=128 fp32: 64 lanes x float2 = 512 B);
//   * a wave holds 64/LPR consecutive rows, a 256-thread workgroup 4 waves => its output tile is
//     one contiguous span of memory;
//   * the group reads its row's colind/val in coalesced chunks of LPR edges and broadcasts them
//     lane-to-lane (ds_bpermute; v_readlane -> SGPR address when LPR == 64), so every index is
//     fetched from memory once;
//   * the gather is issued UNROLL rows at a time (independent loads in flight); accumulation
//     stays strictly in CSR edge order per output element: acc = acc + w*x with separate fp32
//     multiply and add -- no cross-lane reduction over edges;
//   * workgroups are remapped so each XCD (private L2) walks contiguous stripes of rows.
// Skewed degree distributions: rows longer than the long-row threshold take the engine's chunk-parallel
// path (piece sums combined in a FIXED order: deterministic; such rows differ from the reference only by
// re-association, <= 1e-6 relative).
// HBM-bound: algorithmic bytes per edge = 4 (colind) + s_w + F*s, per row 4 + F*s.
#include "spmm_op.h"

namespace cogdl {

// Vector width and lanes per row.  Every lane's VEC columns must stay inside one row (k % VEC == 0), inside
// one head for mhspmm (fdim % VEC == 0), and be naturally aligned.  Measured on MI355X (arxiv-shaped uniform and
// R-MAT graphs, profiles/r01_spmm_vector_width.txt):
//   * at most 4 elements per lane (16 B fp32, 8 B bf16/fp16): 8 x bf16 lanes leave too few lanes per row
//     (bf16 F=64 R-MAT: 164 -> 128 us, F=32: 167 -> 106 us);
//   * a row group of at least 16 lanes where 4-byte lanes allow it (fp32 F=24 R-MAT: 128 -> 112 us), but not
//     narrower lanes than that needs (fp32 F=40 R-MAT: 16-byte lanes 125 us vs 8-byte 150 us);
//   * rows of >= 512 B (fp32 F >= 128) use 8-byte lanes so that a whole wave (or two) shares a row: scalar column
//     broadcast, no intra-wave length divergence (fp32 F=256: 390 -> 359 us with two column tiles per row).
// `align` = guaranteed alignment in bytes of x and out (the workspace query assumes 16: allocator memory).
//   * wide16 (round 6; the XCD-partitioned plans of 16-bit csr_spmm): 16-byte lanes whatever the row width.  On the graphs a
//     plan is built for -- hub-heavy, every gather an L2 hit -- the launch is bound by the CU's memory pipeline, which takes
//     one wave64 memory instruction per 16 cycles whether its lanes load 8 or 16 bytes (Reddit-shaped graph, bf16 F = 64 in
//     8-byte lanes: 4.3e7 vector memory instructions in 1331 us = one per 17.2 cycles and CU, profiles/r06_mem_pipeline.txt):
//     16-byte lanes gather twice the rows per instruction, and halve the per-edge lane arithmetic with it: bf16 F = 64
//     1327 -> 1107 us, F = 128 2295 -> 1836 us, F = 256 3825 -> 3265 us, F = 40 1403 -> 1342 us (fp16 alike;
//     tools/exp/wide16_ab.py, profiles/r06_wide16_ab.txt).  Tuning key 6 != 0 switches it off.
RowGeometry spmm_geometry(int64_t k, int64_t unit, int elem_bytes, int align, bool narrow_groups, bool wide16) {
    const bool wide = wide16 && elem_bytes == 2 && g_tuning[kTuneSpmmVec] == 0;
    const int maxv = wide ? 8 : std::min(4, 16 / elem_bytes);
    auto legal = [&](int v) { return v <= maxv && unit % v == 0 && align % (v * elem_bytes) == 0; };
    int vec = 1;
    while (vec * 2 <= maxv && legal(vec * 2)) vec *= 2;  // widest legal
    if (wide && vec == 8 && k > 32) {  // (rows of at most 64 bytes -- four 16-byte lanes, 16 rows per wave -- measured slower: F = 32 886 -> 1202 us)
        RowGeometry g;
        g.vec = 8;
        g.lpr = 4;
        while (g.lpr < kWave && (int64_t)g.lpr * 8 < k) g.lpr <<= 1;
        g.tiles = (k + (int64_t)g.lpr * 8 - 1) / ((int64_t)g.lpr * 8);
        return g;
    }
    auto lanes = [&](int v) {  // LPR for vector width v
        int l = 4;
        while (l < kWave && (int64_t)l * v < k) l <<= 1;
        return l;
    };
    while (vec > 1 && (vec / 2) * elem_bytes >= 4 && lanes(vec) < 16) vec /= 2;  // >= 16 lanes per row
    // (Round 5: is that still right where every gather is an HBM access?  On one eighth of the papers100M-shaped symmetrised
    //  graph -- X = 7.1 GB -- 16-byte lanes / two rows per wave looked 6 % faster, 36.9 vs 39.1 ms; at FULL size, A/B in one
    //  process on one box (tuning key 6 = -2 against a build with the rule "operands beyond 512 MiB keep 16-byte lanes"):
    //  symmetrised 327-330 vs 333-338 ms, but the DIRECTED graph -- 14.5 edges per row instead of 29 -- 201-202 vs 138-139 ms
    //  with the wider lanes: two rows of different length per wave cost more than the bytes in flight buy.  The rule stays
    //  size-independent; profiles/r05_papers_variants.txt.)
    while (vec > 1 && (vec / 2) * elem_bytes >= 8 && k / vec >= 32) vec /= 2;  // wide rows: 8-byte lanes
    if (g_tuning[kTuneSpmmVec] > 0)  // experiments: cap the vector width
        while (vec > g_tuning[kTuneSpmmVec]) vec /= 2;
    if (g_tuning[kTuneSpmmVec] < 0 && legal(-g_tuning[kTuneSpmmVec])) vec = -g_tuning[kTuneSpmmVec];  // ... or force it
    const int64_t need = (k + vec - 1) / vec;
    int lpr = 4;
    while (lpr < kWave && lpr < need) lpr <<= 1;
    // Rows whose lane count is 5/8 or 3/4 of a power of two (F = 40 fp32: 10 sixteen-byte lanes) would leave 25-37 % of
    // every row group idle; a group of exactly `need` lanes packs floor(64 / need) rows into a wave instead (60 of 64
    // lanes busy, 6 / 5 / 3 rows per wave instead of 4 / 4 / 2).  Measured on MI355X (arxiv-shaped graphs, fp32): R-MAT
    // F=40 136 -> 120 us, F=48 126 -> 113 us, F=80 168 -> 138 us (more rows per wave = better balance on skewed
    // degrees); uniform degrees: +-0 .. -4 % (those are bound by the per-row latency chain, not by lanes -- F=20 takes
    // 68 us where F=40 takes 99 us).  Groups of 5 / 6 lanes (F=20 / 24) gained nothing.  csr_spmm only; tuning key
    // 6 == -99 switches it off for A-B runs.
    // Round 6: groups of 10 and 12 lanes no longer pay -- the power-of-two groups got immediate-pattern broadcasts and fully
    // unrolled chunks (rowreduce.h: group_bcast), the odd ones keep ds_bpermute: fp32 F = 40 R-MAT 105.9 (10 lanes) vs 102.0 us
    // (16), uniform 88.3 vs 85.8, F = 48 102.6 vs 99.1, bf16 F = 40 R-MAT 100.1 vs 91.6 -- but 20 lanes still beat 32 (F = 80
    // R-MAT 129 vs 137 us, bf16 119 vs 132) (tools/exp/f40_ab.py, profiles/r06_narrow_groups.txt).
    if (narrow_groups && g_tuning[kTuneSpmmVec] != -99 && need == 20) lpr = (int)need;
    RowGeometry g;
    g.vec = vec;
    g.lpr = lpr;
    g.tiles = (k + (int64_t)lpr * vec - 1) / ((int64_t)lpr * vec);
    return g;
}

template <typename T>
static int spmm_typed(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x, void *out,
                      int64_t m, int64_t k, int64_t nnz, int acc_mode, void *ws, size_t wsb, hipStream_t s,
                      const int32_t *row_order = nullptr) {
    if (!aligned_to(x, sizeof(T)) || !aligned_to(out, sizeof(T))) return COGDL_HIP_EALIGN;
    SpmmArgs<T> a{rowptr, colind, (const T *)val, nullptr, (const T *)x, (T *)out, m, nnz, (int)k, (int)k, acc_mode, nullptr, {}};
    a.row_order = row_order;
    return val ? spmm_auto<T, 1>(a, ws, wsb, s) : spmm_auto<T, 0>(a, ws, wsb, s);
}

static int check_args(const int32_t *rowptr, const void *x, const void *out, int64_t m, int64_t k, int64_t nnz) {
    if (m < 0 || k < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (m == 0 || k == 0) return COGDL_HIP_OK;
    if (!rowptr || !x || !out) return COGDL_HIP_EINVAL;
    // (the engine's `int` edge arithmetic looks up to one chunk past a row's end: base + LPR + lane, start + thresh - 1)
    if (k > 0x7fffffff || nnz > COGDL_HIP_SEGMENT_MAX_EDGES) return COGDL_HIP_ERANGE;
    return COGDL_HIP_OK;
}

static int csr_spmm_entry(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x, void *out,
                          int64_t m, int64_t k, int64_t nnz, int dtype, int acc_mode, void *ws, size_t wsb,
                          void *stream, const int32_t *row_order = nullptr) {
    int rc = check_args(rowptr, x, out, m, k, nnz);
    if (rc != COGDL_HIP_OK || m == 0 || k == 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32: return spmm_typed<float>(rowptr, colind, val, x, out, m, k, nnz, acc_mode, ws, wsb, s, row_order);
        case COGDL_HIP_F16: return spmm_typed<__half>(rowptr, colind, val, x, out, m, k, nnz, acc_mode, ws, wsb, s, row_order);
        case COGDL_HIP_BF16:
            return spmm_typed<__hip_bfloat16>(rowptr, colind, val, x, out, m, k, nnz, acc_mode, ws, wsb, s, row_order);
        default: return COGDL_HIP_EDTYPE;
    }
}

template <typename T>
static int mhspmm_typed(const int32_t *rowptr, const int32_t *colind, const float *att, const int32_t *eid,
                        const void *feat, void *out, int64_t v, int64_t h, int64_t f, int64_t nnz, void *ws, size_t wsb,
                        hipStream_t s) {
    if (!aligned_to(feat, sizeof(T)) || !aligned_to(out, sizeof(T))) return COGDL_HIP_EALIGN;
    SpmmArgs<T> a{rowptr, colind, nullptr, att, (const T *)feat, (T *)out, v, nnz, (int)(h * f), (int)f, 0, eid, {}};
    return spmm_auto<T, 2>(a, ws, wsb, s);
}

static int elem_bytes_of(int dtype) { return dtype == COGDL_HIP_F32 ? 4 : 2; }

}  // namespace cogdl

using namespace cogdl;

extern "C" size_t cogdl_hip_csr_spmm_workspace_bytes(int64_t nnz, int64_t k, int dtype) {
    if (nnz <= 0 || k <= 0) return 0;
    const RowGeometry g = spmm_geometry(k, k, elem_bytes_of(dtype), 16, true);
    // (the epilogue entry keeps power-of-two groups, which never need less: sized for the larger of the two)
    const RowGeometry g2 = spmm_geometry(k, k, elem_bytes_of(dtype), 16, false);
    return std::max(rowreduce_workspace_bytes(nnz, g.tiles * g.vec * g.lpr), rowreduce_workspace_bytes(nnz, g2.tiles * g2.vec * g2.lpr));
}

extern "C" size_t cogdl_hip_mhspmm_workspace_bytes(int64_t nnz, int64_t h, int64_t f, int dtype) {
    if (nnz <= 0 || h <= 0 || f <= 0) return 0;
    const RowGeometry g = spmm_geometry(h * f, f, elem_bytes_of(dtype), 16);
    return rowreduce_workspace_bytes(nnz, g.tiles * g.vec * g.lpr);
}

extern "C" int cogdl_hip_long_row_threshold(int64_t nnz) { return pick_long_thresh(nnz); }

extern "C" int cogdl_hip_exact_row_edges(int64_t nnz) {
    return pick_long_thresh(nnz);
}

extern "C" int cogdl_hip_csr_spmm(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                  void *out, int64_t m, int64_t k, int64_t nnz, int dtype, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    return csr_spmm_entry(rowptr, colind, val, x, out, m, k, nnz, dtype, 0, workspace, workspace_bytes, stream);
}

extern "C" int cogdl_hip_csr_spmm_acc(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                      void *out, int64_t m, int64_t k, int64_t nnz, int dtype, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    return csr_spmm_entry(rowptr, colind, val, x, out, m, k, nnz, dtype, 1, workspace, workspace_bytes, stream);
}

// ---- XCD-partitioned plan (rowreduce.h: virtual rows) ------------------------------------------------------------------
extern "C" size_t cogdl_hip_csr_spmm_xcd_workspace_bytes(int64_t n_parts, int64_t k, int dtype) {
    if (k <= 0) return 256;
    const RowGeometry g = spmm_geometry(k, k, elem_bytes_of(dtype), 16, false), gw = spmm_geometry(k, k, elem_bytes_of(dtype), 16, false, true);
    return std::max(vrows_workspace_bytes(n_parts, g.tiles * g.vec * g.lpr), vrows_workspace_bytes(n_parts, gw.tiles * gw.vec * gw.lpr));
}

template <typename T>
static int spmm_xcd_typed(const cogdl_hip_vrows *plan, const void *val, const void *x, void *out, int64_t m, int64_t k, int acc,
                          void *ws, size_t wsb, hipStream_t s) {
    if (!aligned_to(x, 16) || !aligned_to(out, 16)) return COGDL_HIP_EALIGN;  // (the workspace query assumes 16-byte operands)
    SpmmArgs<T> a{nullptr, nullptr, (const T *)val, nullptr, (const T *)x, (T *)out, m, plan->nnz, (int)k, (int)k, acc, nullptr, {}, plan};
    return val ? spmm_auto<T, 1>(a, ws, wsb, s) : spmm_auto<T, 0>(a, ws, wsb, s);
}

extern "C" int cogdl_hip_csr_spmm_xcd(const cogdl_hip_vrows *plan, const void *val_plan, const void *x, void *out, int64_t m,
                                      int64_t k, int dtype, int acc, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = vrows_valid(plan);
    if (rc != COGDL_HIP_OK) return rc;
    const int32_t dummy = 0;
    rc = check_args(&dummy, x, out, m, k, plan->nnz);
    if (rc != COGDL_HIP_OK || m == 0 || k == 0) return rc;
    if (k * 4 >= (1 << 24)) return COGDL_HIP_EUNSUPPORTED;  // (24 x 24 -> 32-bit table offsets, common.h: gather_row)
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32: return spmm_xcd_typed<float>(plan, val_plan, x, out, m, k, acc, workspace, workspace_bytes, s);
        case COGDL_HIP_F16: return spmm_xcd_typed<__half>(plan, val_plan, x, out, m, k, acc, workspace, workspace_bytes, s);
        case COGDL_HIP_BF16:
            return spmm_xcd_typed<__hip_bfloat16>(plan, val_plan, x, out, m, k, acc, workspace, workspace_bytes, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

// ---- 64-bit CSR: one launch per row segment (bigcsr.hip) ---------------------------------------------------------------
extern "C" size_t cogdl_hip_csr_spmm_i64_workspace_bytes(const cogdl_hip_segments *seg, int64_t k, int dtype) {
    if (segments_valid(seg) != COGDL_HIP_OK) return 0;
    size_t need = 0;
    for (int s = 0; s < seg->n; ++s)
        need = std::max(need, cogdl_hip_csr_spmm_workspace_bytes(seg->edge[s + 1] - seg->edge[s], k, dtype));
    return need;
}

extern "C" int cogdl_hip_csr_spmm_i64(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                      const void *val, const void *x, void *out, int64_t k, int dtype, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    return cogdl_hip_csr_spmm_i64_ordered(rowptr32, seg, colind, val, x, out, k, dtype, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int cogdl_hip_csr_spmm_ordered(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                          void *out, int64_t m, int64_t k, int64_t nnz, int dtype, int acc,
                                          const int32_t *row_order, void *workspace, size_t workspace_bytes, void *stream) {
    return csr_spmm_entry(rowptr, colind, val, x, out, m, k, nnz, dtype, acc ? 1 : 0, workspace, workspace_bytes, stream, row_order);
}

extern "C" int cogdl_hip_csr_spmm_i64_ordered(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                              const void *val, const void *x, void *out, int64_t k, int dtype,
                                              const int32_t *row_order, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = segments_valid(seg);
    if (rc != COGDL_HIP_OK) return rc;
    if (k < 0) return COGDL_HIP_EINVAL;
    if (dtype != COGDL_HIP_F32 && dtype != COGDL_HIP_F16 && dtype != COGDL_HIP_BF16) return COGDL_HIP_EDTYPE;
    const size_t es = (size_t)elem_bytes_of(dtype);
    for (int s = 0; s < seg->n; ++s) {
        const int64_t r0 = seg->row[s], e0 = seg->edge[s];
        rc = csr_spmm_entry(rowptr32 + r0 + s, colind ? colind + e0 : nullptr, val ? (const char *)val + (size_t)e0 * es : nullptr, x,
                            out ? (char *)out + (size_t)r0 * (size_t)k * es : nullptr, seg->row[s + 1] - r0, k,
                            seg->edge[s + 1] - e0, dtype, 0, workspace, workspace_bytes, stream, row_order ? row_order + r0 : nullptr);
        if (rc != COGDL_HIP_OK) return rc;
    }
    return COGDL_HIP_OK;
}

extern "C" int cogdl_hip_mhspmm(const int32_t *rowptr, const int32_t *colind, const float *att, const void *feat,
                                void *out, int64_t v, int64_t h, int64_t f, int64_t nnz, int dtype, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return cogdl_hip_mhspmm_eid(rowptr, colind, att, nullptr, feat, out, v, h, f, nnz, dtype, workspace, workspace_bytes,
                                stream);
}

extern "C" int cogdl_hip_mhspmm_eid(const int32_t *rowptr, const int32_t *colind, const float *att, const int32_t *eid,
                                    const void *feat, void *out, int64_t v, int64_t h, int64_t f, int64_t nnz, int dtype,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    if (v < 0 || h < 0 || f < 0 || nnz < 0) return COGDL_HIP_EINVAL;
    if (v == 0 || h == 0 || f == 0) return COGDL_HIP_OK;
    if (!rowptr || !att || !feat || !out) return COGDL_HIP_EINVAL;
    if (h * f > 0x7fffffff || nnz > 0x7fffffff) return COGDL_HIP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COGDL_HIP_F32:
            return mhspmm_typed<float>(rowptr, colind, att, eid, feat, out, v, h, f, nnz, workspace, workspace_bytes, s);
        case COGDL_HIP_F16:
            return mhspmm_typed<__half>(rowptr, colind, att, eid, feat, out, v, h, f, nnz, workspace, workspace_bytes, s);
        case COGDL_HIP_BF16:
            return mhspmm_typed<__hip_bfloat16>(rowptr, colind, att, eid, feat, out, v, h, f, nnz, workspace,
                                                workspace_bytes, s);
        default: return COGDL_HIP_EDTYPE;
    }
}

// Benchmark hook: explicit (VEC, LPR, UNROLL, arithmetic) choices for fp32 weighted, k % 4 == 0.
extern "C" int cogdl_hip_csr_spmm_variant(const int32_t *rowptr, const int32_t *colind, const void *val,
                                          const void *x, void *out, int64_t m, int64_t k, int64_t nnz, int dtype,
                                          int variant, void *workspace, size_t workspace_bytes, void *stream) {
    if (variant < 0)
        return cogdl_hip_csr_spmm(rowptr, colind, val, x, out, m, k, nnz, dtype, workspace, workspace_bytes, stream);
    int rc = check_args(rowptr, x, out, m, k, nnz);
    if (rc != COGDL_HIP_OK || m == 0 || k == 0) return rc;
    if (dtype != COGDL_HIP_F32 || !val) return COGDL_HIP_EDTYPE;
    if (k % 4 != 0 || !aligned_to(x, 16) || !aligned_to(out, 16)) return COGDL_HIP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    SpmmArgs<float> a{rowptr, colind, (const float *)val, nullptr, (const float *)x, (float *)out, m, nnz, (int)k, (int)k, 0, nullptr, {}};
    // The variants may need more workspace than the automatic geometry: only pass it on when it is big enough.
#define V(id, VEC, LPR, UNR, EX)                                                                                   \
    case id: {                                                                                                     \
        const int64_t tiles = (k + (int64_t)LPR * VEC - 1) / ((int64_t)LPR * VEC);                                  \
        const bool fits = workspace && workspace_bytes >= rowreduce_workspace_bytes(nnz, tiles * VEC * LPR);        \
        return launch_spmm<float, VEC, LPR, UNR, 1, EX>(a, fits ? workspace : nullptr, fits ? workspace_bytes : 0, s); \
    }
    switch (variant) {
        V(0, 4, 32, 8, true)
        V(1, 4, 32, 4, true)
        V(2, 4, 16, 8, true)
        V(3, 2, 64, 8, true)
        V(4, 2, 64, 4, true)
        V(5, 2, 64, 8, false)
        V(6, 2, 32, 8, true)
        V(7, 4, 64, 8, true)
        V(8, 1, 64, 8, true)
        V(9, 1, 64, 16, true)
        V(10, 2, 64, 16, true)
        V(11, 2, 32, 16, true)
        V(12, 4, 16, 16, true)
        V(13, 4, 10, 16, true)
        V(14, 4, 10, 8, true)
        V(15, 2, 32, 12, true)
        default: return COGDL_HIP_EINVAL;
    }
#undef V
}
