/*
 * cogdl_hip.h -- C ABI of libcogdl_hip.so: the MI355X (gfx950) sparse message-passing
 * backend that sits underneath CogDL's `cogdl.operators` API.
 *
 * Every entry point replaces one function of the reference's per-op pybind11/CUDA
 * extensions (cited as file:line under /root/reference/).  The reference binds those with
 * torch::Tensor arguments; this ABI is torch-free: raw device pointers, sizes, a dtype
 * enum and the hipStream_t (as void*) to launch on.  Ownership: all buffers are
 * caller-allocated and borrowed for the duration of the enqueued work; nothing is
 * allocated, freed or synchronised inside (so calls are hipGraph-capturable), except the
 * *_workspace_bytes queries which are pure host functions.
 *
 * Error convention: 0 = COGDL_HIP_OK, otherwise a COGDL_HIP_E* code (the reference
 * assert()s / exit()s instead: operators/spmm/computeUtil.h:13-27).  cogdl_hip_strerror
 * maps a code to text.  Launch errors are reported from hipGetLastError() right after the
 * launch; asynchronous faults surface at the caller's next synchronisation as usual.
 *
 * Conventions: CSR index arrays are int32 (what the callers pass: rowptr.int(),
 * colind.int(), cogdl/utils/spmm_utils.py:106); all row*width offset arithmetic inside
 * the kernels is 64-bit.  Dense matrices are row-major contiguous.
 */
#ifndef COGDL_HIP_H
#define COGDL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COGDL_HIP_ABI_VERSION 9

/* Exported with default visibility (the library is built -fvisibility=hidden). */
#if defined(COGDL_HIP_BUILD)
#define COGDL_API __attribute__((visibility("default")))
#else
#define COGDL_API
#endif

enum cogdl_hip_status {
    COGDL_HIP_OK = 0,
    COGDL_HIP_EINVAL = 1,      /* bad argument (null pointer, negative size, ...) */
    COGDL_HIP_EDTYPE = 2,      /* unsupported dtype for this entry point */
    COGDL_HIP_EALIGN = 3,      /* pointer not aligned to the element size */
    COGDL_HIP_ELAUNCH = 4,     /* hipLaunchKernel / runtime error (see cogdl_hip_last_hip_error) */
    COGDL_HIP_EWORKSPACE = 5,  /* workspace too small */
    COGDL_HIP_ERANGE = 6,      /* size exceeds what int32 CSR indices can address */
    COGDL_HIP_EUNSUPPORTED = 7 /* valid call, but a shape this entry point declines (see its comment): the caller
                                * is expected to take its documented alternative -- never returned for a bad call */
};

enum cogdl_hip_dtype {
    COGDL_HIP_F32 = 0,
    COGDL_HIP_F16 = 1,
    COGDL_HIP_BF16 = 2
};

COGDL_API int cogdl_hip_abi_version(void);
COGDL_API const char *cogdl_hip_strerror(int status);
/* hipError_t of the most recent COGDL_HIP_ELAUNCH on this thread (0 if none). */
COGDL_API int cogdl_hip_last_hip_error(void);
/* Run-time tuning knobs for experiments and tests (defaults are the measured optima; the negative results behind the
 * retired keys are tabled in DESIGN.md section 8):
 *    0  XCD stripe of the row-block -> workgroup map (0 = hardware round-robin, default 32)
 *    1  long-row threshold override (0 = automatic)          2  1 = natural row -> lane-group assignment inside a workgroup
 *    3  cap on the number of long-row workgroups (1024)      4  cap on the fused-GAT vector width (0 = widest)
 *    5  fused-GAT forward kernel (0 = automatic, 1 = edge-wise online softmax, 2 = chunk-wise softmax where it applies,
 *       3 = plan kernels without the per-chunk scalars through LDS: attn_col / stats gathered per edge as before)
 *    6  csr_spmm / mhspmm vector width cap (negative: force; -99: power-of-two lane groups only; any value but 0 also keeps the
 *       16-bit plan kernels at 8-byte lanes)
 *    7  edge_softmax lanes (bit 0: 4-byte lanes in the row kernels, bit 1: in the hub-row path, bit 2: row kernels only)
 *    8  polls before the flat edge_softmax kernel's cross-tile wait recomputes the row statistics (0 = 4096; < 0: at once)
 *    9  flat edge_softmax experiments (bit 0: no cross-tile exchange -- WRONG results; bit 2: half-size 16-bit tiles; bit 3: no
 *       kept exp values; bit 4: one 16-bit element per LDS access; bit 6: full-size tiles for under-filled launches; bit 7:
 *       phase stamps, tools/es_phase_probe.py; bit 8: round 5's two-kernel forward, measured slower)
 *   10  csr2csc (0 = by size: one single-workgroup launch up to 16 k slots, the radix sort above; 2 = the radix sort at
 *       every size; 3 = with packed intermediate records at every size)
 *   11  sampler relabelling (1 = the sort-based form)       13  timing: 1 = row blocks exit, 2 = long-row workgroups exit (WRONG results)
 *   15  64-bit CSR: edges per row segment (0 = 2^29)
 *   12, 14, 16  retired in round 6 (wave-scope split of medium rows, row tiles, per-workgroup row queue: all measured <= +-5 %). */
COGDL_API int cogdl_hip_set_tuning(int key, int value);
/* Measurement hook (bench.py `roofline.measured_read_GBs`, SURVEY.md section 8d: the box's own roof beside the spec
 * peak): one read-only pass over `bytes` of device memory in 16-byte vectors; sink: COGDL_HIP_PROBE_BLOCKS * 16 bytes. */
#define COGDL_HIP_PROBE_BLOCKS 4096
COGDL_API int cogdl_hip_probe_read_stream(const void *p, size_t bytes, void *sink, void *stream);
/* ... and one 16-byte-vector copy of `bytes` from src to dst (`roofline.measured_copy_GBs` counts read + written bytes). */
COGDL_API int cogdl_hip_probe_copy_stream(const void *src, void *dst, size_t bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * csr_spmm:  out[i,:] = sum_{e in row i, CSR order} val[e] * x[colind[e],:]
 * Replaces spmm.csr_spmm (operators/spmm/spmm.cpp:22-45 -> spmm_cuda, spmm_kernel.cu:534-594)
 * and, with val == NULL, spmm.csr_spmm_no_edge_value (spmm.cpp:47-65, spmm_kernel.cu:155-190).
 * CPU twin / oracle: csr_spmm_cpu (operators/spmm/spmm_cpu.cpp:6-58).
 * f32: each output element is accumulated sequentially in CSR edge order with a separate
 * fp32 multiply and add -- bit-identical to the reference CPU path as CogDL builds it.
 * f16/bf16: val has the dtype of x; products and sums are fp32, rounded once on store.
 * x: [n_src, k], out: [m, k]; rowptr: [m+1]; colind/val: [nnz], nnz == rowptr[m].
 * workspace (optional, device, 256-B aligned, >= cogdl_hip_csr_spmm_workspace_bytes(nnz, k, dtype)):
 * enables the chunk-parallel treatment of rows longer than cogdl_hip_long_row_threshold(nnz)
 * edges (power-law graphs); such rows are summed as fixed-order partial sums (deterministic,
 * re-associated).  With workspace == NULL every row is summed strictly sequentially.
 * The same convention holds for every row-wise operator below: `workspace` is optional scratch
 * for the long-row path (csrc/rowreduce.h), sized by the operator's *_workspace_bytes query
 * (pure host functions; they assume 16-byte aligned operands, as every allocator provides).
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_csr_spmm_workspace_bytes(int64_t nnz, int64_t k, int dtype);
COGDL_API int cogdl_hip_long_row_threshold(int64_t nnz);
/* Rows of at most this many edges are ALWAYS reduced sequentially in CSR order (fp32: bit-identical to the reference's
 * csr_spmm_cpu loop, spmm_cpu.cpp:24-35): cogdl_hip_long_row_threshold(nnz).  Longer rows are cut into contiguous pieces
 * whose partial sums are merged in order by whole workgroups: deterministic, re-associated at the piece borders only
 * (<= 1e-6 relative). */
COGDL_API int cogdl_hip_exact_row_edges(int64_t nnz);
COGDL_API int cogdl_hip_csr_spmm(const int32_t *rowptr, const int32_t *colind, const void *val,
                       const void *x, void *out, int64_t m, int64_t k, int64_t nnz, int dtype,
                       void *workspace, size_t workspace_bytes, void *stream);

/* out += A x: same kernels, the per-element accumulator starts from the existing out[i,:] (then the
 * row's edges are added in CSR order).  Used by the vertex-sharded SpMM, where the remote-column
 * block is applied after the local-column block (no reference counterpart). */
COGDL_API int cogdl_hip_csr_spmm_acc(const int32_t *rowptr, const int32_t *colind, const void *val,
                           const void *x, void *out, int64_t m, int64_t k, int64_t nnz, int dtype,
                           void *workspace, size_t workspace_bytes, void *stream);

/* Tuning hook (benchmarks only): force a kernel variant for csr_spmm; <0 = automatic. */
COGDL_API int cogdl_hip_csr_spmm_variant(const int32_t *rowptr, const int32_t *colind, const void *val,
                               const void *x, void *out, int64_t m, int64_t k, int64_t nnz, int dtype,
                               int variant, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * XCD-partitioned columns (ABI v8; no reference counterpart -- the reference's GE-SpMM kernels have no notion of the chip's
 * eight private L2s).  For gathered tables that fit the eight 4 MiB L2s of the MI355X together but not one of them
 * (BASELINE configs[2]: 233 k rows of 128-256 bytes, rows of hundreds of edges) a PLAN gives every column an owner XCD
 * (hash of its id), cuts every long row into the sub-rows of its edges by owner and lays these virtual rows out so that a
 * workgroup on XCD x only gathers columns XCD x owns.  The plan is built once per structure by the caller
 * (cogdl_amd/xcdplan.py); this struct is its device view, all arrays int32 device memory:
 *   vrowptr [n_slots + 1]  edge offsets of the virtual rows ("slots") in the plan's edge order
 *   vcol    [nnz]          column ids in that order
 *   vdesc   [n_slots][2]   {row, record}: row < 0 = padding slot; record < 0 = the row's only part (finished by the main
 *                          kernel), otherwise the index of this part's state record
 *   eid     [nnz]          what a per-edge operand is indexed by for the edge at a plan position (the CSR position of
 *                          the edge; for a CSC view: the caller's own edge id) -- NULL if the operator has none
 *   mrow [n_multi], mptr [n_multi + 1]   rows with several parts and their records [mptr[i], mptr[i+1]) in merge order
 *   big  [n_big]           the indices i (into mrow) of ALL rows with more than 32 parts: merged by a whole workgroup each
 * n_slots is a multiple of 512 (units of 64 slots, round-robin over the 8 XCDs).  Results are deterministic; a row cut into
 * parts is re-associated (fp32 <= 1e-6 relative) -- which is why csr_spmm takes this path only when asked to. */
typedef struct cogdl_hip_vrows {
    const int32_t *vrowptr, *vcol, *vdesc, *eid, *mrow, *mptr, *big;
    int64_t n_slots, n_multi, n_parts, n_big, nnz;
} cogdl_hip_vrows;
/* out = A x (acc != 0: out += A x) over a plan; val in PLAN order (val_plan[j] = val[eid[j]]) or NULL.
 * workspace >= cogdl_hip_csr_spmm_xcd_workspace_bytes(n_parts, k, dtype). */
COGDL_API size_t cogdl_hip_csr_spmm_xcd_workspace_bytes(int64_t n_parts, int64_t k, int dtype);
COGDL_API int cogdl_hip_csr_spmm_xcd(const cogdl_hip_vrows *plan, const void *val_plan, const void *x, void *out,
                                     int64_t m, int64_t k, int dtype, int acc, void *workspace, size_t workspace_bytes,
                                     void *stream);

/* ---------------------------------------------------------------------------------------
 * 64-bit CSR ("big CSR", ABI v7): graphs of 2^31 edges and more -- ogbn-papers100M as CogDL feeds it to GCN
 * (symmetrised + coalesced, cogdl/datasets/ogb.py:50-55: 3.2e9 edges).  The reference cannot represent them: the
 * dispatcher casts row pointers to int32 (utils/spmm_utils.py:106), csr_spmm_cpu walks `int` edge offsets and
 * `int ik = i * k` (operators/spmm/spmm_cpu.cpp:24-33), the CUDA kernels take `int nnz` (spmm_kernel.cu:534-594).
 * Row pointers are int64, column ids stay int32.  The rows are cut into SEGMENTS of about 2^29 edges; each segment is
 * one launch of the 32-bit kernels on a rebased int32 copy of its row pointers (csrc/bigcsr.hip), so results are
 * those of cogdl_hip_csr_spmm row for row (same summation order, same long-row rule with the SEGMENT's edge count).
 *   1. cogdl_hip_csr_segments(rowptr64, m, nnz, 0, &seg, scratch, stream)      -- plan time; SYNCHRONISES `stream`
 *   2. cogdl_hip_csr_rebase_rowptr(rowptr64, &seg, rowptr32, stream)           -- rowptr32: [m + seg.n] int32
 *   3. cogdl_hip_csr_spmm_i64 / cogdl_hip_csr_sddmm_i64 / cogdl_hip_csr2csc_i64 with (rowptr32, &seg)
 * Segment s covers rows [row[s], row[s+1]) = edges [edge[s], edge[s+1]); its rebased row pointers are
 * rowptr32[row[s] + s .. row[s+1] + s] (rows + 1 entries, starting at 0).
 * ------------------------------------------------------------------------------------- */
#define COGDL_HIP_MAX_SEGMENTS 64
#define COGDL_HIP_SEGMENT_MAX_EDGES 0x7ff00000ll /* what one 32-bit launch takes (2^31 - 2^20) */
typedef struct cogdl_hip_segments {
    int32_t n;
    int64_t row[COGDL_HIP_MAX_SEGMENTS + 1];
    int64_t edge[COGDL_HIP_MAX_SEGMENTS + 1];
} cogdl_hip_segments;
/* max_edges: target edges per segment (0 = default: tuning key 15, else 2^29; capped at 2^30).  scratch: device,
 * >= 2 * (COGDL_HIP_MAX_SEGMENTS + 1) * 8 bytes.  rowptr[m] must equal nnz (COGDL_HIP_EINVAL otherwise); a single row
 * of more than COGDL_HIP_SEGMENT_MAX_EDGES edges is COGDL_HIP_ERANGE. */
COGDL_API int cogdl_hip_csr_segments(const int64_t *rowptr, int64_t m, int64_t nnz, int64_t max_edges,
                                     cogdl_hip_segments *out, void *scratch, void *stream);
COGDL_API int cogdl_hip_csr_rebase_rowptr(const int64_t *rowptr, const cogdl_hip_segments *seg, int32_t *rowptr32,
                                          void *stream);
/* csr_spmm over all segments (x: [n_src, k], out: [row[n], k], colind / val: [edge[n]]).  Workspace as cogdl_hip_csr_spmm,
 * sized for the largest segment by the query below; NULL = every row sequential. */
COGDL_API size_t cogdl_hip_csr_spmm_i64_workspace_bytes(const cogdl_hip_segments *seg, int64_t k, int dtype);
COGDL_API int cogdl_hip_csr_spmm_i64(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                     const void *val, const void *x, void *out, int64_t k, int dtype, void *workspace,
                                     size_t workspace_bytes, void *stream);
/* The same with a plan-time ROW SCHEDULE (round 6; no reference counterpart: spmm_kernel.cu walks rows in id order).
 * row_order: [m] int32, a permutation of 0 .. m-1 (for the segmented form: per segment a permutation of the segment's LOCAL
 * row ids 0 .. rows-1, stored at row_order[row[s] ..]); row block b of the launch walks rows row_order[b * G ..] instead of
 * b * G .. -- e.g. rows by decreasing degree inside windows, so that the lane groups of a wave carry rows of one length.
 * Every row is still reduced by one lane group in CSR order and written to its own place: results are those of the
 * unordered call bit for bit.  NULL = id order.  The entries are not validated (a plan-time artefact of the caller, like
 * colind: an id outside [0, m) reads out of bounds). */
COGDL_API int cogdl_hip_csr_spmm_ordered(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                         void *out, int64_t m, int64_t k, int64_t nnz, int dtype, int acc /* != 0: out += A x */,
                                         const int32_t *row_order, void *workspace, size_t workspace_bytes, void *stream);
COGDL_API int cogdl_hip_csr_spmm_i64_ordered(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                             const void *val, const void *x, void *out, int64_t k, int dtype,
                                             const int32_t *row_order, void *workspace, size_t workspace_bytes, void *stream);
/* csr_sddmm over all segments (d1: [row[n], k], d2: [n_src, k], out: [edge[n]] fp32). */
COGDL_API int cogdl_hip_csr_sddmm_i64(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                      const float *d1, const float *d2, float *out, int64_t k, void *stream);
/* Stable transpose -> colptr [n_cols + 1] int64, rowind [nnz] int32, and either or both of
 *   perm [nnz] int64 (CSR position of CSC entry j; may be NULL: 8 bytes per edge are 26 GB at 3.2e9 edges) and
 *   val_t [nnz] = val[perm] (val_bytes in {2, 4}; val and val_t both NULL or both given),
 * ordered inside a column by ascending CSR position like cogdl_hip_csr2csc.  Every segment is transposed by
 * cogdl_hip_csr2csc (twice: column counts, then entries) and merged. */
COGDL_API size_t cogdl_hip_csr2csc_i64_workspace_bytes(const cogdl_hip_segments *seg, int64_t n_cols);
COGDL_API int cogdl_hip_csr2csc_i64(const int32_t *rowptr32, const cogdl_hip_segments *seg, const int32_t *colind,
                                    int64_t n_cols, int64_t *colptr, int32_t *rowind, int64_t *perm, const void *val,
                                    void *val_t, int val_bytes, void *workspace, size_t workspace_bytes, void *stream);
/* out[i, 0:h] = src[perm[i], 0:h] with 64-bit positions (elem_bytes in {2,4}). */
COGDL_API int cogdl_hip_gather_rows_i64(const int64_t *perm, const void *src, void *out, int64_t n, int64_t h,
                                        int elem_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * csr_spmm with a fused normalisation / bias / activation epilogue (fp32):
 *   out[i,:] = act( dst_scale[i] * sum_e val[e] * (src_scale[colind[e]] * x[colind[e],:]) + bias[:] )
 * i.e. the GPU branch of cogdl.utils.spmm_utils.spmm for a graph that carries its normalisation as out_norm / in_norm
 * vectors (utils/spmm_utils.py:98-109: `x = out_norm * x`, kernel, `x = in_norm * x` -- CSR-only graphs such as sampled
 * blocks after row_norm(), cogdl/data/data.py:240-258) plus the bias / ReLU a layer applies behind it
 * (layers/gcn_layer.py:51-64), in ONE pass: no scaled copy of x, no second and third sweep over the output.
 * src_scale [n_src], dst_scale [m], bias [k] are fp32 device vectors, each may be NULL; act: 0 none, 1 ReLU; val may be
 * NULL (unweighted).  Every product is a separately rounded fp32 multiply in the reference's order: bit-identical to
 * the unfused composition for rows up to the long-row threshold.  Workspace: as cogdl_hip_csr_spmm (same query).
 * Returns COGDL_HIP_EUNSUPPORTED for dtype != COGDL_HIP_F32 (callers compose the unfused operators).
 * ------------------------------------------------------------------------------------- */
COGDL_API int cogdl_hip_csr_spmm_epilogue(const int32_t *rowptr, const int32_t *colind, const void *val, const void *x,
                                void *out, int64_t m, int64_t k, int64_t nnz, int dtype, const float *src_scale,
                                const float *dst_scale, const float *bias, int act, void *workspace,
                                size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * csr2csc: stable transpose of the CSR structure.
 * Replaces spmm.csr2csc (operators/spmm/spmm.cpp:72-90 -> cusparseCsr2cscEx2 ALG1,
 * spmm_kernel.cu:514-532,596-614) and mhtranspose.csr2csc (mhTranspose.cu:51-111).
 * Outputs colptr [n_cols+1], rowind [nnz] and perm [nnz] (perm[j] = CSR position of CSC
 * entry j; within a column entries keep ascending CSR position, i.e. ascending row).
 * Values are moved with cogdl_hip_gather_rows(perm, val, ...) -- one gather serves both
 * the scalar weights of csr2csc and the [E,H] attention of mhtranspose.
 * workspace: device scratch of at least cogdl_hip_csr2csc_workspace_bytes(...) bytes.
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_csr2csc_workspace_bytes(int64_t m, int64_t n_cols, int64_t nnz);
COGDL_API int cogdl_hip_csr2csc(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols,
                      int64_t nnz, int32_t *colptr, int32_t *rowind, int32_t *perm,
                      void *workspace, size_t workspace_bytes, void *stream);
/* The same for a fixed-capacity block (cogdl_hip_sample_adj_padded): colind holds `nnz` slots, the first rowptr[m]
 * (read on the device) are edges, the rest is ignored -- colptr[n_cols] = rowptr[m]; rowind / perm hold that many
 * meaningful entries (the tail is unspecified but in range).  Launch shapes depend on the capacity only. */
COGDL_API size_t cogdl_hip_csr2csc_padded_workspace_bytes(int64_t m, int64_t n_cols, int64_t nnz);
COGDL_API int cogdl_hip_csr2csc_padded(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t n_cols,
                             int64_t nnz, int32_t *colptr, int32_t *rowind, int32_t *perm, void *workspace,
                             size_t workspace_bytes, void *stream);

/* out[i, 0:h] = src[perm[i], 0:h]  (elem_bytes in {2,4}).
 * Replaces mhtranspose.mhtranspose (operators/spmm/mhTranspose.cu:6-49) and the value leg
 * of csr2csc. */
COGDL_API int cogdl_hip_gather_rows(const int32_t *perm, const void *src, void *out, int64_t n, int64_t h,
                          int elem_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * csr_sddmm: out[e] = < d1[row(e),:], d2[colind[e],:] >      (gradient of edge weights)
 * Replaces sddmm.csr_sddmm (operators/spmm/sddmm.cpp:47-70, sddmm_kernel.cu:249-417,451-476).
 * ------------------------------------------------------------------------------------- */
/* Per-edge outputs: hub rows are split over whole workgroups without any scratch. */
COGDL_API int cogdl_hip_csr_sddmm(const int32_t *rowptr, const int32_t *colind, const float *d1,
                        const float *d2, float *out, int64_t m, int64_t k, int64_t nnz, void *stream);

/* ---------------------------------------------------------------------------------------
 * edge_softmax: per (destination row, head) softmax over the row's edges; values [E,H].
 * Replaces edge_softmax.edge_softmax / edge_softmax_backward
 * (operators/edge_softmax/edge_softmax.cc:16-49, edge_softmax.cu:7-60,63-98).
 * Any H >= 1 (the reference's block (32,H) caps H at 32).  values / softmax / grad / out share the element type
 * `dtype` (f32, f16, bf16; fp32 arithmetic, rounded once on store).
 * With a workspace of cogdl_hip_edge_softmax_workspace_bytes(nnz, h) bytes (device, 256-B aligned), H a power of
 * two <= 64 and 16-byte aligned operands the call is ONE streaming pass (every value read from HBM once, written
 * once; rows that cross tile borders are combined inside the launch, csrc/edge_softmax_flat.hip): two launches (a
 * small init kernel + the main kernel), no host synchronisation, hipGraph-capturable.  Other shapes use
 * row-parallel kernels (every dtype since ABI v5: 2-byte values with H not a power of two / H > 64 are read and written
 * natively by the generic row kernel; without a workspace hub rows are reduced sequentially).
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_edge_softmax_workspace_bytes(int64_t nnz, int64_t h);
COGDL_API int cogdl_hip_edge_softmax_fwd(const int32_t *rowptr, const void *values, void *out, int64_t m,
                               int64_t nnz, int64_t h, int dtype, void *workspace, size_t workspace_bytes,
                               void *stream);
COGDL_API int cogdl_hip_edge_softmax_bwd(const int32_t *rowptr, const void *softmax, const void *grad,
                               void *grad_in, int64_t m, int64_t nnz, int64_t h, int dtype, void *workspace,
                               size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * mhspmm:  out[v,h,:] = sum_e att[e,h] * feat[colind[e],h,:]     feat [n_src,H,F]
 * Replaces mhspmm.mhspmm (operators/spmm/multiheadSpmm.cpp, multiheadSpmm.cu:6-77).
 * att is always f32; feat/out have `dtype` (f32: sequential fp32 mul+add per element).
 * workspace: cogdl_hip_mhspmm_workspace_bytes(nnz, H, F, dtype) (mhsddmm needs none).
 * mhsddmm: out[e,h] = < grad[row(e),h,:], feat[colind[e],h,:] >
 * Replaces mhsddmm.mhsddmm (operators/spmm/multiheadSddmm.cu:6-113).
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_mhspmm_workspace_bytes(int64_t nnz, int64_t h, int64_t f, int dtype);
COGDL_API int cogdl_hip_mhspmm(const int32_t *rowptr, const int32_t *colind, const float *att,
                     const void *feat, void *out, int64_t v, int64_t h, int64_t f, int64_t nnz, int dtype,
                     void *workspace, size_t workspace_bytes, void *stream);
/* mhspmm over a permuted view: the attention row of CSR position e is att[eid[e],:] (eid NULL = identity).  With
 * (rowptr, colind, eid) = (colptr, rowind, perm) of cogdl_hip_csr2csc this is the backward product A^T with the
 * attention left in its forward (CSR) order -- no transposed copy of the [E, H] tensor (the reference permutes it
 * with mhtranspose first, operators/mhspmm.py:57-60). */
COGDL_API int cogdl_hip_mhspmm_eid(const int32_t *rowptr, const int32_t *colind, const float *att, const int32_t *eid,
                         const void *feat, void *out, int64_t v, int64_t h, int64_t f, int64_t nnz, int dtype,
                         void *workspace, size_t workspace_bytes, void *stream);
COGDL_API int cogdl_hip_mhsddmm(const int32_t *rowptr, const int32_t *colind, const float *grad,
                      const float *feat, float *out, int64_t v, int64_t h, int64_t f, int64_t nnz,
                      void *stream);

/* ---------------------------------------------------------------------------------------
 * scatter_max: out[r,c] = max_{e in row r} feat[colind[e],c]; max_id[r,c] = the colind of
 * the FIRST maximum in CSR order (-1 and 0.0 for an empty row).
 * Replaces scatter_max.scatter_max_fp / scatter_max_bp
 * (operators/scatter_max/scatter_max.cc:19-38, scatter_max.cu:5-75) -- without its
 * FLT_MIN initial value, uninitialised argmax and uninitialised gradient buffer.
 * Backward: grad_src[max_id[r,c], c] += grad[r,c]; grad_src [n_src,k] is zeroed inside.
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_scatter_max_workspace_bytes(int64_t nnz, int64_t k);
COGDL_API int cogdl_hip_scatter_max_fwd(const int32_t *rowptr, const int32_t *colind, const float *feat,
                              float *out, int32_t *max_id, int64_t m, int64_t k, int64_t nnz,
                              void *workspace, size_t workspace_bytes, void *stream);
COGDL_API int cogdl_hip_scatter_max_bwd(const float *grad, const int32_t *max_id, float *grad_src, int64_t m,
                              int64_t k, int64_t n_src, void *stream);
/* The same backward as a gather over the transposed structure (colptr[n_src+1], rowind[nnz] of cogdl_hip_csr2csc):
 * grad_src[u,c] = sum over the out-edges (u -> v) with max_id[v,c] == u of grad[v,c], in ascending v.  No atomics and no
 * zero-fill: deterministic, and bit-identical to the sequential reference loop for rows up to the long-row threshold.
 * A multi-edge counts once.  workspace: cogdl_hip_scatter_max_bwd_workspace_bytes (optional, hub rows). */
COGDL_API size_t cogdl_hip_scatter_max_bwd_workspace_bytes(int64_t nnz, int64_t k);
COGDL_API int cogdl_hip_scatter_max_bwd_csc(const int32_t *colptr, const int32_t *rowind, const float *grad,
                                  const int32_t *max_id, float *grad_src, int64_t n_src, int64_t k, int64_t nnz,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * gspmm ("source OP edge feature, then aggregate"): the s_{add,sub,mul}_e_{sum,mean} operators and scatter_add of
 * cogdl/operators/ops.py:4-11 (scatter_add), :19-26 (op_src_edge), :28-40 (op_aggr), :43-52 (src_op_e_aggr_coo),
 * :55-103, which the reference composes from torch ops over the COO edge list (an [E, F] message tensor, then
 * scatter_add_).  Over the destination-sorted (CSR) view of the same edges:
 *   out[v,:] = scale_v * sum_{j in row v} weight[id_j] * ( x[colind[j],:] OP efeat[id_j,:] ),  id_j = eid ? eid[j] : j
 *   eid     CSR position -> edge id of efeat/weight (perm of cogdl_hip_coo2csr_index, as int32), NULL = identity;
 *   x       [n_src, k] or NULL (message = the edge feature alone: scatter_add(efeat, dst));
 *   efeat   [E, k], or [E] with efeat_is_scalar != 0 (ops.py:45-46 views a 1-D e_feat as [E, 1]), or NULL (message =
 *           the source row alone);   weight [E] or NULL (ops.py:49-50);   op: COGDL_HIP_GSPMM_*;
 *   mean    != 0: scale_v = 1 / deg(v) (0 for an empty row), else 1.
 * fp32; per output element the edges are added in row order with every step rounded like the torch expression, so
 * rows up to the long-row threshold equal a sequential CPU scatter_add_ bit for bit.  No atomics (deterministic).
 * ------------------------------------------------------------------------------------- */
/* WMUL: (x * weight) * efeat -- the rounding order of autograd's chain for the MUL operator's source gradient (x := the
 * upstream gradient rows over the SOURCE-sorted view of the edges), so that the fused backward equals the torch one bit
 * for bit; needs x and efeat. */
enum { COGDL_HIP_GSPMM_ADD = 0, COGDL_HIP_GSPMM_SUB = 1, COGDL_HIP_GSPMM_MUL = 2, COGDL_HIP_GSPMM_WMUL = 3 };
COGDL_API size_t cogdl_hip_gspmm_workspace_bytes(int64_t nnz, int64_t k);
COGDL_API int cogdl_hip_gspmm(const int32_t *rowptr, const int32_t *colind, const int32_t *eid, const float *x,
                    const float *efeat, int efeat_is_scalar, const float *weight, int op, int mean, float *out,
                    int64_t m, int64_t k, int64_t nnz, void *workspace, size_t workspace_bytes, void *stream);

/* gspmm over an XCD-partitioned plan of the destination-sorted view (round 6, ABI v9; cogdl_hip_vrows as built by
 * cogdl_amd/xcdplan.py: build(rowptr, colind_sorted, eid_base = the view's perm, split = cogdl_hip_exact_row_edges(nnz))): the
 * same sums -- rows up to the exact-row bound in the caller's edge order, longer rows in pieces like the ordinary launch's
 * long-row path -- with rows of one length per wave.  rowptr: the view's row pointer (used by `mean`).  The Python front
 * takes it for memoised edge lists of skewed graphs from their second use on (operators/ops.py: EdgePlan.xcd). */
COGDL_API size_t cogdl_hip_gspmm_xcd_workspace_bytes(int64_t n_parts, int64_t k);
COGDL_API int cogdl_hip_gspmm_xcd(const cogdl_hip_vrows *plan, const int32_t *rowptr, const float *x, const float *efeat,
                        int efeat_is_scalar, const float *weight, int op, int mean, float *out, int64_t m, int64_t k,
                        void *workspace, size_t workspace_bytes, void *stream);


/* Per-edge gradients of the family (autograd of src_op_e_aggr_coo, ops.py:43-52) without [E, k] temporaries, over the
 * caller's COO list (row = destination, col = source, int64 as CogDL keeps them), from the upstream gradient grad [m, k]:
 *   g = grad[row[e], :] * scale[row[e]]            (scale [m]: 1 / deg for "mean", NULL for "sum")
 *   grad_weight[e]   = sum_k (x[col[e], k] OP efeat[e, k]) * g[k]                     ([E]; NULL = not wanted)
 *   grad_efeat[e, :] = MUL: (g * weight[e]) * x[col[e], :];  ADD: g * weight[e];  SUB: -(g * weight[e])
 *                      ([E, k], or [E] summed over k when efeat_is_scalar; NULL = not wanted; weight NULL = 1)
 * with autograd's roundings in autograd's order (bit-identical to the torch composition for [E, k] outputs).
 * op: COGDL_HIP_GSPMM_ADD | SUB | MUL.  fp32. */
COGDL_API int cogdl_hip_gspmm_edge_grad(const int64_t *row, const int64_t *col, const float *grad, const float *scale,
                              const float *weight, const float *x, const float *efeat, int efeat_is_scalar, int op,
                              float *grad_efeat, float *grad_weight, int64_t n_edges, int64_t k, void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused GAT attention + aggregation (no [E,H] tensor is materialised in forward):
 *   s[e,h] = LeakyReLU(attn_row[row(e),h] + attn_col[colind[e],h]);  a = softmax_row(s)
 *   out[v,h,:] = sum_e a[e,h] * feat[colind[e],h,:]
 * Replaces fused_gatconv.gat_forward / gat_backward as bound by
 * operators/fused_gat.py:14-41 (dgNN; source absent from the reference tree -- semantics
 * are those of the unfused path, layers/gat_layer.py:73-77).
 * Forward also emits edge_max/edge_sum [v,H] (row max of s and sum exp(s-max)), which the
 * backward consumes like the reference's ctx.save_for_backward does (fused_gat.py:20).
 * Forward workspace (optional): cogdl_hip_gat_fwd_workspace_bytes -- hub rows are then split over whole
 * workgroups and their (max, sum, acc) states merged like flash-attention blocks.
 * Backward needs the CSC view (colptr,rowind from cogdl_hip_csr2csc), the forward
 * output and a workspace of cogdl_hip_gat_bwd_workspace_bytes(v, n_src, h, f, nnz, dtype) bytes (D[v,h], the per-tile
 * partials of the tiled shapes, plus the long-row scratch of its two passes; a workspace without the latter disables
 * the long-row path).
 * feat / out / grad_out / grad_feat have element type `dtype` (f32, f16 or bf16: read natively, fp32
 * arithmetic, grad_feat rounded once on store); attention vectors, statistics and their gradients are fp32.
 * Any H and F: rows of up to 64 lanes x one 16-byte vector whose heads are a power-of-two number of lanes take one lane
 * group per row; everything else (8 heads x 64 features, 6 x 12, ...) runs in column tiles with the per-(row, head)
 * scalars finished by a second small kernel (csrc/gat_tiled.hip) -- the reference's backward has no shape limit
 * (operators/fused_gat.py:28-40).
 *
 * Attention dropout (ABI v5).  CogDL's gat model runs with attn_drop = 0.5 by default (models/nn/gat.py:30), which
 * sends GATLayer.forward through leaky_relu(h_l[row] + h_r[col]) -> edge_softmax -> nn.Dropout -> mhspmm
 * (layers/gat_layer.py:72-77).  cogdl_hip_gat_dropout_fwd / _bwd are that composition as ONE forward kernel and two
 * backward passes:  out[v,h,:] = sum_e d[e,h] * a[e,h] * feat[colind[e],h,:],  d[e,h] = keep(e,h) ? scale : 0.
 * The mask is a pure function of (seed, e, h) -- Philox4x32-10, csrc/philox.h: counter (e, 0, h / 8, 0), key = seed, the
 * (h % 8)-th 16-bit piece compared with thresh = round(p * 65536); scale = 65536 / (65536 - thresh) (p = 0.5: 2, as
 * torch) -- with e the edge's position in (rowptr, colind), so nothing of size [E,H] is stored between forward and
 * backward: the backward regenerates it (its column pass reads e from `perm`, the CSC-slot -> CSR-position array of
 * cogdl_hip_csr2csc).  cogdl_hip_edge_dropout_mask writes d[e,h] as a dense [nnz,h] fp32 tensor (tests; callers that
 * want the same mask on the unfused operators).  h <= 64.  p = 0 is the plain operator.
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_gat_fwd_workspace_bytes(int64_t nnz, int64_t h, int64_t f, int dtype);
COGDL_API int cogdl_hip_gat_fwd(const int32_t *rowptr, const int32_t *colind, const float *attn_row,
                      const float *attn_col, const void *feat, float negative_slope, void *out,
                      float *edge_max, float *edge_sum, int64_t v, int64_t h, int64_t f, int64_t nnz,
                      int dtype, void *workspace, size_t workspace_bytes, void *stream);
COGDL_API size_t cogdl_hip_gat_bwd_workspace_bytes(int64_t v, int64_t n_src, int64_t h, int64_t f, int64_t nnz,
                      int dtype);
COGDL_API int cogdl_hip_gat_bwd(const int32_t *rowptr, const int32_t *colind, const int32_t *colptr,
                      const int32_t *rowind, const float *attn_row, const float *attn_col,
                      const void *feat, float negative_slope, const float *edge_max,
                      const float *edge_sum, const void *out, const void *grad_out,
                      void *grad_feat, float *grad_attn_row, float *grad_attn_col,
                      void *workspace, size_t workspace_bytes, int64_t v, int64_t n_src, int64_t h,
                      int64_t f, int64_t nnz, int dtype, void *stream);
COGDL_API int cogdl_hip_gat_dropout_fwd(const int32_t *rowptr, const int32_t *colind, const float *attn_row,
                      const float *attn_col, const void *feat, float negative_slope, float p, uint64_t seed,
                      void *out, float *edge_max, float *edge_sum, int64_t v, int64_t h, int64_t f, int64_t nnz,
                      int dtype, void *workspace, size_t workspace_bytes, void *stream);
COGDL_API int cogdl_hip_gat_dropout_bwd(const int32_t *rowptr, const int32_t *colind, const int32_t *colptr,
                      const int32_t *rowind, const int32_t *perm, const float *attn_row, const float *attn_col,
                      const void *feat, float negative_slope, float p, uint64_t seed, const float *edge_max,
                      const float *edge_sum, const void *out, const void *grad_out,
                      void *grad_feat, float *grad_attn_row, float *grad_attn_col,
                      void *workspace, size_t workspace_bytes, int64_t v, int64_t n_src, int64_t h,
                      int64_t f, int64_t nnz, int dtype, void *stream);
/* The fused GAT operator over XCD-partitioned plans (cogdl_hip_vrows above; p == 0: no dropout, p > 0: the mask of
 * cogdl_hip_gat_dropout_fwd -- a function of (seed, CSR position, head), so plan->eid must be the CSR position of every plan
 * position; for the CSC plan of the backward: the CSR position of the transposed entry, i.e. perm composed with the plan).
 * Shapes the ordinary entries run through the chunk-wise forward / the column-tiled backward return COGDL_HIP_EUNSUPPORTED:
 * the caller keeps the ordinary entry for them.  Same results as the ordinary entries up to re-association. */
COGDL_API size_t cogdl_hip_gat_fwd_xcd_workspace_bytes(int64_t n_parts, int64_t h, int64_t f, int dtype);
COGDL_API int cogdl_hip_gat_fwd_xcd(const cogdl_hip_vrows *plan, const float *attn_row, const float *attn_col,
                      const void *feat, float negative_slope, float p, uint64_t seed, void *out, float *edge_max,
                      float *edge_sum, int64_t v, int64_t h, int64_t f, int dtype, void *workspace,
                      size_t workspace_bytes, void *stream);
COGDL_API size_t cogdl_hip_gat_bwd_xcd_workspace_bytes(int64_t v, int64_t h, int64_t f, int64_t n_parts_row,
                      int64_t n_parts_col, int dtype);
COGDL_API int cogdl_hip_gat_bwd_xcd(const cogdl_hip_vrows *plan_csr, const cogdl_hip_vrows *plan_csc,
                      const float *attn_row, const float *attn_col, const void *feat, float negative_slope, float p,
                      uint64_t seed, const float *edge_max, const float *edge_sum, const void *out,
                      const void *grad_out, void *grad_feat, float *grad_attn_row, float *grad_attn_col,
                      void *workspace, size_t workspace_bytes, int64_t v, int64_t n_src, int64_t h, int64_t f,
                      int dtype, void *stream);
COGDL_API int cogdl_hip_edge_dropout_mask(int64_t nnz, int64_t h, float p, uint64_t seed, float *mask, void *stream);
/* the same mask into HOST memory, computed on the host by the same code (no GPU needed: CPU tests pin the generator) */
COGDL_API int cogdl_hip_edge_dropout_mask_host(int64_t nnz, int64_t h, float p, uint64_t seed, float *mask);

/* ---------------------------------------------------------------------------------------
 * Helpers used by the graph-plan cache and the vertex-sharded (multi-GPU) SpMM.
 * fingerprint: 64-bit content hash of (rowptr[0..m], colind[0..nnz)) = the sum modulo 2^64 of the
 * COGDL_HIP_FINGERPRINT_PARTS partials written to out_parts (device memory, or host-mapped pinned memory so that
 * no device-to-host copy is needed; plain stores, no memset, no atomics).  Not part of the reference.
 * ------------------------------------------------------------------------------------- */
#define COGDL_HIP_FINGERPRINT_PARTS 256
COGDL_API int cogdl_hip_csr_fingerprint(const int32_t *rowptr, const int32_t *colind, int64_t m, int64_t nnz,
                              uint64_t *out_parts, void *stream);

/* ---------------------------------------------------------------------------------------
 * coo2csr_index on the GPU: stable sort of the edges by row -> (row_ptr[num_nodes+1], perm[nnz]), perm[j] = COO
 * position of CSR entry j (edges of a row keep their COO order).  int64 in and out, like the reference's
 * sampler.coo2csr_cpu_index (cogdl/operators/sample/sample.cpp:234-270), which cogdl/utils/graph_utils.py:133-142
 * reaches through a GPU -> CPU -> GPU round trip and a single-threaded counting sort.  *bad_flag (device int) is set
 * to 1 if a row id lies outside [0, num_nodes).  workspace: cogdl_hip_coo2csr_index_workspace_bytes.
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_coo2csr_index_workspace_bytes(int64_t nnz, int64_t num_nodes);
COGDL_API int cogdl_hip_coo2csr_index(const int64_t *row, int64_t nnz, int64_t num_nodes, int64_t *row_ptr,
                            int64_t *perm, int *bad_flag, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * sample_adj on the GPU: neighbour sampling + relabelling of a GPU-resident CSR graph, the contract of
 * sampler.sample_adj (cogdl/operators/sample/sample.cpp:6-144; the reference runs it single-threaded on the CPU
 * with libc rand()) and of cogdl_host_sample_adj (include/cogdl_host.h), int64 in and out, device pointers:
 *   -> out_indptr[batch+1], out_indices[E'] (LOCAL ids), out_nodes[N'] (global id of every local id: the seeds
 *      first, then new nodes in discovery order), out_edges[E'] (CSR positions of the picked edges);
 *   num_neighbors < 0: all neighbours (identical to the reference); replace != 0: num_neighbors uniform draws per
 *   seed with a neighbour; else min(deg, num_neighbors) distinct neighbours (Floyd), emitted in ascending CSR
 *   position; num_neighbors <= 1024 without replacement (else COGDL_HIP_ERANGE).  Seeds must be distinct.
 *   Randomness: a counter-based generator keyed by (seed, seed row, draw) -- reproducible, scheduling-independent.
 *   cap_edges >= sum of the per-seed counts (batch * num_neighbors always suffices for num_neighbors >= 0);
 *   out_nodes holds batch + cap_edges entries.  out_counts (DEVICE int64[3]) = {N', E', flags}; flags != 0 marks
 *   an invalid result: bit 0 seed id out of range, bit 1 neighbour id out of range, bit 2 capacity exceeded.
 *   Nothing synchronises; the caller reads out_counts when it needs the sizes.
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_sample_adj_workspace_bytes(int64_t batch, int64_t cap_edges, int64_t num_nodes);
COGDL_API int cogdl_hip_sample_adj(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                         const int64_t *node_idx, int64_t batch, int64_t num_neighbors, int replace,
                         uint64_t seed, int64_t *out_indptr, int64_t *out_indices, int64_t *out_nodes,
                         int64_t *out_edges, int64_t cap_edges, int64_t *out_counts, void *workspace,
                         size_t workspace_bytes, void *stream);

/* sample_adj into buffers of a FIXED capacity, for mini-batch steps captured in a hipGraph (every launch shape is a
 * function of the capacities, never of what was sampled):
 *   batch        = the number of seed SLOTS; *batch_count (device, NULL = all) of them are in use -- hop 2 passes hop 1's
 *                  out_nodes with out_counts[0] as its count;
 *   seed_dev     = device pointer to the RNG seed, NULL = none: the draws use seed + *seed_dev, so the caller bumps the
 *                  device word between replays and tells the hops of one batch apart by the immediate `seed`;
 *   out_indptr   holds batch + cap_edges + 1 entries: rows beyond the seeds in use are empty (= E'), so the block is a
 *                  well-formed CSR over all batch + cap_edges possible nodes (the padding of cogdl/data/data.py:828-830
 *                  at full capacity);
 *   out_indices / out_edges beyond E' are 0, out_nodes beyond N' are 0 (valid ids: a feature gather over the whole
 *                  capacity reads row 0 for them).
 * Everything else as cogdl_hip_sample_adj (same workspace query). */
COGDL_API int cogdl_hip_sample_adj_padded(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                const int64_t *node_idx, int64_t batch, const int64_t *batch_count,
                                int64_t num_neighbors, int replace, uint64_t seed, const uint64_t *seed_dev,
                                int64_t *out_indptr, int64_t *out_indices, int64_t *out_nodes, int64_t *out_edges,
                                int64_t cap_edges, int64_t *out_counts, void *workspace, size_t workspace_bytes,
                                void *stream);
/* cogdl_hip_sample_adj_padded that ALSO leaves the block as the SpMM takes it -- rowptr32 [batch + 1] (the seed rows
 * only: the rows a layer keeps, graphsage.py:99), col32 [cap_edges] (local ids, 0 behind E') and, unless NULL,
 * inv_deg [batch] = 1 / sampled in-degree (0 for an empty row: Graph.row_norm, cogdl/data/data.py:240-258) -- written by
 * the sampler's own kernels: what cogdl_hip_block_prepare would produce from out_indptr / out_indices, without its
 * launch (a captured mini-batch step pays per dependent kernel node, not per byte). */
COGDL_API int cogdl_hip_sample_adj_block(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                               const int64_t *node_idx, int64_t batch, const int64_t *batch_count,
                               int64_t num_neighbors, int replace, uint64_t seed, const uint64_t *seed_dev,
                               int64_t *out_indptr, int64_t *out_indices, int64_t *out_nodes, int64_t *out_edges,
                               int64_t cap_edges, int64_t *out_counts, int32_t *rowptr32, int32_t *col32,
                               float *inv_deg, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Graph preprocessing on the GPU (what cogdl.data.Graph does once per graph before its first SpMM), int64 COO in and out:
 * add_remaining_self_loops (cogdl/utils/graph_utils.py:40-70 via Adjacency.add_remaining_self_loops,
 *   cogdl/data/data.py:175-191): existing self loops are dropped, one loop per node is appended behind the kept edges
 *   (which keep their order); a node that had loops keeps the weight of its LAST one, the others get fill_value;
 *   val == NULL means unit weights.  Outputs hold nnz + num_nodes entries; *out_count (device) = the number written.
 * coo_norm_weights (graph_utils.py:72-89, degrees as in :10-17 = edges per row): mode 0 (sym)
 *   out[e] = d^-1/2[col[e]] * val[e] * d^-1/2[row[e]], mode 1 (row) out[e] = val[e] / d[row[e]]; 1/0 -> 0.
 * *bad_flag (device int, zeroed by the caller) gets bit 0 if an index lies outside [0, num_nodes).
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_add_remaining_self_loops_workspace_bytes(int64_t nnz, int64_t num_nodes);
COGDL_API int cogdl_hip_add_remaining_self_loops(const int64_t *row, const int64_t *col, const float *val, int64_t nnz,
                                       int64_t num_nodes, float fill_value, int64_t *out_row, int64_t *out_col,
                                       float *out_val, int64_t *out_count, int *bad_flag, void *workspace,
                                       size_t workspace_bytes, void *stream);
COGDL_API size_t cogdl_hip_coo_norm_weights_workspace_bytes(int64_t num_nodes);
COGDL_API int cogdl_hip_coo_norm_weights(const int64_t *row, const int64_t *col, const float *val, int64_t nnz,
                               int64_t num_nodes, int mode, float *out, int *bad_flag, void *workspace,
                               size_t workspace_bytes, void *stream);

/* block_prepare: a sampled block (int64, as sample_adj returns it) as the SpMM takes it, in one launch:
 *   rowptr32[0..n_rows] = row_ptr[0..n_rows], col32[0..n_slots) = col[0..n_slots) narrowed to int32, and (inv_deg != NULL)
 *   inv_deg[r] = 1 / (row_ptr[r+1] - row_ptr[r]), 0 for a row without edges -- the in_norm of Graph.row_norm()
 *   (cogdl/data/data.py:240-258), i.e. the mean aggregator's weights.  n_rows may be smaller than the block's row count
 *   (only the target rows are aggregated), n_slots is the capacity of col (>= row_ptr[n_rows]). */
COGDL_API int cogdl_hip_block_prepare(const int64_t *row_ptr, const int64_t *col, int64_t n_rows, int64_t n_slots,
                            int32_t *rowptr32, int32_t *col32, float *inv_deg, void *stream);

/* ---------------------------------------------------------------------------------------
 * subgraph on the GPU: the node-induced subgraph of a GPU-resident CSR graph, the contract of sampler.subgraph
 * (cogdl/operators/sample/sample.cpp:146-188, reached from Graph.csr_subgraph, cogdl/data/data.py:850-872) and of
 * cogdl_host_subgraph, int64 in and out, device pointers:
 *   row i of the result = row node_idx[i] restricted to the sources that are in node_idx, relabelled to their
 *   position in node_idx (a duplicated id keeps its last position), edges in CSR order;
 *   out_indptr[batch+1], out_indices[E'] (local ids), out_edges[E'] (CSR positions of the kept edges);
 *   cap_edges >= E' (the summed degree of the selected rows always suffices);
 *   out_counts (DEVICE int64[2]) = {E', flags}; flags: bit 0 node id out of range, bit 1 neighbour id out of range,
 *   bit 2 capacity exceeded.  Nothing synchronises.
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_subgraph_workspace_bytes(int64_t batch, int64_t num_nodes);
COGDL_API int cogdl_hip_subgraph(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                       const int64_t *node_idx, int64_t batch, int64_t *out_indptr, int64_t *out_indices,
                       int64_t *out_edges, int64_t cap_edges, int64_t *out_counts, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Row gathers by node id.
 * gather_feature_rows: out[i,:] = src[ids[i],:], rows of row_bytes bytes (a multiple of 4) -- the mini-batch feature
 *   gather x[n_id] of the sampling pipeline (cogdl/data/sampler.py:82-116, cogdl/models/nn/graphsage.py:86-99).
 *   `src` is a device pointer OR a pointer into PINNED host memory (device-mapped): the selected rows are then read
 *   straight over the host link, overlapping other streams ("zero copy"); `out` is device memory.  ids int64 (what
 *   the sampler returns) or int32 (_i32).  *bad_flag (device int, may be NULL; the caller zeroes it) gets bit 0 if
 *   an id lies outside [0, n_src): such rows are skipped.
 * add_rows_at_f32: out[ids[i],:] += src[i,:] for DISTINCT ids (no atomics): accumulation of returned halo gradients
 *   in the vertex-sharded SpMM backward (cogdl_amd/dist.py).
 * ------------------------------------------------------------------------------------- */
COGDL_API int cogdl_hip_gather_feature_rows(const int64_t *ids, const void *src, void *out, int64_t n,
                                  int64_t row_bytes, int64_t n_src, int *bad_flag, void *stream);
COGDL_API int cogdl_hip_gather_feature_rows_i32(const int32_t *ids, const void *src, void *out, int64_t n,
                                      int64_t row_bytes, int64_t n_src, int *bad_flag, void *stream);
COGDL_API int cogdl_hip_add_rows_at_f32(const int64_t *ids, const float *src, float *out, int64_t n, int64_t k,
                              int64_t n_dst, int *bad_flag, void *stream);

/* ---------------------------------------------------------------------------------------
 * linear_wgrad: grad_w[out, in] = grad_out[k_rows, out]^T . x[k_rows, in], grad_b[out] = column sums of grad_out
 * (grad_b may be NULL) -- the weight/bias gradient of the `self.linear(x)` inside every CogDL layer
 * (cogdl/layers/gcn_layer.py:52, gat_layer.py:60, sage_layer.py:72; torch.nn.Linear's backward, which torch hands to
 * hipBLASLt).  fp32, row-major, k_rows = number of nodes: a split-K v_mfma_f32_32x32x2_f32 kernel that streams both
 * operands once (csrc/linear_wgrad.hip).  SURVEY.md section 8f rank 3: the one dense product of the path worth a
 * hand-written MFMA kernel.  workspace: cogdl_hip_linear_wgrad_workspace_bytes; x, grad_out 16-byte aligned.
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_linear_wgrad_workspace_bytes(int64_t k_rows, int64_t in_features, int64_t out_features);
COGDL_API int cogdl_hip_linear_wgrad_f32(const float *x, const float *grad_out, float *grad_w, float *grad_b,
                               int64_t k_rows, int64_t in_features, int64_t out_features, void *workspace,
                               size_t workspace_bytes, void *stream);

/* linear_fwd: out[rows, n] = x[rows, k] . B (+ bias[n]) for tall-skinny x (rows = number of nodes), fp32.
 * w_is_n_by_k != 0: B = w^T, w stored [n, k] -- torch.nn.Linear's forward x . W^T + b;
 * w_is_n_by_k == 0: B = w stored [k, n]      -- its grad_input = grad_out . W.
 * Every wave stages 32-row tiles of x through LDS into v_mfma_f32_32x32x2_f32; B stays resident in LDS
 * (csrc/linear_fwd.hip).  Returns COGDL_HIP_EUNSUPPORTED for shapes it does not cover (n > 64 or B larger than 96 KB):
 * the caller then keeps its BLAS call.  x 16-byte aligned. */
COGDL_API int cogdl_hip_linear_fwd_f32(const float *x, const float *w, const float *bias, float *out, int64_t rows,
                             int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream);

/* linear_fwd_bf16 (ABI v9): out[rows, n] (bf16) = bf16(x[rows, k]) . bf16(B) (+ bias[n], fp32 added before the final
 * rounding) -- the same product under bf16 autocast, as torch computes `torch.matmul(x, self.W)` / nn.Linear there
 * (cogdl/layers/gat_layer.py:59 on BASELINE configs[2]: 232,965 x 602 -> 64).  x_dtype / w_dtype: COGDL_HIP_F32 (rounded
 * to bf16 in registers, to nearest even: no bf16 copy of x is made) or COGDL_HIP_BF16; fp32 accumulation in
 * v_mfma_f32_32x32x16_bf16, whose operands are eight CONSECUTIVE k per lane: rows of x are read straight from global
 * memory, B sits in LDS in operand order (csrc/linear_fwd16.hip).  w_is_n_by_k as above.  COGDL_HIP_EUNSUPPORTED: n > 64,
 * more than 128 KB of bf16 B operands, bf16 rows of odd length.  x 16-byte aligned. */
COGDL_API int cogdl_hip_linear_fwd_bf16(const void *x, int x_dtype, const void *w, int w_dtype, const float *bias, void *out,
                              int64_t rows, int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream);
/* head_projection_fwd (ABI v9): h_l[v, h] = sum_f a_l[h, f] * feat[v, h, f] and h_r likewise with a_r -- both attention
 * projections of GATLayer.forward (cogdl/layers/gat_layer.py:65-66) in one pass over feat [n_rows, heads, f_dim] (f32 / f16 /
 * bf16); a_l, a_r [heads, f_dim] fp32; h_l, h_r [n_rows, heads] fp32.  fp32 products summed left to right
 * (csrc/head_proj.hip). */
COGDL_API int cogdl_hip_head_projection_fwd(const void *feat, int dtype, const float *a_l, const float *a_r, float *h_l, float *h_r,
                                  int64_t n_rows, int64_t heads, int64_t f_dim, void *stream);

/* linear_fwd_f16: the same kernel on v_mfma_f32_32x32x16_f16 -- the autocast dtype of the reference's own Trainer(fp16=True)
 * (cogdl/trainer/trainer.py: torch.cuda.amp.autocast); x_dtype / w_dtype: COGDL_HIP_F32 or COGDL_HIP_F16, out f16. */
COGDL_API int cogdl_hip_linear_fwd_f16(const void *x, int x_dtype, const void *w, int w_dtype, const float *bias, void *out,
                             int64_t rows, int64_t k_dim, int64_t n_dim, int w_is_n_by_k, void *stream);

/* ---------------------------------------------------------------------------------------
 * Vertex-sharded graphs (BASELINE.json configs[4]; no reference counterpart -- CogDL only partitions on the host, with
 * METIS, for ClusterGCN: cogdl/data/sampler.py:188-243).  csrc/shard.hip.
 * shard_count / shard_fill: one rank's shard of a 1-D row-partitioned CSR matrix.  The rank owns the rows [lo, hi) =
 *   n_local rows, `rowptr` [n_local + 1] (int64; it may start at any offset: rowptr[0] is subtracted), `col` [nnz]
 *   GLOBAL column ids in [0, n_global), `weight` [nnz] fp32 or NULL.
 *   count: counts[4] (DEVICE int64) = {edges with a column inside [lo, hi), edges outside, distinct outside columns
 *          (= halo rows), flags (bit 0: a column id outside [0, n_global))}.  The caller reads them (the one
 *          synchronisation of building a shard) and allocates the outputs of
 *   fill : rowptr_loc / rowptr_rem [n_local + 1] int32, colind_loc [counts[0]] = col - lo, colind_rem [counts[1]] =
 *          index into the halo table, w_loc / w_rem (NULL iff weight is NULL), halo_ids [counts[2]] = the distinct
 *          outside columns ascending (int64), cut [n_bounds] (int64) = halo ids below bounds[q] -- with
 *          bounds = the partition's row ranges that is where every owner's part of the halo table starts.
 *   A row's edges keep their CSR order inside both blocks.  The same workspace (cogdl_hip_shard_workspace_bytes) must
 *   be passed to both calls, untouched in between.  n_global < 2^31 - 1, nnz < 2^31.
 * bfs_step: one level of a breadth-first search over a CSR graph (int64 rowptr / col, n vertices): every vertex with
 *   level[v] == cur gives its unvisited (level < 0) neighbours the level cur + 1; *changed (device int, zeroed by the
 *   caller) is set when it did.  The caller loops over the levels (cogdl_amd/dist.py: bfs_order, the locality
 *   reordering in front of a contiguous partition).
 * ------------------------------------------------------------------------------------- */
COGDL_API size_t cogdl_hip_shard_workspace_bytes(int64_t n_local, int64_t n_global);
COGDL_API int cogdl_hip_shard_count(const int64_t *rowptr, const int64_t *col, int64_t n_local, int64_t nnz, int64_t lo,
                          int64_t hi, int64_t n_global, int64_t *counts, void *workspace, size_t workspace_bytes,
                          void *stream);
COGDL_API int cogdl_hip_shard_fill(const int64_t *rowptr, const int64_t *col, const float *weight, int64_t n_local,
                         int64_t nnz, int64_t lo, int64_t hi, int64_t n_global, const int64_t *bounds, int64_t n_bounds,
                         int32_t *rowptr_loc, int32_t *colind_loc, float *w_loc, int32_t *rowptr_rem,
                         int32_t *colind_rem, float *w_rem, int64_t *halo_ids, int64_t *cut, void *workspace,
                         size_t workspace_bytes, void *stream);
COGDL_API int cogdl_hip_bfs_step(const int64_t *rowptr, const int64_t *col, int64_t n, int32_t *level, int32_t cur,
                       int *changed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COGDL_HIP_H */
