/*
 * stin_hip.h - C ABI of libstin_hip.so: the MI355X (gfx950) kernels behind the
 * STINet graph-convolution hot path.
 *
 * The reference (johnpeterflynn/surface-texture-inpainting-net) is 100 % Python and
 * owns no native ABI: every entry point below replaces a call the reference makes
 * into the third-party torch_geometric / torch_scatter / ATen kernels.  The cited
 * file:line is the reference call site whose arithmetic the entry point takes over
 * (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless noted.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Every call
 *     only ENQUEUES work on that stream: no allocation, no synchronisation, no
 *     ownership transfer; all buffers (incl. workspaces) are caller-owned and must
 *     outlive the enqueued work.  Re-entrant per stream, no global mutable state.
 *   - return value: 0 = ok, < 0 = argument error (STIN_E_*), > 0 = a hipError_t.
 *   - feature matrices are row-major fp32 with an explicit leading dimension `ld*`
 *     (in elements) so that column slices of a wider matrix can be passed.
 *   - graph structure is CSR with int32 `rowptr[N+1]` / `col[E]` (built once per
 *     sample by stin_csr_from_coo_i64 from the int64 index tensors at the Python
 *     boundary).  Within a row, entries keep the ORIGINAL edge order (stable),
 *     so every segmented reduction has a fixed, reproducible summation order -
 *     no float atomics anywhere.
 */
#ifndef STIN_HIP_H
#define STIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STIN_VERSION 100           /* major*10000 + minor*100 + patch */

#define STIN_OK 0
#define STIN_E_NULL (-1)           /* required pointer is NULL */
#define STIN_E_SIZE (-2)           /* negative / inconsistent size or leading dimension */
#define STIN_E_ALIGN (-3)          /* pointer / ld not aligned as the kernel requires */
#define STIN_E_WORKSPACE (-4)      /* workspace too small */
#define STIN_E_UNSUPPORTED (-5)    /* shape outside what this build supports */

typedef void* stin_stream_t;
typedef void* stin_event_t;  /* a hipEvent_t owned by the caller (only stin_edgeconv_block_bwd's optional side stream uses events) */
/* bf16 STORAGE variants (*_bf16): the same operation on row-major bfloat16 matrices (raw 16-bit patterns, the
 * upper half of an fp32).  Every kernel widens to fp32 on load, computes and accumulates in fp32 and rounds to
 * nearest-even on store; statistics, weights, biases, index plans and weight gradients stay fp32.  Halves the HBM
 * bytes of the streaming/gather kernels and puts the GEMMs on one bf16 MFMA per k-step.  The reference is
 * fp32-only: this is the build's mixed-precision extension for BASELINE configs 3 and 5 (stated tolerance, not
 * the 1e-4 fp32 bar).  bf16 rows must be 4-channel vectorisable: C % 4 == 0, ld % 4 == 0, 8-byte aligned. */
typedef uint16_t stin_bf16_t;

int stin_version(void);
/* Static, NUL-terminated description of a return code (host pointer). */
const char* stin_error_string(int code);

/* ------------------------------------------------------------------ graph plan --
 * COO -> CSR, grouping `E` (key, val) pairs by key in [0, N).
 *   rowptr[n] .. rowptr[n+1] delimit the entries of key n, in ORIGINAL pair order;
 *   col[e]  = val[perm[e]]  (or perm[e] when val == NULL);
 *   perm[e] = index of the pair now at CSR slot e (may be NULL);
 *   inv_deg[n] = 1 / max(1, rowptr[n+1]-rowptr[n])       (may be NULL);
 *   *bad (device int32, may be NULL) is set non-zero when a key is outside [0, N)
 *   or a val outside [0, val_limit) - the analogue of the IndexError the reference's
 *   index_select / scatter raise on CPU; such pairs are left out of the CSR.
 * Implementation: counting sort (integer-atomic histogram, one scan, atomic fill, rank within
 * row) - the result is the stable order above regardless of atomic arrival order.
 * Replaces the implicit indexing of PyG MessagePassing.propagate
 * (models/modules/edge_conv_filter.py:57 via torch_geometric) and of
 * torch_scatter (models/surfacetextureinpaintingnet.py:384-386,:422):
 *   destination CSR: key = edge_index[1], val = edge_index[0]
 *   source CSR     : key = edge_index[0], val = edge_index[1]
 *   children CSR   : key = hierarchy_trace_index_l, val = NULL
 */
size_t stin_csr_workspace_bytes(int64_t E, int64_t N);
int stin_csr_from_coo_i64(const int64_t* key, const int64_t* val, int64_t E, int64_t N,
                          int64_t val_limit, int32_t* rowptr, int32_t* col, int32_t* perm,
                          float* inv_deg, int32_t* bad, void* workspace, size_t workspace_bytes,
                          stin_stream_t stream);
/* Both CSRs of one directed edge set in the same launches: by destination (col = sources, inv_deg =
 * 1/max(1, in-degree)) for the forward gather and dA, by source (col = targets) for dB.
 * xslot[s] (may be NULL) = destination-CSR slot of the edge at source-CSR slot s: lets the backward
 * find an edge's saved ReLU mask (stored per destination-CSR slot) from the source side;
 * w_src[s] (may be NULL; needs xslot and inv_deg_dst) = inv_deg_dst[col_src[s]], the mean weight of
 * that edge's target, laid out sequentially for the dB pass. */
int stin_csr_pair_from_edges_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N,
                                 int32_t* rowptr_dst, int32_t* col_dst, float* inv_deg_dst,
                                 int32_t* rowptr_src, int32_t* col_src, int32_t* xslot, float* w_src,
                                 int32_t* bad, void* workspace, size_t workspace_bytes, stin_stream_t stream);
/* Every CSR structure of a sample in ONE batch of four kernel launches behind one memset of the counters (round 3; the two entry points above are batches of one job).
 * A job = one CSR grouped by a[] with the value b[] (b == NULL: the pair id itself, e.g. the children of every coarse vertex
 * of hierarchy_trace_index_l) or, with pair != 0, both CSRs of one directed edge set (a = edge_index[1] = targets,
 * b = edge_index[0] = sources; outputs as stin_csr_pair_from_edges_i64).  narrow_out (may be NULL): the int32 copy of a[]
 * (0 where out of range) - the `trace` the pool / unpool kernels gather with, written by the counting pass.
 * jobs: HOST array (copied into the kernel arguments); out-of-range pairs set *bad and are left out, as above.
 * workspace >= stin_plan_build_workspace_bytes(sum of E, sum over jobs of (pair ? 2 : 1) * N + 1). */
#define STIN_PLAN_MAX_JOBS 16
typedef struct stin_plan_job {
    const int64_t *a, *b;
    int64_t E, N, b_limit;
    int32_t *rowptr0, *col0, *perm0;
    float* inv_deg0;
    int32_t *rowptr1, *col1, *xslot;
    float* w_src;
    int32_t* narrow_out;
    int32_t pair, reserved;
} stin_plan_job_t;                                    /* 11 pointers + 3 int64 + 2 int32 = 120 bytes */
size_t stin_plan_build_workspace_bytes(int64_t total_E, int64_t total_counters);
int stin_plan_build_many(const stin_plan_job_t* jobs, int n_jobs, int32_t* bad, void* workspace, size_t workspace_bytes,
                         stin_stream_t stream);
/* Vertex renumbering by locality (round 3; optional - SURVEY 7.3 "optional vertex reordering, inverted at the boundary").  The
 * reference has no counterpart: its kernels (torch_scatter / PyG gathers, models/surfacetextureinpaintingnet.py:384-391 and the
 * EdgeConv message passing) index rows in dataset order; on the GPU the gathered neighbour rows then all cross the fabric.
 *   stin_vertex_order_f32: pos [n0, >= 3] (ld_pos floats per row: x, y, z at columns 0..2 of the pointer passed) -> per level
 *     rank int32 [n + 1] (old id -> new id, rank[n] = n) and order int32 [n] (new -> old; may be NULL).  Level 0: stable sort
 *     by the 30-bit Morton code on the bounding box; level l >= 1: stable sort by the smallest new id among the children under
 *     levels[l].trace (the level l - 1 -> l map, int64 [n_{l-1}]).  Deterministic.
 *   stin_relabel_many_i64: out[i] = rank[in[i]] (limit where in[i] is outside [0, limit): the CSR build then flags and drops
 *     it exactly like an out-of-range original id) for up to 16 index arrays in one launch.  A job with rank_fine != NULL is a
 *     TRACE (in = trace [n_fine], rank = the coarse level's): it also writes fine_out[i] = rank_fine[i] (the pair's second
 *     member, int64 for stin_plan_build_many) and trace_out[rank_fine[i]] = out[i] (int32: new fine id -> new coarse id, 0
 *     where out of range).  Index arrays keep their ORDER: CSR rows and children lists built from them keep the reference's. */
typedef struct stin_order_level {
    int64_t n;
    const int64_t* trace;
    int32_t* rank;
    int32_t* order;
} stin_order_level_t;
size_t stin_vertex_order_workspace_bytes(int64_t n_max);
int stin_vertex_order_f32(const float* pos, int64_t ld_pos, const stin_order_level_t* levels, int n_levels, void* workspace,
                          size_t workspace_bytes, stin_stream_t stream);
#define STIN_RELABEL_MAX_JOBS 16
typedef struct stin_relabel_job {
    const int64_t* in;
    int64_t n;
    const int32_t* rank;
    int64_t limit;
    int64_t* out;
    const int32_t* rank_fine;
    int64_t* fine_out;
    int32_t* trace_out;
} stin_relabel_job_t;                                  /* 64 bytes */
int stin_relabel_many_i64(const stin_relabel_job_t* jobs, int n_jobs, stin_stream_t stream);
/* dst[i] = (int32) src[i]; *bad set when a value is outside [0, limit). */
int stin_narrow_i64_to_i32(const int64_t* src, int64_t n, int64_t limit, int32_t* dst, int32_t* bad,
                           stin_stream_t stream);

/* -------------------------------------------------------- segmented reductions --
 * out[n, :] = sum_{e in row n} src[col ? col[e] : e, :]        (mean: / max(1, deg))
 * The standalone scatter-add / scatter-mean (torch_scatter.scatter_{sum,mean},
 * models/surfacetextureinpaintingnet.py:384; SAGEConv mean aggregation,
 * models/modules/sage_conv_filter.py:122; backward of `x[traces]`, :391; PyG
 * aggr='add' in utils/metrics/graph_metrics.py:6-16).
 */
#define STIN_SEG_NONTEMPORAL 2   /* OR-ed into `mean`: the gathered source is read once and is larger than the 256 MB Infinity Cache
                                   (the standalone scatter-add of BASELINE: 307 MB) -> non-temporal loads, +5 %; on a
                                   cache-resident source they cost 25-40 %, so the caller decides by the source's size */
int stin_segment_sum_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* col,
                         int64_t N, int C, int mean, float* out, int64_t ld_out, stin_stream_t stream);

/* ------------------------------------------------------------ fused edge stage --
 * EdgeConv(aggr='mean') after the exact algebraic restructure:
 *   h[i, :] = (1/max(1,deg i)) * sum_{j in N(i)} ReLU(A[i, :] + B[j, :])
 * with A = x (Wa-Wb)^T + b1, B = x Wb^T per-VERTEX GEMM outputs.  Takes over the
 * per-edge gather/cat/Linear/ReLU/scatter chain of PyG EdgeConv
 * (models/modules/edge_conv_filter.py:46-57,
 *  models/modules/edge_conv_translation_invariance.py:19-21).
 *   fwd     : destination CSR.  With indicator != 0 the kernel also writes columns
 *             H..H+3 of `out` = ([deg i > 0], 0, 0, 0) (needs ldo >= H+4): the following
 *             per-vertex GEMM against [W2 | b2 | 0 0 0] then yields W2 h + b2 [deg > 0],
 *             i.e. PyG's "vertices without in-edges aggregate to exactly 0".
 *             With mask != NULL (needs H % 128 == 0) it also stores, per destination-CSR edge slot e,
 *             the H ReLU decisions [A[i,c] + B[j,c] > 0] as H bits at mask[e * H/32 ...] (E*H/8 bytes,
 *             32x smaller than an fp32 [E,H] tensor) for the masked backward below.  Bit order inside
 *             a slot is the wave-ballot order of the kernel ([chunk][x,y,z,w][lane]); the three
 *             kernels agree on it, it is not meant to be read elsewhere.
 *   bwd_dst : dA[i,:] = inv_deg[i] * G[i,:] * #{j in N(i) : A[i,:]+B[j,:] > 0}
 *   bwd_src : dB[j,:] = sum_{i : j in N(i)} inv_deg[i] * G[i,:] * [A[i,:]+B[j,:] > 0]
 *             (source CSR; the backward scatter-add becomes a gather, no atomics)
 */
int stin_edge_relu_mean_fwd_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                const int32_t* rowptr, const int32_t* col, int64_t N, int H,
                                float* out, int64_t ldo, int indicator, uint32_t* mask, stin_stream_t stream);
/* Backward from the saved mask instead of recomputing it (H % 128 == 0):
 *   bwd_dst_mask: dA[i,:] = inv_deg[i] * G[i,:] * popcount over the in-edge slots of i   (streams H/8 bytes
 *                 per edge instead of gathering a B row)
 *   bwd_src_mask: dB[j,:] = sum_{out-edge slots s: j->i} w_src[s] * G[i,:] * mask[xslot[s]] (gathers G rows and
 *                 mask words: half the bytes of the recompute form) */
int stin_edge_relu_mean_bwd_dst_mask_f32(const float* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr,
                                         int64_t N, int H, float* dA, int64_t ldda, stin_stream_t stream);
int stin_edge_relu_mean_bwd_src_mask_f32(const float* G, int64_t ldg, const float* w_src, const uint32_t* mask,
                                         const int32_t* rowptr_src, const int32_t* col_src, const int32_t* xslot,
                                         int64_t N, int H, float* dB, int64_t lddb, stin_stream_t stream);
/* bwd_dst_mask and bwd_src_mask in ONE launch (the dB blocks first, then the dA blocks); results bit-identical to the two
 * calls above.  Optional rider (copy_src != NULL): copy_dst[i, :C_copy] = copy_src[i, :C_copy] for all N rows in the same
 * launch (C_copy <= H, multiple of 4 (bf16: 8), 16-byte aligned rows) - the block backward's dY[:, 2H:] = g. */
int stin_edge_relu_mean_bwd_mask_f32(const float* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst,
                                     const float* w_src, const int32_t* rowptr_src, const int32_t* col_src,
                                     const int32_t* xslot, int64_t N, int H, float* dA, int64_t ldda, float* dB,
                                     int64_t lddb, const float* copy_src, int64_t ld_copy_src, float* copy_dst,
                                     int64_t ld_copy_dst, int C_copy, stin_stream_t stream);
/* Translation-invariant blocks in the COMPACT layout (round 6; STIN_TI_COMPACT below).  The reference's message is
 * nn(x_j - x_i) (models/modules/edge_conv_translation_invariance.py:20-22): W1 (x_j - x_i) + b1 = A_i + B_j with B = x W1^T and
 * A_i = b1 - B_i.  A is therefore not a GEMM output at all:
 *   fwd_ti:      out / mask exactly as stin_edge_relu_mean_fwd_f32 given A_i = fl(b1 - B_i), formed per row (b1 [H] or NULL = zeros);
 *                a GEMM output A = x (-W1)^T + b1 differs from that by <= 1 ulp of the matrix-core accumulator (the MFMA adder is not
 *                symmetric under negation) - both are fp32 evaluations of the same expression;
 *   bwd_mask_ti: D [N, H] = dB - dA (both halves of stin_edge_relu_mean_bwd_mask_f32 for the same row, one rounding for the
 *                difference) - the gradient w.r.t. B in the compact layout - plus colsum [colsum_rows][H]: per-workgroup column sums
 *                of dA in a fixed order (db1 = their sum; sum_i D_i itself is ~ 0).  colsum_rows >= stin_edge_bwd_ti_colsum_rows(N, H).
 *                Same optional row-copy rider as the pair launch.  fp32 rows, H in {128, 256, 512, 1024, 2048}. */
int stin_edge_relu_mean_fwd_ti_f32(const float* b1, const float* B, int64_t ldb, const int32_t* rowptr, const int32_t* col,
                                   int64_t N, int H, float* out, int64_t ldo, int indicator, uint32_t* mask, stin_stream_t stream);
int64_t stin_edge_bwd_ti_colsum_rows(int64_t N, int H);
/* db1 [H] = the sum of the `rows` rows of colsum, in the order stin_edgeconv_wgrad_ti's finalize launch folds them (same bits) */
int stin_edge_bwd_ti_colsum_fold_f32(const float* colsum, int64_t rows, int H, float* db1, stin_stream_t stream);
int stin_edge_relu_mean_bwd_mask_ti_f32(const float* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst,
                                        const float* w_src, const int32_t* rowptr_src, const int32_t* col_src,
                                        const int32_t* xslot, int64_t N, int H, float* D, int64_t ldd, const float* copy_src,
                                        int64_t ld_copy_src, float* copy_dst, int64_t ld_copy_dst, int C_copy, float* colsum,
                                        int64_t colsum_rows, stin_stream_t stream);
int stin_edge_relu_mean_bwd_dst_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                    const float* G, int64_t ldg, const int32_t* rowptr,
                                    const int32_t* col, int64_t N, int H, float* dA, int64_t ldda,
                                    stin_stream_t stream);
int stin_edge_relu_mean_bwd_src_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                    const float* G, int64_t ldg, const float* inv_deg,
                                    const int32_t* rowptr_src, const int32_t* col_src, int64_t N, int H,
                                    float* dB, int64_t lddb, stin_stream_t stream);

/* --------------------------------------------------------------- pool / unpool --
 * Max pool over the children CSR of each coarse vertex with torch_scatter's CPU
 * rule: strict '>' walking children in ascending fine-vertex order => the FIRST
 * maximum wins ties; empty clusters give value 0 and arg = -1
 * (models/surfacetextureinpaintingnet.py:386).  Backward routes g to arg only.
 */
int stin_pool_max_fwd_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                          int64_t n_coarse, int C, float* out, int64_t ldo, int32_t* arg,
                          stin_stream_t stream);
int stin_pool_max_bwd_f32(const float* g, int64_t ldg, const int32_t* arg, const int32_t* trace,
                          int64_t n_fine, int C, float* gx, int64_t ldgx, stin_stream_t stream);
/* out[v, :] = src[idx[v], :] * (row_scale ? row_scale[idx[v]] : 1): the unpool gather
 * `x[traces]` (models/surfacetextureinpaintingnet.py:390-391) and the mean-pool backward. */
int stin_gather_rows_f32(const float* src, int64_t ld_src, const int32_t* idx, const float* row_scale,
                         int64_t n_out, int C, float* out, int64_t ldo, stin_stream_t stream);
/* out[e, :] = a[idx_a[e], :] + b[idx_b[e], :]: the per-edge pre-activation Lin1([x_i ; x_j - x_i]) = A[dst] + B[src]
 * of the BatchNorm edge MLP (models/modules/edge_conv_filter.py:34-44), one pass instead of two gathers and an add. */
int stin_gather_add_rows_f32(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb,
                             const int32_t* idx_b, int64_t n_out, int C, float* out, int64_t ldo, stin_stream_t stream);
/* (round 5) the same pass WITH the first stage of the column moments of its output - the BatchNorm1d over the E edge rows that
 * follows it (edge_conv_filter.py:36-38): partial [groups][2][C] doubles (per block: sum, sum of squares), groups =
 * stin_gather_add_rows_stats_groups(N, C) (0: not supported for this shape); second stage stin_moments_final_f32.  Saves the
 * separate moments pass over the [E, C] matrix just written. */
int64_t stin_segment_mean_stats_groups(int64_t N, int C);
/* segment MEAN of the rows src[col[e]] over the CSR slots of every output row (stin_segment_sum_f32(mean)'s result, bit for bit)
 * AND partial [groups][2][C] = per-block column sums of the visited source rows and of their squares: the batch statistics of all
 * E edge rows for scatter_mean(BatchNorm1d(m)) (edge_conv_filter.py:40-44 + aggr='mean') from the pass that aggregates them. */
int stin_segment_mean_stats_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* col, int64_t N, int C,
                                float* out, int64_t ldo, double* partial, size_t partial_bytes, stin_stream_t stream);
int64_t stin_gather_add_rows_stats_groups(int64_t N, int C);
int stin_gather_add_rows_stats_f32(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb,
                                   const int32_t* idx_b, int64_t N, int C, float* out, int64_t ldo, double* partial,
                                   size_t partial_bytes, stin_stream_t stream);
/* Per-level graph-id vector (int64, bit-exact): scatter_max(batch, trace) and
 * batch.index_select(0, trace) (models/surfacetextureinpaintingnet.py:421-422,:446-447). */
int stin_batch_pool_i64(const int64_t* batch, const int32_t* rowptr, const int32_t* col, int64_t n_coarse,
                        int64_t* out, stin_stream_t stream);
int stin_gather_i64(const int64_t* src, const int32_t* idx, int64_t n_out, int64_t* out,
                    stin_stream_t stream);
/* Row ids of a batched FastInstanceNorm level in one pass (round 4; was five framework launches per level and step): gid[r] =
 * (int32) batch[r], and - sid != NULL - sid[r] = the reference's linspace slice of row r = the number of boundaries
 * ptr_sum[1 .. B] that are <= r (models/modules/fastinstancenorm.py:53-82 sums over torch.linspace(0, N, B + 1) slices;
 * ptr_sum: device int32 [B + 1]).  B <= 8192. */
int stin_norm_group_ids_i64(const int64_t* batch, const int32_t* ptr_sum, int B, int64_t n_rows, int32_t* gid, int32_t* sid,
                            stin_stream_t stream);

/* ----------------------------------------------- norm statistics and epilogues --
 * Column reductions over contiguous row ranges ptr[b]..ptr[b+1] (ptr == NULL: one
 * range [0, N)), fp64 accumulation in a fixed order, float results out[b, c]:
 *   STIN_RED_SUM     : sum x
 *   STIN_RED_CSQ     : sum (x - mean[gid])^2
 *   STIN_RED_DOT_ELU : out0 = sum dY * xc, out1 = sum dY  with xc = x - mean[gid],
 *                      dY = gout * ELU'(xc * rstd[gid])
 *   STIN_RED_COEF_XC : sum coef[sid] * (x - mean[gid])
 *   STIN_RED_MOMENTS : mean and rstd of each range in ONE pass over x (needs inv_cnt; ignores post)
 * `gid`/`sid` are per-row int32 group ids (NULL = 0).  They implement
 * FastInstanceNorm (models/modules/fastinstancenorm.py:42-107) including its
 * linspace-slice quirk (sums over `ptr` slices, centring through `gid`).
 * `post` finalises out0 (and out1) from the fp64 sums with the per-range factor
 * inv_cnt[b] (device float[B]; required unless post == STIN_POST_NONE).
 * workspace: stin_colreduce_workspace_bytes(C, B) bytes.
 */
#define STIN_RED_SUM 0
#define STIN_RED_CSQ 1
#define STIN_RED_DOT_ELU 2
#define STIN_RED_COEF_XC 3
#define STIN_RED_MOMENTS 4          /* one pass: out0 = mean, out1 = 1/sqrt(biased var + eps) from fp64 sum x, sum x^2 */
#define STIN_RED_DOT_BN 5           /* BatchNorm-with-affine backward sums (one range): d = gout (* [gamma n + beta > 0] with
                                       STIN_RED_DOT_BN_RELU), n = (x - mean) rstd; out0 = sum d n (= dgamma), out1 = sum d
                                       (= dbeta); coef = [gamma ; beta] as 2 x C floats                          */
#define STIN_RED_DOT_BN_RELU 6
#define STIN_POST_NONE 0            /* out = s                                  */
#define STIN_POST_SCALE 1           /* out = s * inv_cnt[b]            (mean)    */
#define STIN_POST_RSTD 2            /* out = 1/sqrt(s * inv_cnt[b] + eps)        */
#define STIN_POST_NORM_COEF 3       /* DOT_ELU only, ranges == graphs: out0 = -rstd^3 s0 inv_cnt, out1 = -rstd s1 inv_cnt:
                                       the k / m coefficients of stin_norm_act_bwd_* in the same launch (stin_norm_bwd_coef_f32) */
size_t stin_colreduce_workspace_bytes(int C, int B);
int stin_colreduce_f32(int mode, const float* x, int64_t ldx, const float* gout, int64_t ldg, int64_t N,
                       int C, const int32_t* ptr, int B, const int32_t* gid, const int32_t* sid,
                       const float* mean, const float* rstd, const float* coef, int post,
                       const float* inv_cnt, float eps, float* out0, float* out1,
                       void* workspace, size_t workspace_bytes, stin_stream_t stream);
/* y = res + ELU((x - mean[gid]) * rstd[gid])    (res may be NULL; act: 1 = ELU, 0 = none)
 * GraphResnetBlock epilogue (models/surfacetextureinpaintingnet.py:510-521) and the
 * tail norm+ELU (:465-466). */
int stin_norm_act_res_fwd_f32(const float* x, int64_t ldx, const float* mean, const float* rstd,
                              const int32_t* gid, const float* res, int64_t ldres, int64_t N, int C, int act,
                              float* y, int64_t ldy, stin_stream_t stream);
/* dx = a[gid] * dY + k[sid] * (x - mean[gid]) + m[sid],  dY = gout * act'((x-mean[gid])*rstd[gid]) */
int stin_norm_act_bwd_f32(const float* x, int64_t ldx, const float* gout, int64_t ldg, const float* mean,
                          const float* rstd, const float* a, const float* k, const float* m,
                          const int32_t* gid, const int32_t* sid, int64_t N, int C, int act, float* dx,
                          int64_t lddx, stin_stream_t stream);

/* ------------------------------------------------------------ per-vertex GEMMs --
 * Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) GEMMs for the tall-skinny per-VERTEX products the
 * restructure leaves (reference: the per-EDGE aten::addmm of Lin1/Lin2 in
 * models/modules/edge_conv_filter.py:47-52, the shortcut Linear at
 * models/surfacetextureinpaintingnet.py:515-516 and the tail Linears :464,:467).
 *   nt: C[M, Nc] = A[M, K] . W[Nc, K]^T + bias[Nc] * (row_mask ? row_mask[m * ld_mask] : 1) (+ residual[M, Nc])
 *       forward, and dgrad with W := W^T.  row_mask = the [deg > 0] indicator gives PyG's
 *       "no in-edges -> exactly 0" (bias only where a vertex aggregated something).
 *       `precision` picks the matrix-core path: exact fp32 MFMA, or the bf16 cores (16x the fp32
 *       MFMA rate) on operands split on the fly into 2 or 3 bf16 pieces with fp32 accumulation.
 *   tn: dW[Nc, K (+1)] = G[M, Nc]^T . [X[M, K] | w]       weight gradient; with ones_column the
 *       extra last column is the bias gradient sum_m w[m] G[m, :] (w = row_weight or 1).
 *       Split over M into slabs that are summed in a fixed order (deterministic, no atomics).
 */
#define STIN_GEMM_F32 0      /* v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain                          */
#define STIN_GEMM_BF16X3 2   /* fp32 operands split into 2 bf16 pieces, 3 bf16 MFMAs, ~2^-17 / product */
#define STIN_GEMM_BF16X6 3   /* 3 pieces (exact split), 6 bf16 MFMAs, ~2^-22 / product                 */
#define STIN_GEMM_F16X3 4    /* nt only: 2 fp16 pieces (11 bits each), 3 fp16 MFMAs, ~2^-22 / product for
                                |a| >= 2^-6, |w| >= 2^-9 (absolute 2^-28 / 2^-31 below: made for normalised
                                activations); A, W pre-scaled by 2^3, 2^6 internally (exact); needs
                                |A| < 8188, |W| < 1023 (outside: inf/NaN, never a silently wrong value)   */
#define STIN_GEMM_W_BF16 0x200     /* stin_gemm_nt_bf16 flag / stin_edgeconv_pack_f32 mode: the weight operand holds bf16 [Nc][K]
                                       (ldw in bf16 elements, K % 8 == 0, 16-byte aligned rows) instead of fp32 */
#define STIN_GEMM_W_PRESPLIT 0x100 /* nt, OR-ed into BF16X3 / F16X3: W already holds its two 16-bit pieces, per 4-wide
                                      k-group [hi x 4 | lo x 4] in the 16 bytes of the fp32 values it replaces (same
                                      shape / ld / footprint; made by stin_gemm_split_weights_f32 or the pack kernel) -
                                      the weight split is then done once per step instead of once per block    */
#define STIN_GEMM_W_FRAG 0x400     /* nt, OR-ed into PRESPLIT: the pre-split W is stored in MFMA FRAGMENT order where the shape
                                      takes it (stin_gemm_w_is_frag(Nc, K): the shapes of the resident-strip kernel - K % 64 == 0,
                                      Nc % 32 == 0, 128 <= K <= 256, Nc >= 320 - and of the all-columns kernel - Nc = 256 with
                                      K >= 512 or Nc = 128 with K >= 256, K % 64 == 0; other shapes keep the k-group form above,
                                      so the flag may be set unconditionally): for the 32-column tile t
                                      and the 16-wide k-step s, 2 KB at byte ((t * K/16 + s) * 64 + lane) * 32 hold lane
                                      (k-half h, column r) = h * 32 + r's [hi x 8 | lo x 8] of k = 16 s + 8 h .. + 7 - what the
                                      resident-strip kernel loads straight into its B fragments (same footprint as fp32; ldw
                                      is ignored).  Producer and consumer must agree: pass the same flag to
                                      stin_gemm_split_weights_f32 / stin_edgeconv_pack_f32 and to stin_gemm_nt_f32             */
int stin_gemm_w_is_frag(int Nc, int K);
int stin_gemm_split_weights_f32(const float* W, int64_t ldw, int Nc, int K, int precision, float* out, int64_t ldo,
                                stin_stream_t stream);
int stin_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                     const float* row_mask, int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M,
                     int Nc, int K, float* C, int64_t ldc, int precision, stin_stream_t stream);
size_t stin_gemm_tn_workspace_bytes(int64_t M, int Nc, int K, int ones_column);
int stin_gemm_tn_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int Nc, int K,
                     int ones_column, const float* row_weight, int64_t ld_weight, float* dW, int64_t lddw,
                     int precision, void* workspace, size_t workspace_bytes, stin_stream_t stream);

/* The same product with the weight gradient dW [Nc, K] (row pitch lddw >= K) and the bias gradient db [Nc] as SEPARATE
 * destinations (ones column implied): what torch.nn.Linear's backward hands to weight.grad / bias.grad
 * (models/surfacetextureinpaintingnet.py:461-466 `final_linear1`, and every generic nn.Linear of the build) without an
 * [Nc, K + 1] intermediate and the two slicing copies behind it. */
int stin_gemm_tn_wb_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int Nc, int K,
                        const float* row_weight, int64_t ld_weight, float* dW, int64_t lddw, float* db, int precision,
                        void* workspace, size_t workspace_bytes, stin_stream_t stream);

/* The network's last layer in one launch per direction: y = tanh(x W^T + b), W [Nc, K], Nc <= 4 output (colour) channels,
 * K % 4 == 0, K <= 256 (STIN_E_UNSUPPORTED otherwise: callers fall back to stin_gemm_nt_* + a tanh) - replaces
 * `torch.tanh(self.final_linear2(vertex_features))` of models/surfacetextureinpaintingnet.py:470-471 and its autograd backward.
 *   fwd: x [N, K] rows (ldx), y [N, Nc] fp32 contiguous.  Exact fp32 products, per-row sums in a fixed order.
 *   bwd: g = dL/dy [N, Nc] and y [N, Nc] fp32 contiguous; dz = g (1 - y^2); dx [N, K] = dz W (NULL: not wanted),
 *        dW [Nc, K] = dz^T x, db [Nc] = column sums of dz (NULL: no bias) - block partials folded in a fixed order by the last
 *        block to arrive (deterministic, one launch); workspace >= stin_linear_tanh_bwd_workspace_bytes.
 * The bf16 variants read / write bf16 activation rows (x, dx); W, b, g, y and the gradients stay fp32. */
size_t stin_linear_tanh_bwd_workspace_bytes(int64_t N, int K, int Nc);
int stin_linear_tanh_fwd_f32(const float* x, int64_t ldx, const float* W, const float* b, int64_t N, int K, int Nc, float* y,
                             stin_stream_t stream);
int stin_linear_tanh_bwd_f32(const float* g, const float* y, const float* x, int64_t ldx, const float* W, int64_t N, int K, int Nc,
                             float* dx, int64_t lddx, float* dW, float* db, void* workspace, size_t workspace_bytes,
                             stin_stream_t stream);

/* dst[n, c] += alpha * src[n, c] * [rowptr[n + 1] > rowptr[n]] for c in [c0, c1), in place: the translation-invariant SAGE
 * message of models/modules/sage_conv_filter.py:87-90 (x_j[:, 3:9] -= x_i[:, 3:9]) under the mean aggregation, applied to the
 * aggregated rows (alpha = -1, rowptr = destination CSR) and, in the backward pass, to the input gradient. */
int stin_cols_axpy_rowmask_f32(float* dst, int64_t ldd, const float* src, int64_t lds, const int32_t* rowptr, int64_t N, int c0,
                               int c1, float alpha, stin_stream_t stream);
int stin_cols_axpy_rowmask_bf16(stin_bf16_t* dst, int64_t ldd, const stin_bf16_t* src, int64_t lds, const int32_t* rowptr,
                                int64_t N, int c0, int c1, float alpha, stin_stream_t stream);

/* out [N, Cp] (contiguous) = [x [N, Cin] (row pitch ldx) | zeros]: the network input - 10 channels per vertex,
 * datasets/scannetcolorgraph_dataloader.py:83-156 - padded to the 16-byte rows the first block's GEMM reads. */
int stin_pad_rows_f32(const float* x, int64_t ldx, int64_t N, int Cin, int Cp, float* out, stin_stream_t stream);
int stin_pad_rows_bf16(const stin_bf16_t* x, int64_t ldx, int64_t N, int Cin, int Cp, stin_bf16_t* out, stin_stream_t stream);

/* stin_gemm_nt_f32 (pre-split fragment-order weights) plus the FIRST stage of the instance-norm + ELU backward statistics of the
 * layer whose output gradient this product IS - inside the block backward the input gradient of block k (dY Wcat, + g for an
 * identity shortcut; models/surfacetextureinpaintingnet.py:507-521) is the output gradient of block k - 1, whose
 * FastInstanceNorm backward (models/modules/fastinstancenorm.py:42-107 through autograd) starts with two column sums over
 * (nx = that block's pre-norm rows, g).  partial [groups][2][Nc] doubles: per row group sum of dy (nx - nmean) and of dy with
 * dy = C ELU'((nx - nmean) nrstd); groups = stin_gemm_nt_dotelu_groups (0: shape / precision not served - keep
 * stin_colreduce_f32(STIN_RED_DOT_ELU)).  stin_norm_coef_from_partials_f32 folds them in a fixed order into the coefficients
 * k, m [C] of stin_norm_act_bwd_f32 for ONE graph (inv_cnt[0] = 1 / rows): the tail of stin_colreduce_f32(.., STIN_POST_NORM_COEF). */
int64_t stin_gemm_nt_dotelu_groups(int64_t M, int Nc, int K, int precision);
int stin_gemm_nt_dotelu_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* residual,
                            int64_t ld_res, int64_t M, int Nc, int K, float* C, int64_t ldc, int precision, const float* nx,
                            int64_t ld_nx, const float* nmean, const float* nrstd, double* partial, size_t partial_bytes,
                            stin_stream_t stream);
int stin_norm_coef_from_partials_f32(const double* partial, int64_t groups, int C, const float* rstd, const float* inv_cnt,
                                     float* k, float* m, stin_stream_t stream);

/* BatchNorm1d-with-affine over ALL rows, optionally followed by ReLU, for the per-edge MLP of SingleConvMeshNet
 * (models/modules/edge_conv_filter.py:34-44: Lin - BatchNorm1d - ReLU - Lin - BatchNorm1d over the E edge rows):
 *   fwd: y = act(gamma * (x - mean) * rstd + beta)            mean / rstd [C] from stin_colreduce_f32(STIN_RED_MOMENTS)
 *   bwd: dx = rstd * gamma * (d - Q * inv_n - n * P * inv_n)   d = gout * act'(.), n = (x - mean) * rstd,
 *        P = sum d n (= dgamma), Q = sum d (= dbeta) from stin_colreduce_f32(STIN_RED_DOT_BN[_RELU]).
 * act: 0 = none, 1 = ReLU.
 */
int stin_bn_act_fwd_f32(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, int64_t N, int C, int act, float* y, int64_t ldy, stin_stream_t stream);
int stin_bn_act_bwd_f32(const float* x, int64_t ldx, const float* gout, int64_t ldg, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const float* P, const float* Q, float inv_n, int64_t N,
                        int C, int act, float* dx, int64_t lddx, stin_stream_t stream);

/* BatchNorm1d over the E edge rows followed by aggr='mean' (the second norm of that edge MLP): the affine map commutes with
 * the mean, so the forward normalises the N aggregated rows (stin_segment_sum_f32(mean) then stin_bn_act_fwd_f32, rows
 * without in-edges zeroed); this is the joint backward - the gradient of the raw edge rows m [E, C] from the VERTEX
 * gradient g [N, C] (zero on rows without in-edges):
 *   dm[e] = gamma rstd (g[dst e] * inv_deg[dst e] - Q inv_e - nhat_e P inv_e),  nhat_e = (m[e] - mean) rstd,
 * with P = sum_i g_i nhat(agg_i), Q = sum_i g_i from stin_colreduce_f32(STIN_RED_DOT_BN) over the N aggregated rows. */
int stin_bn_mean_bwd_f32(const float* m, int64_t ldm, const float* g, int64_t ldg, const int32_t* dst, const float* inv_deg,
                         const float* mean, const float* rstd, const float* gamma, const float* P, const float* Q,
                         float inv_e, int64_t E, int C, float* dm, int64_t lddm, stin_stream_t stream);

/* nn.BatchNorm1d's running-statistics update from the batch statistics (mean, rstd = 1/sqrt(biased var + eps)) in one
 * launch: running_mean <- (1 - m) running_mean + m mean; running_var <- (1 - m) running_var + m * unbias * max(1/rstd^2 -
 * eps, 0), unbias = n / (n - 1). */
int stin_bn_running_stats_f32(const float* mean, const float* rstd, int C, float eps, float unbias, float momentum,
                              float* running_mean, float* running_var, stin_stream_t stream);

/* Round 5: the small passes of one fused EdgeConv(BN) layer of SingleConvMeshNet (singleconvmeshnet.EdgeConvBNLayerFn; reference
 * models/modules/edge_conv_filter.py:34-44 inside models/singleconvmeshnet.py:37-66 `ResBlock`), which were framework kernels:
 * stin_scmn_pack_f32: W1 [h2, 2 cin] (trans_inv: [h2, cin]), W2 [cout, h2], the two BatchNorm1d affine pairs ->
 *   wcat [2 h2, cin] = [Wa - Wb ; Wb] (trans_inv: [-W1 ; W1]), wcatT [cin, 2 h2], w2T [h2, cout],
 *   gb1 [2, h2] = [gamma1 ; beta1], gb2 [2, cout] (the `coef` operand of stin_colreduce_f32(STIN_RED_DOT_BN*)).
 * stin_scmn_unpack_f32: dwcat [2 h2, cin] -> dW1 in the reference layout (d/dWa = top, d/dWb = bottom - top; trans_inv: bottom - top).
 * stin_bn_affine_res_fwd_f32: y = [relu](res + [rowptr[r + 1] > rowptr[r]] (gamma ((x - mean) rstd) + beta)) over N rows - the
 *   BatchNorm affine of the aggregated rows, the "vertex has an in-edge" mask, the ResBlock's `x + f(x)` and its ReLU
 *   (singleconvmeshnet.py:60-66) in one pass; rowptr / res may be NULL.
 * stin_relu_mask_bwd_f32: g_eff = g [y > 0] (relu != 0; else g), g_in = g_eff [row has an in-edge]; g_eff may be NULL; both
 *   outputs are [N, C] with pitch C. */
/* The per-EDGE Linear of that layer with the BatchNorm1d + ReLU of its input applied while the operand rows are staged
 * (edge_conv_filter.py:36-40: Lin -> BN -> ReLU -> Lin): the normalised [E, 2 cout] matrix is never written.
 *   stin_gemm_nt_bn_f32: C [M, Nc] = relu(gamma ((A - mean) rstd) + beta) W^T, W plain fp32 [Nc, K] (no pre-split flags);
 *   stin_gemm_tn_bn_f32: dW [Nc, K] = G^T relu(gamma ((X - mean) rstd) + beta)  (the weight gradient from the pre-norm rows).
 * mean / rstd / gamma / beta: [K] fp32.  The transform is evaluated as relu(v s + t) with s = gamma rstd, t = beta - mean s
 * (3 operations per element on the vector-ALU-bound staging threads; equals stin_bn_act_fwd_f32(act = 1) to fp32 rounding,
 * ~1e-7 relative; both entry points use the same form).  precision as stin_gemm_nt_f32 / stin_gemm_tn_f32; workspace = stin_gemm_tn_workspace_bytes(M, Nc, K, 0). */
int stin_gemm_nt_bn_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, int64_t M, int Nc, int K, float* C, int64_t ldc, int precision,
                        stin_stream_t stream);
int stin_gemm_tn_bn_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, int64_t M, int Nc, int K, float* dW, int64_t lddw, int precision,
                        void* workspace, size_t workspace_bytes, stin_stream_t stream);
/* (round 5) The per-EDGE input-gradient product of that layer, dh = A W^T with A = dm [E, cout], W = W2^T [2 cout, cout], is only
 * ever the output gradient of BatchNorm1d + ReLU over the pre-norm rows X [E, 2 cout] (edge_conv_filter.py:36-38).  Instead of
 * storing it, reducing it (stin_colreduce_f32 DOT_BN_RELU) and rewriting it (stin_bn_act_bwd_f32) the product runs twice:
 *   stin_gemm_nt_bn_bwd_stats_f32: nothing stored; sums [2][Nc] floats = P | Q (P = sum d nhat, Q = sum d, d = dh [gamma nhat + beta > 0],
 *     nhat = (X - mean) rstd: the gradients of gamma | beta) through partial [groups][2][Nc] doubles (fixed-order fp64 fold);
 *   stin_gemm_nt_bn_bwd_apply_f32: dx [M, Nc] = rstd gamma (d - Q inv_n - nhat P inv_n) - stin_bn_act_bwd_f32's expression.
 * groups = stin_gemm_nt_bn_bwd_groups(M, Nc, K, precision); 0: not served (K not 64 or 128, a pre-split precision flag,
 * STIN_NT_BNBWD=0) - keep the three-launch route.  Rows of A / W 16-byte aligned.  The accumulators equal stin_gemm_nt_f32's
 * bit for bit (same tiles, k order, MFMA order); the sums differ from the reduction kernel's only by fp64 summation order. */
/* The streaming-rows kernel behind them, as the plain product (mean == NULL) or with stin_gemm_nt_bn_f32's operand transform:
 * persistent blocks whose waves own 32-row tiles staged through wave-private LDS, the weight slice split once per block - for
 * M >> 1e5 rows with K = 64 .. 512 (a multiple of 64) and plain fp32 weights; STIN_E_UNSUPPORTED otherwise.  Bit-identical to the tiled
 * kernels (same k and MFMA order).  stin_gemm_nt_f32 (no bias / mask / residual) and stin_gemm_nt_bn_f32 route M >= 65 536 rows
 * here (STIN_NT_STREAM=0: never). */
int stin_gemm_nt_stream_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, int64_t M, int Nc, int K, float* C, int64_t ldc, int precision,
                            stin_stream_t stream);
int64_t stin_gemm_nt_bn_bwd_groups(int64_t M, int Nc, int K, int precision);
int stin_gemm_nt_bn_bwd_stats_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* X, int64_t ldx,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M, int Nc,
                                  int K, int precision, double* partial, size_t partial_bytes, float* sums, stin_stream_t stream);
int stin_gemm_nt_bn_bwd_apply_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* X, int64_t ldx,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta, const float* sums,
                                  float inv_n, int64_t M, int Nc, int K, float* dx, int64_t lddx, int precision,
                                  stin_stream_t stream);
int stin_scmn_pack_f32(const float* W1, const float* W2, const float* gamma1, const float* beta1, const float* gamma2,
                       const float* beta2, int cin, int h2, int cout, int trans_inv, float* wcat, float* wcatT, float* w2T,
                       float* gb1, float* gb2, stin_stream_t stream);
int stin_scmn_unpack_f32(const float* dwcat, int cin, int h2, int trans_inv, float* dW1, stin_stream_t stream);
int stin_bn_affine_res_fwd_f32(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const int32_t* rowptr, const float* res, int64_t ldres, int64_t N, int C,
                               int relu, float* y, int64_t ldy, stin_stream_t stream);
int stin_relu_mask_bwd_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const int32_t* rowptr, int64_t N, int C,
                           int relu, float* g_eff, float* g_in, stin_stream_t stream);
/* out[r] = [ skip[r, :cs] | coarse[trace[r], :cu] ], r < N: the decoder's torch.cat((skip, x[trace]), -1) with the unpool gather writing
 * straight into its half (models/singleconvmeshnet.py:146-150); trace values must lie in [0, rows of coarse) (validated when the
 * pooling map is built).  Backward needs no kernel of its own: the skip gradient is the left column block of the incoming gradient,
 * the coarse gradient stin_segment_sum_f32 over the right one. */
int stin_concat_unpool_f32(const float* skip, int64_t ld_skip, const float* coarse, int64_t ld_coarse, const int32_t* trace, int64_t N,
                           int cs, int cu, float* out, int64_t ldo, stin_stream_t stream);

/* ------------------------------------------------------- parameter-side helpers --
 * pack: the reference-layout EdgeConv parameters (first_filter.nn.0.{weight,bias} = W1 [H, 2Cin]
 * (or [H, Cin] for EdgeConvTransInv), first_filter.nn.2.weight = W2 [Cout, H], shortcut.{weight,bias})
 * -> the per-vertex GEMM operands  wcat [Yw, Cin] = [Wa-Wb ; Wb ; Ws] (trans_inv: [-W1 ; W1 ; Ws]),
 * bcat [Yw] = [b1 ; 0 ; bs], wcatT = wcat^T, w2T = W2^T; Yw = 2H (+ Cout).  The inner dimension may be zero-padded
 * from Cin to Cp (a multiple of 4) so that a 10-channel network input still takes the 16-byte GEMM paths.
 * trans_inv = STIN_TI_COMPACT (2, round 6; every entry point that takes trans_inv): the translation-invariant filter with ONLY
 * B = x W1^T materialised - wcat = [W1 ; Ws], bcat = [0 ; bs], Yw = H (+ Cout); A_i = b1 - B_i is formed by the edge stage
 * (stin_edge_relu_mean_fwd_ti_f32: mode 1's values up to one ulp of A) and the backward pass carries D = dB - dA in H columns
 * (stin_edge_relu_mean_bwd_mask_ti_f32) - half the first Linear's GEMM work in every direction.  fp32 rows with a saved-mask
 * width H; unpack then takes dW1 = rows [0, H) as they are and leaves db1 to the caller (the column sums of dA).
 * unpack: dwb [Yw, Cin+1] (gemm_tn output: weight grad | bias grad) -> dW1, db1, dWs, dbs; and
 *   dw2b [Cout, H+1] (optional) -> contiguous dW2 [Cout, H], db2 [Cout].
 * (models/modules/edge_conv_filter.py:46-52, models/surfacetextureinpaintingnet.py:505-506)
 * norm_bwd_coef: k = -rstd^3 T1 inv_cnt, m = -rstd S0 inv_cnt for stin_norm_act_bwd_f32.
 * fwd_split / bwd_split (0, STIN_GEMM_F16X3 or STIN_GEMM_BF16X3, optionally | STIN_GEMM_W_FRAG): write the forward
 * operands (wcat, and a copy w2s [Cout, H] of W2) / the backward operands (wcatT, w2T) directly in the
 * STIN_GEMM_W_PRESPLIT form of that precision (each operand in fragment order where its shape allows, see W_FRAG).
 */
/* pack_many: the pack of EVERY block of a network in one launch.  `jobs_device` = n_jobs records in DEVICE memory (the
 * arguments of stin_edgeconv_pack_f32 as a struct; written once per model - the pointers do not change from step to step),
 * max_elems = max over the jobs of Yw * Cp + H * Cout.  The same argument rules as stin_edgeconv_pack_f32 apply per job
 * (not re-checked on the device).  stin_edgeconv_block_fwd then takes STIN_BLOCK_PACKED in fwd_split. */
typedef struct stin_pack_job {
    const float *W1, *b1, *Ws, *bs, *W2;
    float *wcat, *bcat, *wcatT, *w2T, *w2s;
    int Cin, Cp, H, Cout, has_shortcut, trans_inv, fwd_split, bwd_split;
} stin_pack_job_t;                                   /* 10 pointers + 8 ints = 112 bytes */
int stin_edgeconv_pack_many_f32(const stin_pack_job_t* jobs_device, int n_jobs, int64_t max_elems, stin_stream_t stream);
int stin_edgeconv_pack_f32(const float* W1, const float* b1, const float* Ws, const float* bs, const float* W2,
                           int Cin, int Cp, int H, int Cout, int has_shortcut, int trans_inv, float* wcat,
                           float* bcat, float* wcatT, float* w2T, float* w2s, int fwd_split, int bwd_split,
                           stin_stream_t stream);
int stin_edgeconv_unpack_grads_f32(const float* dwb, const float* dw2b, int Cin, int Cp, int H, int Cout,
                                   int has_shortcut, int trans_inv, float* dW1, float* db1, float* dWs, float* dbs,
                                   float* dW2, float* db2, stin_stream_t stream);
int stin_norm_bwd_coef_f32(const float* T1, const float* S0, const float* rstd, const float* inv_cnt, int B, int C,
                           float* k, float* m, stin_stream_t stream);
/* linspace-slice quirk of FastInstanceNorm (fastinstancenorm.py:53-82): m = -(rstd S0 + U) inv_cnt, where
 * U = stin_colreduce(STIN_RED_COEF_XC, coef = k) - the second coefficient of stin_norm_act_bwd_* when slices != graphs. */
int stin_norm_bwd_coef_m_quirk_f32(const float* S0, const float* U, const float* rstd, const float* inv_cnt, int B, int C,
                                   float* m, stin_stream_t stream);

/* ----------------------------------------------------------- train-step epilogues --
 * masked_l1_loss: the trainer's loss and its gradient in one pass
 * (trainers/inpainting3d_trainer.py:127-137): pred = mask > 0 ? out : color,
 * loss = mean |pred - color| * 0.99^mask (use_weight = trainer.use_mask_weighted_loss),
 * grad = dloss/dout.  fp64 block partials summed in a fixed order.  mask: int64 [N].
 * adam: torch.optim.Adam(amsgrad) on ONE flat parameter/gradient/state buffer (the reference's
 * optimizer, experiments/3d_inpainting/config/...json:95-102), step = 1-based update count; the hyper-parameters are
 * doubles (python floats in torch) and the bias corrections 1 - beta^step are formed in double on the host, as torch does.
 */
size_t stin_masked_l1_workspace_bytes(int64_t N, int C);
int stin_masked_l1_loss_f32(const float* out, const float* color, const int64_t* mask, int64_t N, int C,
                            int use_weight, float* loss, float* grad, void* workspace, size_t workspace_bytes,
                            stin_stream_t stream);
/* graph total variation, the per-step smoothness metric of the trainer (utils/metrics/graph_metrics.py:34-38,
 * trainers/inpainting3d_trainer.py:254-263): out[0] = sum_e sum_c |x[src_e, c] - x[dst_e, c]| / (N * C) over the
 * destination CSR the forward pass has built (rowptr_dst / col_dst), fp64 partial sums in a fixed order. */
size_t stin_total_variation_workspace_bytes(int64_t N);
int stin_total_variation_f32(const float* x, int64_t ldx, const int32_t* rowptr_dst, const int32_t* col_dst, int64_t N, int C,
                             float* out, void* workspace, size_t workspace_bytes, stin_stream_t stream);
/* graph Laplace of the same metrics module (utils/metrics/graph_metrics.py:6-16, GraphLaplaceOperator: propagate [1 | x] with
 * aggr = 'add', then prop[:, 1:] - prop[:, 0:1] * x): out[i, c] = sum_{j in N(i)} x[j, c] - deg_i x[i, c] over the destination CSR,
 * fp32 adds in edge order - the numbers of the reference's composition, bit for bit. */
int stin_graph_laplace_f32(const float* x, int64_t ldx, const int32_t* rowptr_dst, const int32_t* col_dst, int64_t N, int C,
                           float* out, int64_t ldo, stin_stream_t stream);
int stin_adam_f32(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step, int amsgrad, stin_stream_t stream);

/* GEMM + first stage of the instance-norm statistics of its output (the all-columns NT kernel's blocks own whole rows):
 * colstats [groups][2][Nc] doubles = per 64-row group the column sums of the stored values and of their squares;
 * groups = stin_gemm_nt_colstats_groups(M, Nc, K, precision) (0: shape / precision not supported - Nc = 128 / 256 in
 * fragment order only).  stin_moments_final_f32 turns them into mean / rstd of ONE instance (inv_cnt[0] = 1 / M).
 * Replaces GEMM2 + stin_colreduce_f32(MOMENTS) of a single-graph block (fastinstancenorm.py:44-98 on the conv output). */
int64_t stin_gemm_nt_colstats_groups(int64_t M, int Nc, int K, int precision);
int stin_gemm_nt_colstats_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                              const float* row_mask, int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M,
                              int Nc, int K, float* C, int64_t ldc, int precision, double* colstats, size_t colstats_bytes,
                              stin_stream_t stream);
int stin_moments_final_f32(const double* partial, int64_t groups, int C, const float* inv_cnt, float eps, float* mean,
                           float* rstd, stin_stream_t stream);
/* Round 5: the fold of per-row-group statistics partials INSIDE the elementwise launch (single graph, fp32 rows, C % 32 == 0,
 * 16-byte rows): every workgroup folds the partials of its own 32 columns with the arithmetic of stin_moments_final_f32 /
 * stin_norm_coef_from_partials_f32 and then runs stin_norm_act_res_fwd_f32 / stin_norm_act_bwd_f32's expression on its rows:
 * bit-identical to the two-launch route, one launch less on the critical path of a block.  Replaces the same reference lines as
 * those entry points (models/modules/fastinstancenorm.py:44-49 + the ELU / residual of surfacetextureinpaintingnet.py:507-521).
 * stin_norm_fold_rows: rows per workgroup the form would use for (N, C, groups), or 0 when it does not apply / does not pay (the
 * grid would read more fold bytes than half its elementwise traffic; STIN_NORM_FOLD=0).  The two launches return
 * STIN_E_UNSUPPORTED - before anything is enqueued - when the form does not apply: the caller then takes the two-launch route.
 * fwd: partial [groups][2][C] (sum, sum of squares) -> mean, rstd [C] written; y = ELU((x - mean) rstd) (+ res).
 * bwd: partial [groups][2][C] (sum of dy xc, sum of dy) with mean, rstd [C] given -> dx = rstd dy + k xc + m. */
int stin_norm_fold_rows(int64_t N, int C, int64_t groups);
int stin_norm_act_res_fwd_fold_f32(const double* partial, int64_t groups, const float* x, int64_t ldx, const float* res,
                                   int64_t ldres, const float* inv_cnt, float eps, int64_t N, int C, float* mean, float* rstd,
                                   float* y, int64_t ldy, stin_stream_t stream);
int stin_norm_act_bwd_fold_f32(const double* partial, int64_t groups, const float* x, int64_t ldx, const float* gout, int64_t ldg,
                               const float* mean, const float* rstd, const float* inv_cnt, int64_t N, int C, float* dx,
                               int64_t lddx, stin_stream_t stream);

/* --------------------------------------------------- bf16-storage variants of the path --
 * Same semantics, argument order and reference call sites as the *_f32 entry points above, on bf16 rows
 * (stin_bf16_t, see the typedef).  What stays fp32: instance-norm statistics (fp64 accumulation), inv_deg / w_src /
 * row_scale, the weight operand W and bias of the GEMMs, the weight gradients dW and every index plan.
 * Vector kernels only: C % 4 == 0, ld % 4 == 0, 8-byte aligned rows, else STIN_E_UNSUPPORTED; the edge stage
 * needs the saved ReLU mask (H in {128, 256, 512, 1024, 2048}) in backward.  With 16-byte aligned rows (ld % 8 == 0)
 * the edge stage gives every lane 8 channels (the same 16 bytes per lane and request as the fp32 kernels); the mask
 * kernels REQUIRE that alignment (STIN_E_ALIGN otherwise) because the lane geometry fixes the bit order inside a mask
 * slot: a bf16 mask is only meaningful to the bf16 backward kernels, an fp32 mask to the fp32 ones.
 * gemm_nt_bf16: A, row_mask, residual bf16; W, bias fp32 (W is rounded to bf16 while staged); one
 *   v_mfma_f32_32x32x16_bf16 per k-step, fp32 accumulate, bias/mask/residual added in fp32, one rounding;
 *   C is bf16 (c_is_f32 = 0) or fp32 (c_is_f32 = 1: the network's final [N, 3] output); OR-ing STIN_GEMM_W_BF16 into
 *   c_is_f32 declares W as a bf16 [Nc][K] operand (written by stin_edgeconv_pack_f32 in that mode).
 * gemm_tn_bf16: G, X, row_weight bf16 -> fp32 dW; workspace = stin_gemm_tn_workspace_bytes.
 */
int stin_segment_sum_bf16(const stin_bf16_t* src, int64_t ld_src, const int32_t* rowptr, const int32_t* col,
                          int64_t n_rows, int C, int mean, stin_bf16_t* out, int64_t ld_out, stin_stream_t stream);
int stin_edge_relu_mean_fwd_bf16(const stin_bf16_t* A, int64_t lda, const stin_bf16_t* B, int64_t ldb,
                                 const int32_t* rowptr, const int32_t* col, int64_t N, int H, stin_bf16_t* out,
                                 int64_t ldo, int indicator, uint32_t* mask, stin_stream_t stream);
int stin_edge_relu_mean_bwd_dst_mask_bf16(const stin_bf16_t* G, int64_t ldg, const uint32_t* mask,
                                          const int32_t* rowptr, int64_t N, int H, stin_bf16_t* dA, int64_t ldda,
                                          stin_stream_t stream);
int stin_edge_relu_mean_bwd_src_mask_bf16(const stin_bf16_t* G, int64_t ldg, const float* w_src, const uint32_t* mask,
                                          const int32_t* rowptr_src, const int32_t* col_src, const int32_t* xslot,
                                          int64_t N, int H, stin_bf16_t* dB, int64_t lddb, stin_stream_t stream);
int stin_edge_relu_mean_bwd_mask_bf16(const stin_bf16_t* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst,
                                      const float* w_src, const int32_t* rowptr_src, const int32_t* col_src,
                                      const int32_t* xslot, int64_t N, int H, stin_bf16_t* dA, int64_t ldda,
                                      stin_bf16_t* dB, int64_t lddb, const stin_bf16_t* copy_src, int64_t ld_copy_src,
                                      stin_bf16_t* copy_dst, int64_t ld_copy_dst, int C_copy, stin_stream_t stream);
int stin_pool_max_fwd_bf16(const stin_bf16_t* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                           int64_t n_coarse, int C, stin_bf16_t* out, int64_t ldo, int32_t* arg, stin_stream_t stream);
int stin_pool_max_bwd_bf16(const stin_bf16_t* g, int64_t ldg, const int32_t* arg, const int32_t* trace,
                           int64_t n_fine, int C, stin_bf16_t* gx, int64_t ldgx, stin_stream_t stream);
int stin_gather_rows_bf16(const stin_bf16_t* src, int64_t ld_src, const int32_t* idx, const float* row_scale,
                          int64_t n_out, int C, stin_bf16_t* out, int64_t ldo, stin_stream_t stream);
int stin_colreduce_bf16(int mode, const stin_bf16_t* x, int64_t ldx, const stin_bf16_t* gout, int64_t ldg, int64_t N,
                        int C, const int32_t* ptr, int B, const int32_t* gid, const int32_t* sid, const float* mean,
                        const float* rstd, const float* coef, int post, const float* inv_cnt, float eps, float* out0,
                        float* out1, void* workspace, size_t workspace_bytes, stin_stream_t stream);
int stin_norm_act_res_fwd_bf16(const stin_bf16_t* x, int64_t ldx, const float* mean, const float* rstd,
                               const int32_t* gid, const stin_bf16_t* res, int64_t ldres, int64_t N, int C, int act,
                               stin_bf16_t* y, int64_t ldy, stin_stream_t stream);
int stin_norm_act_bwd_bf16(const stin_bf16_t* x, int64_t ldx, const stin_bf16_t* gout, int64_t ldg, const float* mean,
                           const float* rstd, const float* a, const float* k, const float* m, const int32_t* gid,
                           const int32_t* sid, int64_t N, int C, int act, stin_bf16_t* dx, int64_t lddx,
                           stin_stream_t stream);
int stin_gemm_nt_bf16(const stin_bf16_t* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                      const stin_bf16_t* row_mask, int64_t ld_mask, const stin_bf16_t* residual, int64_t ld_res,
                      int64_t M, int Nc, int K, void* C, int64_t ldc, int c_is_f32, stin_stream_t stream);
int stin_gemm_tn_bf16(const stin_bf16_t* G, int64_t ldg, const stin_bf16_t* X, int64_t ldx, int64_t M, int Nc, int K,
                      int ones_column, const stin_bf16_t* row_weight, int64_t ld_weight, float* dW, int64_t lddw,
                      void* workspace, size_t workspace_bytes, stin_stream_t stream);
/* (bf16 activation rows: stin_gemm_tn_wb_f32 and the stin_linear_tanh_* pair above) */
int stin_gemm_tn_wb_bf16(const stin_bf16_t* G, int64_t ldg, const stin_bf16_t* X, int64_t ldx, int64_t M, int Nc, int K,
                         const stin_bf16_t* row_weight, int64_t ld_weight, float* dW, int64_t lddw, float* db,
                         void* workspace, size_t workspace_bytes, stin_stream_t stream);
int stin_linear_tanh_fwd_bf16(const stin_bf16_t* x, int64_t ldx, const float* W, const float* b, int64_t N, int K, int Nc,
                              float* y, stin_stream_t stream);
int stin_linear_tanh_bwd_bf16(const float* g, const float* y, const stin_bf16_t* x, int64_t ldx, const float* W, int64_t N, int K,
                              int Nc, stin_bf16_t* dx, int64_t lddx, float* dW, float* db, void* workspace,
                              size_t workspace_bytes, stin_stream_t stream);

/* ------------------------------------------------------ whole-block launch sequences --
 * One GraphResnetBlock (EdgeConv(mean) -> instance norm -> ELU -> + residual,
 * models/surfacetextureinpaintingnet.py:507-521) per call: these functions only ENQUEUE the entry points above in the
 * order of the fused block (pack -> Y GEMM -> edge stage -> agg GEMM -> moments -> norm/ELU/residual; and its
 * backward), so the arithmetic is identical to calling them one by one - what they remove is ~9 / ~16 host-side
 * foreign calls per block and direction.  storage: 0 = fp32 rows, 1 = bf16 rows (x, Y, hE, agg, out, g, dx).
 * Fast-path precondition (callers fall back to the individual entry points otherwise): H supports the saved ReLU mask.
 * slice_quirk != 0 (fwd) / sid != NULL (bwd): the reference's linspace-slice statistics for batches of unequal graphs
 * (fastinstancenorm.py:53-82) - sums over the `ptr_sum` slices, centring through gid: two-pass statistics forward, the
 * extra COEF_XC reduction backward; ptr_true are the true per-graph row ranges.
 *   fwd: x [N, Cp] (input zero-padded to Cp columns), reference-layout parameters, destination CSR, norm groups
 *        (ptr_sum/gid may be NULL for one graph; inv_cnt [B]); writes what backward needs - wcatT [Cp, Yw], w2T [H, Cout]
 *        (fp32 or pre-split per bwd_split), Y [N, Yw], hE [N, H + pad] (column H = [deg > 0]), mask [E * H / 32],
 *        agg [N, Cout], mean / rstd [B, Cout] - and out [N, Cout].  Yw = 2 H (+ Cout with a shortcut).
 *        mask may be NULL for a forward nobody differentiates (the reference's validation loop,
 *        trainers/inpainting3d_trainer.py:204-263: model(data) under torch.no_grad()): same kernels, same rows bit for
 *        bit, the E * H / 8 mask bytes per block are not written.
 *   bwd: g = dL/dout; dx [N, Cp] may be NULL; parameter gradients in the reference layout (NULL where the parameter
 *        does not exist).  All temporaries live in the caller's workspace.
 *        wgrad_stream (optional, NULL = everything on `stream`): the two weight-gradient GEMMs, their slab reductions
 *        and the unpack are off the dx <- g critical path; given a second stream they are enqueued there, ordered after
 *        `stream` by ev_dagg / ev_dy (recorded on `stream` when their inputs are complete) and followed by ev_done
 *        (recorded on wgrad_stream after the unpack).  join != 0 makes `stream` wait for ev_done before returning to the
 *        caller's next enqueue; with join == 0 the CALLER must order any reader of dW1..dbs after ev_done and keep the
 *        workspace (and x, hE, g) alive until then.  The three events are caller-owned hipEvent_t.
 */
#define STIN_TI_COMPACT 2         /* value of `trans_inv`: translation-invariant, compact layout (see the pack documentation above) */
#define STIN_BLOCK_PACKED 0x800   /* OR-ed into stin_edgeconv_block_fwd's fwd_split: the caller has already run the pack (e.g.
                                     stin_edgeconv_pack_many_f32) with the same modes into wcatT / w2T and into THIS workspace
                                     at the offsets stin_edgeconv_block_fwd_pack_offsets reports - the call then skips it    */
size_t stin_edgeconv_block_fwd_workspace_bytes(int Cin, int Cp, int H, int Cout, int has_shortcut, int B);
/* byte offsets of wcat [Yw, Cp], w2s [Cout, H] and bcat [Yw] from the workspace pointer rounded UP to 256 bytes */
int stin_edgeconv_block_fwd_pack_offsets(int Cp, int H, int Cout, int has_shortcut, size_t* off_wcat, size_t* off_w2s,
                                         size_t* off_bcat);
int stin_edgeconv_block_fwd(int storage, const void* x, int64_t ldx, int64_t N, int Cin, int Cp, int H, int Cout,
                            int has_shortcut, int trans_inv, const float* W1, const float* b1, const float* W2,
                            const float* b2, const float* Ws, const float* bs, const int32_t* rowptr_dst,
                            const int32_t* col_dst, const int32_t* ptr_sum, int B, const int32_t* gid, const float* inv_cnt,
                            int slice_quirk, float eps, int prec_fwd, int fwd_split, int bwd_split, float* wcatT, float* w2T, void* Y,
                            int64_t ldy, void* hE, int64_t ldh, uint32_t* mask, void* agg, float* mean, float* rstd, void* out,
                            int64_t ldo, void* workspace, size_t workspace_bytes, stin_stream_t stream);
size_t stin_edgeconv_block_bwd_workspace_bytes(int64_t N, int Cp, int H, int Cout, int has_shortcut, int B, int storage);
int stin_edgeconv_block_bwd(int storage, const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t N, int Cin, int Cp,
                            int H, int Cout, int has_shortcut, int trans_inv, const void* Y, int64_t ldy, const void* hE,
                            int64_t ldh, const uint32_t* mask, const void* agg, const float* mean, const float* rstd,
                            const float* wcatT, const float* w2T, const int32_t* rowptr_dst, const int32_t* rowptr_src,
                            const int32_t* col_src, const int32_t* xslot, const float* w_src, const int32_t* ptr_true, int B,
                            const int32_t* gid, const int32_t* sid, const float* inv_cnt, int prec_bwd, int bwd_split, void* dx,
                            int64_t lddx,
                            float* dW1, float* db1, float* dW2, float* db2, float* dWs, float* dbs, void* workspace,
                            size_t workspace_bytes, stin_stream_t stream, stin_stream_t wgrad_stream, stin_event_t ev_dagg,
                            stin_event_t ev_dy, stin_event_t ev_done, int join);

/* A CHAIN of fused blocks of one level in one call per direction (round 3): the n consecutive identity-residual blocks of
 * the bottleneck (models/surfacetextureinpaintingnet.py:437-443: `for i in range(n_blocks): x = bottleneck_blocks[i](x, ...)`)
 * share N, the width (Cin = Cout, no shortcut) and the norm groups; block i reads block i - 1's output, the edge set may
 * differ per block (dilations).  The calls only loop over stin_edgeconv_block_fwd / _bwd with the per-block pointers of the
 * HOST job array - identical kernels in identical order, one foreign call and one autograd node instead of n.
 *   fwd: x [N, Cp] feeds job 0; job i writes the tensors backward needs (Y, hE, mask, agg, mean, rstd, wcatT, w2T) and `out`.
 *   bwd: g = dL/d(out of the last job); job i's input gradient goes to scratch[i & 1] ([N, Cp] each) and is job i - 1's g;
 *        job 0's goes to dx (may be NULL).  Every job needs its OWN bwd workspace while a weight-gradient stream is in use
 *        (its kernels read dagg / dY from it after the call has returned).  ev_dy / ev_done per job as in block_bwd. */
typedef struct stin_chain_job {
    const float *W1, *b1, *W2, *b2;
    float *wcatT, *w2T;
    void* fwd_ws;
    const int32_t *rowptr_dst, *col_dst, *rowptr_src, *col_src, *xslot;
    const float* w_src;
    void *Y, *hE;
    uint32_t* mask;
    void* agg;
    float *mean, *rstd;
    void* out;
    float *dW1, *db1, *dW2, *db2;
    void* bwd_ws;
    stin_event_t ev_dy, ev_done;
    int32_t trans_inv, fwd_split, bwd_split, prec_fwd;
} stin_chain_job_t;                                   /* 27 pointers + 4 int32 = 232 bytes */
int stin_edgeconv_chain_fwd(int storage, const stin_chain_job_t* jobs, int n_jobs, const void* x, int64_t ldx, int64_t N,
                            int C, int Cp, int H, const int32_t* ptr_sum, int B, const int32_t* gid, const float* inv_cnt,
                            int slice_quirk, float eps, size_t fwd_ws_bytes, stin_stream_t stream);
int stin_edgeconv_chain_bwd(int storage, const stin_chain_job_t* jobs, int n_jobs, const void* g, int64_t ldg, const void* x,
                            int64_t ldx, int64_t N, int C, int Cp, int H, const int32_t* ptr_true, int B, const int32_t* gid,
                            const int32_t* sid, const float* inv_cnt, int prec_bwd, void* dx, int64_t lddx, void* scratch0,
                            void* scratch1, size_t bwd_ws_bytes, stin_stream_t stream, stin_stream_t wgrad_stream);

/* The WHOLE graph part of the network in one call per direction (round 3): every fused EdgeConv + instance-norm block and the
 * pool / unpool steps between them, in network order (models/surfacetextureinpaintingnet.py:404-455: input blocks, `for`
 * over the encoder levels - `_pooling` :384-386 then a block -, the bottleneck blocks, `for` over the decoder levels -
 * `_unpooling` :390-391 then a block -, the output blocks).  The calls loop over stin_edgeconv_block_fwd / _bwd,
 * stin_pool_max_{fwd,bwd}_*, stin_gather_rows_* and stin_segment_sum_* with the per-op pointers of a HOST op array: identical
 * kernels in identical order (bit-identical to the per-op calls), ONE foreign call and one autograd node per direction
 * instead of ~20 - what it removes is host time (the launch-bound sizes: 20 k-vertex crops, the 8-crop batches).
 *   op i reads `x` (ops[0]: the network input padded to Cp channels; else ops[i - 1].out) and writes `out`;
 *   bwd walks the ops in reverse: the gradient of op i's output is ops[i + 1].dx (the call's `g` for the last op), op i writes
 *   its input gradient to `dx` (NULL for ops[0] when the network input needs none).
 *   STIN_OP_BLOCK: the arguments of stin_edgeconv_block_fwd / _bwd (n_out = n_in = N rows); use_side != 0 puts the block's
 *     weight-gradient work on `wgrad_stream` behind ev_dy and records ev_done there (as stin_edgeconv_block_bwd).
 *   STIN_OP_POOL_MAX: x [n_in, Cout] -> out [n_out, Cout], arg [n_out, Cout]; rowptr_dst / col_dst = the children CSR, trace = the
 *     fine -> coarse map (backward).  STIN_OP_UNPOOL: out[v] = x[trace[v]] (n_out fine rows); backward = the segment sum
 *     over the children CSR. */
#define STIN_OP_BLOCK 0
#define STIN_OP_POOL_MAX 1
#define STIN_OP_UNPOOL 2
typedef struct stin_net_op {
    int32_t kind, Cin, Cp, H, Cout, has_shortcut, trans_inv, prec_fwd, fwd_split, bwd_split, B, slice_quirk, use_side, reserved0,
        reserved1, reserved2;
    float eps;
    int32_t reserved3;
    int64_t n_out, n_in, ldx, ldo, lddx, ldy, ldh;
    uint64_t fwd_ws_bytes, bwd_ws_bytes;
    const void* x;
    void* out;
    void* dx;
    const float *W1, *b1, *W2, *b2, *Ws, *bs;
    float *wcatT, *w2T;
    void* fwd_ws;
    const int32_t *rowptr_dst, *col_dst, *rowptr_src, *col_src, *xslot;
    const float* w_src;
    const int32_t *ptr_sum, *ptr_true, *gid, *sid;
    const float* inv_cnt;
    void *Y, *hE;
    uint32_t* mask;
    void* agg;
    float *mean, *rstd;
    int32_t* arg;
    const int32_t* trace;
    float *dW1, *db1, *dW2, *db2, *dWs, *dbs;
    void* bwd_ws;
    stin_event_t ev_dy, ev_done;
    stin_event_t ev_edge0, ev_edge1;  /* optional (block ops): recorded on `stream` right before / after the block's edge-stage launch
                                         (forward: stin_edge_relu_mean_fwd_*, backward: stin_edge_relu_mean_bwd_mask_*) - a timing
                                         bracket inside the call (bench.py's live roofline figure); NULL = none */
} stin_net_op_t;                      /* 16 int32 + float + int32 + 7 int64 + 2 uint64 + 42 pointers = 480 bytes */
int stin_net_fwd(int storage, const stin_net_op_t* ops, int n_ops, stin_stream_t stream);
int stin_net_bwd(int storage, const stin_net_op_t* ops, int n_ops, const void* g, int64_t ldg, int prec_bwd, stin_stream_t stream,
                 stin_stream_t wgrad_stream);

/* All weight gradients of one fused block in two launches (round 3): BOTH transposed products
 *   dW2 | db2 = dagg^T [hE[:, :H] | hE[:, H]]        (second Linear; db2 weighted by the [deg > 0] column of hE)
 *   [dW1 ; dWs | db1 ; dbs] = dY^T [x | 1]           (first Linear + shortcut in the packed operand layout)
 * as ONE grid of the producer / consumer TN kernel (csrc/stin_wgrad.hip) where both have 128 x 128 tiles and 16-byte rows
 * (otherwise one TN launch each), then ONE kernel that sums the partial slabs in the fixed order of the stand-alone
 * stin_gemm_tn_* reduction and writes the reference-layout gradients directly - bit-identical to
 * stin_gemm_tn_f32 x 2 + stin_edgeconv_unpack_grads_f32, which it replaces inside stin_edgeconv_block_bwd.
 * Replaces the autograd backward of the two nn.Linear modules of edge_conv_filter.py:46-57 and of the shortcut Linear
 * (models/surfacetextureinpaintingnet.py:489-492) for the gradients w.r.t. their parameters.
 * storage: 0 = fp32 rows, 1 = bf16 rows (dagg, hE, dY, x); hE has at least H + 1 columns; precision as stin_gemm_tn_f32. */
size_t stin_edgeconv_wgrad_workspace_bytes(int64_t N, int Cp, int H, int Cout, int has_shortcut);
int stin_edgeconv_wgrad(int storage, const void* dagg, int64_t ld_dagg, const void* hE, int64_t ldh, const void* dY,
                        int64_t ldy, const void* x, int64_t ldx, int64_t N, int Cin, int Cp, int H, int Cout,
                        int has_shortcut, int trans_inv, int precision, float* dW1, float* db1, float* dW2, float* db2,
                        float* dWs, float* dbs, void* workspace, size_t workspace_bytes, stin_stream_t stream);
/* ... for every trans_inv mode.  trans_inv == STIN_TI_COMPACT: dY = [D | g] is [N, H (+ Cout)], dW1 = the first H rows of the packed
 * product as they are, and db1 = the sum of the ti_rows rows of ti_colsum [ti_rows][H] (stin_edge_relu_mean_bwd_mask_ti_f32's column
 * partials of dA), folded by the same finalize launch; other modes ignore the two arguments. */
int stin_edgeconv_wgrad_ti(int storage, const void* dagg, int64_t ld_dagg, const void* hE, int64_t ldh, const void* dY,
                           int64_t ldy, const void* x, int64_t ldx, int64_t N, int Cin, int Cp, int H, int Cout,
                           int has_shortcut, int trans_inv, int precision, float* dW1, float* db1, float* dW2, float* db2,
                           float* dWs, float* dbs, const float* ti_colsum, int64_t ti_rows, void* workspace,
                           size_t workspace_bytes, stin_stream_t stream);

/* ------------------------------------------------- offline preprocessing on the GPU --
 * The dilated-edge walk of preprocessing/graph_dilation.py:85-137 (`compute_dilated_edges`), one thread per
 * directed adjacency entry (centre c = row_of[e], one-hop h = col[e]) of the COALESCED adjacency CSR
 * (`rowptr`/`col`: neighbours of every vertex sorted by id, duplicates removed - pyg.utils.coalesce, :53-56):
 * from h the walker repeatedly moves to the neighbour whose direction, pushed through the reference's
 * plane_projection (:27-28) with the current vertex normal, has the largest cosine with the running direction
 * (candidates adjacent to c or equal to the previous vertex excluded, similarity must be >= 0, the last maximum in
 * adjacency order wins, :104-118) and records the vertex reached after d hops for every requested d.
 *   dilations: HOST array of n_dil ascending ints in [2, 63];
 *   out[i * E + e] = vertex reached for dilations[i] (the edge is [out, c]), -1 where the walk ended earlier.
 * pos / nrm: [N, 3] row-major in the arithmetic type of the call (the pipeline passes float64,
 * preprocessing/graph_level_generation.py:463-465; the reference's dil_test float32).  Bit-identical to
 * oracle/dilation_oracle.py (fixed operation order, IEEE divide / sqrt, no FMA).
 */
int stin_dilated_walk_f32(const int32_t* rowptr, const int32_t* col, const int32_t* row_of, const float* pos,
                          const float* nrm, int64_t N, int64_t E, const int32_t* dilations, int n_dil, int32_t* out,
                          stin_stream_t stream);
int stin_dilated_walk_f64(const int32_t* rowptr, const int32_t* col, const int32_t* row_of, const double* pos,
                          const double* nrm, int64_t N, int64_t E, const int32_t* dilations, int n_dil, int32_t* out,
                          stin_stream_t stream);

/* Voxel (Rossignac) vertex clustering, preprocessing/graph_level_generation.py:193-244 (`vertex_clustering`):
 *   bins = coords // voxel (numpy's floor division, fp64), coarse ids = np.unique(bins, axis=0)'s lexicographic bin order,
 *   trace[v] = coarse id of vertex v, new_coords[c] = fp64 mean of the members' coordinates (in vertex order) as fp32.
 * coords [N, 3] fp64 row-major; trace [N] int64; new_coords [N, 3] fp32 (the first Nc rows are written);
 * state: 5 x int64 DEVICE words written by the call - state[3] != 0: unsupported input (a non-finite coordinate or more
 * than 2^21 bins along an axis), state[4] = Nc.  Stable radix sort of one 63-bit key per vertex + scan: deterministic.
 * stin_coalesce_pairs_i64 = pyg.utils.coalesce on (a[e], b[e]) pairs (graph_level_generation.py:236-241 on the coarse edges,
 * graph_dilation.py:53-56): optionally mapped through map[map_n] first (the trace), self loops dropped when drop_loops, sorted by
 * (a, b), duplicates removed; values in [0, n); out_a / out_b [E] int64 (the first state[4] entries are written);
 * state[3] != 0: an index was outside [0, n), or - with a map - a raw endpoint outside [0, map_n) (checked BEFORE map[] is read:
 * the torch formulation trace[edge_index[0]] this replaces raises there). */
size_t stin_voxel_cluster_workspace_bytes(int64_t N);
int stin_voxel_cluster_f64(const double* coords, int64_t N, double voxel, int64_t* trace, float* new_coords, int64_t* state,
                           void* workspace, size_t workspace_bytes, stin_stream_t stream);
size_t stin_coalesce_workspace_bytes(int64_t E);
int stin_coalesce_pairs_i64(const int64_t* a, const int64_t* b, const int64_t* map, int64_t map_n, int64_t E, int64_t n, int drop_loops,
                            int64_t* out_a, int64_t* out_b, int64_t* state, void* workspace, size_t workspace_bytes,
                            stin_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STIN_HIP_H */
