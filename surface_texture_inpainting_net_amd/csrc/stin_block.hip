// One GraphResnetBlock (EdgeConv(mean) -> instance norm -> ELU -> + residual) per C call: the launch SEQUENCE of the
// fused block, forward and backward, enqueued from native code instead of ~9 / ~16 Python-level ctypes calls.  The
// arithmetic is exactly that of the individual entry points (this file only calls them, in the order
// functional.EdgeConvBlockFn used to); what it removes is host time - at 200k vertices the training step was
// launch-bound on the Python side in its backward half, at 20k vertices entirely.
// Reference composition: models/surfacetextureinpaintingnet.py:507-521 (GraphResnetBlock.forward).
#include "stin_common.h"

namespace {

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
inline char* carve(char*& p, size_t bytes) {
    char* r = p;
    p += up256(bytes);
    return r;
}
#define STIN_TRY(expr)            \
    do {                          \
        const int rc_ = (expr);   \
        if (rc_ != STIN_OK) return rc_; \
    } while (0)

// element-size aware column offset
inline const void* col_off(const void* base, int64_t cols, int storage) {
    return static_cast<const char*>(base) + cols * (storage ? 2 : 4);
}
inline void* col_off(void* base, int64_t cols, int storage) { return static_cast<char*>(base) + cols * (storage ? 2 : 4); }

}  // namespace

extern "C" size_t stin_edgeconv_block_fwd_workspace_bytes(int Cin, int Cp, int H, int Cout, int has_shortcut, int B) {
    if (Cp <= 0 || H <= 0 || Cout <= 0 || B <= 0) return 0;
    const size_t Yw = 2 * (size_t)H + (has_shortcut ? Cout : 0);
    (void)Cin;
    // wcat [Yw, Cp] + w2s [Cout, H] + bcat [Yw] (forward-only weight operands) + column-reduction workspace
    return up256(Yw * Cp * 4) + up256((size_t)Cout * H * 4) + up256(Yw * 4) + up256(stin_colreduce_workspace_bytes(Cout, B)) + 256;
}

extern "C" int stin_edgeconv_block_fwd_pack_offsets(int Cp, int H, int Cout, int has_shortcut, size_t* off_wcat, size_t* off_w2s,
                                                    size_t* off_bcat) {
    if (Cp <= 0 || H <= 0 || Cout <= 0 || !off_wcat || !off_w2s || !off_bcat) return STIN_E_SIZE;
    const size_t Yw = 2 * (size_t)H + (has_shortcut ? Cout : 0);
    *off_wcat = 0;                                                  // the carve order of stin_edgeconv_block_fwd
    *off_w2s = up256(Yw * Cp * 4);
    *off_bcat = *off_w2s + up256((size_t)Cout * H * 4);
    return STIN_OK;
}

// Forward.  storage: 0 = fp32 rows, 1 = bf16 rows (x, Y, hE, agg, out).  Saved for backward by the caller: x, Y, hE,
// mask, agg, mean, rstd, wcatT, w2T.  Requirements of this fast path (the caller falls back to the individual entry
// points otherwise): saved ReLU mask supported for H.  slice_quirk: statistics over the reference's linspace slices
// (fastinstancenorm.py:53-82) instead of the true per-graph ranges - two passes, as the reference computes them.
// Optional HIP-event bracket around the edge-stage launch of the NEXT block call on this host thread (set by stin_net_fwd / _bwd
// from stin_net_op_t::ev_edge0 / ev_edge1, cleared right after): how bench.py times the roofline kernel inside the
// whole-network call without leaving the fast path.
static thread_local hipEvent_t t_edge_ev0 = nullptr, t_edge_ev1 = nullptr;
#define STIN_EDGE_BRACKET(CALL)                                                             \
    do {                                                                                    \
        if (t_edge_ev0) (void)hipEventRecord(t_edge_ev0, (hipStream_t)stream);              \
        STIN_TRY(CALL);                                                                     \
        if (t_edge_ev1) (void)hipEventRecord(t_edge_ev1, (hipStream_t)stream);              \
    } while (0)

extern "C" int stin_edgeconv_block_fwd(int storage, const void* x, int64_t ldx, int64_t N, int Cin, int Cp, int H, int Cout,
                                       int has_shortcut, int trans_inv, const float* W1, const float* b1, const float* W2,
                                       const float* b2, const float* Ws, const float* bs, const int32_t* rowptr_dst,
                                       const int32_t* col_dst, const int32_t* ptr_sum, int B, const int32_t* gid,
                                       const float* inv_cnt, int slice_quirk, float eps, int prec_fwd, int fwd_split, int bwd_split,
                                       float* wcatT, float* w2T, void* Y, int64_t ldy, void* hE, int64_t ldh, uint32_t* mask,
                                       void* agg, float* mean, float* rstd, void* out, int64_t ldo, void* workspace,
                                       size_t workspace_bytes, stin_stream_t stream) {
    STIN_REQUIRE(storage == 0 || storage == 1, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(N >= 0 && Cin > 0 && Cp >= Cin && H > 0 && Cout > 0 && B > 0, STIN_E_SIZE);
    // mask == NULL (round 6): a forward nobody differentiates (torch.no_grad() / evaluation) - the ReLU mask is not stored
    STIN_REQUIRE(x && W1 && W2 && rowptr_dst && wcatT && w2T && Y && hE && agg && mean && rstd && out && workspace,
                 STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_edgeconv_block_fwd_workspace_bytes(Cin, Cp, H, Cout, has_shortcut, B), STIN_E_WORKSPACE);
    // trans_inv == STIN_TI_COMPACT (round 6, fp32 rows): Y = [B | S], the edge stage forms A_i = b1 - B_i (stin_common.h: stin_yw)
    const bool compact = trans_inv == STIN_TI_COMPACT;
    STIN_REQUIRE(!compact || storage == 0, STIN_E_UNSUPPORTED);
    const int Yw = stin_yw(H, Cout, has_shortcut, trans_inv);
    const int Yw_max = 2 * H + (has_shortcut ? Cout : 0);          // the carve keeps the offsets of stin_edgeconv_block_fwd_pack_offsets
    STIN_REQUIRE(ldy >= Yw, STIN_E_SIZE);
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float* wcat = reinterpret_cast<float*>(carve(p, (size_t)Yw_max * Cp * 4));
    float* w2s = reinterpret_cast<float*>(carve(p, (size_t)Cout * H * 4));
    float* bcat = reinterpret_cast<float*>(carve(p, (size_t)Yw_max * 4));
    void* red_ws = p;
    const size_t red_bytes = stin_colreduce_workspace_bytes(Cout, B);

    // bf16 rows: the GEMM weight operands are written as bf16 once here (half the bytes every tile load, no conversion)
    // when every reduction length is a multiple of 8; fwd_split / bwd_split then carry STIN_GEMM_W_BF16
    const bool packed = (fwd_split & STIN_BLOCK_PACKED) != 0;      // the caller ran the pack (pack_many; bf16 rows: with the modes below)
    fwd_split &= ~STIN_BLOCK_PACKED;
    if (storage == 1) fwd_split = bwd_split = (Cp % 8 == 0 && Cout % 8 == 0) ? STIN_GEMM_W_BF16 : 0;
    if (!packed)
        STIN_TRY(stin_edgeconv_pack_f32(W1, b1, Ws, bs, W2, Cin, Cp, H, Cout, has_shortcut, trans_inv, wcat, bcat, wcatT, w2T,
                                        fwd_split ? w2s : nullptr, fwd_split, bwd_split, stream));
    const float* w2_op = fwd_split ? w2s : W2;
    const int wbf = (storage == 1 && fwd_split) ? STIN_GEMM_W_BF16 : 0;
    const int pf = fwd_split ? (prec_fwd | STIN_GEMM_W_PRESPLIT | (fwd_split & STIN_GEMM_W_FRAG)) : prec_fwd;
    const void* res = has_shortcut ? col_off(static_cast<const void*>(Y), (int64_t)(Yw - Cout), storage) : x;
    const int64_t ld_res = has_shortcut ? ldy : ldx;
    if (storage == 0) {
        float* Yf = static_cast<float*>(Y);
        float* hf = static_cast<float*>(hE);
        STIN_TRY(stin_gemm_nt_f32(static_cast<const float*>(x), ldx, wcat, Cp, bcat, nullptr, 0, nullptr, 0, N, Yw, Cp, Yf, ldy,
                                  pf, stream));
        if (compact)
            STIN_EDGE_BRACKET(stin_edge_relu_mean_fwd_ti_f32(b1, Yf, ldy, rowptr_dst, col_dst, N, H, hf, ldh, 1, mask, stream));
        else
            STIN_EDGE_BRACKET(stin_edge_relu_mean_fwd_f32(Yf, ldy, Yf + H, ldy, rowptr_dst, col_dst, N, H, hf, ldh, 1, mask, stream));
        // one graph, all-columns GEMM shape: the column sums of agg come out of GEMM2's epilogue (no pass over agg for them)
        const int64_t stat_groups = (B == 1 && gid == nullptr && !slice_quirk) ? stin_gemm_nt_colstats_groups(N, Cout, H, pf) : 0;
        const bool fused_stats = stat_groups > 0 && (size_t)stat_groups * 2 * Cout * sizeof(double) + 256 <= red_bytes;
        bool normed = false;
        if (fused_stats) {
            double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(red_ws) + 255) & ~(uintptr_t)255);
            STIN_TRY(stin_gemm_nt_colstats_f32(hf, ldh, w2_op, H, b2, hf + H, ldh, nullptr, 0, N, Cout, H, static_cast<float*>(agg),
                                               Cout, pf, partial, (size_t)stat_groups * 2 * Cout * sizeof(double), stream));
            // (round 5) few row groups (the bottleneck level): every workgroup of the normalisation launch folds its own columns'
            // partials - no separate fold launch on the critical path; same sums, same order: bit-identical (k_norm_fold)
            int rc_fold = STIN_E_UNSUPPORTED;
            if (N > 0 && stin_norm_fold_rows(N, Cout, stat_groups) > 0)
                rc_fold = stin_norm_act_res_fwd_fold_f32(partial, stat_groups, static_cast<const float*>(agg), Cout, static_cast<const float*>(res),
                                                         ld_res, inv_cnt, eps, N, Cout, mean, rstd, static_cast<float*>(out), ldo, stream);
            if (rc_fold == STIN_OK) normed = true;
            else if (rc_fold != STIN_E_UNSUPPORTED) return rc_fold;
            else STIN_TRY(stin_moments_final_f32(partial, stat_groups, Cout, inv_cnt, eps, mean, rstd, stream));
        } else {
            STIN_TRY(stin_gemm_nt_f32(hf, ldh, w2_op, H, b2, hf + H, ldh, nullptr, 0, N, Cout, H, static_cast<float*>(agg), Cout,
                                      pf, stream));
        }
        if (fused_stats) {
        } else if (!slice_quirk) {
            STIN_TRY(stin_colreduce_f32(STIN_RED_MOMENTS, static_cast<const float*>(agg), Cout, nullptr, 0, N, Cout, ptr_sum, B,
                                        gid, nullptr, nullptr, nullptr, nullptr, STIN_POST_NONE, inv_cnt, eps, mean, rstd, red_ws,
                                        red_bytes, stream));
        } else {  // sums over the linspace slices, centring through the graph id: two passes as the reference does
            STIN_TRY(stin_colreduce_f32(STIN_RED_SUM, static_cast<const float*>(agg), Cout, nullptr, 0, N, Cout, ptr_sum, B, gid,
                                        nullptr, nullptr, nullptr, nullptr, STIN_POST_SCALE, inv_cnt, eps, mean, nullptr, red_ws,
                                        red_bytes, stream));
            STIN_TRY(stin_colreduce_f32(STIN_RED_CSQ, static_cast<const float*>(agg), Cout, nullptr, 0, N, Cout, ptr_sum, B, gid,
                                        nullptr, mean, nullptr, nullptr, STIN_POST_RSTD, inv_cnt, eps, rstd, nullptr, red_ws,
                                        red_bytes, stream));
        }
        if (!normed)
            STIN_TRY(stin_norm_act_res_fwd_f32(static_cast<const float*>(agg), Cout, mean, rstd, gid, static_cast<const float*>(res),
                                               ld_res, N, Cout, 1, static_cast<float*>(out), ldo, stream));
    } else {
        stin_bf16_t* Yh = static_cast<stin_bf16_t*>(Y);
        stin_bf16_t* hh = static_cast<stin_bf16_t*>(hE);
        STIN_TRY(stin_gemm_nt_bf16(static_cast<const stin_bf16_t*>(x), ldx, wcat, Cp, bcat, nullptr, 0, nullptr, 0, N, Yw, Cp, Yh,
                                   ldy, wbf, stream));
        STIN_EDGE_BRACKET(stin_edge_relu_mean_fwd_bf16(Yh, ldy, Yh + H, ldy, rowptr_dst, col_dst, N, H, hh, ldh, 1, mask, stream));
        STIN_TRY(stin_gemm_nt_bf16(hh, ldh, w2_op, H, b2, hh + H, ldh, nullptr, 0, N, Cout, H, agg, Cout, wbf, stream));
        if (!slice_quirk) {
            STIN_TRY(stin_colreduce_bf16(STIN_RED_MOMENTS, static_cast<const stin_bf16_t*>(agg), Cout, nullptr, 0, N, Cout, ptr_sum,
                                         B, gid, nullptr, nullptr, nullptr, nullptr, STIN_POST_NONE, inv_cnt, eps, mean, rstd,
                                         red_ws, red_bytes, stream));
        } else {
            STIN_TRY(stin_colreduce_bf16(STIN_RED_SUM, static_cast<const stin_bf16_t*>(agg), Cout, nullptr, 0, N, Cout, ptr_sum, B,
                                         gid, nullptr, nullptr, nullptr, nullptr, STIN_POST_SCALE, inv_cnt, eps, mean, nullptr,
                                         red_ws, red_bytes, stream));
            STIN_TRY(stin_colreduce_bf16(STIN_RED_CSQ, static_cast<const stin_bf16_t*>(agg), Cout, nullptr, 0, N, Cout, ptr_sum, B,
                                         gid, nullptr, mean, nullptr, nullptr, STIN_POST_RSTD, inv_cnt, eps, rstd, nullptr, red_ws,
                                         red_bytes, stream));
        }
        STIN_TRY(stin_norm_act_res_fwd_bf16(static_cast<const stin_bf16_t*>(agg), Cout, mean, rstd, gid,
                                            static_cast<const stin_bf16_t*>(res), ld_res, N, Cout, 1,
                                            static_cast<stin_bf16_t*>(out), ldo, stream));
    }
    return STIN_OK;
}

extern "C" size_t stin_edgeconv_block_bwd_workspace_bytes(int64_t N, int Cp, int H, int Cout, int has_shortcut, int B,
                                                          int storage) {
    if (N < 0 || Cp <= 0 || H <= 0 || Cout <= 0 || B <= 0) return 0;
    const size_t Yw = 2 * (size_t)H + (has_shortcut ? Cout : 0);
    const size_t es = storage ? 2 : 4;
    const size_t tn = stin_edgeconv_wgrad_workspace_bytes(N, Cp, H, Cout, has_shortcut);   // the slabs of both products
    // (sized for the wide layout; the compact trans-inv layout uses H of dY's 2 H columns and the slack pays for its dA column partials)
    return up256((size_t)N * Cout * es)      /* dagg */
           + up256((size_t)N * H * es)       /* dhE  */
           + up256((size_t)N * Yw * es)      /* dY   */
           + 5 * up256((size_t)B * Cout * 4) /* k, m (+ T1, S0, U with the linspace-slice quirk) */
           + up256(stin_colreduce_workspace_bytes(Cout, B)) + up256(tn) + 256;
}

// Backward of the same block.  g = dL/dout [N, Cout]; dx may be NULL (block input needs no gradient).  Gradients of the
// reference-layout parameters are written to dW1 [H, Cin or 2 Cin], db1 [H], dW2 [Cout, H], db2 [Cout], dWs [Cout, Cin],
// dbs [Cout] (bias / shortcut outputs may be NULL when the parameter does not exist).
// Hand-off between consecutive blocks of stin_net_bwd (round 4).  The input gradient dx of block k is the output gradient of
// block k - 1, whose instance-norm backward starts with two column sums over (agg_{k-1}, dx): when block k's dx product runs on
// the panel kernel those sums ride on its epilogue (stin_gemm_nt_dotelu_f32) and block k - 1 only folds the partials
// (stin_norm_coef_from_partials_f32) - one short, contention-sensitive launch less on the critical path per hand-off.
struct BwdLink {
    // producer side (this block's dx product computes the NEXT block-in-backward-order's statistics)
    const float* next_agg = nullptr;
    int64_t next_ld = 0;
    const float* next_mean = nullptr;
    const float* next_rstd = nullptr;
    double* next_partial = nullptr;
    size_t next_partial_bytes = 0;
    int64_t produced_groups = 0;         // out: > 0 when the partials were written
    // consumer side (this block's statistics were computed by the block before it in backward order)
    const double* pre_partial = nullptr;
    int64_t pre_groups = 0;
};

// where stin_edgeconv_block_bwd's workspace keeps its column-reduction scratch (the partials of a hand-off are written there)
static void* bwd_ws_red(void* workspace, int64_t N, int Cp, int H, int Cout, int has_shortcut, int B, int storage, size_t* red_bytes) {
    const int Yw = 2 * H + (has_shortcut ? Cout : 0);
    const size_t es = storage ? 2 : 4;
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    (void)carve(p, (size_t)N * Cout * es);
    (void)carve(p, (size_t)N * H * es);
    (void)carve(p, (size_t)N * Yw * es);
    for (int i = 0; i < 5; ++i) (void)carve(p, (size_t)B * Cout * 4);
    *red_bytes = stin_colreduce_workspace_bytes(Cout, B);
    (void)Cp;
    return p;
}

static int block_bwd_impl(int storage, const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t N, int Cin,
                                       int Cp, int H, int Cout, int has_shortcut, int trans_inv, const void* Y, int64_t ldy,
                                       const void* hE, int64_t ldh, const uint32_t* mask, const void* agg, const float* mean,
                                       const float* rstd, const float* wcatT, const float* w2T, const int32_t* rowptr_dst,
                                       const int32_t* rowptr_src, const int32_t* col_src, const int32_t* xslot,
                                       const float* w_src, const int32_t* ptr_true, int B, const int32_t* gid,
                                       const int32_t* sid, const float* inv_cnt, int prec_bwd, int bwd_split, void* dx, int64_t lddx, float* dW1,
                                       float* db1, float* dW2, float* db2, float* dWs, float* dbs, void* workspace,
                                       size_t workspace_bytes, stin_stream_t stream, stin_stream_t wgrad_stream,
                                       stin_event_t ev_dagg, stin_event_t ev_dy, stin_event_t ev_done, int join, BwdLink* link) {
    (void)Y;
    (void)ldy;
    STIN_REQUIRE(storage == 0 || storage == 1, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(N >= 0 && Cin > 0 && Cp >= Cin && H > 0 && Cout > 0 && B > 0, STIN_E_SIZE);
    STIN_REQUIRE(g && x && hE && mask && agg && mean && rstd && wcatT && w2T && rowptr_dst && rowptr_src && col_src && xslot &&
                     w_src && inv_cnt && dW1 && dW2 && workspace && (!has_shortcut || dWs),
                 STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_edgeconv_block_bwd_workspace_bytes(N, Cp, H, Cout, has_shortcut, B, storage),
                 STIN_E_WORKSPACE);
    const bool compact = trans_inv == STIN_TI_COMPACT;
    STIN_REQUIRE(!compact || storage == 0, STIN_E_UNSUPPORTED);
    const int Yw = stin_yw(H, Cout, has_shortcut, trans_inv);
    const int Yw_max = 2 * H + (has_shortcut ? Cout : 0);
    const size_t es = storage ? 2 : 4;
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    void* dagg = carve(p, (size_t)N * Cout * es);
    void* dhE = carve(p, (size_t)N * H * es);
    void* dY = carve(p, (size_t)N * Yw_max * es);
    // compact layout: dY is [N, Yw] inside that region; the column partials of dA ([rows][H] floats, rows ~ N / 16..32) live behind it
    // in the H unused columns' worth of space (N * H * 4 bytes >= rows * H * 4)
    float* ti_colsum = nullptr;
    int64_t ti_rows = 0;
    if (compact) {
        ti_rows = stin_edge_bwd_ti_colsum_rows(N, H);
        ti_colsum = reinterpret_cast<float*>(static_cast<char*>(dY) + up256((size_t)N * Yw * 4));
        STIN_REQUIRE(up256((size_t)N * Yw * 4) + (size_t)ti_rows * H * 4 <= up256((size_t)N * Yw_max * 4), STIN_E_WORKSPACE);
    }
    float* kk = reinterpret_cast<float*>(carve(p, (size_t)B * Cout * 4));
    float* mm = reinterpret_cast<float*>(carve(p, (size_t)B * Cout * 4));
    float* t1 = reinterpret_cast<float*>(carve(p, (size_t)B * Cout * 4));
    float* s0 = reinterpret_cast<float*>(carve(p, (size_t)B * Cout * 4));
    float* uu = reinterpret_cast<float*>(carve(p, (size_t)B * Cout * 4));
    const int32_t* sid_n = sid ? sid : gid;     // slice id of the norm backward (== graph id without the quirk)
    const size_t red_bytes = stin_colreduce_workspace_bytes(Cout, B);
    void* red_ws = carve(p, red_bytes);
    void* tn_ws = p;
    const size_t tn_bytes = workspace_bytes - (size_t)(p - static_cast<char*>(workspace));
    const int pb = bwd_split ? (prec_bwd | STIN_GEMM_W_PRESPLIT | (bwd_split & STIN_GEMM_W_FRAG)) : prec_bwd;
    hipStream_t hs = (hipStream_t)stream;
    // weight-gradient GEMMs are off the critical path dx <- g: with a wgrad_stream they run beside the edge-stage /
    // dx kernels of this block (and the head of the next one), ordered by the caller's events
    const bool side = wgrad_stream != nullptr && wgrad_stream != stream;
    if (side) STIN_REQUIRE(ev_dy && ev_done, STIN_E_NULL);
    (void)ev_dagg;                             // (round 2 forked the dW2 product here; both products now start behind ev_dy)
    stin_stream_t ws_ = side ? wgrad_stream : stream;
    // the fork event is bound to the edge-stage kernel's own completion signal where that launch is the last one before the fork
    // (STIN_LAUNCH_STOP, stin_common.h)
    bool bind_on = false;
    if (side) {                    // (a stream being captured into a hipGraph keeps the event-record node)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        bind_on = hipStreamIsCapturing(hs, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    }
    struct StopEventGuard { ~StopEventGuard() { stin_tl_stop_event = nullptr; } } stop_event_guard;   // never left set on an error return
    bool bound = false;
    auto fork = [&](stin_event_t ev) -> int {
        if (!side) return STIN_OK;
        hipError_t e = bound ? hipSuccess : hipEventRecord((hipEvent_t)ev, hs);
        if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)wgrad_stream, (hipEvent_t)ev, 0);
        return (int)e;
    };

    if (storage == 0) {
        const float* gf = static_cast<const float*>(g);
        const float* hf = static_cast<const float*>(hE);
        float* dYf = static_cast<float*>(dY);
        // instance norm + ELU backward: two column sums finalised straight into the k / m coefficients, one elementwise pass
        bool dagg_done = false;
        if (link != nullptr && link->pre_partial != nullptr && link->pre_groups > 0 && !sid && B == 1 && gid == nullptr) {
            // the two column sums came out of the previous block's dx product (BwdLink): fold its partials - (round 5) inside the
            // normalisation launch itself where the row groups are few (k_norm_fold: every workgroup folds its own columns)
            // Measured (round 5): the forward form is 16 us where norm + fold were 19; the backward twin (stin_norm_act_bwd_fold_f32:
            // two fp64 folds per workgroup in front of the rows) 44 us where they were 39 - the backward keeps the separate fold.
            STIN_TRY(stin_norm_coef_from_partials_f32(link->pre_partial, link->pre_groups, Cout, rstd, inv_cnt, kk, mm, stream));
        } else if (!sid) {
            STIN_TRY(stin_colreduce_f32(STIN_RED_DOT_ELU, static_cast<const float*>(agg), Cout, gf, ldg, N, Cout, ptr_true, B, gid,
                                        nullptr, mean, rstd, nullptr, STIN_POST_NORM_COEF, inv_cnt, 0.f, kk, mm, red_ws, red_bytes,
                                        stream));
        } else {  // linspace-slice quirk: k from the per-graph sums, then U = sum over the slice of k xc, then m
            STIN_TRY(stin_colreduce_f32(STIN_RED_DOT_ELU, static_cast<const float*>(agg), Cout, gf, ldg, N, Cout, ptr_true, B, gid,
                                        nullptr, mean, rstd, nullptr, STIN_POST_NONE, inv_cnt, 0.f, t1, s0, red_ws, red_bytes,
                                        stream));
            STIN_TRY(stin_norm_bwd_coef_f32(t1, s0, rstd, inv_cnt, B, Cout, kk, mm, stream));
            STIN_TRY(stin_colreduce_f32(STIN_RED_COEF_XC, static_cast<const float*>(agg), Cout, nullptr, 0, N, Cout, ptr_true, B,
                                        gid, sid, mean, nullptr, kk, STIN_POST_NONE, inv_cnt, 0.f, uu, nullptr, red_ws, red_bytes,
                                        stream));
            STIN_TRY(stin_norm_bwd_coef_m_quirk_f32(s0, uu, rstd, inv_cnt, B, Cout, mm, stream));
        }
        if (!dagg_done)
            STIN_TRY(stin_norm_act_bwd_f32(static_cast<const float*>(agg), Cout, gf, ldg, mean, rstd, rstd, kk, mm, gid, sid_n, N,
                                           Cout, 1, static_cast<float*>(dagg), Cout, stream));
        // second Linear: weight gradient (+ masked bias gradient) and input gradient
        STIN_TRY(stin_gemm_nt_f32(static_cast<const float*>(dagg), Cout, w2T, Cout, nullptr, nullptr, 0, nullptr, 0, N, H, Cout,
                                  static_cast<float*>(dhE), H, pb, stream));
        // edge stage backward from the saved ReLU mask -> dY = [dA | dB | g]
        // (a shortcut block's dY[:, 2H:] = g rides on the same launch when the rows allow 16-byte copies, else one 2-D memcpy)
        float* dYs = dYf + (Yw - Cout);                             // the shortcut columns of dY (has_shortcut)
        const bool ride = has_shortcut && N > 0 && Cout % 4 == 0 && ldg % 4 == 0 && Yw % 4 == 0 && stin_aligned16(gf) &&
                          stin_aligned16(dYs) && Cout <= H;
        if (side && bind_on && N > 0 && (!has_shortcut || ride) && t_edge_ev1 == nullptr) stin_tl_stop_event = (hipEvent_t)ev_dy;
        if (compact)     // D = dB - dA in ONE row of H columns, + the column partials of dA for db1 (stin_graph.hip: k_edge_bwd_mask_ti)
            STIN_EDGE_BRACKET(stin_edge_relu_mean_bwd_mask_ti_f32(static_cast<const float*>(dhE), H, mask, rowptr_dst, w_src, rowptr_src,
                                                                 col_src, xslot, N, H, dYf, Yw, ride ? gf : nullptr, ldg,
                                                                 ride ? dYs : nullptr, Yw, ride ? Cout : 0, ti_colsum, ti_rows, stream));
        else
            STIN_EDGE_BRACKET(stin_edge_relu_mean_bwd_mask_f32(static_cast<const float*>(dhE), H, mask, rowptr_dst, w_src, rowptr_src, col_src,
                                                              xslot, N, H, dYf, Yw, dYf + H, Yw, ride ? gf : nullptr, ldg,
                                                              ride ? dYs : nullptr, Yw, ride ? Cout : 0, stream));
        if (has_shortcut && N > 0 && !ride) {
            hipError_t e = hipMemcpy2DAsync(dYs, (size_t)Yw * 4, gf, (size_t)ldg * 4, (size_t)Cout * 4, (size_t)N,
                                            hipMemcpyDeviceToDevice, hs);
            if (e != hipSuccess) return (int)e;
        }
        // first Linear (+ shortcut): packed weight gradient and the block-input gradient (+ identity residual)
        // all weight gradients: both transposed products in one grid + one finalize launch (stin_wgrad.hip), off the
        // critical path dx <- g on the caller's weight-gradient stream
        bound = side && bind_on && N > 0 && (!has_shortcut || ride) && t_edge_ev1 == nullptr && stin_tl_stop_event == nullptr;
        stin_tl_stop_event = nullptr;
        STIN_TRY(fork(ev_dy));
        STIN_TRY(stin_edgeconv_wgrad_ti(0, dagg, Cout, hf, ldh, dYf, Yw, x, ldx, N, Cin, Cp, H, Cout, has_shortcut, trans_inv, prec_bwd,
                                        dW1, db1, dW2, db2, dWs, dbs, ti_colsum, ti_rows, tn_ws, tn_bytes, ws_));
        if (dx != nullptr) {
            const bool link_ok = link != nullptr && link->next_agg != nullptr && link->next_ld % 4 == 0 && stin_aligned16(link->next_agg) &&
                                 stin_aligned16(link->next_mean) && stin_aligned16(link->next_rstd) && link->next_partial != nullptr;
            const int64_t lg = link_ok ? stin_gemm_nt_dotelu_groups(N, Cp, Yw, pb) : 0;
            if (lg > 0 && (size_t)lg * 2 * Cp * sizeof(double) <= link->next_partial_bytes) {
                STIN_TRY(stin_gemm_nt_dotelu_f32(dYf, Yw, wcatT, Yw, nullptr, has_shortcut ? nullptr : gf, ldg, N, Cp, Yw,
                                                 static_cast<float*>(dx), lddx, pb, link->next_agg, link->next_ld, link->next_mean,
                                                 link->next_rstd, link->next_partial, link->next_partial_bytes, stream));
                link->produced_groups = lg;
            } else {
                STIN_TRY(stin_gemm_nt_f32(dYf, Yw, wcatT, Yw, nullptr, nullptr, 0, has_shortcut ? nullptr : gf, ldg, N, Cp, Yw,
                                          static_cast<float*>(dx), lddx, pb, stream));
            }
        }
    } else {
        const stin_bf16_t* gh = static_cast<const stin_bf16_t*>(g);
        const stin_bf16_t* hh = static_cast<const stin_bf16_t*>(hE);
        stin_bf16_t* dYh = static_cast<stin_bf16_t*>(dY);
        if (!sid) {
            STIN_TRY(stin_colreduce_bf16(STIN_RED_DOT_ELU, static_cast<const stin_bf16_t*>(agg), Cout, gh, ldg, N, Cout, ptr_true, B,
                                         gid, nullptr, mean, rstd, nullptr, STIN_POST_NORM_COEF, inv_cnt, 0.f, kk, mm, red_ws,
                                         red_bytes, stream));
        } else {
            STIN_TRY(stin_colreduce_bf16(STIN_RED_DOT_ELU, static_cast<const stin_bf16_t*>(agg), Cout, gh, ldg, N, Cout, ptr_true, B,
                                         gid, nullptr, mean, rstd, nullptr, STIN_POST_NONE, inv_cnt, 0.f, t1, s0, red_ws, red_bytes,
                                         stream));
            STIN_TRY(stin_norm_bwd_coef_f32(t1, s0, rstd, inv_cnt, B, Cout, kk, mm, stream));
            STIN_TRY(stin_colreduce_bf16(STIN_RED_COEF_XC, static_cast<const stin_bf16_t*>(agg), Cout, nullptr, 0, N, Cout, ptr_true,
                                         B, gid, sid, mean, nullptr, kk, STIN_POST_NONE, inv_cnt, 0.f, uu, nullptr, red_ws,
                                         red_bytes, stream));
            STIN_TRY(stin_norm_bwd_coef_m_quirk_f32(s0, uu, rstd, inv_cnt, B, Cout, mm, stream));
        }
        STIN_TRY(stin_norm_act_bwd_bf16(static_cast<const stin_bf16_t*>(agg), Cout, gh, ldg, mean, rstd, rstd, kk, mm, gid, sid_n, N,
                                        Cout, 1, static_cast<stin_bf16_t*>(dagg), Cout, stream));
        const int wbb = (Cp % 8 == 0 && Cout % 8 == 0) ? STIN_GEMM_W_BF16 : 0;   // as written by the forward call's pack
        STIN_TRY(stin_gemm_nt_bf16(static_cast<const stin_bf16_t*>(dagg), Cout, w2T, Cout, nullptr, nullptr, 0, nullptr, 0, N, H,
                                   Cout, dhE, H, wbb, stream));
        const bool ride = has_shortcut && N > 0 && Cout % 8 == 0 && ldg % 8 == 0 && Yw % 8 == 0 && stin_aligned16(gh) &&
                          stin_aligned16(dYh + 2 * H);
        if (side && bind_on && N > 0 && (!has_shortcut || ride) && t_edge_ev1 == nullptr) stin_tl_stop_event = (hipEvent_t)ev_dy;
        STIN_EDGE_BRACKET(stin_edge_relu_mean_bwd_mask_bf16(static_cast<const stin_bf16_t*>(dhE), H, mask, rowptr_dst, w_src, rowptr_src,
                                                           col_src, xslot, N, H, dYh, Yw, dYh + H, Yw, ride ? gh : nullptr, ldg,
                                                           ride ? dYh + 2 * H : nullptr, Yw, ride ? Cout : 0, stream));
        if (has_shortcut && N > 0 && !ride) {
            hipError_t e = hipMemcpy2DAsync(dYh + 2 * H, (size_t)Yw * 2, gh, (size_t)ldg * 2, (size_t)Cout * 2, (size_t)N,
                                            hipMemcpyDeviceToDevice, hs);
            if (e != hipSuccess) return (int)e;
        }
        bound = side && bind_on && N > 0 && (!has_shortcut || ride) && t_edge_ev1 == nullptr && stin_tl_stop_event == nullptr;
        stin_tl_stop_event = nullptr;
        STIN_TRY(fork(ev_dy));
        STIN_TRY(stin_edgeconv_wgrad(1, dagg, Cout, hh, ldh, dYh, Yw, x, ldx, N, Cin, Cp, H, Cout, has_shortcut, trans_inv, prec_bwd,
                                     dW1, db1, dW2, db2, dWs, dbs, tn_ws, tn_bytes, ws_));
        if (dx != nullptr)
            STIN_TRY(stin_gemm_nt_bf16(dYh, Yw, wcatT, Yw, nullptr, nullptr, 0, has_shortcut ? nullptr : gh, ldg, N, Cp, Yw, dx,
                                       lddx, wbb, stream));
    }
    if (side) {
        hipError_t e = hipEventRecord((hipEvent_t)ev_done, (hipStream_t)wgrad_stream);
        if (e == hipSuccess && join) e = hipStreamWaitEvent(hs, (hipEvent_t)ev_done, 0);
        if (e != hipSuccess) return (int)e;
    } else if (ev_done != nullptr) {
        // (round 4) no weight-gradient stream for this block: ev_done still marks "this block's parameter gradients are
        // written", on the compute stream - what the overlapped gradient all-reduce of a data-parallel step waits for
        // segment by segment while the rest of stin_net_bwd's kernels are still queued (train_step.FlatGradBucket.blocks_done)
        hipError_t e = hipEventRecord((hipEvent_t)ev_done, hs);
        if (e != hipSuccess) return (int)e;
    }
    return STIN_OK;
}

extern "C" int stin_edgeconv_block_bwd(int storage, const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t N, int Cin,
                                       int Cp, int H, int Cout, int has_shortcut, int trans_inv, const void* Y, int64_t ldy,
                                       const void* hE, int64_t ldh, const uint32_t* mask, const void* agg, const float* mean,
                                       const float* rstd, const float* wcatT, const float* w2T, const int32_t* rowptr_dst,
                                       const int32_t* rowptr_src, const int32_t* col_src, const int32_t* xslot,
                                       const float* w_src, const int32_t* ptr_true, int B, const int32_t* gid,
                                       const int32_t* sid, const float* inv_cnt, int prec_bwd, int bwd_split, void* dx, int64_t lddx, float* dW1,
                                       float* db1, float* dW2, float* db2, float* dWs, float* dbs, void* workspace,
                                       size_t workspace_bytes, stin_stream_t stream, stin_stream_t wgrad_stream,
                                       stin_event_t ev_dagg, stin_event_t ev_dy, stin_event_t ev_done, int join) {
    return block_bwd_impl(storage, g, ldg, x, ldx, N, Cin, Cp, H, Cout, has_shortcut, trans_inv, Y, ldy, hE, ldh, mask, agg, mean, rstd,
                          wcatT, w2T, rowptr_dst, rowptr_src, col_src, xslot, w_src, ptr_true, B, gid, sid, inv_cnt, prec_bwd, bwd_split,
                          dx, lddx, dW1, db1, dW2, db2, dWs, dbs, workspace, workspace_bytes, stream, wgrad_stream, ev_dagg, ev_dy,
                          ev_done, join, nullptr);
}

// ------------------------------------------------------------------------------------------------ chains of blocks
extern "C" int stin_edgeconv_chain_fwd(int storage, const stin_chain_job_t* jobs, int n_jobs, const void* x, int64_t ldx, int64_t N,
                                       int C, int Cp, int H, const int32_t* ptr_sum, int B, const int32_t* gid, const float* inv_cnt,
                                       int slice_quirk, float eps, size_t fwd_ws_bytes, stin_stream_t stream) {
    STIN_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || jobs != nullptr), STIN_E_NULL);
    STIN_REQUIRE(Cp == C || n_jobs <= 1, STIN_E_SIZE);                  // (a chain feeds [N, C] outputs back in as [N, Cp] inputs)
    const int Yw = 2 * H;
    const int pad = storage ? 8 : 4;
    const void* xi = x;
    int64_t ldi = ldx;
    for (int i = 0; i < n_jobs; ++i) {
        const stin_chain_job_t& J = jobs[i];
        STIN_TRY(stin_edgeconv_block_fwd(storage, xi, ldi, N, C, Cp, H, C, 0, J.trans_inv, J.W1, J.b1, J.W2, J.b2, nullptr, nullptr,
                                         J.rowptr_dst, J.col_dst, ptr_sum, B, gid, inv_cnt, slice_quirk, eps, J.prec_fwd, J.fwd_split,
                                         J.bwd_split, J.wcatT, J.w2T, J.Y, Yw, J.hE, H + pad, J.mask, J.agg, J.mean, J.rstd, J.out, C,
                                         J.fwd_ws, fwd_ws_bytes, stream));
        xi = J.out;
        ldi = C;
    }
    return STIN_OK;
}

extern "C" int stin_edgeconv_chain_bwd(int storage, const stin_chain_job_t* jobs, int n_jobs, const void* g, int64_t ldg,
                                       const void* x, int64_t ldx, int64_t N, int C, int Cp, int H, const int32_t* ptr_true, int B,
                                       const int32_t* gid, const int32_t* sid, const float* inv_cnt, int prec_bwd, void* dx,
                                       int64_t lddx, void* scratch0, void* scratch1, size_t bwd_ws_bytes, stin_stream_t stream,
                                       stin_stream_t wgrad_stream) {
    STIN_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || jobs != nullptr), STIN_E_NULL);
    STIN_REQUIRE(n_jobs <= 1 || (scratch0 != nullptr && scratch1 != nullptr && Cp == C), STIN_E_NULL);
    const int Yw = 2 * H;
    const int pad = storage ? 8 : 4;
    const void* gi = g;
    int64_t ldgi = ldg;
    for (int i = n_jobs - 1; i >= 0; --i) {
        const stin_chain_job_t& J = jobs[i];
        const void* xi = i == 0 ? x : jobs[i - 1].out;                  // the block's input = its predecessor's output
        const int64_t ldxi = i == 0 ? ldx : C;
        void* dxi = i == 0 ? dx : ((i & 1) ? scratch1 : scratch0);
        const int64_t lddxi = i == 0 ? lddx : Cp;
        STIN_TRY(stin_edgeconv_block_bwd(storage, gi, ldgi, xi, ldxi, N, C, Cp, H, C, 0, J.trans_inv, J.Y, Yw, J.hE, H + pad, J.mask,
                                         J.agg, J.mean, J.rstd, J.wcatT, J.w2T, J.rowptr_dst, J.rowptr_src, J.col_src, J.xslot,
                                         J.w_src, ptr_true, B, gid, sid, inv_cnt, prec_bwd, J.bwd_split, dxi, lddxi, J.dW1, J.db1,
                                         J.dW2, J.db2, nullptr, nullptr, J.bwd_ws, bwd_ws_bytes, stream, wgrad_stream, J.ev_dy,
                                         J.ev_dy, J.ev_done, 0));
        gi = dxi;
        ldgi = Cp;
    }
    return STIN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The graph part of the network as one op list per direction (include/stin_hip.h: stin_net_op_t).  Only loops: every op is
// one of the existing entry points with the pointers of the host array.
static_assert(sizeof(stin_net_op_t) == 16 * 4 + 8 + 7 * 8 + 2 * 8 + 42 * 8, "stin_net_op_t layout (functional._net_struct packs it)");

extern "C" int stin_net_fwd(int storage, const stin_net_op_t* ops, int n_ops, stin_stream_t stream) {
    STIN_REQUIRE(n_ops >= 0 && (n_ops == 0 || ops != nullptr), STIN_E_NULL);
    STIN_REQUIRE(storage == 0 || storage == 1, STIN_E_UNSUPPORTED);
    for (int i = 0; i < n_ops; ++i) {
        const stin_net_op_t& J = ops[i];
        if (J.kind == STIN_OP_BLOCK) {
            t_edge_ev0 = (hipEvent_t)J.ev_edge0;
            t_edge_ev1 = (hipEvent_t)J.ev_edge1;
            const int rc_blk = stin_edgeconv_block_fwd(storage, J.x, J.ldx, J.n_out, J.Cin, J.Cp, J.H, J.Cout, J.has_shortcut, J.trans_inv, J.W1,
                                             J.b1, J.W2, J.b2, J.Ws, J.bs, J.rowptr_dst, J.col_dst, J.ptr_sum, J.B, J.gid, J.inv_cnt,
                                             J.slice_quirk, J.eps, J.prec_fwd, J.fwd_split, J.bwd_split, J.wcatT, J.w2T, J.Y, J.ldy,
                                             J.hE, J.ldh, J.mask, J.agg, J.mean, J.rstd, J.out, J.ldo, J.fwd_ws, (size_t)J.fwd_ws_bytes,
                                             stream);
            t_edge_ev0 = t_edge_ev1 = nullptr;
            if (rc_blk != STIN_OK) return rc_blk;
        } else if (J.kind == STIN_OP_POOL_MAX) {
            if (storage)
                STIN_TRY(stin_pool_max_fwd_bf16(static_cast<const stin_bf16_t*>(J.x), J.ldx, J.rowptr_dst, J.col_dst, J.n_out, J.Cout,
                                                static_cast<stin_bf16_t*>(J.out), J.ldo, J.arg, stream));
            else
                STIN_TRY(stin_pool_max_fwd_f32(static_cast<const float*>(J.x), J.ldx, J.rowptr_dst, J.col_dst, J.n_out, J.Cout,
                                               static_cast<float*>(J.out), J.ldo, J.arg, stream));
        } else if (J.kind == STIN_OP_UNPOOL) {
            if (storage)
                STIN_TRY(stin_gather_rows_bf16(static_cast<const stin_bf16_t*>(J.x), J.ldx, J.trace, nullptr, J.n_out, J.Cout,
                                               static_cast<stin_bf16_t*>(J.out), J.ldo, stream));
            else
                STIN_TRY(stin_gather_rows_f32(static_cast<const float*>(J.x), J.ldx, J.trace, nullptr, J.n_out, J.Cout,
                                              static_cast<float*>(J.out), J.ldo, stream));
        } else {
            return STIN_E_UNSUPPORTED;
        }
    }
    return STIN_OK;
}

extern "C" int stin_net_bwd(int storage, const stin_net_op_t* ops, int n_ops, const void* g, int64_t ldg, int prec_bwd,
                            stin_stream_t stream, stin_stream_t wgrad_stream) {
    STIN_REQUIRE(n_ops >= 0 && (n_ops == 0 || ops != nullptr), STIN_E_NULL);
    STIN_REQUIRE(storage == 0 || storage == 1, STIN_E_UNSUPPORTED);
    const void* gi = g;
    int64_t ldgi = ldg;
    const double* pre_partial = nullptr;       // statistics of op i computed by op i + 1's dx product (BwdLink)
    int64_t pre_groups = 0;
    for (int i = n_ops - 1; i >= 0; --i) {
        const stin_net_op_t& J = ops[i];
        STIN_REQUIRE(J.dx != nullptr || i == 0, STIN_E_NULL);
        if (J.kind == STIN_OP_BLOCK) {
            t_edge_ev0 = (hipEvent_t)J.ev_edge0;
            t_edge_ev1 = (hipEvent_t)J.ev_edge1;
            BwdLink link;
            link.pre_partial = pre_partial;
            link.pre_groups = pre_groups;
            pre_partial = nullptr;
            pre_groups = 0;
            if (storage == 0 && i > 0 && J.dx != nullptr && ops[i - 1].kind == STIN_OP_BLOCK) {
                // op i - 1 is a block whose output is this block's input: single graph, no slice quirk, same rows, and this
                // block's input width is that block's output width (no channel padding in between)
                const stin_net_op_t& Pn = ops[i - 1];
                if (Pn.B == 1 && Pn.gid == nullptr && Pn.sid == nullptr && Pn.n_out == J.n_out && Pn.Cout == J.Cp && J.Cin == J.Cp &&
                    Pn.bwd_ws != nullptr) {
                    size_t rb = 0;
                    void* red = bwd_ws_red(Pn.bwd_ws, Pn.n_out, Pn.Cp, Pn.H, Pn.Cout, Pn.has_shortcut, Pn.B, storage, &rb);
                    link.next_agg = static_cast<const float*>(Pn.agg);
                    link.next_ld = Pn.Cout;
                    link.next_mean = Pn.mean;
                    link.next_rstd = Pn.rstd;
                    link.next_partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(red) + 255) & ~(uintptr_t)255);
                    link.next_partial_bytes = rb > 256 ? rb - 256 : 0;
                }
            }
            const int rc_blk = block_bwd_impl(storage, gi, ldgi, J.x, J.ldx, J.n_out, J.Cin, J.Cp, J.H, J.Cout, J.has_shortcut, J.trans_inv,
                                             J.Y, J.ldy, J.hE, J.ldh, J.mask, J.agg, J.mean, J.rstd, J.wcatT, J.w2T, J.rowptr_dst,
                                             J.rowptr_src, J.col_src, J.xslot, J.w_src, J.ptr_true, J.B, J.gid, J.sid, J.inv_cnt, prec_bwd,
                                             J.bwd_split, J.dx, J.lddx, J.dW1, J.db1, J.dW2, J.db2, J.dWs, J.dbs, J.bwd_ws,
                                             (size_t)J.bwd_ws_bytes, stream, J.use_side ? wgrad_stream : nullptr, J.ev_dy, J.ev_dy,
                                             J.ev_done, 0, &link);
            t_edge_ev0 = t_edge_ev1 = nullptr;
            if (rc_blk != STIN_OK) return rc_blk;
            if (link.produced_groups > 0) {
                pre_partial = link.next_partial;
                pre_groups = link.produced_groups;
            }
        } else if (J.kind == STIN_OP_POOL_MAX) {
            if (J.dx != nullptr) {
                if (storage)
                    STIN_TRY(stin_pool_max_bwd_bf16(static_cast<const stin_bf16_t*>(gi), ldgi, J.arg, J.trace, J.n_in, J.Cout,
                                                    static_cast<stin_bf16_t*>(J.dx), J.lddx, stream));
                else
                    STIN_TRY(stin_pool_max_bwd_f32(static_cast<const float*>(gi), ldgi, J.arg, J.trace, J.n_in, J.Cout,
                                                   static_cast<float*>(J.dx), J.lddx, stream));
            }
        } else if (J.kind == STIN_OP_UNPOOL) {
            if (J.dx != nullptr) {
                if (storage)
                    STIN_TRY(stin_segment_sum_bf16(static_cast<const stin_bf16_t*>(gi), ldgi, J.rowptr_dst, J.col_dst, J.n_in, J.Cout, 0,
                                                   static_cast<stin_bf16_t*>(J.dx), J.lddx, stream));
                else   // (non-temporal loads on a once-read source beyond the Infinity Cache, as functional.segment_sum asks for)
                    STIN_TRY(stin_segment_sum_f32(static_cast<const float*>(gi), ldgi, J.rowptr_dst, J.col_dst, J.n_in, J.Cout,
                                                  (J.n_out * (int64_t)J.Cout * 4 > ((int64_t)256 << 20)) ? STIN_SEG_NONTEMPORAL : 0,
                                                  static_cast<float*>(J.dx), J.lddx, stream));
            }
        } else {
            return STIN_E_UNSUPPORTED;
        }
        gi = J.dx;
        ldgi = J.lddx;
    }
    return STIN_OK;
}
