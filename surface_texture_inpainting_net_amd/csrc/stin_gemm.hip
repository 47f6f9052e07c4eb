// Per-vertex dense GEMMs of the STINet hot path on the gfx950 matrix cores.  Kernel families in this file:
//   k_gemm_nt / k_gemm_tn              exact fp32 (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf chain, 157 TFLOP/s peak)
//   k_gemm_nt_bf16s / k_gemm_tn_bf16s  fp32 storage, operands split on the fly into 16-bit pieces (bf16 x3 / x6, fp16 x3),
//                                      v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation - what the network runs on
//   k_gemm_nt_b16 / k_gemm_tn_b16      bf16 storage (the *_bf16 entry points), one MFMA per k-step
//
//   stin_gemm_nt_f32 : C[M, Nc] = A[M, K] . W[Nc, K]^T (+ bias)      forward GEMMs and dgrad (with W^T)
//   stin_gemm_tn_f32 : dW[Nc, K(+1)] = G[M, Nc]^T . [X[M, K] | 1]    weight (+bias) gradients, split over M
//
// M is the vertex count (1e4..1e6), Nc and K are channel counts (3..2052): tall-skinny shapes where
// library GEMMs pick poor tiles.  Fragment maps (cdna_hip_programming.md §3): A operand lane l holds
// A[i = l&31][k = l>>5], B operand B[k = l>>5][j = l&31]; C/D reg r of lane l is
// row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
#include <cstdlib>
#include <type_traits>
#include "stin_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BLOCK = 256;
constexpr int BK = 32;

// NT block -> output tile, XCD-aware: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own
// 4 MB L2.  Row tile r lives on XCD r % 8 and its column blocks get CONSECUTIVE slots of that XCD, so the A rows of a tile
// are fetched across the fabric once and then re-read from that L2 by the other column blocks (a 2-D grid dispatches
// x-fastest: the column blocks of one row tile would start thousands of workgroups apart, on different XCDs).
__device__ __forceinline__ bool nt_block_tile(int64_t M, int Nc, int BM, int BN, int64_t& m0, int& n0) {
    const int64_t L = blockIdx.x;
    const int ncol = (Nc + BN - 1) / BN;
    if ((M + BM - 1) / BM < 16) {        // too few row tiles to give every XCD its own (coarsest levels): plain order
        m0 = (L / ncol) * BM;
        n0 = (int)(L % ncol) * BN;
        return true;
    }
    const int64_t j = L >> 3;
    const int64_t rt = (j / ncol) * 8 + (L & 7);
    if (rt * BM >= M) return false;
    m0 = rt * BM;
    n0 = (int)(j % ncol) * BN;
    return true;
}
inline unsigned nt_grid(int64_t M, int Nc, int BM, int BN) {
    const int64_t nrow = (M + BM - 1) / BM, ncol = (Nc + BN - 1) / BN;
    return (unsigned)((nrow < 16 ? nrow : ((nrow + 7) / 8) * 8) * ncol);
}

#include "gemm_nt_tiled.inc"   // NT products: the exact-fp32 tiling and the split-16-bit tiling (k_gemm_nt, k_gemm_nt_bf16s)
#include "gemm_nt_stream.inc"   // NT products, streaming rows through wave-private LDS (k_gemm_nt_stream and its epilogue modes)
#include "gemm_nt_resident.inc"   // NT products on fragment-order weights: resident strip, all-columns and balanced column-panel kernels (k_gemm_nt_strip / _wide / _panel)
#include "gemm_tn.inc"   // TN (weight-gradient) products of fp32 rows: exact-fp32 and split-bf16 tilings (k_gemm_tn, k_gemm_tn_bf16s)
#include "gemm_b16.inc"   // GEMMs of bf16-STORAGE rows: NT tiled / LDS-DMA, TN register-transpose / hardware-transposed reads
#include "gemm_tn_skinny.inc"   // TN products with K <= 16 (k_gemm_tn_skinny), the slab reduction and the host-side tile / chunk rules
}  // namespace

// colstats != NULL: the launch must be the all-columns kernel (its blocks own whole rows) - STIN_E_UNSUPPORTED otherwise
// the streaming-rows kernel for a pre-split (not fragment-ordered) weight operand with the tiled kernels' epilogue; defined below
static int stream_nt_presplit_try(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* row_mask,
                                  int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M, int Nc, int K, float* C,
                                  int64_t ldc, int precision, hipStream_t stream);
static int stream_nt_epi_try(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* row_mask,
                             int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M, int Nc, int K, float* C, int64_t ldc,
                             int precision, hipStream_t stream, int wpre);

static int gemm_nt_f32_impl(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                            const float* row_mask, int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M,
                            int Nc, int K, float* C, int64_t ldc, int precision, double* colstats, stin_stream_t stream_,
                            const NtDotElu* dotelu = nullptr, const stin_bn_tf* tf = nullptr) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(M >= 0 && Nc > 0 && K > 0 && lda >= K && ldw >= K && ldc >= Nc, STIN_E_SIZE);
    STIN_REQUIRE(residual == nullptr || ld_res >= Nc, STIN_E_SIZE);
    const bool wpre = (precision & STIN_GEMM_W_PRESPLIT) != 0;
    const bool wfrag = wpre && (precision & STIN_GEMM_W_FRAG) != 0 && stin_w_frag_shape(Nc, K);
    precision &= ~(STIN_GEMM_W_PRESPLIT | STIN_GEMM_W_FRAG);
    STIN_REQUIRE(precision == STIN_GEMM_F32 || precision == STIN_GEMM_BF16X3 || precision == STIN_GEMM_BF16X6 ||
                     precision == STIN_GEMM_F16X3,
                 STIN_E_UNSUPPORTED);
    STIN_REQUIRE(!wpre || precision == STIN_GEMM_BF16X3 || precision == STIN_GEMM_F16X3, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(tf == nullptr || (!wpre && colstats == nullptr && dotelu == nullptr), STIN_E_UNSUPPORTED);   // (the tiled kernels only)
    if (M == 0) return STIN_OK;
    STIN_REQUIRE(A && W && C, STIN_E_NULL);
    bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldw % 4 == 0) && stin_aligned16(A) && stin_aligned16(W);
    if (tf != nullptr) {
        STIN_REQUIRE(tf->mean && tf->rstd && tf->gamma && tf->beta, STIN_E_NULL);
        vec = vec && stin_aligned16(tf->mean) && stin_aligned16(tf->rstd) && stin_aligned16(tf->gamma) && stin_aligned16(tf->beta);
    }
    // Tile choice, from per-shape sweeps on MI355X (profiles/gemm_tiles.py) and whole-step A/B runs: the skinny GEMMs of the
    // shipped 3-level network (K <= 256, or K >= 512 with only 256 output columns) are latency-bound, so the 64x64 tile (4x
    // the blocks in flight) wins there.  Long reductions with enough tiles (the 1024..4096-wide layers of a 5-level network:
    // K >= 512, >= 500 tiles of 128x128) are 1.15-1.35x faster on 128x128.
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Nc + bn - 1) / bn); };
    constexpr int64_t min_blocks = 500;
    const int force_tile = stin_nt_force_tile();   // tuning aid: 0 = rule above, 1 = 128x128, 2 = 128x64, 3 = 64x64
    const bool big_tile = Nc % 128 == 0 && K >= 512 && blocks(128, 128) >= min_blocks;
    stin_bn_tf tf_arg;
    tf_arg.mean = tf_arg.rstd = tf_arg.gamma = tf_arg.beta = nullptr;
    if (tf != nullptr) tf_arg = *tf;
#define STIN_NT_ARGS A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, C, ldc, tf_arg
#define STIN_NT(KERNEL, BM_, BN_, WM_, WN_, ...)                                                                  \
    do {                                                                                                          \
        dim3 grid(nt_grid(M, Nc, BM_, BN_));                                                                      \
        if (vec) hipLaunchKernelGGL((KERNEL<BM_, BN_, WM_, WN_, ##__VA_ARGS__, true>), grid, dim3(BLOCK), 0, stream, STIN_NT_ARGS); \
        else hipLaunchKernelGGL((KERNEL<BM_, BN_, WM_, WN_, ##__VA_ARGS__, false>), grid, dim3(BLOCK), 0, stream, STIN_NT_ARGS);    \
    } while (0)
#define STIN_NT_PICK(KERNEL, ...)                                                              \
    do {                                                                                       \
        if (Nc <= 32) STIN_NT(KERNEL, 128, 32, 4, 1, ##__VA_ARGS__);                           \
        else if (force_tile == 1 || (force_tile == 0 && big_tile)) STIN_NT(KERNEL, 128, 128, 2, 2, ##__VA_ARGS__); \
        else if (force_tile == 2) STIN_NT(KERNEL, 128, 64, 2, 2, ##__VA_ARGS__);  \
        else STIN_NT(KERNEL, 64, 64, 2, 2, ##__VA_ARGS__);                                     \
    } while (0)
    STIN_REQUIRE(colstats == nullptr || (wfrag && (Nc <= 256 || dotelu != nullptr)), STIN_E_UNSUPPORTED);
    const bool out16 = ldc % 4 == 0 && stin_aligned16(C) && (residual == nullptr || (ld_res % 4 == 0 && stin_aligned16(residual)));
    const int pmts = (wfrag && vec && out16 && !(colstats != nullptr && residual != nullptr && dotelu == nullptr)) ? panel_tiles(M, Nc, K) : 0;
    STIN_REQUIRE(dotelu == nullptr || (pmts > 0 && colstats != nullptr), STIN_E_UNSUPPORTED);      // (the panel kernel's epilogue only)
    NtDotElu de_arg;
    de_arg.x = nullptr;
    de_arg.ldx = 0;
    de_arg.mean = de_arg.rstd = nullptr;
    if (dotelu != nullptr) de_arg = *dotelu;
    // (stin_gemm_nt_colstats_groups promised the panel kernel's group count from the shape alone)
    STIN_REQUIRE(colstats == nullptr || pmts > 0 || !wfrag || panel_tiles(M, Nc, K) == 0, STIN_E_ALIGN);
    if (pmts > 0) {
        // balanced column-panel kernel (k_gemm_nt_panel)
        const int P = Nc / 128, bm = 32 * pmts;
        const int nrb = (int)((M + bm - 1) / bm);
        const int xmap = nrb >= 16 ? 1 : 0;
        const unsigned grid = (unsigned)((xmap ? ((nrb + 7) / 8) * 8 : nrb) * P);
        const size_t lds = (size_t)bm * 512 > (size_t)(8 * 4096 + bm * 4) ? (size_t)bm * 512 : (size_t)(8 * 4096 + bm * 4);
#define STIN_PANEL_L(PT_, A_, B_)                                                                                         \
    do {                                                                                                                  \
        static stin_once_per_device attr_once;                                                                                     \
        if (attr_once.first()) {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_panel<PT_, A_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_gemm_nt_panel<PT_, A_, B_>), dim3(grid), dim3(512), lds, stream, A, lda, W, bias, row_mask, ld_mask, \
                           residual, ld_res, M, Nc, K, C, ldc, colstats, nrb, P, xmap, de_arg);                           \
    } while (0)
#define STIN_PANEL(PT_)                                                                                                   \
    do {                                                                                                                  \
        switch (pmts) {                                                                                                   \
            case 2: STIN_PANEL_L(PT_, 1, 1); break;                                                                       \
            case 3: STIN_PANEL_L(PT_, 2, 1); break;                                                                       \
            case 4: STIN_PANEL_L(PT_, 2, 2); break;                                                                       \
            case 5: STIN_PANEL_L(PT_, 3, 2); break;                                                                       \
            case 6: STIN_PANEL_L(PT_, 3, 3); break;                                                                       \
            case 7: STIN_PANEL_L(PT_, 4, 3); break;                                                                       \
            case 8: STIN_PANEL_L(PT_, 4, 4); break;                                                                       \
            default: STIN_PANEL_L(PT_, 5, 4); break;                                                                      \
        }                                                                                                                 \
    } while (0)
        if (precision == STIN_GEMM_BF16X3) STIN_PANEL(__bf16);
        else STIN_PANEL(_Float16);
#undef STIN_PANEL_L
#undef STIN_PANEL
    } else if (wfrag && Nc <= 256) {
        // all-columns kernel (k_gemm_nt_wide): Nc = 128 / 256, the fragment-order weight operand cannot be read by any other kernel
        STIN_REQUIRE(vec, STIN_E_ALIGN);
        const int wm = wide_waves_m(M, Nc);
        // the LDS-restaged epilogue stores 16 bytes per lane: C (and the residual) rows must allow it
        const int vec_out = (ldc % 4 == 0 && stin_aligned16(C) && (residual == nullptr || (ld_res % 4 == 0 && stin_aligned16(residual))) &&
                             !(colstats != nullptr && residual != nullptr)) ? 1 : 0;
#define STIN_WIDE_L(PT_, NW_, WM_)                                                                                        \
    hipLaunchKernelGGL((k_gemm_nt_wide<PT_, NW_, WM_>), dim3((unsigned)((M + 64 * WM_ - 1) / (64 * WM_))), dim3(64 * NW_ * WM_), 0, stream, \
                       A, lda, W, bias, row_mask, ld_mask, residual, ld_res, M, K, C, ldc, colstats, vec_out)
#define STIN_WIDE(PT_, NW_)                                                                                               \
    do {                                                                                                                  \
        if (wm == 8 / NW_) STIN_WIDE_L(PT_, NW_, 8 / NW_);                                                                \
        else STIN_WIDE_L(PT_, NW_, 4 / NW_);                                                                              \
    } while (0)
        if (precision == STIN_GEMM_BF16X3) {
            if (Nc == 256) STIN_WIDE(__bf16, 4);
            else STIN_WIDE(__bf16, 2);
        } else {
            if (Nc == 256) STIN_WIDE(_Float16, 4);
            else STIN_WIDE(_Float16, 2);
        }
#undef STIN_WIDE_L
#undef STIN_WIDE
    } else if (wfrag) {
        // resident-strip kernel (k_gemm_nt_strip): the fragment-order weight operand cannot be read by any other kernel
        STIN_REQUIRE(vec && ldc % 4 == 0 && stin_aligned16(C) && (residual == nullptr || (ld_res % 4 == 0 && stin_aligned16(residual))),
                     STIN_E_ALIGN);
        const int KC = K < 256 ? K : 256;                             // resident K chunk: 64 rows x 256 k x 4 B = 64 KB -> 2 blocks per CU
        const int P = (Nc + ST_PANEL - 1) / ST_PANEL;
        const int cfg = strip_config(M, Nc, KC);                      // 21 = (MT 2, one quad), 22 = two quads, 41 = MT 4
        const int bm = cfg == 21 ? 64 : 128, waves = cfg == 22 ? 8 : 4;
        size_t lds = (size_t)2 * (KC / 32) * bm * 64 + (size_t)P * ST_PANEL * 4 + bm * 4;
        // the epilogue's restage area (2 KB per wave) only where it does not cost a resident block (K chunks of 256: 2 blocks
        // per CU either way; chunks of 128 would drop from 4 to 3 and lose more than the wide stores gain: 130 -> 155 us at
        // 200 704 x 320 x 128)
        const size_t occ_plain = 160 * 1024 / lds, occ_rest = 160 * 1024 / (lds + waves * 2048);
        const int restage = ((occ_rest > 2 ? 2 : occ_rest) == (occ_plain > 2 ? 2 : occ_plain)) ? 1 : 0;     // (occupancy is capped at 2 below)
        if (restage) lds += waves * 2048;
        const int64_t units = ((M + bm - 1) / bm) * P;
        int occ = (int)(160 * 1024 / lds);
        // (round 4) at most two resident blocks per CU: with three or four (K chunks of 128: 32-44 KB of LDS per block) the blocks
        // of a CU re-stream the W panels against each other - 200 704 x 320 x 128: 136.6 us at four / three, 126.5 at two, 124 at one
        // two-quad block (profiles/probes/nt_probe.py, STIN_STRIP_OCC sweep)
        if (occ > 2) occ = 2;
        if (cfg == 22) occ = 1;                                       // eight-wave blocks: one per CU (18 063 x 1280 x 128: 39.2 -> 35.6 us)
        if (occ < 1) occ = 1;
        int64_t grid = (int64_t)stin_cu_count() * occ;
        if (grid > units) grid = units;
#define STIN_STRIP_L(PT_, MT_, QM_)                                                                                       \
    do {                                                                                                                  \
        static stin_once_per_device attr_once;                                                                                     \
        if (attr_once.first()) {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_strip<PT_, MT_, QM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_gemm_nt_strip<PT_, MT_, QM_>), dim3((unsigned)grid), dim3(256 * QM_), lds, stream, A, lda, W, bias, row_mask, \
                           ld_mask, residual, ld_res, M, Nc, K, C, ldc, KC, units, P, restage);                          \
    } while (0)
#define STIN_STRIP(PT_)                                                                                                   \
    do {                                                                                                                  \
        if (cfg == 22) STIN_STRIP_L(PT_, 2, 2);                                                                           \
        else if (cfg == 41) STIN_STRIP_L(PT_, 4, 1);                                                                      \
        else STIN_STRIP_L(PT_, 2, 1);                                                                                     \
    } while (0)
        if (precision == STIN_GEMM_BF16X3) STIN_STRIP(__bf16);
        else STIN_STRIP(_Float16);
#undef STIN_STRIP_L
#undef STIN_STRIP
    } else if (wpre) {
        // pre-split W: the 16-byte vector path only (K % 4 == 0, aligned rows) - one tile shape, the data is per-network
        STIN_REQUIRE(vec, STIN_E_ALIGN);
        if (tf == nullptr && colstats == nullptr) {       // (round 5) very tall products: the streaming-rows kernel, bit-identical
            const int rc = stream_nt_presplit_try(A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, C, ldc, precision, stream);
            if (rc != STIN_E_UNSUPPORTED) return rc;
        }
        if (force_tile == 1 || (force_tile == 0 && big_tile)) {
            dim3 grid(nt_grid(M, Nc, 128, 128));
            if (precision == STIN_GEMM_BF16X3)
                hipLaunchKernelGGL((k_gemm_nt_bf16s<128, 128, 2, 2, 2, __bf16, true, true>), grid, dim3(BLOCK), 0, stream, STIN_NT_ARGS);
            else
                hipLaunchKernelGGL((k_gemm_nt_bf16s<128, 128, 2, 2, 2, _Float16, true, true>), grid, dim3(BLOCK), 0, stream, STIN_NT_ARGS);
        } else {
            dim3 grid(nt_grid(M, Nc, 64, 64));
            if (precision == STIN_GEMM_BF16X3)
                hipLaunchKernelGGL((k_gemm_nt_bf16s<64, 64, 2, 2, 2, __bf16, true, true>), grid, dim3(BLOCK), 0, stream, STIN_NT_ARGS);
            else
                hipLaunchKernelGGL((k_gemm_nt_bf16s<64, 64, 2, 2, 2, _Float16, true, true>), grid, dim3(BLOCK), 0, stream, STIN_NT_ARGS);
        }
    } else if (precision == STIN_GEMM_BF16X3) STIN_NT_PICK(k_gemm_nt_bf16s, 2, __bf16);
    else if (precision == STIN_GEMM_BF16X6) STIN_NT_PICK(k_gemm_nt_bf16s, 3, __bf16);
    else if (precision == STIN_GEMM_F16X3) STIN_NT_PICK(k_gemm_nt_bf16s, 2, _Float16);
    else STIN_NT_PICK(k_gemm_nt);
#undef STIN_NT_PICK
#undef STIN_NT
#undef STIN_NT_ARGS
    return stin_launch_status();
}

extern "C" int stin_gemm_nt_stream_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, int64_t M, int Nc, int K, float* C, int64_t ldc,
                                       int precision, stin_stream_t stream);
namespace {
constexpr int64_t STREAM_MIN_ROWS = 65536;      // below this the tiling's many short blocks fill the chip better than 32-row wave tiles
inline bool stream_enabled() {
    const char* e = getenv("STIN_NT_STREAM");                                  // A/B switch, re-read per call (tests flip it)
    return e == nullptr || atoi(e) != 0;
}
}  // namespace

extern "C" int stin_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                const float* row_mask, int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M,
                                int Nc, int K, float* C, int64_t ldc, int precision, stin_stream_t stream) {
    if (M >= STREAM_MIN_ROWS && bias == nullptr && row_mask == nullptr && residual == nullptr && stream_enabled()) {
        const int rc = stin_gemm_nt_stream_f32(A, lda, W, ldw, nullptr, nullptr, nullptr, nullptr, M, Nc, K, C, ldc, precision, stream);
        if (rc != STIN_E_UNSUPPORTED) return rc;                               // (plain fp32 weights, K = 64 .. 512: the streaming kernel)
    } else if (M >= STREAM_MIN_ROWS && stream_enabled()) {                      // ... with the tiled kernels' epilogue (bias [* mask], residual)
        const int rc = stream_nt_epi_try(A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, C, ldc, precision, (hipStream_t)stream, 0);
        if (rc != STIN_E_UNSUPPORTED) return rc;
    }
    return gemm_nt_f32_impl(A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, C, ldc, precision, nullptr, stream);
}

// C = relu(gamma ((A - mean) rstd) + beta) W^T: BatchNorm1d + ReLU over the columns of A applied while the rows are staged
// (SingleConvMeshNet's per-EDGE product: the normalised [E, 2 cout] matrix never exists in memory).  W plain fp32 [Nc, K].
extern "C" int stin_gemm_nt_bn_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, int64_t M, int Nc, int K, float* C, int64_t ldc,
                                   int precision, stin_stream_t stream) {
    STIN_REQUIRE((precision & (STIN_GEMM_W_PRESPLIT | STIN_GEMM_W_FRAG)) == 0, STIN_E_UNSUPPORTED);
    if (M >= STREAM_MIN_ROWS && mean != nullptr && stream_enabled()) {
        const int rc = stin_gemm_nt_stream_f32(A, lda, W, ldw, mean, rstd, gamma, beta, M, Nc, K, C, ldc, precision, stream);
        if (rc != STIN_E_UNSUPPORTED) return rc;
    }
    stin_bn_tf tf;
    tf.mean = mean;
    tf.rstd = rstd;
    tf.gamma = gamma;
    tf.beta = beta;
    return gemm_nt_f32_impl(A, lda, W, ldw, nullptr, nullptr, 0, nullptr, 0, M, Nc, K, C, ldc, precision, nullptr, stream, nullptr, &tf);
}

// dh = A W^T used ONLY as the output gradient of BatchNorm1d + ReLU over the rows X (k_gemm_nt_stream MODE 1 / 2; SingleConvMeshNet's
// per-edge backward, edge_conv_filter.py:34-44): `stats` = the product with the two column sums on its epilogue (nothing stored:
// partial [groups][2][Nc] doubles, then sums [2][Nc] floats = P | Q, the gradients of gamma | beta); `apply` = the product again,
// stored as dx = rstd gamma (d - Q / n - nhat P / n).  groups = stin_gemm_nt_bn_bwd_groups (0: shape / precision not served -
// the caller keeps stin_gemm_nt_f32 + stin_colreduce_f32(DOT_BN_RELU) + stin_bn_act_bwd_f32).  W plain fp32 [Nc, K].
namespace {
// geometry of the streaming kernel for a shape: KC (staged k chunk), NT (32-column tiles per block), LDS bytes, blocks per CU
struct StreamGeo {
    int kc, nt, ncb, bpc;
    size_t lds;
    int64_t gx;
};
inline bool stream_geo(int64_t M, int Nc, int K, bool tf, StreamGeo* g, int mode = 0, int precision = STIN_GEMM_BF16X3) {
    if (M <= 0 || Nc <= 0 || Nc % 4 != 0 || K < 64 || K > 512 || K % 64 != 0) return false;
    // staged chunk: 64 columns (8 KB per wave in flight, 32 KB of staging per block: two blocks per CU up to K = 128 -
    // 1 200 642 x 64 x 128 with the BatchNorm transform: 184 us against 212 with 128-column chunks, profiles/probes/kc_probe.sh)
    g->kc = 64;
    const char* e = getenv("STIN_NT_STREAM_NT");
    int nt = e != nullptr ? atoi(e) : 0;
    // (the statistics pass is faster with two column tiles although the A rows then come from L2 a second time: 207 against 232 us at
    // 1 200 642 x 128 x 64, 178 against 277 at 361 000 x 256 x 128 - STIN_NT_STREAM_STATS4=1 is the four-tile form, a tuning aid)
    if (nt != 2 && nt != 4) nt = (Nc <= 64 || mode == 1) ? 2 : 4;
    const size_t esz = precision == STIN_GEMM_BF16X6 ? 6 : 4;                   // bytes per operand element in LDS: 2 or 3 pieces of 16 bits
    auto lds_of = [&](int nt_) { return esz * ((size_t)K * 32 * nt_ + (size_t)128 * g->kc) + (tf ? (size_t)8 * K : 0) + (mode != 0 ? (size_t)24 * 32 * nt_ : 0); };
    if (lds_of(nt) > 160 * 1024) nt = 2;
    if (lds_of(nt) > 160 * 1024) return false;
    g->nt = nt;
    g->lds = lds_of(nt);
    if (g->lds < 4 * 2 * 32 * nt * sizeof(double)) g->lds = 4 * 2 * 32 * nt * sizeof(double);
    g->ncb = (Nc + 32 * nt - 1) / (32 * nt);
    g->bpc = (int)((160 * 1024) / g->lds);
    if (g->bpc > 2) g->bpc = 2;
    int64_t gx = (int64_t)stin_cu_count() * g->bpc / g->ncb;
    if (gx < 1) gx = 1;
    const int64_t need = ((M + 31) / 32 + 3) / 4;                                 // blocks that have a tile for every wave
    g->gx = need < gx ? need : gx;
    return true;
}
inline bool stream_ok(const float* A, int64_t lda, const float* W, int64_t ldw, const float* X, int64_t ldx, const float* C, int64_t ldc,
                      int precision) {
    return lda % 4 == 0 && ldw % 4 == 0 && stin_aligned16(A) && stin_aligned16(W) && ldx % 4 == 0 && ldc % 4 == 0 && stin_aligned16(X) &&
           stin_aligned16(C) &&
           (precision == STIN_GEMM_BF16X3 || precision == STIN_GEMM_BF16X6 || precision == STIN_GEMM_F16X3);
}
struct StreamEpi {                     // MODE 0 only: C = (acc + bias [* row_mask]) + res; wpre: W pre-split (not in fragment order)
    const float *bias = nullptr, *row_mask = nullptr, *res = nullptr;
    int64_t ld_mask = 0, ld_res = 0;
    int wpre = 0;
};
template <int MODE, bool TF>
int stream_launch(const float* A, int64_t lda, const float* W, int64_t ldw, const float* X, int64_t ldx, const stin_bn_tf& tf,
                  const float* P, const float* Q, float inv_n, int64_t M, int Nc, int K, double* partial, float* C, int64_t ldc,
                  int precision, hipStream_t stream, const StreamEpi epi = StreamEpi()) {
    StreamGeo g;
    if (!stream_geo(M, Nc, K, TF, &g, MODE, precision)) return STIN_E_UNSUPPORTED;
    const dim3 grid((unsigned)g.gx, (unsigned)g.ncb);
#define STIN_STREAM(KC_, NT_, NS_, PT_)                                                                                              \
    do {                                                                                                                             \
        static stin_once_per_device attr_once;                                                                                                \
        if (attr_once.first()) {                                                                                                             \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_stream<KC_, NT_, NS_, PT_, MODE, TF>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                                   \
        }                                                                                                                            \
        hipLaunchKernelGGL((k_gemm_nt_stream<KC_, NT_, NS_, PT_, MODE, TF>), grid, dim3(BLOCK), g.lds, stream, A, lda, W, ldw, M, Nc, K, X, \
                           ldx, tf, P, Q, inv_n, partial, C, ldc, epi.bias, epi.row_mask, epi.ld_mask, epi.res, epi.ld_res, epi.wpre);       \
    } while (0)
#define STIN_STREAM_P(KC_, NT_)                                                  \
    do {                                                                         \
        if (precision == STIN_GEMM_BF16X3) STIN_STREAM(KC_, NT_, 2, __bf16);     \
        else if (precision == STIN_GEMM_BF16X6) STIN_STREAM(KC_, NT_, 3, __bf16); \
        else STIN_STREAM(KC_, NT_, 2, _Float16);                                 \
    } while (0)
    if (g.kc == 64 && g.nt == 2) STIN_STREAM_P(64, 2);
    else if (g.kc == 64) STIN_STREAM_P(64, 4);
    else if (g.nt == 2) STIN_STREAM_P(128, 2);
    else STIN_STREAM_P(128, 4);
#undef STIN_STREAM_P
#undef STIN_STREAM
    return stin_launch_status();
}
}  // namespace
static int stream_nt_presplit_try(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* row_mask,
                                  int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M, int Nc, int K, float* C,
                                  int64_t ldc, int precision, hipStream_t stream) {
    // in-step A/B (profiles/r05_experiments_not_shipped.md): at 200 704 rows the headline step is 0.03 ms SLOWER with it (7.36-7.39
    // against 7.34 ms), at 1 M rows (config 5 in fp32) 0.35 ms faster (41.45 against 41.81 ms) - the default sits between
    const char* er = getenv("STIN_NT_STREAM_PRE_ROWS");                        // (re-read per call: tests flip it)
    const int64_t min_rows = er != nullptr ? atoll(er) : 500000;
    const char* e = getenv("STIN_NT_STREAM");
    if (M < min_rows || (e != nullptr && atoi(e) == 0)) return STIN_E_UNSUPPORTED;
    return stream_nt_epi_try(A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, C, ldc, precision, stream, 1);
}
static int stream_nt_epi_try(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* row_mask,
                             int64_t ld_mask, const float* residual, int64_t ld_res, int64_t M, int Nc, int K, float* C, int64_t ldc,
                             int precision, hipStream_t stream, int wpre) {
    StreamGeo g;
    if (!(precision == STIN_GEMM_BF16X3 || precision == STIN_GEMM_F16X3) || !stream_geo(M, Nc, K, false, &g, 0, precision) ||
        !stream_ok(A, lda, W, ldw, nullptr, 0, C, ldc, precision) || (bias != nullptr && !stin_aligned16(bias)) ||
        (residual != nullptr && (ld_res % 4 != 0 || !stin_aligned16(residual))))
        return STIN_E_UNSUPPORTED;
    StreamEpi epi;
    epi.bias = bias;
    epi.row_mask = row_mask;
    epi.ld_mask = ld_mask;
    epi.res = residual;
    epi.ld_res = ld_res;
    epi.wpre = wpre;
    stin_bn_tf tf;
    tf.mean = tf.rstd = tf.gamma = tf.beta = nullptr;
    return stream_launch<0, false>(A, lda, W, ldw, nullptr, 0, tf, nullptr, nullptr, 0.f, M, Nc, K, nullptr, C, ldc, precision, stream, epi);
}
extern "C" int64_t stin_gemm_nt_bn_bwd_groups(int64_t M, int Nc, int K, int precision) {
    if (!(precision == STIN_GEMM_BF16X3 || precision == STIN_GEMM_BF16X6 || precision == STIN_GEMM_F16X3)) return 0;
    const char* e = getenv("STIN_NT_BNBWD");                                   // A/B switch, re-read per call (tests flip it)
    if (e != nullptr && atoi(e) == 0) return 0;
    // K = 256 (Nc = 512, the 18 063-vertex level of SingleConvMeshNet): the weight slice leaves room for two column tiles per block, so the
    // A rows are re-read by 8 column blocks and the two passes (436 us) lose to the three launches (377 us, profiles/probes/bnbwd_probe.py)
    StreamGeo g;
    return (K <= 128 && stream_geo(M, Nc, K, false, &g, 1, precision)) ? g.gx : 0;
}
extern "C" int stin_gemm_nt_bn_bwd_stats_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* X, int64_t ldx,
                                             const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t M,
                                             int Nc, int K, int precision, double* partial, size_t partial_bytes, float* sums,
                                             stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(lda >= K && ldw >= K && ldx >= Nc, STIN_E_SIZE);
    StreamGeo g;
    STIN_REQUIRE(K <= 128 && stream_geo(M, Nc, K, false, &g, 1, precision) && stream_ok(A, lda, W, ldw, X, ldx, nullptr, 0, precision), STIN_E_UNSUPPORTED);
    STIN_REQUIRE(A && W && X && mean && rstd && gamma && beta && partial && sums, STIN_E_NULL);
    STIN_REQUIRE(partial_bytes >= (size_t)g.gx * 2 * (size_t)Nc * sizeof(double), STIN_E_WORKSPACE);
    stin_bn_tf tf;
    tf.mean = mean;
    tf.rstd = rstd;
    tf.gamma = gamma;
    tf.beta = beta;
    const int rc = stream_launch<1, false>(A, lda, W, ldw, X, ldx, tf, nullptr, nullptr, 0.f, M, Nc, K, partial, nullptr, 0, precision,
                                           (hipStream_t)stream);
    if (rc != STIN_OK) return rc;
    hipLaunchKernelGGL(k_partial_sums_final, dim3((unsigned)((2 * Nc + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, partial, g.gx, 2 * Nc,
                       sums);
    return stin_launch_status();
}
extern "C" int stin_gemm_nt_bn_bwd_apply_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* X, int64_t ldx,
                                             const float* mean, const float* rstd, const float* gamma, const float* beta,
                                             const float* sums, float inv_n, int64_t M, int Nc, int K, float* dx, int64_t lddx,
                                             int precision, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(lda >= K && ldw >= K && ldx >= Nc && lddx >= Nc, STIN_E_SIZE);
    StreamGeo g;
    STIN_REQUIRE(K <= 128 && stream_geo(M, Nc, K, false, &g, 2, precision) && stream_ok(A, lda, W, ldw, X, ldx, dx, lddx, precision), STIN_E_UNSUPPORTED);
    STIN_REQUIRE(A && W && X && mean && rstd && gamma && beta && sums && dx, STIN_E_NULL);
    stin_bn_tf tf;
    tf.mean = mean;
    tf.rstd = rstd;
    tf.gamma = gamma;
    tf.beta = beta;
    return stream_launch<2, false>(A, lda, W, ldw, X, ldx, tf, sums, sums + Nc, inv_n, M, Nc, K, nullptr, dx, lddx, precision,
                                   (hipStream_t)stream);
}
// The plain product C = A W^T (mean == NULL) or C = relu(gamma ((A - mean) rstd) + beta) W^T on the streaming kernel; returns
// STIN_E_UNSUPPORTED for the shapes it does not serve (K not 64 / 128 / 256, rows not 16-byte aligned, a pre-split precision flag):
// stin_gemm_nt_f32 / stin_gemm_nt_bn_f32 try it first for M >= 65 536 rows (STIN_NT_STREAM=0: never).
extern "C" int stin_gemm_nt_stream_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, int64_t M, int Nc, int K, float* C, int64_t ldc,
                                       int precision, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(lda >= K && ldw >= K && ldc >= Nc, STIN_E_SIZE);
    StreamGeo g;
    STIN_REQUIRE(stream_geo(M, Nc, K, mean != nullptr, &g, 0, precision) && stream_ok(A, lda, W, ldw, nullptr, 0, C, ldc, precision), STIN_E_UNSUPPORTED);
    STIN_REQUIRE(A && W && C && (mean == nullptr || (rstd && gamma && beta)), STIN_E_NULL);
    stin_bn_tf tf;
    tf.mean = mean;
    tf.rstd = rstd;
    tf.gamma = gamma;
    tf.beta = beta;
    if (mean != nullptr)
        return stream_launch<0, true>(A, lda, W, ldw, nullptr, 0, tf, nullptr, nullptr, 0.f, M, Nc, K, nullptr, C, ldc, precision,
                                      (hipStream_t)stream);
    return stream_launch<0, false>(A, lda, W, ldw, nullptr, 0, tf, nullptr, nullptr, 0.f, M, Nc, K, nullptr, C, ldc, precision,
                                   (hipStream_t)stream);
}

// The GEMM plus the FIRST stage of the instance-norm statistics of its output: colstats [groups][2][Nc] doubles, groups =
// ceil(M / 64) (stin_gemm_nt_colstats_groups; 0 = this shape / precision does not support it: only the all-columns kernel's
// blocks own whole rows).  Second stage: stin_moments_final_f32.
extern "C" int64_t stin_gemm_nt_colstats_groups(int64_t M, int Nc, int K, int precision) {
    const bool ok = (precision & STIN_GEMM_W_PRESPLIT) && (precision & STIN_GEMM_W_FRAG) && Nc <= 256 && stin_w_frag_shape(Nc, K);
    const int p = precision & ~(STIN_GEMM_W_PRESPLIT | STIN_GEMM_W_FRAG);
    if (!ok || M <= 0 || (p != STIN_GEMM_BF16X3 && p != STIN_GEMM_F16X3)) return 0;
    const int pmts = panel_tiles(M, Nc, K);                 // (the colstats launches carry 16-byte aligned rows: the same choice
    if (pmts > 0) return 2 * ((M + 32 * pmts - 1) / (32 * pmts));   //  as gemm_nt_f32_impl) two row groups per panel-kernel row block
    return (M + 63) / 64;                                   // one statistics group per wave row group (64 rows)
}
extern "C" int stin_gemm_nt_colstats_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                         const float* row_mask, int64_t ld_mask, const float* residual, int64_t ld_res,
                                         int64_t M, int Nc, int K, float* C, int64_t ldc, int precision, double* colstats,
                                         size_t colstats_bytes, stin_stream_t stream) {
    const int64_t groups = stin_gemm_nt_colstats_groups(M, Nc, K, precision);
    STIN_REQUIRE(groups > 0, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(colstats != nullptr, STIN_E_NULL);
    STIN_REQUIRE(colstats_bytes >= (size_t)groups * 2 * (size_t)Nc * sizeof(double), STIN_E_WORKSPACE);
    return gemm_nt_f32_impl(A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, C, ldc, precision, colstats, stream);
}

// The GEMM plus the first stage of the instance-norm + ELU BACKWARD statistics of the layer whose output gradient it produces
// (NtDotElu above): partial [groups][2][Nc] doubles, groups = stin_gemm_nt_dotelu_groups (0: this shape / precision is not served
// by the panel kernel - the caller keeps its separate reduction).  Second stage: stin_norm_coef_from_partials_f32.
extern "C" int64_t stin_gemm_nt_dotelu_groups(int64_t M, int Nc, int K, int precision) {
    const bool ok = (precision & STIN_GEMM_W_PRESPLIT) && (precision & STIN_GEMM_W_FRAG) && stin_w_frag_shape(Nc, K);
    const int p = precision & ~(STIN_GEMM_W_PRESPLIT | STIN_GEMM_W_FRAG);
    if (!ok || M <= 0 || (p != STIN_GEMM_BF16X3 && p != STIN_GEMM_F16X3)) return 0;
    const char* e = getenv("STIN_DOTELU_FUSED");                           // A/B switch, re-read per call (tests flip it)
    if (e != nullptr && atoi(e) == 0) return 0;
    const int pmts = panel_tiles(M, Nc, K);
    return pmts > 0 ? 2 * ((M + 32 * pmts - 1) / (32 * pmts)) : 0;
}
extern "C" int stin_gemm_nt_dotelu_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                       const float* residual, int64_t ld_res, int64_t M, int Nc, int K, float* C, int64_t ldc,
                                       int precision, const float* nx, int64_t ld_nx, const float* nmean, const float* nrstd,
                                       double* partial, size_t partial_bytes, stin_stream_t stream) {
    const int64_t groups = stin_gemm_nt_dotelu_groups(M, Nc, K, precision);
    STIN_REQUIRE(groups > 0, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(nx && nmean && nrstd && partial, STIN_E_NULL);
    STIN_REQUIRE(ld_nx >= Nc, STIN_E_SIZE);
    STIN_REQUIRE(ld_nx % 4 == 0 && stin_aligned16(nx) && stin_aligned16(nmean) && stin_aligned16(nrstd), STIN_E_ALIGN);
    STIN_REQUIRE(partial_bytes >= (size_t)groups * 2 * (size_t)Nc * sizeof(double), STIN_E_WORKSPACE);
    NtDotElu de;
    de.x = nx;
    de.ldx = ld_nx;
    de.mean = nmean;
    de.rstd = nrstd;
    return gemm_nt_f32_impl(A, lda, W, ldw, bias, nullptr, 0, residual, ld_res, M, Nc, K, C, ldc, precision, partial, stream, &de);
}

extern "C" size_t stin_gemm_tn_workspace_bytes(int64_t M, int Nc, int K, int ones_column) {
    if (M < 0 || Nc <= 0 || K <= 0) return 0;
    (void)ones_column;
    const int TI = tn_tile(Nc), TJ = tn_tile(K);
    const int tiles = ((Nc + TI - 1) / TI) * ((K + TJ - 1) / TJ);
    int64_t chunks = 1;
    for (int rule = 0; rule < 2; ++rule) {                       // (the chunk rule depends on storage / precision: take the larger)
        const int rows = tn_rows_per_chunk(M, tiles, rule == 1);
        const int64_t c = (M + rows - 1) / rows;
        if (c > chunks) chunks = c;
    }
    if (K <= 16) {                                               // k_gemm_tn_skinny's chunking (fp32 rows)
        const int rows = tn_skinny_rows(M);
        const int64_t c = (M + rows - 1) / rows;
        if (c > chunks) chunks = c;
    }
    if (Nc >= 256 && K >= 256) {                                 // the 256 x 256 tiles of the bf16-storage products: fewer tiles, more chunks
        const int rows = tn_rows_per_chunk(M, ((Nc + 255) / 256) * ((K + 255) / 256), true);
        const int64_t c = (M + rows - 1) / rows;
        if (c > chunks) chunks = c;
    }
    return (size_t)chunks * (size_t)tn_chunk_stride(Nc, (K + 3) & ~3) * sizeof(float) + 256;
}

// Geometry of one TN product (shared with stin_wgrad.hip).  ws_eligible: fp32 storage, 2-piece bf16 split, 128 x 128 tiles,
// 16-byte rows - what k_gemm_tn_ws is written for (STIN_TN_WS=0 keeps the four-wave kernel, A/B aid).
int stin_tn_problem_init(stin_tn_problem* p, int storage, const void* G, int64_t ldg, const void* X, int64_t ldx, int64_t M,
                         int Nc, int K, int ones_column, const void* row_w, int64_t ld_w, int precision, float* slab,
                         int* ws_eligible) {
    STIN_REQUIRE(M >= 0 && Nc > 0 && K > 0 && ldg >= Nc && ldx >= K, STIN_E_SIZE);
    STIN_REQUIRE(slab && (M == 0 || (G && X)), STIN_E_NULL);
    p->xtf.mean = p->xtf.rstd = p->xtf.gamma = p->xtf.beta = nullptr;      // (set by the caller after init: stin_gemm_tn_bn_f32)
    p->G = static_cast<const float*>(G);
    p->X = static_cast<const float*>(X);
    p->row_w = static_cast<const float*>(row_w);
    p->slab = slab;
    p->ldg = ldg;
    p->ldx = ldx;
    p->ld_w = ld_w;
    p->M = M;
    p->Nc = Nc;
    p->K = K;
    const int a16 = storage ? 8 : 4;                               // elements per 16-byte vector
    const bool vec16 = (Nc % a16 == 0) && (K % a16 == 0) && (ldg % a16 == 0) && (ldx % a16 == 0) && stin_aligned16(G) && stin_aligned16(X);
    const bool big = storage == 1 && vec16 && tn_b16_big_tile(Nc, K);
    p->TI = big ? 256 : tn_tile(Nc);
    p->TJ = big ? 256 : tn_tile(K);
    // (round 5) fp32 bf16x3 products whose narrow side is <= 64 columns (the level-0 dW2 = dagg^T h: 64 x 128; SingleConvMeshNet's
    // per-edge dW2) ran on the four-wave kernel (its 64-wide tiles) at ~3 TB/s; on the producer / consumer kernel the same
    // product is a 128 x 128 tile that is half or a quarter empty - the wasted MFMAs are free beside the operand stream
    // (200 704 x 64 x 128: 50 -> 3x us; 1.2 M x 64 x 128: 267 -> 1xx us).  One tile either way: same chunks, same slabs.
    if (!big && storage == 0 && precision == STIN_GEMM_BF16X3 && vec16 && M > 0 && Nc >= 32 && K >= 32 && stin_tn_ws_enabled()) {
        p->TI = 128;
        p->TJ = 128;
    }
    p->tiles_i = (Nc + p->TI - 1) / p->TI;
    p->tiles_j = (K + p->TJ - 1) / p->TJ;
    p->rows_per_chunk = tn_rows_per_chunk(M, p->tiles_i * p->tiles_j, big || tn_one_per_cu(storage, precision, p->TI, p->TJ, M));
    if (tn_skinny_shape(storage, Nc, K, ldg, ldx, G, X)) {         // k_gemm_tn_skinny: a block owns all columns of its row chunk
        p->TI = p->TJ = 0;
        p->tiles_i = p->tiles_j = 1;
        p->rows_per_chunk = tn_skinny_rows(M);
    }
    p->chunks = M > 0 ? (M + p->rows_per_chunk - 1) / p->rows_per_chunk : 0;
    p->Kq = (K + 3) & ~3;
    p->has_bias = ones_column ? 1 : 0;
    p->block0 = 0;
    const int a = storage ? 8 : 4;                                 // elements per 16-byte vector
    p->vec = (Nc % a == 0) && (K % a == 0) && (ldg % a == 0) && (ldx % a == 0) && stin_aligned16(G) && stin_aligned16(X);
    if (ws_eligible)
        *ws_eligible = (storage == 0 && precision == STIN_GEMM_BF16X3 && p->TI == 128 && p->TJ == 128 && p->vec && M > 0 &&
                        stin_tn_ws_enabled()) ? 1 : 0;
    return STIN_OK;
}

// The TN kernel of one product: partial slabs only (k_reduce_slabs / k_wgrad_finalize add them).
int stin_tn_slabs(const stin_tn_problem* p, int storage, int precision, stin_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (p->chunks <= 0) return STIN_OK;
    const int64_t M = p->M, ldg = p->ldg, ldx = p->ldx, ld_weight = p->ld_w, chunks = p->chunks;
    const int Nc = p->Nc, K = p->K, Kq = p->Kq, has_bias = p->has_bias, rows = p->rows_per_chunk, tiles_i = p->tiles_i,
              tiles_j = p->tiles_j, TI = p->TI, TJ = p->TJ;
    const bool vec = p->vec != 0;
    float* slab = p->slab;
    const int64_t blocks = (chunks >= 8 ? ((chunks + 7) / 8) * 8 : chunks) * (int64_t)tiles_i * tiles_j;   // 8 chunks (one per XCD) per round
    if (storage == 1) {
        const stin_bf16* G = reinterpret_cast<const stin_bf16*>(p->G);
        const stin_bf16* X = reinterpret_cast<const stin_bf16*>(p->X);
        const stin_bf16* row_weight = reinterpret_cast<const stin_bf16*>(p->row_w);
#define STIN_TNK(TI_, TJ_)                                                                                            \
    do {                                                                                                              \
        if (vec) hipLaunchKernelGGL((k_gemm_tn_b16<TI_, TJ_, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, stream, G, ldg, X, ldx, M, Nc, K, Kq, has_bias, row_weight, ld_weight, rows, tiles_i, tiles_j, chunks, slab); \
        else hipLaunchKernelGGL((k_gemm_tn_b16<TI_, TJ_, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, stream, G, ldg, X, ldx, M, Nc, K, Kq, has_bias, row_weight, ld_weight, rows, tiles_i, tiles_j, chunks, slab);    \
    } while (0)
#define STIN_TNTR(WI_, WJ_, MT_, NT_)                                                                                  \
    do {                                                                                                              \
        typedef TrGeom<WI_, WJ_, MT_, NT_> Geo_;                                                                      \
        static stin_once_per_device attr_once;                                                                                 \
        if (attr_once.first()) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)k_gemm_tn_b16_tr<WI_, WJ_, MT_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo_::LDS); \
        }                                                                                                             \
        hipLaunchKernelGGL((k_gemm_tn_b16_tr<WI_, WJ_, MT_, NT_>), dim3((unsigned)blocks), dim3(Geo_::THREADS), Geo_::LDS, stream, G, ldg, X, \
                           ldx, M, Nc, K, Kq, has_bias, row_weight, ld_weight, rows, tiles_i, tiles_j, chunks, slab);  \
    } while (0)
        if (TI == 256 && TJ == 256) {
            STIN_TNTR(2, 4, 4, 2);
        } else if (TI == 128 && TJ == 128 && vec && Nc % 8 == 0 && K % 8 == 0 && tn_b16_tr_enabled(Nc, K)) {
            STIN_TNTR(2, 2, 2, 2);
        } else if (TI == 128 && TJ == 128) STIN_TNK(128, 128);
        else if (TI == 128) STIN_TNK(128, 64);
        else if (TJ == 128) STIN_TNK(64, 128);
        else STIN_TNK(64, 64);
#undef STIN_TNK
#undef STIN_TNTR
        return stin_launch_status();
    }
    const float *G = p->G, *X = p->X, *row_weight = p->row_w;
    if (TI == 0) {                                                 // skinny K (set by stin_tn_problem_init): exact fp32 on the VALU
        const int kp = (K + 3) & ~3;
        const size_t lds_x = (size_t)rows * (kp + 4) * sizeof(float), lds_o = (size_t)Nc * (kp + 1) * sizeof(float);
        const size_t lds = lds_x > lds_o ? lds_x : lds_o;
#define STIN_TNS(KP_)                                                                                                 \
    hipLaunchKernelGGL((k_gemm_tn_skinny<KP_>), dim3((unsigned)chunks), dim3(BLOCK), lds, stream, G, ldg, X, ldx, M, Nc, K, Kq, has_bias, \
                       row_weight, ld_weight, rows, slab)
        if (K <= 4) STIN_TNS(4);
        else if (K <= 8) STIN_TNS(8);
        else if (K <= 12) STIN_TNS(12);
        else STIN_TNS(16);
#undef STIN_TNS
        return stin_launch_status();
    }
    if (precision == STIN_GEMM_BF16X3 && TI == 128 && TJ == 128 && vec && stin_tn_ws_enabled()) {
        stin_tn_batch batch;
        batch.p[0] = *p;
        batch.p[1] = *p;
        batch.n = 1;
        return stin_tn_ws_launch(batch, stream_);
    }
#define STIN_TN(TI_, TJ_)                                                                                            \
    do {                                                                                                             \
        if (vec) hipLaunchKernelGGL((k_gemm_tn<TI_, TJ_, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, stream, G, ldg, X, ldx, M, Nc, K, Kq, has_bias, row_weight, ld_weight, rows, tiles_i, tiles_j, chunks, slab, p->xtf); \
        else hipLaunchKernelGGL((k_gemm_tn<TI_, TJ_, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, stream, G, ldg, X, ldx, M, Nc, K, Kq, has_bias, row_weight, ld_weight, rows, tiles_i, tiles_j, chunks, slab, p->xtf);    \
    } while (0)
#define STIN_TNB(TI_, TJ_, NS_)                                                                                      \
    do {                                                                                                             \
        if (vec) hipLaunchKernelGGL((k_gemm_tn_bf16s<TI_, TJ_, NS_, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, stream, G, ldg, X, ldx, M, Nc, K, Kq, has_bias, row_weight, ld_weight, rows, tiles_i, tiles_j, chunks, slab, p->xtf); \
        else hipLaunchKernelGGL((k_gemm_tn_bf16s<TI_, TJ_, NS_, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, stream, G, ldg, X, ldx, M, Nc, K, Kq, has_bias, row_weight, ld_weight, rows, tiles_i, tiles_j, chunks, slab, p->xtf);    \
    } while (0)
#define STIN_TN_PICK(LAUNCH, ...)                                   \
    do {                                                            \
        if (TI == 128 && TJ == 128) LAUNCH(128, 128, ##__VA_ARGS__); \
        else if (TI == 128) LAUNCH(128, 64, ##__VA_ARGS__);          \
        else if (TJ == 128) LAUNCH(64, 128, ##__VA_ARGS__);          \
        else LAUNCH(64, 64, ##__VA_ARGS__);                          \
    } while (0)
    if (precision == STIN_GEMM_BF16X3) STIN_TN_PICK(STIN_TNB, 2);
    else if (precision == STIN_GEMM_BF16X6) STIN_TN_PICK(STIN_TNB, 3);
    else STIN_TN_PICK(STIN_TN);
#undef STIN_TN_PICK
#undef STIN_TNB
#undef STIN_TN
    return stin_launch_status();
}

static int gemm_tn_f32_impl(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int Nc, int K,
                           int ones_column, const float* row_weight, int64_t ld_weight, float* dW, int64_t lddw, float* db,
                           int precision, void* workspace, size_t workspace_bytes, stin_stream_t stream_,
                           const stin_bn_tf* xtf = nullptr) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    const int Kp = K + ((ones_column && db == nullptr) ? 1 : 0);
    STIN_REQUIRE(M >= 0 && Nc > 0 && K > 0 && ldg >= Nc && ldx >= K && lddw >= Kp, STIN_E_SIZE);
    STIN_REQUIRE(dW && workspace && (M == 0 || (G && X)), STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_gemm_tn_workspace_bytes(M, Nc, K, ones_column), STIN_E_WORKSPACE);
    STIN_REQUIRE(precision == STIN_GEMM_F32 || precision == STIN_GEMM_BF16X3 || precision == STIN_GEMM_BF16X6,
                 STIN_E_UNSUPPORTED);
    float* slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    stin_tn_problem p;
    int rc = stin_tn_problem_init(&p, 0, G, ldg, X, ldx, M, Nc, K, ones_column, row_weight, ld_weight, precision, slab, nullptr);
    if (rc != STIN_OK) return rc;
    // the slab the kernels will write must fit what the caller was told to allocate (tile choice and workspace bound are two functions)
    STIN_REQUIRE((size_t)p.chunks * (size_t)tn_chunk_stride(Nc, p.Kq) * sizeof(float) + 256 <= workspace_bytes, STIN_E_WORKSPACE);
    if (xtf != nullptr) {                     // X read as relu(bn(X)) per column (the tiled kernels and the producer / consumer kernel)
        STIN_REQUIRE(xtf->mean && xtf->rstd && xtf->gamma && xtf->beta, STIN_E_NULL);
        STIN_REQUIRE(p.TI != 0, STIN_E_UNSUPPORTED);                                      // (not the skinny-K kernel)
        p.xtf = *xtf;
        if (!(stin_aligned16(xtf->mean) && stin_aligned16(xtf->rstd) && stin_aligned16(xtf->gamma) && stin_aligned16(xtf->beta))) p.vec = 0;
    }
    rc = stin_tn_slabs(&p, 0, precision, stream_);
    if (rc != STIN_OK) return rc;
    const int64_t n4 = (int64_t)Nc * p.Kq / 4 + (p.has_bias ? (Nc + 3) / 4 : 0);
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n4 + RS_COLS - 1) / RS_COLS)), dim3(BLOCK), 0, stream, slab, p.chunks, Nc, K,
                       p.Kq, p.has_bias, dW, lddw, db);
    return stin_launch_status();
}
extern "C" int stin_gemm_tn_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int Nc, int K,
                                int ones_column, const float* row_weight, int64_t ld_weight, float* dW, int64_t lddw,
                                int precision, void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    return gemm_tn_f32_impl(G, ldg, X, ldx, M, Nc, K, ones_column, row_weight, ld_weight, dW, lddw, nullptr, precision, workspace,
                            workspace_bytes, stream_);
}
// dW [Nc, K] = G^T relu(gamma ((X - mean) rstd) + beta): the weight gradient of SingleConvMeshNet's per-edge Linear from the
// PRE-normalisation rows (the BatchNorm1d + ReLU of stin_gemm_nt_bn_f32 applied to the X operand while it is staged)
extern "C" int stin_gemm_tn_bn_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, int64_t M, int Nc, int K, float* dW, int64_t lddw,
                                   int precision, void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_bn_tf tf;
    tf.mean = mean;
    tf.rstd = rstd;
    tf.gamma = gamma;
    tf.beta = beta;
    return gemm_tn_f32_impl(G, ldg, X, ldx, M, Nc, K, 0, nullptr, 0, dW, lddw, nullptr, precision, workspace, workspace_bytes, stream_, &tf);
}
// weight gradient dW [Nc, K] (row pitch lddw >= K) and bias gradient db [Nc] as SEPARATE destinations - e.g. the views of an
// nn.Linear's weight.grad / bias.grad in a flat gradient bucket: no [Nc, K + 1] intermediate and no slicing copies afterwards
extern "C" int stin_gemm_tn_wb_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int Nc, int K,
                                   const float* row_weight, int64_t ld_weight, float* dW, int64_t lddw, float* db, int precision,
                                   void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    STIN_REQUIRE(db != nullptr, STIN_E_NULL);
    return gemm_tn_f32_impl(G, ldg, X, ldx, M, Nc, K, 1, row_weight, ld_weight, dW, lddw, db, precision, workspace, workspace_bytes,
                            stream_);
}

// ------------------------------------------------------------------ bf16-storage entry points
extern "C" int stin_gemm_nt_bf16(const stin_bf16_t* A_, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                 const stin_bf16_t* row_mask_, int64_t ld_mask, const stin_bf16_t* residual_,
                                 int64_t ld_res, int64_t M, int Nc, int K, void* C, int64_t ldc, int c_is_f32,
                                 stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    const stin_bf16* A = reinterpret_cast<const stin_bf16*>(A_);
    const stin_bf16* row_mask = reinterpret_cast<const stin_bf16*>(row_mask_);
    const stin_bf16* residual = reinterpret_cast<const stin_bf16*>(residual_);
    STIN_REQUIRE(M >= 0 && Nc > 0 && K > 0 && lda >= K && ldw >= K && ldc >= Nc, STIN_E_SIZE);
    STIN_REQUIRE(residual == nullptr || ld_res >= Nc, STIN_E_SIZE);
    if (M == 0) return STIN_OK;
    STIN_REQUIRE(A && W && C, STIN_E_NULL);
    const bool wb = (c_is_f32 & STIN_GEMM_W_BF16) != 0;          // flags: bit 0 = fp32 output, STIN_GEMM_W_BF16 = bf16 weight operand
    c_is_f32 &= 1;
    const bool vec = (K % 8 == 0) && (lda % 8 == 0) && (ldw % (wb ? 8 : 4) == 0) && stin_aligned16(A) && stin_aligned16(W);
    STIN_REQUIRE(!wb || vec, STIN_E_ALIGN);
    const int vec_out = (!c_is_f32 && Nc % 8 == 0 && ldc % 8 == 0 && stin_aligned16(C)) ? 1 : 0;
    // 64x64 is the best tile for every shape of the shipped 3-level network; long reductions with enough tiles (the wide
    // layers of a 5-level network) are 1.1-1.5x faster on 128x64 (profiles/gemm_tiles.py)
    const int force_tile = stin_nt_force_tile();
    const bool tall_tile = K >= 512 && ((M + 127) / 128) * ((Nc + 63) / 64) >= 1000;
#define STIN_NTB(BM_, BN_, WM_, WN_, OUT_)                                                                             \
    do {                                                                                                               \
        dim3 grid(nt_grid(M, Nc, BM_, BN_));                                                                           \
        if (wb) hipLaunchKernelGGL((k_gemm_nt_b16<BM_, BN_, WM_, WN_, OUT_, true, true>), grid, dim3(BLOCK), 0, stream, A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, (OUT_*)C, ldc, vec_out); \
        else if (vec) hipLaunchKernelGGL((k_gemm_nt_b16<BM_, BN_, WM_, WN_, OUT_, true>), grid, dim3(BLOCK), 0, stream, A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, (OUT_*)C, ldc, vec_out); \
        else hipLaunchKernelGGL((k_gemm_nt_b16<BM_, BN_, WM_, WN_, OUT_, false>), grid, dim3(BLOCK), 0, stream, A, lda, W, ldw, bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, (OUT_*)C, ldc, vec_out);    \
    } while (0)
#define STIN_NTB_PICK(OUT_)                                                                          \
    do {                                                                                             \
        if (Nc <= 32) STIN_NTB(128, 32, 4, 1, OUT_);                                                 \
        else if (force_tile == 1) STIN_NTB(128, 128, 2, 2, OUT_);                                    \
        else if (force_tile == 2 || (force_tile == 0 && tall_tile)) STIN_NTB(128, 64, 2, 2, OUT_);   \
        else STIN_NTB(64, 64, 2, 2, OUT_);                                                           \
    } while (0)
    // fat shapes with bf16 weights: the LDS-DMA kernel (STIN_NT_GLDS=0 keeps the register-staged tiles: tuning aid / A-B)
    if (wb && K % BKB == 0 && nt_b16_glds_pays(M, Nc, K)) {
        const stin_bf16* Wb = reinterpret_cast<const stin_bf16*>(W);
#define STIN_GLDS(OUT_, WM_, WN_, MT_, NT_)                                                                             \
    do {                                                                                                               \
        typedef GlGeom<WM_, WN_, MT_, NT_> Geo_;                                                                       \
        static stin_once_per_device attr_once;                                                                                  \
        if (attr_once.first()) {                                                                                               \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_b16_glds<OUT_, WM_, WN_, MT_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo_::LDS); \
        }                                                                                                              \
        const int64_t nrow = (M + Geo_::BM - 1) / Geo_::BM, ncol = (Nc + Geo_::BN - 1) / Geo_::BN;                     \
        dim3 grid(ncol % 8 == 0 ? (unsigned)(nrow * ncol) : nt_grid(M, Nc, Geo_::BM, Geo_::BN));                       \
        hipLaunchKernelGGL((k_gemm_nt_b16_glds<OUT_, WM_, WN_, MT_, NT_>), grid, dim3(Geo_::THREADS), Geo_::LDS, stream, A, lda, Wb, ldw, \
                           bias, row_mask, ld_mask, residual, ld_res, M, Nc, K, (OUT_*)C, ldc, vec_out);                \
    } while (0)
        const bool big = nt_b16_big_tile(M, Nc, K);
        if (c_is_f32) {
            if (big) STIN_GLDS(float, 2, 4, 4, 2);
            else STIN_GLDS(float, 2, 2, 2, 2);
        } else {
            if (big) STIN_GLDS(stin_bf16, 2, 4, 4, 2);
            else STIN_GLDS(stin_bf16, 2, 2, 2, 2);
        }
#undef STIN_GLDS
        return stin_launch_status();
    }
    if (c_is_f32) STIN_NTB_PICK(float);
    else STIN_NTB_PICK(stin_bf16);
#undef STIN_NTB_PICK
#undef STIN_NTB
    return stin_launch_status();
}

static int gemm_tn_bf16_impl(const stin_bf16_t* G_, int64_t ldg, const stin_bf16_t* X_, int64_t ldx, int64_t M, int Nc,
                            int K, int ones_column, const stin_bf16_t* row_weight_, int64_t ld_weight, float* dW,
                            int64_t lddw, float* db, void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    const int Kp = K + ((ones_column && db == nullptr) ? 1 : 0);
    STIN_REQUIRE(M >= 0 && Nc > 0 && K > 0 && ldg >= Nc && ldx >= K && lddw >= Kp, STIN_E_SIZE);
    STIN_REQUIRE(dW && workspace && (M == 0 || (G_ && X_)), STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_gemm_tn_workspace_bytes(M, Nc, K, ones_column), STIN_E_WORKSPACE);
    float* slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    stin_tn_problem p;
    int rc = stin_tn_problem_init(&p, 1, G_, ldg, X_, ldx, M, Nc, K, ones_column, row_weight_, ld_weight, STIN_GEMM_BF16X3, slab, nullptr);
    if (rc != STIN_OK) return rc;
    rc = stin_tn_slabs(&p, 1, STIN_GEMM_BF16X3, stream_);
    if (rc != STIN_OK) return rc;
    const int64_t n4 = (int64_t)Nc * p.Kq / 4 + (p.has_bias ? (Nc + 3) / 4 : 0);
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n4 + RS_COLS - 1) / RS_COLS)), dim3(BLOCK), 0, stream, slab, p.chunks, Nc, K,
                       p.Kq, p.has_bias, dW, lddw, db);
    return stin_launch_status();
}
extern "C" int stin_gemm_tn_bf16(const stin_bf16_t* G_, int64_t ldg, const stin_bf16_t* X_, int64_t ldx, int64_t M, int Nc,
                                 int K, int ones_column, const stin_bf16_t* row_weight_, int64_t ld_weight, float* dW,
                                 int64_t lddw, void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    return gemm_tn_bf16_impl(G_, ldg, X_, ldx, M, Nc, K, ones_column, row_weight_, ld_weight, dW, lddw, nullptr, workspace,
                             workspace_bytes, stream_);
}
extern "C" int stin_gemm_tn_wb_bf16(const stin_bf16_t* G_, int64_t ldg, const stin_bf16_t* X_, int64_t ldx, int64_t M, int Nc,
                                    int K, const stin_bf16_t* row_weight_, int64_t ld_weight, float* dW, int64_t lddw, float* db,
                                    void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    STIN_REQUIRE(db != nullptr, STIN_E_NULL);
    return gemm_tn_bf16_impl(G_, ldg, X_, ldx, M, Nc, K, 1, row_weight_, ld_weight, dW, lddw, db, workspace, workspace_bytes, stream_);
}
