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

// ----------------------------------------------------------------------------- NT
// Block tile BM x BN, 4 waves as WM x WN, each wave (BM/WM) x (BN/WN) = MT x NT MFMA tiles of 32x32.
// A and W tiles are staged K-major in LDS ([BK][rows + 1]: conflict-free transposed stores and
// stride-1 fragment reads); the next tile's global loads are issued before the current tile's MFMAs.
template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_gemm_nt(const float* __restrict__ A, int64_t lda,
                                                   const float* __restrict__ W, int64_t ldw,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ row_mask, int64_t ld_mask,
                                                   const float* __restrict__ res, int64_t ld_res, int64_t M,
                                                   int Nc, int K, float* __restrict__ C, int64_t ldc, const stin_bn_tf tf) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int A_F4 = BM * BK / 4 / BLOCK, W_F4 = BN * BK / 4 / BLOCK;   // float4 per thread per tile
    static_assert(A_F4 >= 1 && W_F4 >= 1, "tile too small for 256 threads");
    __shared__ float As[BK][BM + 1];
    __shared__ float Ws[BK][BN + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int64_t m0;
    int n0;
    if (!nt_block_tile(M, Nc, BM, BN, m0, n0)) return;                 // block-uniform
    const int kq = tid % (BK / 4), r0 = tid / (BK / 4);   // staging: float4 index along k, first row
    constexpr int RSTEP = BLOCK / (BK / 4);                // rows covered per staging pass (32)

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_F4], rw[W_F4];
    // (round 5) optional operand transform relu(bn(.)) of the A columns: applied when the tile is STORED into LDS, not where it is
    // loaded - the loads must stay in flight during the MFMAs of the previous tile (a use right behind the load drains them:
    // measured +25 % on this kernel, +70 % on the TN kernels).  Zero padding may pass through it: padded rows are never stored
    // and padded k-columns meet zero columns of W.
    stin_bn_coef4 cq;                                                             // (s, t) of this thread's four columns of the tile in `ra`
    cq.s = cq.t = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tiles = [&](int k0) {
        const int k = k0 + kq * 4;
        if (tf.mean != nullptr) {
            if (VEC) cq = stin_bn_coef4_load(tf, k < K ? k : 0);
            else {
                float* sp = reinterpret_cast<float*>(&cq.s);
                float* tp = reinterpret_cast<float*>(&cq.t);
#pragma unroll
                for (int e = 0; e < 4; ++e) stin_bn_st(tf, k + e < K ? k + e : 0, sp[e], tp[e]);
            }
        }
#pragma unroll
        for (int s = 0; s < A_F4; ++s) {
            const int64_t row = m0 + r0 + s * RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) {
                const float* p = A + row * lda + k;
                if (VEC) {
                    if (k < K) v = ld4(p);
                } else {
                    if (k + 0 < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
            ra[s] = v;
        }
#pragma unroll
        for (int s = 0; s < W_F4; ++s) {
            const int row = n0 + r0 + s * RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < Nc) {
                const float* p = W + (int64_t)row * ldw + k;
                if (VEC) {
                    if (k < K) v = ld4(p);
                } else {
                    if (k + 0 < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
            rw[s] = v;
        }
    };
    auto store_tiles = [&]() {
        if (tf.mean != nullptr) {                                                     // block-uniform
#pragma unroll
            for (int s = 0; s < A_F4; ++s) ra[s] = stin_bn_relu4(ra[s], cq);
        }
#pragma unroll
        for (int s = 0; s < A_F4; ++s) {
            const int row = r0 + s * RSTEP;
            As[kq * 4 + 0][row] = ra[s].x;
            As[kq * 4 + 1][row] = ra[s].y;
            As[kq * 4 + 2][row] = ra[s].z;
            As[kq * 4 + 3][row] = ra[s].w;
        }
#pragma unroll
        for (int s = 0; s < W_F4; ++s) {
            const int row = r0 + s * RSTEP;
            Ws[kq * 4 + 0][row] = rw[s].x;
            Ws[kq * 4 + 1][row] = rw[s].y;
            Ws[kq * 4 + 2][row] = rw[s].z;
            Ws[kq * 4 + 3][row] = rw[s].w;
        }
    };

    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();           // previous tile's fragment reads are done
        store_tiles();
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK);   // in flight during the MFMAs below
        const int kh = lane >> 5, li = lane & 31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = As[kk + kh][wm * TM + i * 32 + li];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = Ws[kk + kh][wn * TN + j * 32 + li];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    const int kh = lane >> 5, li = lane & 31;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + li;
        if (col >= Nc) continue;
        const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M)
                    C[row * ldc + col] = acc[i][j][r] + (row_mask != nullptr ? bv * row_mask[row * ld_mask] : bv) +
                                         (res != nullptr ? res[row * ld_res + col] : 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------- NT, split-bf16
// The same product on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) with each
// fp32 operand split on the fly into NS bf16 pieces (x = x0 + x1 (+ x2), piece p = bf16(residual)):
//   NS = 2: A0 B0 + A0 B1 + A1 B0                     (3 MFMAs, ~2^-17 per-product error: backward GEMMs)
//   NS = 3: + A0 B2 + A2 B0 + A1 B1                   (6 MFMAs, x0+x1+x2 is exact, dropped terms 2^-24)
// fp32 accumulation inside the MFMA.  LDS tiles are [rows][BK] bf16 with K contiguous (as in global
// memory, no transpose), XOR-swizzled 64-byte rows = conflict-free staging stores and ds_read_b128 fragments
// (lane l: row l&31, k = 8*(l>>5) .. +7 of a 16-wide k-step).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BKH = 32;          // k per LDS tile (two MFMA k-steps of 16)

// Piece type PT = __bf16 (above) or _Float16: an fp32 value is hi + lo with 11-bit pieces, hi*hi + hi*lo + lo*hi
// leaves 2^-22 relative error - fp32-grade products from THREE MFMAs (v_mfma_f32_32x32x16_f16, same rate as
// bf16).  fp16's narrow exponent is handled by fixed power-of-two pre-scales (exact, undone on the
// accumulator): A * 2^3 and W * 2^6, so that the low pieces of unit-scale activations and of typical weights
// (|w| ~ 1e-2) stay normal fp16 numbers.  Relative precision 2^-22 for |a| >= 2^-6, |w| >= 2^-9; below that the
// low piece is an fp16 subnormal and the precision becomes ABSOLUTE (2^-28 for a, 2^-31 for w) - fp32-grade for
// the normalised activations of this network, not for arbitrarily scaled data (use BF16X6 there).  Requires
// |A| < 8188 and |W| < 1023: out-of-range operands give inf/NaN, never a silently wrong value.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename PT> struct PieceTraits;
template <> struct PieceTraits<__bf16> {
    typedef bf16x8 vec8;
    static constexpr float ascale = 1.f, wscale = 1.f;
};
template <> struct PieceTraits<_Float16> {
    typedef f16x8 vec8;
    static constexpr float ascale = 8.f, wscale = 64.f;
};
__device__ __forceinline__ f32x16 mfma_k16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_k16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int NS, typename PT>
__device__ __forceinline__ void split_store(float4 v, PT* dst, int plane_stride, float scale) {
    typedef PT pt4 __attribute__((ext_vector_type(4)));
    float r[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        pt4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = (PT)r[i];
            r[i] -= (float)h[i];
        }
        *reinterpret_cast<pt4*>(dst + p * plane_stride) = h;
    }
}

template <int BM, int BN, int WM, int WN, int NS, typename PT, bool VEC, bool WPRE = false>
__global__ __launch_bounds__(BLOCK) void k_gemm_nt_bf16s(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ row_mask, int64_t ld_mask,
                                                         const float* __restrict__ res, int64_t ld_res,
                                                         int64_t M, int Nc, int K, float* __restrict__ C,
                                                         int64_t ldc, const stin_bn_tf tf) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int A_F4 = BM * BKH / 4 / BLOCK, W_F4 = BN * BKH / 4 / BLOCK;
    static_assert(A_F4 >= 1 && W_F4 >= 1, "tile too small for 256 threads");
    // 64-byte rows (32 bf16), the 16-byte chunk c of row r stored at chunk position c ^ ((r >> 2) & 3): conflict-free for
    // the 8-byte staging stores (two consecutive rows tile one 128-byte bank span) AND for the ds_read_b128 fragment
    // reads (any 16 rows of one read group land on 16 distinct 16-byte slots) - no padding.
    typedef typename PieceTraits<PT>::vec8 vec8;
    constexpr float ASCALE = PieceTraits<PT>::ascale, WSCALE = PieceTraits<PT>::wscale;
    __shared__ __attribute__((aligned(16))) PT As[NS][BM][BKH];
    __shared__ __attribute__((aligned(16))) PT Ws[NS][BN][BKH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int64_t m0;
    int n0;
    if (!nt_block_tile(M, Nc, BM, BN, m0, n0)) return;                 // block-uniform
    const int kq = tid % (BKH / 4), r0 = tid / (BKH / 4);
    constexpr int RSTEP = BLOCK / (BKH / 4);
    auto swz = [](int row, int chunk) { return ((chunk ^ ((row >> 2) & 3)) << 3); };   // bf16 offset of a 16-byte chunk

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_F4], rw[W_F4];
    // (round 5) optional operand transform relu(bn(.)) of the A columns: applied when the tile is STORED into LDS, not where it is
    // loaded - the loads must stay in flight during the MFMAs of the previous tile (a use right behind the load drains them:
    // measured +25 % on this kernel, +70 % on the TN kernels).  Zero padding may pass through it: padded rows are never stored
    // and padded k-columns meet zero columns of W.
    stin_bn_coef4 cq;                                                             // (s, t) of this thread's four columns of the tile in `ra`
    cq.s = cq.t = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tiles = [&](int k0) {
        const int k = k0 + kq * 4;
        if (tf.mean != nullptr) {
            if (VEC) cq = stin_bn_coef4_load(tf, k < K ? k : 0);
            else {
                float* sp = reinterpret_cast<float*>(&cq.s);
                float* tp = reinterpret_cast<float*>(&cq.t);
#pragma unroll
                for (int e = 0; e < 4; ++e) stin_bn_st(tf, k + e < K ? k + e : 0, sp[e], tp[e]);
            }
        }
#pragma unroll
        for (int s = 0; s < A_F4; ++s) {
            const int64_t row = m0 + r0 + s * RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) {
                const float* p = A + row * lda + k;
                if (VEC) {
                    if (k < K) v = ld4(p);
                } else {
                    if (k + 0 < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
            ra[s] = v;
        }
#pragma unroll
        for (int s = 0; s < W_F4; ++s) {
            const int row = n0 + r0 + s * RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < Nc) {
                const float* p = W + (int64_t)row * ldw + k;
                if (VEC) {
                    if (k < K) v = ld4(p);
                } else {
                    if (k + 0 < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
            rw[s] = v;
        }
    };
    auto store_tiles = [&]() {
        if (tf.mean != nullptr) {                                                     // block-uniform
#pragma unroll
            for (int s = 0; s < A_F4; ++s) ra[s] = stin_bn_relu4(ra[s], cq);
        }
#pragma unroll
        for (int s = 0; s < A_F4; ++s) {
            const int row = r0 + s * RSTEP;
            split_store<NS, PT>(ra[s], &As[0][row][swz(row, kq >> 1) + (kq & 1) * 4], BM * BKH, ASCALE);
        }
#pragma unroll
        for (int s = 0; s < W_F4; ++s) {
            const int row = r0 + s * RSTEP;
            PT* dst = &Ws[0][row][swz(row, kq >> 1) + (kq & 1) * 4];
            if (WPRE) {          // W arrives pre-split: [hi x 4 | lo x 4] per k-group (stin_pack.hip put_weight) - no VALU work
                *reinterpret_cast<float2*>(dst) = make_float2(rw[s].x, rw[s].y);
                *reinterpret_cast<float2*>(dst + BN * BKH) = make_float2(rw[s].z, rw[s].w);
            } else {
                split_store<NS, PT>(rw[s], dst, BN * BKH, WSCALE);
            }
        }
    };

    const int kh = lane >> 5, li = lane & 31;
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BKH) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (k0 + BKH < K) load_tiles(k0 + BKH);
#pragma unroll
        for (int ks = 0; ks < BKH; ks += 16) {
            vec8 a[NS][MT], b[NS][NT];
#pragma unroll
            for (int p = 0; p < NS; ++p) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = wm * TM + i * 32 + li;
                    a[p][i] = *reinterpret_cast<const vec8*>(&As[p][row][swz(row, (ks >> 3) + kh)]);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int row = wn * TN + j * 32 + li;
                    b[p][j] = *reinterpret_cast<const vec8*>(&Ws[p][row][swz(row, (ks >> 3) + kh)]);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    // smallest terms first
                    if (NS == 3) {
                        acc[i][j] = mfma_k16(a[1][i], b[1][j], acc[i][j]);
                        acc[i][j] = mfma_k16(a[0][i], b[2][j], acc[i][j]);
                        acc[i][j] = mfma_k16(a[2][i], b[0][j], acc[i][j]);
                    }
                    acc[i][j] = mfma_k16(a[0][i], b[1][j], acc[i][j]);
                    acc[i][j] = mfma_k16(a[1][i], b[0][j], acc[i][j]);
                    acc[i][j] = mfma_k16(a[0][i], b[0][j], acc[i][j]);
                }
        }
    }

#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + li;
        if (col >= Nc) continue;
        const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M)
                    C[row * ldc + col] = acc[i][j][r] * (1.f / (ASCALE * WSCALE)) + (row_mask != nullptr ? bv * row_mask[row * ld_mask] : bv) +
                                         (res != nullptr ? res[row * ld_res + col] : 0.f);
            }
        }
    }
}

// ------------------------------------------------------------ NT, split 16-bit, streaming rows through wave-private LDS (round 5)
// SingleConvMeshNet's per-EDGE products (models/modules/edge_conv_filter.py:34-44: Lin - BN - ReLU - Lin - BN over the E edge rows):
// M = 1e5..1e6 rows, K and Nc 64..256, plain fp32 weights - a few flops per byte, yet the 64 x 64 tiling above runs them at half the
// HBM rate: a block lives for K / 32 = 2-4 k-tiles, each a global-load latency + two block barriers, and every block splits the
// same weight tile again.  Here the loop nest is turned around for rows that only stream:
//   * persistent blocks of 4 waves; the block's weight slice [BN = 32 NT columns, all K] is split ONCE into LDS (fragment reads as in
//     the tiling: [piece][32-wide k chunk][row][64 B], XOR-swizzled);
//   * every wave owns whole 32-row tiles: it loads its tile's KC-wide row chunk with fully coalesced 16-byte loads (8 rows x 128 B
//     per instruction), splits it into a wave-PRIVATE LDS region and multiplies - no block barrier in the loop, the waves of a block
//     run free of each other; the next tile's loads are issued right after the registers are stored and stay in flight during the
//     MFMAs and the epilogue (8-16 KB per wave, 64-128 KB per CU);
//   * k order, MFMA order and the epilogue expression of k_gemm_nt_bf16s: bit-identical accumulators.
// Epilogue modes: MODE 0 stores acc (the plain product); MODE 1 / 2 are the two passes of "the product is only the output gradient
// of BatchNorm1d + ReLU over the pre-norm rows X" (the edge MLP's backward): that backward needs two column sums over ALL rows before
// any row can be finished (P = sum d nhat, Q = sum d with d = dh [gamma nhat + beta > 0]), so the product runs TWICE instead of
// being written, re-read by a reduction, and re-read + rewritten by the elementwise pass:
//   MODE 1: per-lane fp64 sums (a lane owns its columns for the whole loop), nothing stored: partial [gridDim.x][2][Nc] doubles,
//           folded in a fixed order by k_partial_sums_final;
//   MODE 2: the product again, stored as dx = rstd gamma (d - Q / n - nhat P / n) - k_bn_bwd's expression on the accumulator.
// TF: relu(v s + t) (BatchNorm1d + ReLU of the A columns, stin_bn_relu) applied when the tile is stored into LDS - the forward product.
template <int KC, int NT, int NS, typename PT, int MODE, bool TF>
__global__ __launch_bounds__(BLOCK, KC == 64 ? 2 : 1) void k_gemm_nt_stream(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                          int64_t ldw, int64_t M, int Nc, int K, const float* __restrict__ X,
                                                          int64_t ldx, const stin_bn_tf tf, const float* __restrict__ P,
                                                          const float* __restrict__ Q, float inv_n, double* __restrict__ partial,
                                                          float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                                          const float* __restrict__ row_mask, int64_t ld_mask,
                                                          const float* __restrict__ res, int64_t ld_res, int wpre) {
    constexpr int BN = 32 * NT, CH = KC / 32;                                       // staged k chunks of 32 per tile
    typedef typename PieceTraits<PT>::vec8 vec8;
    constexpr float ASCALE = PieceTraits<PT>::ascale, WSCALE = PieceTraits<PT>::wscale;
    extern __shared__ __attribute__((aligned(16))) unsigned char stream_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = K / 32;                                                          // (K % KC == 0)
    const int wplane = nch * BN * 32;                                                // elements per weight piece plane
    constexpr int APLANE = CH * 32 * 32;
    PT* Wl = reinterpret_cast<PT*>(stream_smem);                                    // [NS][nch][BN][32]
    PT* Al = Wl + (size_t)NS * wplane + (size_t)wave * NS * APLANE;                 // [NS][CH][32][32], this wave's
    float* coef = reinterpret_cast<float*>(Wl + (size_t)NS * wplane + (size_t)4 * NS * APLANE);   // TF: s [K] | t [K]
    const int n0 = blockIdx.y * BN;
    auto swz = [](int row, int chunk) { return ((chunk ^ ((row >> 2) & 3)) << 3); };
    const int kh = lane >> 5, li = lane & 31;
    const int kq = lane & 7, rr = lane >> 3;

    // ---- prologue: the block's weight slice, split once
    {
        const int k4n = K / 4;
        for (int idx = tid; idx < BN * k4n; idx += BLOCK) {
            const int row = idx / k4n, k4 = idx % k4n;
            const float4 v = (n0 + row < Nc) ? ld4(W + (int64_t)(n0 + row) * ldw + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = k4 >> 3, q = k4 & 7;
            PT* dst = Wl + ((size_t)c * BN + row) * 32 + swz(row, q >> 1) + (q & 1) * 4;
            if (wpre) {          // pre-split weights (stin_pack.hip put_weight): [hi x 4 | lo x 4] per k-group, scaled already
                *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(dst + wplane) = make_float2(v.z, v.w);
            } else {
                split_store<NS, PT>(v, dst, wplane, WSCALE);
            }
        }
        if (TF) {
            for (int k = tid; k < K; k += BLOCK) {
                float s, t;
                stin_bn_st(tf, k, s, t);
                coef[k] = s;
                coef[K + k] = t;
            }
        }
        if (MODE != 0) {
            for (int i = tid; i < BN; i += BLOCK) {
                const int cc = n0 + i < Nc ? n0 + i : 0;
                coef[i] = tf.mean[cc];
                coef[BN + i] = tf.rstd[cc];
                coef[2 * BN + i] = tf.gamma[cc];
                coef[3 * BN + i] = tf.beta[cc];
                coef[4 * BN + i] = MODE == 2 ? P[cc] * inv_n : 0.f;
                coef[5 * BN + i] = MODE == 2 ? Q[cc] * inv_n : 0.f;
            }
        }
    }
    __syncthreads();

    // the lane's output columns; their BatchNorm coefficients (MODE 1 / 2) live in LDS - [mean | rstd | gamma | beta | P / n | Q / n][BN],
    // read per column tile in the epilogue: six registers per tile across the MFMA loop are what pushes four tiles over 256
    bool cok[NT];
    double ps[NT], qs[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        cok[j] = n0 + j * 32 + li < Nc;
        ps[j] = 0.0;
        qs[j] = 0.0;
    }

    const int64_t tiles = (M + 31) / 32;
    const int64_t wstride = (int64_t)gridDim.x * 4;
    const int kcn = K / KC;
    float4 ra[CH][4];
    auto load_chunk = [&](int64_t tile, int kc) {                                   // rows [32 tile, +32), columns [KC kc, +KC)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t row = tile * 32 + g * 8 + rr;
                ra[c][g] = row < M ? ld4(A + row * lda + kc * KC + c * 32 + kq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    };
    auto store_chunk = [&](int kc) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            stin_bn_coef4 cq;
            if (TF) {
                cq.s = *reinterpret_cast<const float4*>(coef + kc * KC + c * 32 + kq * 4);
                cq.t = *reinterpret_cast<const float4*>(coef + K + kc * KC + c * 32 + kq * 4);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = g * 8 + rr;
                const float4 v = TF ? stin_bn_relu4(ra[c][g], cq) : ra[c][g];
                split_store<NS, PT>(v, Al + (c * 32 + row) * 32 + swz(row, kq >> 1) + (kq & 1) * 4, APLANE, ASCALE);
            }
        }
    };

    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t < tiles) load_chunk(t, 0);
    for (; t < tiles; t += wstride) {
        const int64_t m0 = t * 32;
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // X rows 8 u + rr, columns 32 j + 4 kq .. + 3 of column tile j in slot j & 1: two tiles requested before the MFMAs, tile
        // j + 2 when tile j has gone into LDS (all NT at once do not fit 256 registers beside the accumulators)
        float4 xq[2][4];
        auto load_x = [&](int j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t row = m0 + u * 8 + rr;
                const int col = n0 + j * 32 + kq * 4;
                xq[j & 1][u] = (row < M && col < Nc) ? ld4(X + row * ldx + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        for (int kc = 0; kc < kcn; ++kc) {
            __builtin_amdgcn_wave_barrier();                                        // (the previous chunk's fragment reads precede these stores)
            store_chunk(kc);
            if (MODE != 0 && kc + 1 == kcn) {                                       // the epilogue's X rows: requested before the MFMAs
                load_x(0);
                if (NT > 1) load_x(1);
            }
            if (kc + 1 < kcn) load_chunk(t, kc + 1);
            else if (t + wstride < tiles) load_chunk(t + wstride, 0);               // in flight during the MFMAs and the epilogue
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const PT* wb = Wl + (size_t)(kc * CH + c) * BN * 32;
#pragma unroll
                for (int ks = 0; ks < 32; ks += 16) {
                    vec8 a[NS];
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        a[p] = *reinterpret_cast<const vec8*>(Al + p * APLANE + (c * 32 + li) * 32 + swz(li, (ks >> 3) + kh));
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        vec8 b[NS];
#pragma unroll
                        for (int p = 0; p < NS; ++p)
                            b[p] = *reinterpret_cast<const vec8*>(wb + (size_t)p * wplane + (j * 32 + li) * 32 + swz(li, (ks >> 3) + kh));
                        if (NS == 3) {
                            acc[j] = mfma_k16(a[1], b[1], acc[j]);
                            acc[j] = mfma_k16(a[0], b[2], acc[j]);
                            acc[j] = mfma_k16(a[2], b[0], acc[j]);
                        }
                        acc[j] = mfma_k16(a[0], b[1], acc[j]);
                        acc[j] = mfma_k16(a[1], b[0], acc[j]);
                        acc[j] = mfma_k16(a[0], b[0], acc[j]);
                    }
                }
            }
        }
        // ---- epilogue: the accumulator layout (lane = column, 16 rows) meets row-major memory (16-byte accesses, 8 rows x 128 B per
        // instruction) through a 4 KB [32][32] tile of the wave's staging region - X comes in through it, the result goes out through it
        float* stg = reinterpret_cast<float*>(Al);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[j][r] * (1.f / (ASCALE * WSCALE));
            if (MODE != 0) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(stg + (u * 8 + rr) * 32 + kq * 4) = xq[j & 1][u];
                if (j + 2 < NT) load_x(j + 2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float mu = coef[j * 32 + li], rs = coef[BN + j * 32 + li], ga = coef[2 * BN + j * 32 + li], be = coef[3 * BN + j * 32 + li];
                const float pn = MODE == 2 ? coef[4 * BN + j * 32 + li] : 0.f, qn = MODE == 2 ? coef[5 * BN + j * 32 + li] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tr = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float n = (stg[tr * 32 + li] - mu) * rs;
                    const float d = !(ga * n + be > 0.f) ? 0.f : v[r];
                    if (MODE == 1) {
                        if (cok[j] && m0 + tr < M) {
                            ps[j] += (double)(d * n);
                            qs[j] += (double)d;
                        }
                    } else {
                        v[r] = rs * ga * (d - qn - n * pn);
                    }
                }
            }
            if (MODE != 1) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + li] = v[r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t row = m0 + u * 8 + rr;
                    const int col = n0 + j * 32 + kq * 4;
                    float4 o = *reinterpret_cast<const float4*>(stg + (u * 8 + rr) * 32 + kq * 4);
                    if (row < M && col < Nc) {
                        if (MODE == 0 && (bias != nullptr || res != nullptr)) {      // k_gemm_nt_bf16s's expression: (v + bias [* mask]) + res
                            const float4 bq = bias != nullptr ? ld4(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                            const float mk = row_mask != nullptr ? row_mask[row * ld_mask] : 1.f;
                            const float4 rv = res != nullptr ? ld4(res + row * ld_res + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                            o.x = o.x + (row_mask != nullptr ? bq.x * mk : bq.x) + rv.x;
                            o.y = o.y + (row_mask != nullptr ? bq.y * mk : bq.y) + rv.y;
                            o.z = o.z + (row_mask != nullptr ? bq.z * mk : bq.z) + rv.z;
                            o.w = o.w + (row_mask != nullptr ? bq.w * mk : bq.w) + rv.w;
                        }
                        st4(C + row * ldc + col, o);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (MODE == 1) {
        // fold: the two row halves of a wave (same column), then the four waves of the block in wave order
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            ps[j] += __shfl_xor(ps[j], 32);
            qs[j] += __shfl_xor(qs[j], 32);
        }
        __syncthreads();                                                            // (every wave is done with its LDS region)
        double* red = reinterpret_cast<double*>(stream_smem);                        // [4 waves][2][BN]
        if (kh == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                red[(wave * 2 + 0) * BN + j * 32 + li] = ps[j];
                red[(wave * 2 + 1) * BN + j * 32 + li] = qs[j];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += BLOCK) {
            const int o = i / BN, lc = i % BN;
            if (n0 + lc < Nc)
                partial[((int64_t)blockIdx.x * 2 + o) * Nc + n0 + lc] =
                    ((red[(0 * 2 + o) * BN + lc] + red[(1 * 2 + o) * BN + lc]) + red[(2 * 2 + o) * BN + lc]) + red[(3 * 2 + o) * BN + lc];
        }
    }
}

// out[i] = (float) sum over the groups of partial[g][i], i < n: 64 columns x 16 group lanes per block, each lane a fixed
// interleaved chain over g, the lanes combined in lane order - deterministic
__global__ __launch_bounds__(1024) void k_partial_sums_final(const double* __restrict__ partial, int64_t groups, int n,
                                                               float* __restrict__ out) {
    __shared__ double sm[16][65];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (i < n) {
        int64_t g = gl;
        for (; g + 16 < groups; g += 32) {
            s0 += partial[g * n + i];
            s1 += partial[(g + 16) * n + i];
        }
        if (g < groups) s0 += partial[g * n + i];
    }
    sm[gl][cl] = s0 + s1;
    __syncthreads();
    if (gl == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][cl];
        out[i] = (float)t;
    }
}

// ------------------------------------------------------------ NT, split 16-bit, resident row strip
// The tall-skinny shapes of this network (M = 1e4..1e6 rows, K <= 1280, Nc <= 1280) spend their time moving operands, not
// multiplying: with 64x64 output tiles every A tile is fetched from L2 and split into its 16-bit pieces once per COLUMN
// block (16 times at Nc = 1024), every block pays a load-latency prologue for 8 k-steps of work, and a per-k-step barrier
// keeps the waves of a block in lock-step.  This kernel turns the loop nest around:
//   * a block of 4 waves owns a STRIP of 64 rows and keeps the split A strip (one K chunk of <= 256) resident in LDS: every A
//     element crosses the fabric once per strip and is split once (64 KB at K = 256: two blocks per CU, one stages while
//     the other multiplies).  LDS image: [piece][k-step of 16][row][2 x 16 B], the two 16-byte k-halves of a row swapped
//     for rows with bit 3 set - conflict-free ds_read_b128 fragments whose address is ONE per-lane register plus immediates;
//   * the weight operand comes pre-split in MFMA FRAGMENT ORDER (STIN_GEMM_W_FRAG, stin_pack.hip): the B fragment of a
//     32-column tile and 16-wide k-step is 2 KB contiguous, lane l's 32 bytes = [hi x 8 | lo x 8], so each wave fetches its
//     own fragments straight from L2 into registers with two fully coalesced 16-byte loads per lane (scalar base + lane
//     offset) - no LDS staging, no barrier: wave w computes columns [32 w, 32 w + 32) of a 128-column panel for all 64
//     rows (2 accumulator tiles, 6 MFMAs per fragment), and the waves run free of each other until the strip changes;
//   * the fragment loads are a 4-deep register ring inside a panel (a fragment is requested 4 k-steps = 24 MFMAs before its
//     use); what a wave cannot hide at a panel or strip boundary the second block on the CU does;
//   * work = (strip, 128-column panel) units in strip-major order, dealt to a grid sized to the chip (blocks resident at
//     once) in contiguous ranges: every block gets the same number of units +-1, a range that crosses a strip boundary
//     stages two strips.  M = 18 063 rows no longer means "283 row tiles on 256 CUs".
// K longer than the resident chunk is processed chunk by chunk: the first chunk stores C, later chunks add to it.
// Same arithmetic, same k order and same epilogue expression as k_gemm_nt_bf16s<.., NS = 2, .., WPRE = true>: bit-identical
// results while K fits one chunk (tests/test_hip_parity.py::test_gemm_nt_strip_kernel_equals_tiled_kernel).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// MT = 32-row MFMA tiles per wave, QM = wave quads stacked along the rows: strips of BM = 32 MT QM rows, 256 QM threads.
// (2, 1) is the kernel described above.  Round 3: every wave pulls its W fragments (2 KB per 16-wide k-step) from L2 and
// spends 3 MT MFMAs on them; at MT = 2 with eight waves on the CU the MFMA rate asks for ~31 B/clk of fragment traffic and
// the L2 -> CU path delivers ~27 (profiles/nt_stamps.hip, profiles/micro/load_pattern.hip): the kernel was bound by that
// path.  (2, 2) = two quads on the two 64-row halves of a 128-row strip walking the same panels (the second quad's
// fragment loads hit the CU's vector L1 when the quads stay close), (4, 1) = 128-row strips on one quad (half the
// fragment bytes per MFMA outright, one wave per SIMD).  Selection: strip_config().
constexpr int ST_PANEL = 128, ST_RING = 4;

// fragment of one k-step: 2 x 16 bytes per lane at base + lane * 32 (base wave-uniform -> scalar registers)
struct StFrag {
    u32x4 hi, lo;
};
__device__ __forceinline__ StFrag st_wload(const unsigned char* base, unsigned lane_off) {
    const u32x4* p = reinterpret_cast<const u32x4*>(base + lane_off);
    StFrag f;
    f.hi = p[0];
    f.lo = p[1];
    return f;
}

template <typename PT, int MT, int QM>
__global__ __launch_bounds__(256 * QM) void k_gemm_nt_strip(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ Wf,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ row_mask, int64_t ld_mask,
                                                              const float* __restrict__ res, int64_t ld_res, int64_t M,
                                                              int Nc, int K, float* __restrict__ C, int64_t ldc, int KC,
                                                              int64_t units, int P, int restage) {
    typedef typename PieceTraits<PT>::vec8 vec8;
    constexpr float ASCALE = PieceTraits<PT>::ascale, WSCALE = PieceTraits<PT>::wscale;
    constexpr int BM = 32 * MT * QM, ST_THREADS = 256 * QM;
    constexpr int ST_STEP_BYTES = BM * 32;          // LDS bytes of one piece of one k-step (BM rows x 32 B)
    constexpr int RPP = 32 * QM;                    // rows per staging pass (8 threads along k per row)
    extern __shared__ __attribute__((aligned(16))) unsigned char strip_smem[];
    const int plane_bytes = (KC / 16) * ST_STEP_BYTES;                    // one piece of the resident chunk
    float* bias_s = reinterpret_cast<float*>(strip_smem + 2 * plane_bytes);   // [Nc rounded up to 128]
    float* mask_s = bias_s + P * ST_PANEL;                                // [BM]
    float* stage_s = mask_s + BM;                                         // [waves][16 rows][32 columns]: the epilogue's restage

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform: the sequence logic stays scalar
    const int wave = wave_all & 3, quad = wave_all >> 2;                  // column tile of the panel, row quad of the strip
    const int qrow = quad * MT * 32;                                      // first strip row of this quad
    const int kh = lane >> 5, li = lane & 31;
    const int kq = tid & 7, r0 = tid >> 3;                                // staging: float4 index along k, first row (32 rows per pass)
    const int KS_total = K / 16;                                          // k-steps of the whole K (K is a multiple of 64)
    const int nchunk = (K + KC - 1) / KC;

    const int64_t u0 = (int64_t)blockIdx.x * units / gridDim.x, u1 = (int64_t)(blockIdx.x + 1) * units / gridDim.x;
    if (u0 >= u1) return;
    for (int c = tid; c < P * ST_PANEL; c += ST_THREADS) bias_s[c] = (bias != nullptr && c < Nc) ? bias[c] : 0.f;

    // ---- the sequence of (strip, chunk, panel) this block walks
    struct Seq {
        int64_t u, s;        // first unit of the current strip segment, strip index
        int pa, pb;          // panels [pa, pb) of that strip
        int c, p;            // chunk, panel
        bool live;
    };
    auto ks_of = [&](int c) { const int len = K - c * KC; return (len < KC ? len : KC) / 16; };
    auto seq_begin = [&]() {
        Seq q;
        q.u = u0;
        q.s = u0 / P;
        q.pa = (int)(u0 % P);
        q.pb = (u1 - u0 < P - q.pa) ? q.pa + (int)(u1 - u0) : P;
        q.c = 0;
        q.p = q.pa;
        q.live = true;
        return q;
    };
    auto seq_next_panel = [&](Seq& q) {          // live = false at the end of the block's range
        if (++q.p < q.pb) return;
        q.p = q.pa;
        if (++q.c < nchunk) return;
        q.c = 0;
        q.u += q.pb - q.pa;
        if (q.u >= u1) {
            q.live = false;
            return;
        }
        ++q.s;
        q.pa = 0;
        q.pb = (u1 - q.u < P) ? (int)(u1 - q.u) : P;
        q.p = 0;
    };
    auto frag_base = [&](const Seq& q) {         // first fragment of the panel's chunk for this wave's column tile (uniform)
        int tile = q.p * 4 + wave;
        if (tile * 32 >= Nc) tile = Nc / 32 - 1;                          // (a panel past Nc: fetch something valid, store nothing)
        return reinterpret_cast<const unsigned char*>(Wf) + ((int64_t)tile * KS_total + q.c * (KC / 16)) * 2048;
    };
    const unsigned lane_off = (unsigned)lane * 32u;

    Seq run = seq_begin();

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // A fragment of (row tile i, k-step ks, piece p): a_frag + p * plane_bytes + ks * ST_STEP_BYTES + i * 1024
    const unsigned char* a_frag = strip_smem + qrow * 32 + li * 32 + ((kh ^ ((li >> 3) & 1)) << 4);
    int staged_c = -1;
    int64_t staged_s = -1;
    while (run.live) {
        // ---- (strip, chunk) changed: every wave is done with the old strip -> stage the new one (split once)
        if (run.s != staged_s || run.c != staged_c) {
            __syncthreads();
            const int KTc = ks_of(run.c) / 2;
            constexpr int SH = MT >= 4 ? 2 : 4;         // k-tiles in flight (SH x MT float4 per thread)
            for (int kt0 = 0; kt0 < KTc; kt0 += SH) {
                float4 ra[SH][MT];
#pragma unroll
                for (int h = 0; h < SH; ++h) {
                    const int ktc = kt0 + h < KTc ? kt0 + h : KTc - 1;
                    const int k = run.c * KC + ktc * 32 + kq * 4;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const int64_t row = run.s * BM + r0 + t * RPP;
                        const float4 v = ld4(A + (row < M ? row : M - 1) * lda + k);
                        ra[h][t] = row < M ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int h = 0; h < SH; ++h) {
                    if (kt0 + h >= KTc) break;
                    const int ks = (kt0 + h) * 2 + (kq >> 2), kh_ = (kq >> 1) & 1;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const int row = r0 + t * RPP;
                        PT* dst = reinterpret_cast<PT*>(strip_smem + ks * ST_STEP_BYTES + row * 32 + ((kh_ ^ ((row >> 3) & 1)) << 4) + (kq & 1) * 8);
                        split_store<2, PT>(ra[h][t], dst, plane_bytes / 2, ASCALE);
                    }
                }
            }
            if (row_mask != nullptr && tid < BM) {
                const int64_t row = run.s * BM + tid;
                mask_s[tid] = row_mask[(row < M ? row : M - 1) * ld_mask];
            }
            staged_s = run.s;
            staged_c = run.c;
            __syncthreads();
        }
        // ---- one panel: groups of ST_RING k-steps, 3 MT MFMAs per step; the ring slot a step has consumed is refilled with
        // the fragment ST_RING steps further along.  The ring does not reach across panels: a wave stalls once per panel
        // on its first fragments (and on its last panel's stores, which are older in the in-order memory queue) while the
        // other wave of the SIMD - the CU holds two blocks - keeps the matrix pipe busy.
        const int G = ks_of(run.c) / ST_RING;
        const unsigned char* wrun = frag_base(run);
        StFrag wf[ST_RING];
#pragma unroll
        for (int j = 0; j < ST_RING; ++j) wf[j] = st_wload(wrun + j * 2048, lane_off);
        auto kgroup = [&](const unsigned char* ag, const unsigned char* fetch, auto REFILL) {
#pragma unroll
            for (int j = 0; j < ST_RING; ++j) {
                vec8 a0[MT], a1[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    a0[i] = *reinterpret_cast<const vec8*>(ag + j * ST_STEP_BYTES + i * 1024);
                    a1[i] = *reinterpret_cast<const vec8*>(ag + plane_bytes + j * ST_STEP_BYTES + i * 1024);
                }
                const vec8 b0 = __builtin_bit_cast(vec8, wf[j].hi), b1 = __builtin_bit_cast(vec8, wf[j].lo);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    acc[i] = mfma_k16(a0[i], b1, acc[i]);
                    acc[i] = mfma_k16(a1[i], b0, acc[i]);
                    acc[i] = mfma_k16(a0[i], b0, acc[i]);
                }
                if (decltype(REFILL)::value) wf[j] = st_wload(fetch + j * 2048, lane_off);
            }
        };
        for (int g = 0; g + 1 < G; ++g)
            kgroup(a_frag + g * (ST_RING * ST_STEP_BYTES), wrun + (g + 1) * (ST_RING * 2048), std::true_type());
        kgroup(a_frag + (G - 1) * (ST_RING * ST_STEP_BYTES), wrun, std::false_type());
        // ---- epilogue of the panel's chunk
        const int col0 = run.p * ST_PANEL + wave * 32;                    // wave-uniform
        const bool tile_ok = col0 < Nc;                                   // Nc is a multiple of 32: a whole tile is in or out
        const bool full_rows = (run.s + 1) * BM <= M;
        const float sc = 1.f / (ASCALE * WSCALE);
        if (restage && run.c == 0 && res == nullptr && tile_ok && full_rows) {
            // the common case loads nothing.  Round 3: the 64 x 32 wave tile is restaged through 2 KB of LDS per wave (16 rows x
            // 32 columns at a time: one dword per lane and ds_write, conflict-free) and leaves as 16-byte stores, 8 rows x 128
            // contiguous bytes per instruction - 8 store instructions per tile instead of 32 dword stores (in the all-columns
            // kernel the dword-store epilogue was 10 k of a block's 62 k cycles, profiles/nt_stamps.hip).  Same values.
            const float bv = bias_s[col0 + li];
            float* stage_f = stage_s + wave_all * 512;
            const int r16 = lane >> 3, c8 = lane & 7;
            float* cbase = C + (run.s * BM + qrow + r16) * ldc + col0 + c8 * 4;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = p * 8 + r8;
                        const int tr = (r & 3) + 8 * ((r >> 2) & 1) + 4 * kh;             // row inside this 16-row pass
                        const float m = row_mask != nullptr ? mask_s[qrow + i * 32 + p * 16 + tr] : 1.f;
                        stage_f[tr * 32 + li] = row_mask != nullptr ? acc[i][r] * sc + bv * m + 0.f : acc[i][r] * sc + bv + 0.f;
                        acc[i][r] = 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(stage_f + (q * 8 + r16) * 32 + c8 * 4);
                        st4(cbase + (int64_t)(i * 32 + p * 16 + q * 8) * ldc, v);
                    }
                }
        } else if (run.c == 0 && res == nullptr && tile_ok && full_rows) {
            // (no LDS to spare for the restage without losing a block of occupancy: K chunks of 128) 32 dword stores, row base
            // pointers wave-uniform (scalar), the lane's offset one register
            const float bv = bias_s[col0 + li];
            float* cbase = C + (run.s * BM + qrow) * ldc + col0;
            const unsigned coff = (unsigned)(4 * kh) * (unsigned)ldc + (unsigned)li;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = i * 32 + (r & 3) + 8 * (r >> 2);
                    float* rp = cbase + (int64_t)lr * ldc;
                    rp[coff] = row_mask != nullptr ? acc[i][r] * sc + bv * mask_s[qrow + lr + 4 * kh] + 0.f : acc[i][r] * sc + bv + 0.f;
                    acc[i][r] = 0.f;
                }
        } else {
            // partial strip / tile past Nc / residual / later K chunk: every load first (clamped addresses), then the stores
            const int col = col0 + li;
            const int cc = tile_ok ? col : 0;
            const float bv = bias_s[cc];
            float ld[MT][16];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = run.s * BM + qrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int64_t rc = row < M ? row : M - 1;
                    ld[i][r] = run.c > 0 ? C[rc * ldc + cc] : (res != nullptr ? res[rc * ld_res + cc] : 0.f);
                }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = qrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int64_t row = run.s * BM + lr;
                    float v;
                    if (run.c == 0) v = acc[i][r] * sc + (row_mask != nullptr ? bv * mask_s[lr] : bv) + ld[i][r];
                    else v = ld[i][r] + acc[i][r] * sc;                   // later K chunk: add to what this lane stored before
                    if (row < M && tile_ok) C[row * ldc + col] = v;
                    acc[i][r] = 0.f;
                }
        }
        seq_next_panel(run);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Narrow outputs (Nc = 128 or 256 columns) with any K: the block owns ALL columns of its rows.  A 64x64 tiling reads and
// splits every A tile once per 64-column block (4x at Nc = 256) and runs 3 MFMAs per 4 LDS fragment reads; here
//   * 4 waves as (4 / NW) x NW, each 64 rows x 64 columns (2 x 2 accumulator tiles, 12 MFMAs per 16-wide k-step against
//     4 A-fragment reads from LDS and 4 B-fragment loads), NW = Nc / 64: BM = 64 rows at Nc = 256, 128 rows at Nc = 128;
//   * A is streamed ONCE: 64-wide K chunks, split into the two 16-bit pieces once and double-buffered in LDS in the strip
//     kernel's image ([piece][k-step][row][2 x 16 B swizzled]); the next chunk's global loads are in flight during the 48
//     MFMAs of the current one, one barrier per chunk;
//   * the weight operand is read in MFMA FRAGMENT order straight from L2 (as in k_gemm_nt_strip), a 4-step register ring
//     that runs ahead across chunk boundaries;
//   * accumulators live across the whole K: no partial sums through memory, same k order and epilogue expression as the
//     other split kernels -> bit-identical results (tests/test_hip_parity.py::test_gemm_nt_strip_kernel_equals_tiled_kernel).
// Measured (profiles/r02_gemm_shapes.md): 18 063 x 256 x 1024 52 -> 45 us, 18 063 x 256 x 512 36 -> 31, 18 063 x 128 x 1280
// 43 -> 37, 60 211 x 256 x 640 91 -> 72, 60 211 x 128 x 256 37 -> 31.  At 18 063 rows the grid is 283 blocks on 256 CUs - one wave
// per SIMD, so nothing hides a stall for free.  Compile-time ablations of the plain loop (load chunk c+1, 48 MFMAs, split +
// store, barrier): MFMA issue 12.6 us + weight-fragment waits 10 + A staging 8 + the rest 21 = the measured 52 - the costs
// ADD UP.  Hence the explicit software pipeline below (-6 us).  What remains of the 45: 27 of the 256 CUs run two blocks
// whose waves share the matrix pipes - a 2 x 12 us MFMA makespan - plus launch / prologue / epilogue (~12 us).  Tried and
// dropped: a second group of 4 waves per block splitting K (two waves per SIMD; the block-wide barrier keeps the groups in
// lock-step: 59 us), rotating the K order per block (L2 channel hot spots: no change), padding the 4 KB row pitch (no change);
// SQ counters of the first version: profiles/r02_pmc_nt_wide.md.
constexpr int WD_KC = 64, WD_STEPS = WD_KC / 16;
// profiling build only (profiles/nt_stamps.hip compiles this file with -DSTIN_NT_STAMPS): s_memtime stamps per wave
#ifdef STIN_NT_STAMPS
__device__ unsigned long long* stin_nt_stamp_buf = nullptr;
#define NT_STAMP(i)                                                                                                    \
    do {                                                                                                               \
        if ((threadIdx.x & 63) == 0 && stin_nt_stamp_buf != nullptr && (i) < 32)                                        \
            stin_nt_stamp_buf[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define NT_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// compile-time ablation mask of the panel kernel (profiles/nt_stamps.hip builds one binary per mask; a run-time flag inside
// the loop perturbs the code it measures): 1 second row group re-reads W step 0, 2 every wave does, 4 A loads re-read chunk 0,
// 8 no split / LDS store, 16 no A-fragment LDS reads after the chunk's first, 32 no per-chunk barrier, 64 no MFMAs
#ifndef STIN_NT_ABLATE_MASK
#define STIN_NT_ABLATE_MASK 0
#endif
#define NT_ABLATE(bit) ((STIN_NT_ABLATE_MASK) & (bit))
#else
#define NT_ABLATE(bit) 0
#define NT_STAMP(i)
#define NT_STAMP_DRAIN()
#endif

// WAVES_M = waves along the rows of the block (each wave owns 64 rows x 64 columns): the block has NW * WAVES_M waves and
// BM = 64 * WAVES_M rows.  Round 3, from in-kernel stamps (profiles/nt_stamps.hip, 18 063 x 256 x 1024, BM = 64): a block alone
// on its CU needs 62 k cycles - 5.5 k of first-load latency, 8 x 5.9 k for the K loop and 10 k for an epilogue of 64 dword
// stores per wave - and the K loop runs at 60 cycles per MFMA where a register-fed loop runs 32 and an LDS-fed one 43
// (profiles/micro/mfma_rate.hip).  A block with HALF the rows per wave (32 x 64 tiles) needs the SAME 5.5 k cycles per two
// chunks: the loop is bound by what the CU can pull out of L2 - every block streams the whole fragment-order weight operand
// (1 MB at 256 x 1024) plus its A rows, ~27 bytes per cycle and CU, the rate the guide gives for L2-resident gathers - not
// by the matrix pipe.  So (1) blocks of 128 rows on 8 waves (2 per SIMD): the weight bytes per output row halve, and the
// 283-blocks-on-256-CUs second round disappears (142 blocks, one per CU, the idle CUs cost less than a second round did);
// (2) the epilogue restages the tile through the (then idle) LDS and stores whole 256-byte row segments with 16 bytes per
// lane: 8 store instructions per 32 x 64 tile instead of 32.
template <typename PT, int NW, int WAVES_M>
__global__ __launch_bounds__(64 * NW * WAVES_M) void k_gemm_nt_wide(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ Wf,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ row_mask, int64_t ld_mask,
                                                             const float* __restrict__ res, int64_t ld_res, int64_t M,
                                                             int K, float* __restrict__ C, int64_t ldc,
                                                             double* __restrict__ colstats, int vec_out) {
    typedef typename PieceTraits<PT>::vec8 vec8;
    constexpr float ASCALE = PieceTraits<PT>::ascale, WSCALE = PieceTraits<PT>::wscale;
    constexpr int MT = 2, WROWS = 32 * MT, BM = WROWS * WAVES_M, THREADS = 64 * NW * WAVES_M;
    constexpr int RPP = THREADS / 8, PASSES = BM / RPP;                         // staging: THREADS / 8 rows per pass, 8 lanes per row
    constexpr int STEP_BYTES = BM * 32, PLANE = WD_STEPS * STEP_BYTES, BUF = 2 * PLANE;
    constexpr int SMEM = 2 * BUF > NW * WAVES_M * 8192 ? 2 * BUF : NW * WAVES_M * 8192;   // (the epilogue restages 8 KB per wave)
    static_assert(PASSES >= 2 && PASSES % 2 == 0, "staging is spread as PASSES / 2 pieces per k-step");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    __shared__ float mask_s[BM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NW, wn = wave % NW;                                  // wave-uniform
    const int kh = lane >> 5, li = lane & 31;
    const int kq = tid & 7, r0 = tid >> 3;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int nchunk = K / WD_KC, KS_total = K / 16;

    NT_STAMP(0);
    if (row_mask != nullptr && tid < BM) {
        const int64_t row = row0 + tid;
        mask_s[tid] = row_mask[(row < M ? row : M - 1) * ld_mask];
    }

    // Software pipeline (one wave per SIMD at M = 18 k: nothing else hides a stall).  Measured by ablation at 18 063 x 256 x
    // 1024: MFMA issue 12.6 us, weight-fragment waits 10, A staging 8, fixed 21 - and in the straightforward loop they ADD UP
    // (52 us).  So the staging work of chunk c+1 is spread over the MFMA groups of chunk c (split + LDS store of one 16-byte
    // piece after the first MFMAs of every k-step), its global loads are issued a whole chunk earlier (two register sets),
    // and the A fragments of k-step j+1 are read from LDS while k-step j multiplies.
    float4 ra[2][2][PASSES];                                                   // [register set][k-tile of the chunk][row pass]
    auto gload = [&](float4 (&r)[2][PASSES], int c) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < PASSES; ++t) {
                const int64_t row = row0 + r0 + t * RPP;
                // rows past M read row M - 1 (valid memory): their accumulators are never stored, so no zeroing - a select
                // here would make the compiler wait for the load right where it is issued
                r[h][t] = ld4(A + (row < M ? row : M - 1) * lda + c * WD_KC + h * 32 + kq * 4);
            }
    };
    auto sstore_piece = [&](const float4 (&r)[2][PASSES], int buf, int h, int t) {
        const int ks = h * 2 + (kq >> 2), kh_ = (kq >> 1) & 1;
        const int row = r0 + t * RPP;
        PT* dst = reinterpret_cast<PT*>(smem + buf * BUF + ks * STEP_BYTES + row * 32 + ((kh_ ^ ((row >> 3) & 1)) << 4) + (kq & 1) * 8);
        split_store<2, PT>(r[h][t], dst, PLANE / 2, ASCALE);
    };

    // B fragments of this wave's two 32-column tiles: step ks of tile t at ((t * KS_total + ks) * 2048) + lane * 32
    const unsigned char* wb0 = reinterpret_cast<const unsigned char*>(Wf) + (int64_t)(wn * 2) * KS_total * 2048;
    const unsigned char* wb1 = wb0 + (int64_t)KS_total * 2048;
    const unsigned lane_off = (unsigned)lane * 32u;
    StFrag wf[WD_STEPS][2];
#pragma unroll
    for (int j = 0; j < WD_STEPS; ++j) {
        wf[j][0] = st_wload(wb0 + j * 2048, lane_off);
        wf[j][1] = st_wload(wb1 + j * 2048, lane_off);
    }
    gload(ra[0], 0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < PASSES; ++t) sstore_piece(ra[0], 0, h, t);
    gload(ra[1], nchunk > 1 ? 1 : 0);

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned char* a_frag = smem + (wm * WROWS + li) * 32 + ((kh ^ ((li >> 3) & 1)) << 4);
    struct AFrag {
        vec8 a0[MT], a1[MT];
    };
    auto aread = [&](const unsigned char* ab, int j) {
        AFrag f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f.a0[i] = *reinterpret_cast<const vec8*>(ab + j * STEP_BYTES + i * 1024);
            f.a1[i] = *reinterpret_cast<const vec8*>(ab + PLANE + j * STEP_BYTES + i * 1024);
        }
        return f;
    };
    // one chunk: compute from buffer (c & 1); meanwhile split + store `stage` (chunk c + 1, loaded an iteration ago) into the
    // other buffer and request chunk c + 2 into `fetch` (the set whose content was stored during the previous iteration)
    // (branch-free body: past the end of K the fetches re-read the last chunk and the staging writes a buffer nobody reads
    // any more - conditionals between the MFMAs made the compiler shuttle the accumulators between AGPRs and VGPRs)
    auto chunk = [&](int c, const float4 (&stage)[2][PASSES], float4 (&fetch)[2][PASSES]) {
        gload(fetch, c + 2 < nchunk ? c + 2 : nchunk - 1);
        const unsigned char* ab = a_frag + (c & 1) * BUF;
        const int nxt = (c + 1 < nchunk ? c + 1 : c) * WD_STEPS;               // ring refill: same step of the next chunk (clamped)
        AFrag cur = aread(ab, 0);
#pragma unroll
        for (int j = 0; j < WD_STEPS; ++j) {
            AFrag nx = cur;
            if (j + 1 < WD_STEPS) nx = aread(ab, j + 1);                       // next k-step's fragments in flight during this one
            const vec8 b00 = __builtin_bit_cast(vec8, wf[j][0].hi), b01 = __builtin_bit_cast(vec8, wf[j][0].lo);
            const vec8 b10 = __builtin_bit_cast(vec8, wf[j][1].hi), b11 = __builtin_bit_cast(vec8, wf[j][1].lo);
            acc[0][0] = mfma_k16(cur.a0[0], b01, acc[0][0]);
            acc[0][0] = mfma_k16(cur.a1[0], b00, acc[0][0]);
            acc[0][0] = mfma_k16(cur.a0[0], b00, acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            // this k-step's share of the staging: 2 * PASSES pieces per chunk over the WD_STEPS = 4 k-steps
#pragma unroll
            for (int q = 0; q < PASSES / 2; ++q) sstore_piece(stage, (c + 1) & 1, j >> 1, (j & 1) * (PASSES / 2) + q);
            __builtin_amdgcn_sched_barrier(0);
            if (MT == 2) {
                acc[MT - 1][0] = mfma_k16(cur.a0[MT - 1], b01, acc[MT - 1][0]);
                acc[MT - 1][0] = mfma_k16(cur.a1[MT - 1], b00, acc[MT - 1][0]);
                acc[MT - 1][0] = mfma_k16(cur.a0[MT - 1], b00, acc[MT - 1][0]);
            }
            wf[j][0] = st_wload(wb0 + (int64_t)(nxt + j) * 2048, lane_off);
            acc[0][1] = mfma_k16(cur.a0[0], b11, acc[0][1]);
            acc[0][1] = mfma_k16(cur.a1[0], b10, acc[0][1]);
            acc[0][1] = mfma_k16(cur.a0[0], b10, acc[0][1]);
            if (MT == 2) {
                acc[MT - 1][1] = mfma_k16(cur.a0[MT - 1], b11, acc[MT - 1][1]);
                acc[MT - 1][1] = mfma_k16(cur.a1[MT - 1], b10, acc[MT - 1][1]);
                acc[MT - 1][1] = mfma_k16(cur.a0[MT - 1], b10, acc[MT - 1][1]);
            }
            wf[j][1] = st_wload(wb1 + (int64_t)(nxt + j) * 2048, lane_off);
            __builtin_amdgcn_sched_barrier(0);
            cur = nx;
        }
        __syncthreads();
    };
    __syncthreads();
    NT_STAMP(1);
    int c = 0;
    for (; c + 1 < nchunk; c += 2) {
        chunk(c, ra[1], ra[0]);
        chunk(c + 1, ra[0], ra[1]);
        NT_STAMP(2 + (c >> 1));
    }
    if (c < nchunk) chunk(c, ra[1], ra[0]);
    NT_STAMP(30);

    // ---- epilogue: v = acc / (ascale wscale) + bias [* row mask] + residual
    // colstats (optional): per WROWS-row group (blockIdx * WAVES_M + wm) the column sums of the STORED values and of their
    // squares in fp64, [group][2][Nc] - the first stage of the instance-norm statistics (stin_moments_final_f32 is the second),
    // fixed summation order: rows of the lane in storage order, then the two 32-lane halves.
    const float sc = 1.f / (ASCALE * WSCALE);
    if (vec_out) {
        // (every wave has passed the barrier that ends the last chunk: the staging buffers are free)  Per 32-row tile the wave
        // writes its 32 x 64 values into its own 8 KB of LDS (one dword per lane and instruction: conflict-free) and reads
        // them back as whole 256-byte row segments, 16 lanes per row: 4 rows per 16-byte store instruction.  The residual is
        // loaded in the same shape and added after the bias term, as before: bit-identical values.
        float* stage_f = reinterpret_cast<float*>(smem + wave * 8192);
        double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
        float bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = bias != nullptr ? bias[wn * 64 + t * 32 + li] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int lrow0 = wm * WROWS + i * 32;                             // first block-local row of this tile
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tr = (r & 3) + 8 * (r >> 2) + 4 * kh;            // row inside the tile
                    const float v = acc[i][t][r] * sc + (row_mask != nullptr ? bv[t] * mask_s[lrow0 + tr] : bv[t]);
                    stage_f[tr * 64 + t * 32 + li] = v;
                    if (colstats != nullptr && row0 + lrow0 + tr < M) {        // (colstats launches carry no residual)
                        const double d = (double)v;
                        s1[t] += d;
                        s2[t] += d * d;
                    }
                }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int tr = q * 4 + (lane >> 4), c16 = lane & 15;
                const int64_t grow = row0 + lrow0 + tr;
                float4 v = *reinterpret_cast<const float4*>(stage_f + tr * 64 + c16 * 4);
                if (grow < M) {
                    const int col = wn * 64 + c16 * 4;
                    if (res != nullptr) {
                        const float4 rv = ld4(res + grow * ld_res + col);
                        v.x += rv.x;
                        v.y += rv.y;
                        v.z += rv.z;
                        v.w += rv.w;
                    }
                    st4(C + grow * ldc + col, v);
                }
            }
        }
        if (colstats != nullptr) {                                             // block-uniform
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s1[t] += __shfl_xor(s1[t], 32);
                s2[t] += __shfl_xor(s2[t], 32);
                if (kh == 0 && row0 + wm * WROWS < M) {                        // (a group wholly past M does not exist)
                    double* dst = colstats + ((int64_t)blockIdx.x * WAVES_M + wm) * 2 * (NW * 64) + wn * 64 + t * 32 + li;
                    dst[0] = s1[t];
                    dst[NW * 64] = s2[t];
                }
            }
        }
        NT_STAMP(31);
        return;
    }
    const bool full_rows = row0 + BM <= M;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = wn * 64 + t * 32 + li;
        const float bv = bias != nullptr ? bias[col] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int lr0 = wm * WROWS + i * 32 + 4 * kh;
            if (full_rows) {
                float ld[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ld[r] = res != nullptr ? res[(row0 + lr0 + (r & 3) + 8 * (r >> 2)) * ld_res + col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                    const float v = acc[i][t][r] * sc + (row_mask != nullptr ? bv * mask_s[lr] : bv) + ld[r];
                    C[(row0 + lr) * ldc + col] = v;
                    if (colstats != nullptr) {
                        const double d = (double)v;
                        s1 += d;
                        s2 += d * d;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = lr0 + (r & 3) + 8 * (r >> 2);
                    const int64_t row = row0 + lr;
                    if (row < M) {
                        const float v = acc[i][t][r] * sc + (row_mask != nullptr ? bv * mask_s[lr] : bv) +
                                        (res != nullptr ? res[row * ld_res + col] : 0.f);
                        C[row * ldc + col] = v;
                        if (colstats != nullptr) {
                            const double d = (double)v;
                            s1 += d;
                            s2 += d * d;
                        }
                    }
                }
            }
        }
        if (colstats != nullptr) {                                             // block-uniform
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (kh == 0 && row0 + wm * WROWS < M) {                            // (a group wholly past M does not exist)
                double* dst = colstats + ((int64_t)blockIdx.x * WAVES_M + wm) * 2 * (NW * 64) + col;
                dst[0] = s1;
                dst[NW * 64] = s2;
            }
        }
    }
    NT_STAMP(31);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: balanced column-panel tiles (k_gemm_nt_panel).  What bounds the split-precision NT kernels at the bottleneck level
// (M = 18 063) is the L2 -> CU path (~27-30 B/clk per CU, profiles/nt_stamps.hip), so the tile a CU owns should be as square as
// the chip allows AND every CU should own exactly one: bytes pulled per CU = (rows + columns) x K x 4.  The all-columns kernel
// gives a CU 128 rows x 256 columns (1.5 MB at K = 1024) and uses 142 of the 256 CUs; the strip kernel re-streams a 128-column
// W panel for every 64-row strip.  Here
//   * the output is cut into 128-column panels x row blocks of BM = 32 (MT0 + MT1) rows, MT0 + MT1 chosen on the host so that
//     (row blocks) x (panels) fills the chip in ONE round (panel_tiles(): 18 063 x 256: 113 blocks of 160 rows x 2 panels = 226
//     CUs; x 512: 63 blocks of 288 rows x 4 = 252) - "283 tiles on 256 CUs" is gone; shapes that would need two or more rounds
//     measured no better than the strip kernel (one 147 KB block per CU at a time: nothing overlaps a block's prologue and its
//     store phase) and stay there;
//   * a block is 8 waves = 2 row groups x 4 column tiles: wave (q, wn) owns columns [32 wn, 32 wn + 32) of the panel and MT0
//     (q = 0) or MT1 (q = 1) 32-row tiles.  The two waves of a SIMD share its matrix pipe, so MT0 != MT1 costs nothing; they
//     run as two role programs behind a scalar branch (5 and 4 accumulator tiles at most);
//   * A is streamed ONCE per block through the wide kernel's double-buffered LDS image (64-wide K chunks, split into the two
//     16-bit pieces while staging, one barrier per chunk, ONE register set: the rows of chunk c + 1 are split + stored behind the
//     row tiles' MFMAs of k-steps 0 and 1, the rows of chunk c + 2 requested at the top of k-step 2);
//   * W fragments (fragment order, STIN_GEMM_W_FRAG) go straight from L2 to registers through a 4-step ring, 2 KB per k-step per
//     wave for 3 MT MFMAs (the strip kernel: 6); the second row group requests the same fragments (copying them once per block
//     into LDS by LDS-DMA instead measured SLOWER, profiles/r04_nt_panel.md);
//   * the A fragments of the whole next k-step are in flight during a k-step's MFMAs;
//   * epilogue restaged through LDS (16-byte row-contiguous stores), optional column statistics as in the wide kernel
//     (groups = 2 per row block).
// What bounds it (in-kernel stamps + compile-time ablations, STIN_NT_ABLATE_MASK): profiles/r04_nt_panel.md.
// Same k order, MFMA order and epilogue expression as the other split kernels: bit-identical results
// (tests/test_hip_parity.py::test_gemm_nt_panel_kernel_equals_tiled_kernel).
// Optional epilogue of the panel kernel (round 4): the FIRST stage of the instance-norm + ELU backward statistics of the layer
// that consumes this product as its output gradient.  With g = the stored values (residual included), x / mean / rstd that
// layer's pre-norm activations and statistics: dy = g ELU'((x - mean) rstd), partial sums of dy (x - mean) and of dy per column
// and row group in fp64 -> colstats [group][2][Nc] (the layout of the moment statistics; k_colreduce_final folds either).
// The block backward hands the input gradient dx of block k straight to block k - 1: its separate pass over (agg, g) - a short
// launch that runs 3x slower beside the weight-gradient stream than alone - rides on the rows while they are in registers.
struct NtDotElu {
    const float* x;          // NULL: off
    int64_t ldx;
    const float* mean;
    const float* rstd;
};
__device__ __forceinline__ float nt_elu_grad_from_pre(float n) { return n > 0.f ? 1.f : __expf(n); }     // (= stin_norm.hip)

template <typename PT, int MT0, int MT1>
__global__ __launch_bounds__(512) void k_gemm_nt_panel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Wf,
                                                       const float* __restrict__ bias, const float* __restrict__ row_mask,
                                                       int64_t ld_mask, const float* __restrict__ res, int64_t ld_res, int64_t M,
                                                       int Nc, int K, float* __restrict__ C, int64_t ldc,
                                                       double* __restrict__ colstats, int nrb, int P, int xcd_map,
                                                       const NtDotElu de) {
    typedef typename PieceTraits<PT>::vec8 vec8;
    constexpr float ASCALE = PieceTraits<PT>::ascale, WSCALE = PieceTraits<PT>::wscale;
    constexpr int MTS = MT0 + MT1, BM = 32 * MTS, THREADS = 512;
    constexpr int NF4 = BM * 16 / THREADS;                                      // float4 per thread and 64-wide chunk (= MTS)
    constexpr int STEP_BYTES = BM * 32, PLANE = WD_STEPS * STEP_BYTES, BUF = 2 * PLANE;
    static_assert(2 * BUF >= 8 * 4096, "the epilogue restages 4 KB per wave (BM = 64: the launch adds the mask's BM floats)");
    extern __shared__ __attribute__((aligned(16))) unsigned char panel_smem[];
    unsigned char* smem = panel_smem;
    // (the row mask of the block goes through a register and, after the K loop, into the then idle staging area behind the
    // epilogue's restage slots: the block needs exactly 2 BUF bytes of LDS - 80 KB at 160 rows, two blocks per CU)
    float* mask_s = reinterpret_cast<float*>(panel_smem + 8 * 4096);            // [BM], valid from the end of the K loop on

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave >> 2, wn = wave & 3;                                     // row group, column tile of the panel (wave-uniform)
    const int kh = lane >> 5, li = lane & 31;
    int rb, p;
    if (xcd_map) {                                                          // row block rb on XCD rb % 8, its panels on consecutive slots
        const int slot = blockIdx.x >> 3;
        rb = (slot / P) * 8 + (blockIdx.x & 7);
        p = slot % P;
    } else {
        rb = blockIdx.x / P;
        p = blockIdx.x % P;
    }
    if (rb >= nrb) return;                                                      // block-uniform
    NT_STAMP(0);
    const int64_t row0 = (int64_t)rb * BM;
    const int nchunk = K / WD_KC, KS_total = K / 16;
    const int col0 = p * 128 + wn * 32;                                         // first column of this wave's tile

    float mask_r = 1.f;
    if (row_mask != nullptr && tid < BM) {
        const int64_t row = row0 + tid;
        mask_r = row_mask[(row < M ? row : M - 1) * ld_mask];
    }

    // staging map: item i = tid + s * THREADS of the chunk's BM x 16 float4: first all rows of the chunk's first 32-wide k-tile,
    // then all rows of the second (8 lanes per row and k-tile = one 128-byte line, 8 rows per wave: the wide kernel's map)
    int goff[NF4], soff[NF4];
#pragma unroll
    for (int s = 0; s < NF4; ++s) {
        const int i = tid + s * THREADS;
        const int h = i / (BM * 8), rem = i % (BM * 8), row = rem >> 3, kq = rem & 7;
        const int64_t grow = row0 + row;
        goff[s] = (int)((grow < M ? grow : M - 1) - row0) * (int)lda + h * 32 + kq * 4;     // rows past M re-read row M - 1
        const int ks = h * 2 + (kq >> 2), kh_ = (kq >> 1) & 1;
        soff[s] = ks * STEP_BYTES + row * 32 + ((kh_ ^ ((row >> 3) & 1)) << 4) + (kq & 1) * 8;
    }
    const float* Ab = A + row0 * lda;
    float4 ra[NF4];
    auto gload = [&](int c) {
#pragma unroll
        for (int s = 0; s < NF4; ++s) ra[s] = ld4(Ab + c * WD_KC + goff[s]);
    };
    auto sstore = [&](int buf, int s) {
        split_store<2, PT>(ra[s], reinterpret_cast<PT*>(smem + buf * BUF + soff[s]), PLANE / 2, ASCALE);
    };

    const unsigned char* wb = reinterpret_cast<const unsigned char*>(Wf) + (int64_t)(p * 4 + wn) * KS_total * 2048;
    const unsigned lane_off = (unsigned)lane * 32u;
    const float sc = 1.f / (ASCALE * WSCALE);

    auto role = [&](auto MTc) {
        constexpr int MT = decltype(MTc)::value;
        // Staging schedule (one register set): the rows of chunk c + 1 sit in `ra` when chunk c starts (requested during the
        // second half of chunk c - 1); they are split and stored into the other LDS buffer after the row tiles' MFMAs of k-steps
        // 0 and 1, and the loads of chunk c + 2 are issued into the same registers at the top of k-step 2: every load has half a
        // chunk plus a barrier of lead time.  (The first version issued at the top of the chunk and stored 1-2 k-steps later; the
        // stamps read 3.9 k cycles per chunk of 160 x 128 x 64 with either schedule - load latency is not what the loop waits for.)
        constexpr int NS = 2 * MT;                                              // staging slots of a chunk
        const int trow = q * (MT0 * 32);                                        // first block-local row of this wave's tiles
        StFrag wf[WD_STEPS];
#pragma unroll
        for (int j = 0; j < WD_STEPS; ++j) wf[j] = st_wload(wb + j * 2048, lane_off);
        gload(0);
        f32x16 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NF4; ++s) sstore(0, s);
        gload(nchunk > 1 ? 1 : 0);
        const unsigned char* a_frag = smem + (trow + li) * 32 + ((kh ^ ((li >> 3) & 1)) << 4);
        __syncthreads();
        NT_STAMP(1);
        for (int c = 0; c < nchunk; ++c) {
            // branch-free body: past the end of K the fetches re-read the last chunk and the staging writes a buffer nobody reads
            const unsigned char* ab = a_frag + (c & 1) * BUF;
            const int nxt = (c + 1 < nchunk ? c + 1 : c) * WD_STEPS;            // ring refill: same step of the next chunk (clamped)
            // A fragments: the whole next k-step in flight (tile i of step j + 1 is requested in front of tile i's MFMAs of step j:
            // MT tiles = 96 MT cycles of lead; one tile ahead measured the same time - the LDS round trip is not the bound either)
            vec8 ch[MT], cl[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ch[i] = *reinterpret_cast<const vec8*>(ab + i * 1024);
                cl[i] = *reinterpret_cast<const vec8*>(ab + PLANE + i * 1024);
            }
#pragma unroll
            for (int j = 0; j < WD_STEPS; ++j) {
                const vec8 b0 = __builtin_bit_cast(vec8, wf[j].hi), b1 = __builtin_bit_cast(vec8, wf[j].lo);
                if (j == 2) gload(NT_ABLATE(4) ? 0 : (c + 2 < nchunk ? c + 2 : nchunk - 1));
                vec8 nh[MT], nl[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    nh[i] = ch[i];
                    nl[i] = cl[i];
                    if (j + 1 < WD_STEPS && !NT_ABLATE(16)) {
                        nh[i] = *reinterpret_cast<const vec8*>(ab + (j + 1) * STEP_BYTES + i * 1024);
                        nl[i] = *reinterpret_cast<const vec8*>(ab + PLANE + (j + 1) * STEP_BYTES + i * 1024);
                    }
                    if (!NT_ABLATE(64)) {
                        acc[i] = mfma_k16(ch[i], b1, acc[i]);
                        acc[i] = mfma_k16(cl[i], b0, acc[i]);
                        acc[i] = mfma_k16(ch[i], b0, acc[i]);
                    } else {
                        acc[i][0] += (float)ch[i][0] + (float)cl[i][0] + (float)b0[0] + (float)b1[0];
                    }
                    if (j < 2) {
                        const int slot = j * MT + i;
#pragma unroll
                        for (int s = 0; s < NF4; ++s)
                            if ((s * NS) / NF4 == slot && !NT_ABLATE(8)) sstore((c + 1) & 1, s);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    ch[i] = nh[i];
                    cl[i] = nl[i];
                }
                wf[j] = st_wload(wb + (int64_t)((NT_ABLATE(2) || (NT_ABLATE(1) && q)) ? 0 : nxt + j) * 2048, lane_off);
            }
            if (!NT_ABLATE(32)) __syncthreads();
            NT_STAMP(2 + (c < 26 ? c : 26));
        }
        if (tid < BM) mask_s[tid] = mask_r;                                     // (behind the last chunk's barrier: the buffers are idle)
        __syncthreads();
        // ---- epilogue: v = acc / (ascale wscale) + bias [* row mask] (+ residual), restaged per 32 x 32 tile through 4 KB of the
        // (now idle) staging buffers: 8 lanes x 16 B per row, 8 rows per store instruction
        float* stage_f = reinterpret_cast<float*>(smem + wave * 4096);
        const float bv = bias != nullptr ? bias[col0 + li] : 0.f;
        double s1 = 0.0, s2 = 0.0;
        const int r8 = lane >> 3, c8 = lane & 7;
        const bool dotelu = de.x != nullptr;                                    // block-uniform
        double d0[4] = {0.0, 0.0, 0.0, 0.0}, d1[4] = {0.0, 0.0, 0.0, 0.0};
        float4 dmu = make_float4(0.f, 0.f, 0.f, 0.f), drs = dmu;
        if (dotelu) {
            dmu = ld4(de.mean + col0 + c8 * 4);
            drs = ld4(de.rstd + col0 + c8 * 4);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int lrow0 = trow + i * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tr = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float v = acc[i][r] * sc + (row_mask != nullptr ? bv * mask_s[lrow0 + tr] : bv);
                stage_f[tr * 32 + li] = v;
                if (colstats != nullptr && !dotelu && row0 + lrow0 + tr < M) {
                    const double d = (double)v;
                    s1 += d;
                    s2 += d * d;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int tr = t * 8 + r8;
                const int64_t grow = row0 + lrow0 + tr;
                float4 v = *reinterpret_cast<const float4*>(stage_f + tr * 32 + c8 * 4);
                if (grow < M) {
                    if (res != nullptr) {
                        const float4 rv = ld4(res + grow * ld_res + col0 + c8 * 4);
                        v.x += rv.x;
                        v.y += rv.y;
                        v.z += rv.z;
                        v.w += rv.w;
                    }
                    st4(C + grow * ldc + col0 + c8 * 4, v);
                    if (dotelu) {
                        const float4 xa = ld4(de.x + grow * de.ldx + col0 + c8 * 4);
                        const float gv[4] = {v.x, v.y, v.z, v.w}, xv[4] = {xa.x, xa.y, xa.z, xa.w};
                        const float mu[4] = {dmu.x, dmu.y, dmu.z, dmu.w}, rs[4] = {drs.x, drs.y, drs.z, drs.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xc = xv[e] - mu[e];
                            const float dy = gv[e] * nt_elu_grad_from_pre(xc * rs[e]);
                            d0[e] += (double)(dy * xc);
                            d1[e] += (double)dy;
                        }
                    }
                }
            }
        }
        if (dotelu) {                                                          // block-uniform: fold the 8 row lanes of every column quad
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int m = 8; m < 64; m <<= 1) {
                    d0[e] += __shfl_xor(d0[e], m);
                    d1[e] += __shfl_xor(d1[e], m);
                }
            }
            if (r8 == 0) {
                double* dst = colstats + ((int64_t)rb * 2 + q) * 2 * Nc + col0 + c8 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dst[e] = d0[e];
                    dst[Nc + e] = d1[e];
                }
            }
        } else if (colstats != nullptr) {                                      // block-uniform
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (kh == 0) {
                double* dst = colstats + ((int64_t)rb * 2 + q) * 2 * Nc + col0 + li;
                dst[0] = s1;
                dst[Nc] = s2;
            }
        }
        NT_STAMP(29);
        NT_STAMP_DRAIN();
        NT_STAMP(30);
    };
    if (q == 0) role(std::integral_constant<int, MT0>());
    else role(std::integral_constant<int, MT1>());
}

// ----------------------------------------------------------------------------- TN
// dW tile TI x TJ per 256-thread block, reduction over a chunk of rows m.  Both operands are
// row-major with m as the slow index, so a 32-row slab of G (TI columns) and X (TJ columns) is staged
// in LDS exactly as it lies in memory (16-byte global loads, ds_write_b128, double buffered, the next
// slab's loads in flight during the MFMAs) and the MFMA fragments are stride-1 ds_read_b32:
// A[i][k] = Gs[k][i], B[k][j] = Xs[k][j] with k = the row inside the slab.  4 waves as 2 x 2, each
// (TI/2) x (TJ/2).  The bias gradient sum_m w[m] G[m, :] is accumulated on the VALU by the staging
// threads of the j-tile-0 blocks.  Partial tiles go to the chunk block of the slab workspace (tn_chunk_stride); k_reduce_slabs adds them
// in a fixed order.  Blocks are numbered so that all tiles of one row chunk share an XCD
// (blockIdx % 8 is the observed XCD round-robin): the chunk's rows are re-read from that L2.
constexpr int TN_R = 32;   // rows per LDS slab
// One chunk's partial result in the slab workspace: the weight block [Nc][Kq] (Kq = K rounded up to 4: rows stay
// 16-byte aligned, and 128-byte aligned for the usual K % 32 == 0, so the tile stores are whole cache lines) followed
// by the bias-gradient partials [Nc] - NOT an odd-pitched [Nc][K + 1] matrix.
__host__ __device__ __forceinline__ int64_t tn_chunk_stride(int Nc, int Kq) { return (int64_t)Nc * Kq + ((Nc + 3) & ~3); }

template <int TI, int TJ, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_gemm_tn(const float* __restrict__ G, int64_t ldg,
                                                   const float* __restrict__ X, int64_t ldx, int64_t M, int Nc,
                                                   int K, int Kq, int has_bias, const float* __restrict__ row_w, int64_t ld_w,
                                                   int rows_per_chunk, int tiles_i, int tiles_j, int64_t chunks,
                                                   float* __restrict__ slab, const stin_bn_tf xtf) {
    constexpr int MT = TI / 64, NT = TJ / 64;                    // 32x32 MFMA tiles per wave
    constexpr int GF4 = TN_R * TI / 4 / BLOCK, XF4 = TN_R * TJ / 4 / BLOCK;   // float4 per thread per slab
    constexpr int GC4 = TI / 4, XC4 = TJ / 4;                    // float4 columns
    __shared__ float Gs[2][TN_R][TI];
    __shared__ float Xs[2][TN_R][TJ];
    __shared__ float bsum[BLOCK / GC4][TI + 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles = tiles_i * tiles_j;
    const int64_t b = blockIdx.x;
    // chunks >= 8: XCD x (blocks b = x mod 8) owns the row chunks x, x+8, ... - its tiles re-read the same rows from its own L2.
    // Fewer chunks than XCDs (the wide layers: >= 64 output tiles): plain order, so that every XCD gets tiles of every chunk.
    const int64_t xcd = b % 8, q = b / 8;
    const int64_t chunk = chunks >= 8 ? (q / tiles) * 8 + xcd : b / tiles;
    const int tile = (int)(chunks >= 8 ? q % tiles : b % tiles);
    if (chunk >= chunks) return;                                  // block-uniform
    const int tj = tile % tiles_j, ti = tile / tiles_j;
    const int i0 = ti * TI, j0 = tj * TJ;
    const int64_t mb = chunk * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;
    const bool want_bias = has_bias && (tj == 0);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);

    const int gc = tid % GC4, gr = tid / GC4;    // staging coordinates in the G slab (rows gr + s * BLOCK/GC4)
    const int xc = tid % XC4, xr = tid / XC4;
    float4 rg[GF4], rx[XF4];
    float rwt[GF4];

    stin_bn_coef4 xq;                                                  // (s, t) of this thread's fixed X columns (transform at STORE time)
    xq.s = xq.t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xtf.mean != nullptr) {
        float* sp = reinterpret_cast<float*>(&xq.s);
        float* tp = reinterpret_cast<float*>(&xq.t);
#pragma unroll
        for (int e = 0; e < 4; ++e) stin_bn_st(xtf, j0 + xc * 4 + e < K ? j0 + xc * 4 + e : 0, sp[e], tp[e]);
    }
    auto load_slab = [&](int64_t m0) {
#pragma unroll
        for (int s = 0; s < GF4; ++s) {
            const int64_t row = m0 + gr + s * (BLOCK / GC4);
            const int col = i0 + gc * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float w = 0.f;
            if (row < me) {
                const float* p = G + row * ldg + col;
                if (VEC) {
                    if (col < Nc) v = ld4(p);
                } else {
                    if (col + 0 < Nc) v.x = p[0];
                    if (col + 1 < Nc) v.y = p[1];
                    if (col + 2 < Nc) v.z = p[2];
                    if (col + 3 < Nc) v.w = p[3];
                }
                if (want_bias) w = row_w != nullptr ? row_w[row * ld_w] : 1.f;
            }
            rg[s] = v;
            rwt[s] = w;
        }
#pragma unroll
        for (int s = 0; s < XF4; ++s) {
            const int64_t row = m0 + xr + s * (BLOCK / XC4);
            const int col = j0 + xc * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < me) {
                const float* p = X + row * ldx + col;
                if (VEC) {
                    if (col < K) v = ld4(p);
                } else {
                    if (col + 0 < K) v.x = p[0];
                    if (col + 1 < K) v.y = p[1];
                    if (col + 2 < K) v.z = p[2];
                    if (col + 3 < K) v.w = p[3];
                }
            }
            rx[s] = v;
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int s = 0; s < GF4; ++s) {
            st4(&Gs[buf][gr + s * (BLOCK / GC4)][gc * 4], rg[s]);
            bs.x += rwt[s] * rg[s].x;
            bs.y += rwt[s] * rg[s].y;
            bs.z += rwt[s] * rg[s].z;
            bs.w += rwt[s] * rg[s].w;
        }
#pragma unroll
        for (int s = 0; s < XF4; ++s) st4(&Xs[buf][xr + s * (BLOCK / XC4)][xc * 4], xtf.mean != nullptr ? stin_bn_relu4(rx[s], xq) : rx[s]);
    };

    const int kh = lane >> 5, li = lane & 31;
    load_slab(mb);
    store_slab(0);
    __syncthreads();
    int buf = 0;
    for (int64_t m0 = mb; m0 < me; m0 += TN_R) {
        const bool more = m0 + TN_R < me;
        if (more) load_slab(m0 + TN_R);                           // in flight during the MFMAs
#pragma unroll
        for (int kk = 0; kk < TN_R; kk += 2) {
            float a[MT], c[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = Gs[buf][kk + kh][wi * (TI / 2) + t * 32 + li];
#pragma unroll
            for (int t = 0; t < NT; ++t) c[t] = Xs[buf][kk + kh][wj * (TJ / 2) + t * 32 + li];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], c[u], acc[t][u], 0, 0, 0);
        }
        if (more) store_slab(buf ^ 1);                            // the other buffer: last read one iteration ago
        __syncthreads();
        buf ^= 1;
    }

    float* out = slab + chunk * tn_chunk_stride(Nc, Kq);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col = j0 + wj * (TJ / 2) + u * 32 + li;
        if (col >= K) continue;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * (TI / 2) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < Nc) out[(int64_t)row * Kq + col] = acc[t][u][r];
            }
    }
    if (want_bias) {                                              // block-uniform branch
        st4(&bsum[gr][gc * 4], bs);
        __syncthreads();
        if (tid < TI) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < BLOCK / GC4; ++r) t += bsum[r][tid];
            if (i0 + tid < Nc) out[(int64_t)Nc * Kq + i0 + tid] = t;
        }
    }
}

// ------------------------------------------------------------------- TN, split-bf16
// Same contract as k_gemm_tn on the bf16 matrix cores.  The reduction index is the ROW m, so the MFMA
// fragments need 8 consecutive m for one column: each staging thread loads a 4(rows) x 4(cols) fp32 patch
// (4 x 16-byte loads), splits it into NS bf16 pieces, transposes it in registers and writes one 8-byte
// (4 consecutive m) run per column into Gt[piece][col][m] / Xt[piece][col][m] (row pitch 80 B; lanes of a
// 16-lane group differ in the row group first => conflict-free ds_write_b64 and ds_read_b128).
constexpr int TNB_R = 32;                  // rows (m) per LDS slab = two MFMA k-steps of 16
constexpr int TNB_PITCH = TNB_R + 8;       // bf16 per LDS row (80 bytes)

template <int TI, int TJ, int NS, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_gemm_tn_bf16s(const float* __restrict__ G, int64_t ldg,
                                                         const float* __restrict__ X, int64_t ldx, int64_t M,
                                                         int Nc, int K, int Kq, int has_bias, const float* __restrict__ row_w,
                                                         int64_t ld_w, int rows_per_chunk, int tiles_i, int tiles_j,
                                                         int64_t chunks, float* __restrict__ slab, const stin_bn_tf xtf) {
    constexpr int MT = TI / 64, NT = TJ / 64;
    constexpr int ITEMS = 2 * (TI + TJ);                          // 4x4 patches per slab (G then X)
    constexpr int PASSES = ITEMS / BLOCK;                          // 1, 1.5 -> handled as 2 with a guard, or 2
    constexpr int NPASS = (ITEMS + BLOCK - 1) / BLOCK;
    __shared__ __attribute__((aligned(16))) __bf16 Gt[NS][TI][TNB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Xt[NS][TJ][TNB_PITCH];
    __shared__ float bsum[8][TI + 4];
    (void)PASSES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles = tiles_i * tiles_j;
    const int64_t b = blockIdx.x;
    // chunks >= 8: XCD x (blocks b = x mod 8) owns the row chunks x, x+8, ... - its tiles re-read the same rows from its own L2.
    // Fewer chunks than XCDs (the wide layers: >= 64 output tiles): plain order, so that every XCD gets tiles of every chunk.
    const int64_t xcd = b % 8, q = b / 8;
    const int64_t chunk = chunks >= 8 ? (q / tiles) * 8 + xcd : b / tiles;
    const int tile = (int)(chunks >= 8 ? q % tiles : b % tiles);
    if (chunk >= chunks) return;
    const int tj = tile % tiles_j, ti = tile / tiles_j;
    const int i0 = ti * TI, j0 = tj * TJ;
    const int64_t mb = chunk * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;
    const bool want_bias = has_bias && (tj == 0);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);                  // bias-gradient partial (columns of this thread's G patch)

    // Two register sets (128x128 tiles): the loads of slab s + 2 are issued while slab s multiplies - one slab of cover (24 MFMAs
    // = 0.35 us per wave) is less than an HBM / Infinity-Cache load takes under load.  Pays for M >= 60 k; at M = 18 k the
    // kernel is bound by the LDS round trip of the transposing split (2 barriers, 32 KB written + 64 KB read per slab), not by loads.
    struct Slab {
        float4 patch[NPASS][4];
        float pw[4];
        bool pv[4];
    };
    Slab S0, S1;
    stin_bn_coef4 xq[NPASS];                                       // (s, t) of the X patches' columns: fixed per thread and pass
    if (xtf.mean != nullptr) {                                     // (the transform itself runs at STORE time, see k_gemm_nt_bf16s)
#pragma unroll
        for (int s = 0; s < NPASS; ++s) {
            const int item = tid + s * BLOCK;
            const int c = j0 + ((item >= 2 * TI ? item - 2 * TI : 0) / 8) * 4;
            float* sp = reinterpret_cast<float*>(&xq[s].s);
            float* tp = reinterpret_cast<float*>(&xq[s].t);
#pragma unroll
            for (int e = 0; e < 4; ++e) stin_bn_st(xtf, c + e < K ? c + e : 0, sp[e], tp[e]);
        }
    }
    auto load_slab = [&](Slab& P, int64_t m0) {
#pragma unroll
        for (int s = 0; s < NPASS; ++s) {
            const int item = tid + s * BLOCK;
            const bool isG = item < 2 * TI;
            const int it = isG ? item : item - 2 * TI;
            const int rg = it % 8, c4 = it / 8;
            const bool live = item < ITEMS;
            const float* base = isG ? G : X;
            const int64_t ld = isG ? ldg : ldx;
            const int col = (isG ? i0 : j0) + c4 * 4;
            const int lim = isG ? Nc : K;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + rg * 4 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live && row < me) {
                    const float* p = base + row * ld + col;
                    if (VEC) {
                        if (col < lim) v = ld4(p);
                    } else {
                        if (col + 0 < lim) v.x = p[0];
                        if (col + 1 < lim) v.y = p[1];
                        if (col + 2 < lim) v.z = p[2];
                        if (col + 3 < lim) v.w = p[3];
                    }
                }
                P.patch[s][r] = v;
                // row weight of the bias-gradient column: a PLAIN load (clamped row, no select on the loaded value), so that
                // the compiler does not have to drain the prefetched patch loads before the MFMAs; masked in store_slab
                if (s == 0) {
                    const int64_t rc = row < me ? row : me - 1;
                    const float* wp = (want_bias && row_w != nullptr) ? row_w + rc * ld_w : G;   // always a valid address
                    P.pw[r] = *wp;
                    P.pv[r] = row < me;
                }
            }
        }
    };
    auto store_slab = [&](const Slab& P) {
#pragma unroll
        for (int s = 0; s < NPASS; ++s) {
            const int item = tid + s * BLOCK;
            if (item >= ITEMS) continue;
            const bool isG = item < 2 * TI;
            const int it = isG ? item : item - 2 * TI;
            const int rg = it % 8, c4 = it / 8;
            if (s == 0 && isG) {                                   // 2*TI >= 128: pass 0 holds every G patch of TI=128; see below for TI=64
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float w = (want_bias && P.pv[r]) ? (row_w != nullptr ? P.pw[r] : 1.f) : 0.f;
                    bs.x += w * P.patch[s][r].x;
                    bs.y += w * P.patch[s][r].y;
                    bs.z += w * P.patch[s][r].z;
                    bs.w += w * P.patch[s][r].w;
                }
            }
            __bf16* dst = isG ? &Gt[0][c4 * 4][rg * 4] : &Xt[0][c4 * 4][rg * 4];
            const int plane = (isG ? TI : TJ) * TNB_PITCH;
            float4 pr[4] = {P.patch[s][0], P.patch[s][1], P.patch[s][2], P.patch[s][3]};
            if (!isG && xtf.mean != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[r] = stin_bn_relu4(pr[r], xq[s]);
            }
            float col[4][4] = {{pr[0].x, pr[1].x, pr[2].x, pr[3].x},
                               {pr[0].y, pr[1].y, pr[2].y, pr[3].y},
                               {pr[0].z, pr[1].z, pr[2].z, pr[3].z},
                               {pr[0].w, pr[1].w, pr[2].w, pr[3].w}};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    bf16x4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h[r] = (__bf16)col[c][r];
                        col[c][r] -= (float)h[r];
                    }
                    *reinterpret_cast<bf16x4*>(dst + p * plane + c * TNB_PITCH) = h;
                }
            }
        }
    };

    const int kh = lane >> 5, li = lane & 31;
    auto multiply = [&]() {
#pragma unroll
        for (int ks = 0; ks < TNB_R; ks += 16) {
            bf16x8 a[NS][MT], c[NS][NT];
#pragma unroll
            for (int p = 0; p < NS; ++p) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    a[p][t] = *reinterpret_cast<const bf16x8*>(&Gt[p][wi * (TI / 2) + t * 32 + li][ks + 8 * kh]);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    c[p][t] = *reinterpret_cast<const bf16x8*>(&Xt[p][wj * (TJ / 2) + t * 32 + li][ks + 8 * kh]);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    if (NS == 3) {
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][t], c[1][u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][t], c[2][u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][t], c[0][u], acc[t][u], 0, 0, 0);
                    }
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][t], c[1][u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][t], c[0][u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][t], c[0][u], acc[t][u], 0, 0, 0);
                }
        }
    };
    // 128x128 tiles only: the second register set costs the narrower tiles a wave of occupancy (measured: 200 704 x 320 x 12
    // 91 -> 118 us with it, while 60 211 x 640 x 256 goes 161 -> 139 and 200 704 x 320 x 128 186 -> 143)
    constexpr bool DEEP = (TI == 128 && TJ == 128);
    load_slab(S0, mb);
    if (DEEP) {
        load_slab(S1, mb + TNB_R);                                 // (rows >= me load nothing)
        // (branch-free body: with an odd slab count the last half-iteration multiplies a slab of zeros)
        auto step = [&](Slab& P, int64_t m0) {
            __syncthreads();                                       // previous slab's fragment reads are done
            store_slab(P);
            __syncthreads();
            load_slab(P, m0 + 2 * TNB_R);                          // two slabs ahead
            multiply();
        };
        for (int64_t m0 = mb; m0 < me; m0 += 2 * TNB_R) {
            step(S0, m0);
            step(S1, m0 + TNB_R);
        }
    } else {
        for (int64_t m0 = mb; m0 < me; m0 += TNB_R) {
            __syncthreads();
            store_slab(S0);
            __syncthreads();
            if (m0 + TNB_R < me) load_slab(S0, m0 + TNB_R);        // next slab's global loads in flight during the MFMAs
            multiply();
        }
    }

    float* out = slab + chunk * tn_chunk_stride(Nc, Kq);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col = j0 + wj * (TJ / 2) + u * 32 + li;
        if (col >= K) continue;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * (TI / 2) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < Nc) out[(int64_t)row * Kq + col] = acc[t][u][r];
            }
    }
    if (want_bias) {                                              // block-uniform
        // pass-0 G patches: item = tid < 2*TI, row group tid % 8, column group tid / 8
        if (tid < 2 * TI) st4(&bsum[tid % 8][(tid / 8) * 4], bs);
        __syncthreads();
        if (tid < TI) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += bsum[r][tid];
            if (i0 + tid < Nc) out[(int64_t)Nc * Kq + i0 + tid] = t;
        }
    }
}

// =================================================================== bf16-STORAGE GEMMs
// Operands already live in HBM as bf16 rows (the *_bf16 pipeline): no split, ONE v_mfma_f32_32x32x16_bf16 per
// k-step, fp32 accumulation.  The weight operand W stays fp32 in memory (it is tiny) and is rounded to bf16 while
// it is staged.  LDS tiles are [rows][64 bf16] = 128-byte rows, the 16-byte chunk c of row r stored at chunk
// position c ^ ((r >> 1) & 7): conflict-free ds_write_b128 staging (8 lanes = one row) and ds_read_b128 fragments
// (the 16-lane read groups {0-3,12-15,20-27}, {4-11,16-19,28-31} hit 16 distinct 16-byte slots of the 256-byte
// bank span).  bf16 output tiles go through LDS so that the global stores are 16 bytes per lane, row-contiguous.
constexpr int BKB = 64;

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 h = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<uint32_t*>(&h);
}
__device__ __forceinline__ uint4 f8_to_bf16x8(float4 a, float4 b) {
    return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// Epilogue shared by the bf16-storage NT kernels: + bias * row_mask + residual in fp32, one rounding; bf16 tiles leave through
// LDS (smem: the operand tiles, free once every wave has passed the caller's last barrier) as 16-byte row-contiguous stores.
template <int BM, int BN, int WM, int WN, typename OUT, int THREADS = BLOCK>
__device__ __forceinline__ void nt_b16_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], unsigned char* smem, int64_t m0, int n0,
                                                const float* __restrict__ bias, const stin_bf16* __restrict__ row_mask,
                                                int64_t ld_mask, const stin_bf16* __restrict__ res, int64_t ld_res, int64_t M,
                                                int Nc, OUT* __restrict__ C, int64_t ldc, int vec_out) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int CPITCH = BN + 32;                               // output staging pitch (bf16): rows r, r+1 on disjoint banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kh = lane >> 5, li = lane & 31;
    // ---- epilogue: + bias * row_mask + residual in fp32, one rounding
    constexpr bool OUT_BF16 = sizeof(OUT) == 2;
    const bool staged = OUT_BF16 && vec_out;
    if (staged) __syncthreads();                                  // every wave is done with the operand tiles
    stin_bf16(*Cs)[CPITCH] = reinterpret_cast<stin_bf16(*)[CPITCH]>(smem);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int lcol = wn * TN + j * 32 + li;
        const int col = n0 + lcol;
        const bool col_ok = col < Nc;
        const float bv = (bias != nullptr && col_ok) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int64_t row = m0 + lrow;
                float t = acc[i][j][r];
                if (row < M && col_ok) {
                    if (bias != nullptr) t += row_mask != nullptr ? bv * (float)row_mask[row * ld_mask] : bv;
                    if (res != nullptr) t += (float)res[row * ld_res + col];
                }
                v[r] = t;
            }
            if (staged) {
                // lanes (li, li^1) trade one value per register pair so that each writes two adjacent columns of ONE row
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool odd = li & 1;
                    const float got = __shfl_xor(odd ? v[2 * q] : v[2 * q + 1], 1);
                    const int lrow = wm * TM + i * 32 + ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * kh + (odd ? 1 : 0);
                    const uint32_t pr = odd ? pack_bf16x2(got, v[2 * q + 1]) : pack_bf16x2(v[2 * q], got);
                    *reinterpret_cast<uint32_t*>(&Cs[lrow][lcol & ~1]) = pr;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (row < M && col_ok) st1(C + row * ldc + col, v[r]);
                }
            }
        }
    }
    if (staged) {
        __syncthreads();
        constexpr int CHUNKS = BM * BN / 8;                       // 16-byte chunks of the output tile
#pragma unroll
        for (int s = 0; s < CHUNKS / THREADS; ++s) {
            const int c = tid + s * THREADS;
            const int lrow = c / (BN / 8), lc = (c % (BN / 8)) * 8;
            const int64_t row = m0 + lrow;
            const int col = n0 + lc;
            if (row < M && col < Nc)
                *reinterpret_cast<uint4*>(reinterpret_cast<stin_bf16*>(C) + row * ldc + col) =
                    *reinterpret_cast<const uint4*>(&Cs[lrow][lc]);
        }
    }
}

template <int BM, int BN, int WM, int WN, typename OUT, bool VEC, bool WB = false>   // WB: W already holds bf16 (ldw in bf16 elements)
__global__ __launch_bounds__(BLOCK) void k_gemm_nt_b16(const stin_bf16* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ W, int64_t ldw,
                                                       const float* __restrict__ bias,
                                                       const stin_bf16* __restrict__ row_mask, int64_t ld_mask,
                                                       const stin_bf16* __restrict__ res, int64_t ld_res, int64_t M,
                                                       int Nc, int K, OUT* __restrict__ C, int64_t ldc, int vec_out) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int A_CH = BM * 8 / BLOCK, W_CH = BN * 8 / BLOCK;   // 16-byte (8 x bf16) chunks per thread per tile
    static_assert(A_CH >= 1 && W_CH >= 1, "tile too small for 256 threads");
    constexpr int CPITCH = BN + 32;                               // output staging pitch (bf16): rows r, r+1 on disjoint banks
    constexpr int TILE_BYTES = (BM + BN) * BKB * 2, OUT_BYTES = BM * CPITCH * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_BYTES > OUT_BYTES ? TILE_BYTES : OUT_BYTES];
    stin_bf16(*As)[BKB] = reinterpret_cast<stin_bf16(*)[BKB]>(smem);
    stin_bf16(*Ws)[BKB] = reinterpret_cast<stin_bf16(*)[BKB]>(smem + BM * BKB * 2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int64_t m0;
    int n0;
    if (!nt_block_tile(M, Nc, BM, BN, m0, n0)) return;                 // block-uniform
    const int ch = tid & 7, r0 = tid >> 3;                       // staging: chunk along k, first row (32 rows per pass)
    auto swz = [](int row, int chunk) { return (chunk ^ ((row >> 1) & 7)) << 3; };   // bf16 offset of a 16-byte chunk

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[A_CH], rw[W_CH];
    auto load_tiles = [&](int k0) {
        const int k = k0 + ch * 8;
#pragma unroll
        for (int s = 0; s < A_CH; ++s) {
            const int64_t row = m0 + r0 + s * 32;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < M) {
                const stin_bf16* p = A + row * lda + k;
                if (VEC) {
                    if (k < K) v = *reinterpret_cast<const uint4*>(p);
                } else {
                    uint32_t h[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (k + e < K) ? (uint32_t)*reinterpret_cast<const uint16_t*>(p + e) : 0u;
                    v = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                }
            }
            ra[s] = v;
        }
#pragma unroll
        for (int s = 0; s < W_CH; ++s) {
            const int row = n0 + r0 + s * 32;
            if constexpr (WB) {                                   // pre-converted weights: 16 bytes = 8 k-values, no conversion
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (row < Nc && k < K) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const stin_bf16*>(W) + (int64_t)row * ldw + k);
                rw[s] = v;
                continue;
            }
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (row < Nc) {
                const float* p = W + (int64_t)row * ldw + k;
                if (VEC) {
                    if (k < K) {
                        a = ld4(p);
                        b = ld4(p + 4);
                    }
                } else {
                    if (k + 0 < K) a.x = p[0];
                    if (k + 1 < K) a.y = p[1];
                    if (k + 2 < K) a.z = p[2];
                    if (k + 3 < K) a.w = p[3];
                    if (k + 4 < K) b.x = p[4];
                    if (k + 5 < K) b.y = p[5];
                    if (k + 6 < K) b.z = p[6];
                    if (k + 7 < K) b.w = p[7];
                }
            }
            rw[s] = f8_to_bf16x8(a, b);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int s = 0; s < A_CH; ++s) {
            const int row = r0 + s * 32;
            *reinterpret_cast<uint4*>(&As[row][swz(row, ch)]) = ra[s];
        }
#pragma unroll
        for (int s = 0; s < W_CH; ++s) {
            const int row = r0 + s * 32;
            *reinterpret_cast<uint4*>(&Ws[row][swz(row, ch)]) = rw[s];
        }
    };

    const int kh = lane >> 5, li = lane & 31;
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BKB) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (k0 + BKB < K) load_tiles(k0 + BKB);
#pragma unroll
        for (int ks = 0; ks < BKB / 16; ++ks) {
            bf16x8 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = wm * TM + i * 32 + li;
                a[i] = *reinterpret_cast<const bf16x8*>(&As[row][swz(row, 2 * ks + kh)]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = wn * TN + j * 32 + li;
                b[j] = *reinterpret_cast<const bf16x8*>(&Ws[row][swz(row, 2 * ks + kh)]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    nt_b16_epilogue<BM, BN, WM, WN, OUT>(acc, smem, m0, n0, bias, row_mask, ld_mask, res, ld_res, M, Nc, C, ldc, vec_out);
}

// Fat shapes of the deep hierarchies (BASELINE config 5: 8 100 rows x 1024..4096 channels, 27 k x 512..2048): both operands
// bf16, K a multiple of 64, where the register-staged 64 x 64 / 128 x 64 tiles above ran at 0.11-0.18 of the MFMA peak (one
// LDS buffer: load -> wait -> write -> barrier per k-tile).  128 x 128 tile, 4 waves of 64 x 64, the two operand tiles of a
// k-step (2 x 16 KB) staged by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave-instruction = 8 rows x 128 B, no VGPRs, no
// ds_write) into one of TWO buffers while the MFMAs of the previous k-tile run; one barrier per k-tile.  An LDS-DMA writes
// wave-linear (base + lane * 16), so the bank swizzle of the image - the 16-byte chunk c of row r sits at position
// c ^ ((r >> 1) & 7), the image k_gemm_nt_b16 reads conflict-free - is applied to the per-lane SOURCE address.  Rows past M
// / Nc are clamped to the last valid row (their products are never stored).  Same MFMA sequence per output element as
// k_gemm_nt_b16 (k ascending in steps of 16): bit-identical results.
// Block -> tile: with a multiple of 8 column tiles every XCD owns Nc / 8 columns (its W panel, 1 MB at K = 1024 and Nc = 4096,
// stays in that XCD's L2 while the A row tiles stream through, each read by the XCD's consecutive blocks); otherwise the
// row-tile-per-XCD order of nt_block_tile.
// Tile = (WM MT 32) x (WN NT 32): 128 x 128 on 2 x 2 waves of 64 x 64 (static 64 KB, two blocks per CU) or 256 x 256 on 2 x 4 waves
// of 128 x 64 (one block per CU; halves the operand re-reads through L2 and reads 6 KB of LDS per 8 MFMAs instead of 4 per 4).
template <int WM, int WN, int MT, int NT>
struct GlGeom {
    static constexpr int BM = WM * MT * 32, BN = WN * NT * 32, THREADS = 64 * WM * WN;
    static constexpr int BUF = (BM + BN) * BKB * 2;               // one k-tile of both operands
    static constexpr int OUT_BYTES = BM * (BN + 32) * 2;          // the epilogue's bf16 staging image
    static constexpr int LDS = 2 * BUF > OUT_BYTES ? 2 * BUF : OUT_BYTES;
    static constexpr int RPP = THREADS / 8;                       // rows per staging pass (8 lanes = one 128-byte row)
    static constexpr int NSA = BM / RPP, NSW = BN / RPP;
};

template <typename OUT, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN, 2) void k_gemm_nt_b16_glds(const stin_bf16* __restrict__ A, int64_t lda,
                                                                   const stin_bf16* __restrict__ W, int64_t ldw,
                                                                   const float* __restrict__ bias,
                                                                   const stin_bf16* __restrict__ row_mask, int64_t ld_mask,
                                                                   const stin_bf16* __restrict__ res, int64_t ld_res, int64_t M,
                                                                   int Nc, int K, OUT* __restrict__ C, int64_t ldc, int vec_out) {
    typedef GlGeom<WM, WN, MT, NT> Geo;
    constexpr int BM = Geo::BM, BN = Geo::BN, TM = MT * 32, TN = NT * 32, BUF = Geo::BUF;
    extern __shared__ __attribute__((aligned(1024))) unsigned char gl_smem[];
    unsigned char* smem = gl_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int64_t m0;
    int n0;
    const int ncol = (Nc + BN - 1) / BN;
    if (ncol % 8 == 0) {
        const int cpx = ncol / 8;
        const int64_t j = blockIdx.x >> 3;
        m0 = (j / cpx) * BM;
        n0 = (int)((blockIdx.x & 7) * cpx + j % cpx) * BN;
    } else if (!nt_block_tile(M, Nc, BM, BN, m0, n0)) {
        return;                                                   // block-uniform
    }
    // staging: lane -> LDS position (row tid >> 3 of a pass of RPP rows, 16-byte slot tid & 7), source chunk = slot ^ swizzle(row)
    const int slot = tid & 7, r0 = tid >> 3;
    const stin_bf16* asrc[Geo::NSA];
    const stin_bf16* wsrc[Geo::NSW];
#pragma unroll
    for (int s = 0; s < Geo::NSA; ++s) {
        const int row = r0 + Geo::RPP * s;
        const int64_t ar = m0 + row < M ? m0 + row : M - 1;
        asrc[s] = A + ar * lda + (slot ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int s = 0; s < Geo::NSW; ++s) {
        const int row = r0 + Geo::RPP * s;
        const int wr = n0 + row < Nc ? n0 + row : Nc - 1;
        wsrc[s] = W + (int64_t)wr * ldw + (slot ^ ((row >> 1) & 7)) * 8;
    }
    auto stage = [&](int buf, int k0) {
        unsigned char* base = smem + buf * BUF + wave * 1024;     // this wave's 8 rows of every pass
#pragma unroll
        for (int s = 0; s < Geo::NSA; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[s] + k0),
                                             (__attribute__((address_space(3))) void*)(base + s * (Geo::RPP * 128)), 16, 0, 0);
#pragma unroll
        for (int s = 0; s < Geo::NSW; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[s] + k0),
                                             (__attribute__((address_space(3))) void*)(base + BM * 128 + s * (Geo::RPP * 128)), 16, 0, 0);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kh = lane >> 5, li = lane & 31;
    // fragment byte offsets inside a tile: row * 128 + ((2 ks + kh) ^ ((row >> 1) & 7)) * 16
    int aoff[MT], boff[NT], asw[MT], bsw[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = wm * TM + i * 32 + li;
        aoff[i] = row * 128;
        asw[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 32 + li;
        boff[j] = BM * 128 + row * 128;
        bsw[j] = (row >> 1) & 7;
    }
    const int nt = K / BKB;
    stage(0, 0);
    __syncthreads();                                              // (hipcc drains the LDS-DMAs before the barrier)
    for (int t = 0; t < nt; ++t) {
        const unsigned char* tile = smem + (t & 1) * BUF;
        if (t + 1 < nt) stage((t + 1) & 1, (t + 1) * BKB);        // the buffer every wave finished reading before the last barrier
#pragma unroll
        for (int ks = 0; ks < BKB / 16; ++ks) {
            bf16x8 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(tile + aoff[i] + (((2 * ks + kh) ^ asw[i]) << 4));
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(tile + boff[j] + (((2 * ks + kh) ^ bsw[j]) << 4));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                          // next tile landed (vmcnt(0) of the DMAs) and this one is free
    }
    nt_b16_epilogue<BM, BN, WM, WN, OUT, Geo::THREADS>(acc, smem, m0, n0, bias, row_mask, ld_mask, res, ld_res, M, Nc, C, ldc, vec_out);
}

// dW[Nc, K(+1)] = G^T [X | w] with G, X (and the optional row weight w) stored as bf16; fp32 slabs as k_gemm_tn.
// The reduction index is the ROW m: each staging thread loads a 4(rows) x 8(cols) bf16 patch (4 x 16-byte loads),
// transposes it in registers (v_perm byte selects) and writes one 8-byte run (4 consecutive m) per column into
// Gt[col][m] / Xt[col][m] (row pitch 144 B: conflict-free ds_write_b64 across 16 row groups and ds_read_b128).
constexpr int TNK_R = 64;                  // rows (m) per LDS slab = four MFMA k-steps
constexpr int TNK_PITCH = TNK_R + 8;

template <int TI, int TJ, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_gemm_tn_b16(const stin_bf16* __restrict__ G, int64_t ldg,
                                                       const stin_bf16* __restrict__ X, int64_t ldx, int64_t M, int Nc,
                                                       int K, int Kq, int has_bias, const stin_bf16* __restrict__ row_w, int64_t ld_w,
                                                       int rows_per_chunk, int tiles_i, int tiles_j, int64_t chunks,
                                                       float* __restrict__ slab) {
    constexpr int MT = TI / 64, NT = TJ / 64;
    constexpr int ITEMS = 2 * (TI + TJ);                           // 4x8 patches per 64-row slab (G then X): 16 row groups x T/8
    constexpr int NPASS = (ITEMS + BLOCK - 1) / BLOCK;
    __shared__ __attribute__((aligned(16))) stin_bf16 Gt[TI][TNK_PITCH];
    __shared__ __attribute__((aligned(16))) stin_bf16 Xt[TJ][TNK_PITCH];
    __shared__ float bsum[16][TI + 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles = tiles_i * tiles_j;
    const int64_t b = blockIdx.x;
    // chunks >= 8: XCD x (blocks b = x mod 8) owns the row chunks x, x+8, ... - its tiles re-read the same rows from its own L2.
    // Fewer chunks than XCDs (the wide layers: >= 64 output tiles): plain order, so that every XCD gets tiles of every chunk.
    const int64_t xcd = b % 8, q = b / 8;
    const int64_t chunk = chunks >= 8 ? (q / tiles) * 8 + xcd : b / tiles;
    const int tile = (int)(chunks >= 8 ? q % tiles : b % tiles);
    if (chunk >= chunks) return;
    const int tj = tile % tiles_j, ti = tile / tiles_j;
    const int i0 = ti * TI, j0 = tj * TJ;
    const int64_t mb = chunk * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;
    const bool want_bias = has_bias && (tj == 0);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float bs[8];                                                    // bias-gradient partial: the 8 columns of this thread's G patches
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;

    uint4 patch[NPASS][4];
    uint16_t pw[NPASS][4];
    bool pv[NPASS][4];
    auto load_slab = [&](int64_t m0) {
#pragma unroll
        for (int s = 0; s < NPASS; ++s) {
            const int item = tid + s * BLOCK;
            const bool isG = item < 2 * TI;
            const int it = isG ? item : item - 2 * TI;
            const int rg = it % 16, c8 = it / 16;
            const bool live = item < ITEMS;
            const stin_bf16* base = isG ? G : X;
            const int64_t ld = isG ? ldg : ldx;
            const int col = (isG ? i0 : j0) + c8 * 8;
            const int lim = isG ? Nc : K;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + rg * 4 + r;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (live && row < me) {
                    const stin_bf16* p = base + row * ld + col;
                    if (VEC) {
                        if (col < lim) v = *reinterpret_cast<const uint4*>(p);
                    } else {
                        uint32_t h[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (col + e < lim) ? (uint32_t)*reinterpret_cast<const uint16_t*>(p + e) : 0u;
                        v = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                    }
                }
                patch[s][r] = v;
                const int64_t rc = row < me ? row : me - 1;        // plain load, masked in store_slab (see k_gemm_tn_bf16s)
                const stin_bf16* wp = (want_bias && row_w != nullptr) ? row_w + rc * ld_w : G;   // always a valid address
                pw[s][r] = *reinterpret_cast<const uint16_t*>(wp);   // raw bits: widened in store_slab, no use of the value here
                pv[s][r] = isG && live && row < me;
            }
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int s = 0; s < NPASS; ++s) {
            const int item = tid + s * BLOCK;
            if (item >= ITEMS) continue;
            const bool isG = item < 2 * TI;
            const int it = isG ? item : item - 2 * TI;
            const int rg = it % 16, c8 = it / 16;
            if (isG && want_bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint4 v = patch[s][r];
                    const float w = pv[s][r] ? (row_w != nullptr ? bf16_lo(pw[s][r]) : 1.f) : 0.f;
                    bs[0] += w * bf16_lo(v.x); bs[1] += w * bf16_hi(v.x);
                    bs[2] += w * bf16_lo(v.y); bs[3] += w * bf16_hi(v.y);
                    bs[4] += w * bf16_lo(v.z); bs[5] += w * bf16_hi(v.z);
                    bs[6] += w * bf16_lo(v.w); bs[7] += w * bf16_hi(v.w);
                }
            }
            stin_bf16* dst = isG ? &Gt[c8 * 8][rg * 4] : &Xt[c8 * 8][rg * 4];
            const uint32_t w[4][4] = {{patch[s][0].x, patch[s][0].y, patch[s][0].z, patch[s][0].w},
                                      {patch[s][1].x, patch[s][1].y, patch[s][1].z, patch[s][1].w},
                                      {patch[s][2].x, patch[s][2].y, patch[s][2].z, patch[s][2].w},
                                      {patch[s][3].x, patch[s][3].y, patch[s][3].z, patch[s][3].w}};
#pragma unroll
            for (int d = 0; d < 4; ++d) {                          // dword d of a row = columns 2d, 2d+1
                // column 2d: low halves of rows 0..3; column 2d+1: high halves
                const uint32_t lo01 = __builtin_amdgcn_perm(w[1][d], w[0][d], 0x05040100u);
                const uint32_t lo23 = __builtin_amdgcn_perm(w[3][d], w[2][d], 0x05040100u);
                const uint32_t hi01 = __builtin_amdgcn_perm(w[1][d], w[0][d], 0x07060302u);
                const uint32_t hi23 = __builtin_amdgcn_perm(w[3][d], w[2][d], 0x07060302u);
                *reinterpret_cast<uint2*>(dst + (2 * d) * TNK_PITCH) = make_uint2(lo01, lo23);
                *reinterpret_cast<uint2*>(dst + (2 * d + 1) * TNK_PITCH) = make_uint2(hi01, hi23);
            }
        }
    };

    const int kh = lane >> 5, li = lane & 31;
    load_slab(mb);
    for (int64_t m0 = mb; m0 < me; m0 += TNK_R) {
        __syncthreads();                                           // previous slab's fragment reads are done
        store_slab();
        __syncthreads();
        if (m0 + TNK_R < me) load_slab(m0 + TNK_R);                // next slab's global loads in flight during the MFMAs
#pragma unroll
        for (int ks = 0; ks < TNK_R; ks += 16) {
            bf16x8 a[MT], c[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const bf16x8*>(&Gt[wi * (TI / 2) + t * 32 + li][ks + 8 * kh]);
#pragma unroll
            for (int t = 0; t < NT; ++t) c[t] = *reinterpret_cast<const bf16x8*>(&Xt[wj * (TJ / 2) + t * 32 + li][ks + 8 * kh]);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], c[u], acc[t][u], 0, 0, 0);
        }
    }

    float* out = slab + chunk * tn_chunk_stride(Nc, Kq);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col = j0 + wj * (TJ / 2) + u * 32 + li;
        if (col >= K) continue;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * (TI / 2) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < Nc) out[(int64_t)row * Kq + col] = acc[t][u][r];
            }
    }
    if (want_bias) {                                              // block-uniform
        // G patches are items [0, 2*TI): item = tid (2*TI <= 256), row group tid % 16, column group tid / 16
        if (tid < 2 * TI) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[tid % 16][(tid / 16) * 8 + e] = bs[e];
        }
        __syncthreads();
        if (tid < TI) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += bsum[r][tid];
            if (i0 + tid < Nc) out[(int64_t)Nc * Kq + i0 + tid] = t;
        }
    }
}

// ------------------------------------------------------------ TN, bf16 storage, hardware-transposed LDS reads (round 3)
// dW tile [128 x 128] += G[m, i]^T X[m, j] over a chunk of rows m.  The reduction index is the ROW of both operands, so the
// MFMA fragments (8 consecutive k per lane) run DOWN the columns of the row-major tiles.  k_gemm_tn_b16 transposes 4 x 8
// patches in registers (v_perm) and scatters 8-byte runs into a k-major LDS image; gfx950 reads the transpose for free:
// the tiles are staged as they lie in memory ([64 rows][128 cols] bf16 = 256-byte rows, one ds_write_b128 per 16-byte chunk,
// no shuffles) and ds_read_b64_tr_b16 hands each lane of a 16-lane group one COLUMN of a 4-row x 16-column block - two of
// them are the 32x32x16 operand of the transposed tile.  Image: chunk ch of row r at position ch ^ (((r & 3) << 2) |
// ((r >> 2) & 3)) (conflict-free for the row writes and for the transposed reads; addressing checked on the device by
// profiles/micro/tr_read_check.hip).  Two LDS buffers, ONE barrier per 64-row slab: the next slab's global loads are issued
// before the MFMAs of the current one and written to the other buffer after them.  Same products in the same k order
// as k_gemm_tn_b16 (bit-identical slabs); the bias column (sum_m w[m] G[m, i]) is accumulated from the A fragments in fp32
// (other association than the staged form: equal to rounding).  128 x 128 tiles, 16-byte rows (Nc, K, ldg, ldx % 8 == 0).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lo, const unsigned char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi);
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// Tile = (WI MT 32) x (WJ NT 32) on WI x WJ waves: 128 x 128 on 2 x 2 waves of 64 x 64 (two blocks per CU), or 256 x 256 on
// 2 x 4 waves of 128 x 64 (one block per CU).  The big tile is what the fat shapes need: with 128 x 128 tiles every operand
// element is re-read Nc / 128 (K / 128) times - 1.06 GB through the L2 -> CU paths for the 8 100 x 4096 x 1024 product, more
// time than its MFMAs - the 256 x 256 tile halves that and reads 6 KB of LDS per 8 MFMAs instead of 4 KB per 4.
template <int WI, int WJ, int MT, int NT>
struct TrGeom {
    static constexpr int TI = WI * MT * 32, TJ = WJ * NT * 32, THREADS = 64 * WI * WJ;
    static constexpr int ROWB_G = TI * 2, ROWB_X = TJ * 2;                      // bytes of a tile row
    static constexpr int TILE_G = 64 * ROWB_G, TILE_X = 64 * ROWB_X, BUF = TILE_G + TILE_X;
    static constexpr int LDS = 2 * BUF + 2 * 64 * 4;                            // two buffers + two slabs of row weights
    static_assert(TI == TJ, "the staging map assumes square tiles");
    static_assert((64 * (TI / 8)) % THREADS == 0, "whole staging passes");
};
template <int ROWB>
__device__ __forceinline__ int tr_offw(int row, int ch) { return ROWB * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int WI, int WJ, int MT, int NT>
__global__ __launch_bounds__(64 * WI * WJ) void k_gemm_tn_b16_tr(const stin_bf16* __restrict__ G, int64_t ldg,
                                                                 const stin_bf16* __restrict__ X, int64_t ldx, int64_t M, int Nc,
                                                                 int K, int Kq, int has_bias, const stin_bf16* __restrict__ row_w,
                                                                 int64_t ld_w, int rows_per_chunk, int tiles_i, int tiles_j,
                                                                 int64_t chunks, float* __restrict__ slab) {
    typedef TrGeom<WI, WJ, MT, NT> Geo;
    constexpr int TI = Geo::TI, TJ = Geo::TJ, THREADS = Geo::THREADS, CI = TI / 8;
    constexpr int NS = 64 * CI / THREADS;                                      // staging passes (16 rows each)
    constexpr int RPP = THREADS / CI;                                          // rows per pass
    extern __shared__ __attribute__((aligned(16))) unsigned char tr_smem[];
    float* wrow = reinterpret_cast<float*>(tr_smem + 2 * Geo::BUF);            // [2][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave / WJ, wj = wave % WJ;
    const int tiles = tiles_i * tiles_j;
    const int64_t b = blockIdx.x;
    const int64_t xcd = b % 8, q8 = b / 8;                                     // (block -> chunk / tile as k_gemm_tn_b16)
    const int64_t chunk = chunks >= 8 ? (q8 / tiles) * 8 + xcd : b / tiles;
    const int tile = (int)(chunks >= 8 ? q8 % tiles : b % tiles);
    if (chunk >= chunks) return;
    const int tj = tile % tiles_j, ti = tile / tiles_j;
    const int i0 = ti * TI, j0 = tj * TJ;
    const int64_t mb = chunk * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;
    const bool want_bias = has_bias && (tj == 0);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float bias[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) bias[t] = 0.f;

    // staging: chunk idx = tid + THREADS s of the 64 x CI chunk grid of a tile -> row idx / CI, chunk idx % CI (whole rows per wave)
    const int sch = tid % CI, srow = tid / CI;                                 // rows srow + RPP s
    const bool g_ok = i0 + sch * 8 < Nc, x_ok = j0 + sch * 8 < K;
    const stin_bf16* gsrc = G + i0 + (g_ok ? sch * 8 : 0);
    const stin_bf16* xsrc = X + j0 + (x_ok ? sch * 8 : 0);
    int soff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) soff[s] = tr_offw<Geo::ROWB_G>(srow + RPP * s, sch);
    uint4 rg[NS], rx[NS];
    float rw = 0.f;
    auto load_slab = [&](int64_t m0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t row = m0 + srow + RPP * s;
            const int64_t rc = row < me ? row : me - 1;                         // clamped address, masked value
            const uint4 vg = *reinterpret_cast<const uint4*>(gsrc + rc * ldg);
            const uint4 vx = *reinterpret_cast<const uint4*>(xsrc + rc * ldx);
            const bool live = row < me;
            rg[s] = (live && g_ok) ? vg : make_uint4(0u, 0u, 0u, 0u);
            rx[s] = (live && x_ok) ? vx : make_uint4(0u, 0u, 0u, 0u);
        }
        if (want_bias && tid < 64) {
            const int64_t row = m0 + tid;
            rw = row < me ? (row_w != nullptr ? (float)row_w[row * ld_w] : 1.f) : 0.f;
        }
    };
    auto store_slab = [&](int buf) {
        unsigned char* base = tr_smem + buf * Geo::BUF;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            *reinterpret_cast<uint4*>(base + soff[s]) = rg[s];
            *reinterpret_cast<uint4*>(base + Geo::TILE_G + soff[s]) = rx[s];
        }
        if (want_bias && tid < 64) wrow[buf * 64 + tid] = rw;
    };

    // transposed-read addresses of this lane for k-step 0 (a k-step further down = + 16 rows)
    const int g16 = lane >> 4, tl = lane & 15, tq = tl >> 2, tp = tl & 3;
    int aoff[MT][2], boff[NT][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * (g16 >> 1) + 4 * h + tq;
#pragma unroll
        for (int t = 0; t < MT; ++t)
            aoff[t][h] = tr_offw<Geo::ROWB_G>(row, (wi * (MT * 32) + t * 32 + 16 * (g16 & 1)) / 8 + (tp >> 1)) + 8 * (tp & 1);
#pragma unroll
        for (int u = 0; u < NT; ++u)
            boff[u][h] = Geo::TILE_G + tr_offw<Geo::ROWB_X>(row, (wj * (NT * 32) + u * 32 + 16 * (g16 & 1)) / 8 + (tp >> 1)) + 8 * (tp & 1);
    }
    const int kh = lane >> 5, li = lane & 31;

    load_slab(mb);
    store_slab(0);
    __syncthreads();
    int buf = 0;
    for (int64_t m0 = mb; m0 < me; m0 += TNK_R) {
        const bool more = m0 + TNK_R < me;
        if (more) load_slab(m0 + TNK_R);                                       // in flight during the MFMAs below
        const unsigned char* base = tr_smem + buf * Geo::BUF;
#pragma unroll
        for (int ks = 0; ks < TNK_R / 16; ++ks) {
            bf16x8 a[MT], c[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t)
                a[t] = tr_frag(base + ks * 16 * Geo::ROWB_G + aoff[t][0], base + ks * 16 * Geo::ROWB_G + aoff[t][1]);
#pragma unroll
            for (int u = 0; u < NT; ++u)
                c[u] = tr_frag(base + ks * 16 * Geo::ROWB_X + boff[u][0], base + ks * 16 * Geo::ROWB_X + boff[u][1]);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], c[u], acc[t][u], 0, 0, 0);
            if (want_bias && wj == 0) {                                        // wave-uniform
                const float* w = wrow + buf * 64 + ks * 16 + 8 * kh;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bias[t] += w[e] * (float)a[t][e];
            }
        }
        if (more) store_slab(buf ^ 1);                                         // the buffer nobody reads in this iteration
        __syncthreads();
        buf ^= 1;
    }

    float* out = slab + chunk * tn_chunk_stride(Nc, Kq);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col = j0 + wj * (NT * 32) + u * 32 + li;
        if (col >= K) continue;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * (MT * 32) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < Nc) out[(int64_t)row * Kq + col] = acc[t][u][r];
            }
    }
    if (want_bias && wj == 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const float tot = bias[t] + __shfl_xor(bias[t], 32);               // the two k-halves of the wave
            const int row = i0 + wi * (MT * 32) + t * 32 + li;
            if (kh == 0 && row < Nc) out[(int64_t)Nc * Kq + row] = tot;
        }
    }
}

// ----------------------------------------------------------------------------- TN, skinny K (round 4)
// dW[Nc, K] = G^T [X | w] with K <= 16 (the first block of the network: [dA | dB | g] against the 12 padded input channels,
// 200 704 x 320 x 12).  On the MFMA kernels above this shape stages 128 x 64 tiles that are four fifths empty and streams G at
// 2.5 TB/s (102 us for 267 MB).  It is pure streaming with 13 multiply-adds per G element, so: no matrix cores, no LDS staging -
// a block owns ALL Nc columns of a chunk of rows; thread (row lane, column lane) holds 4 columns x (KP + 1) fp32 accumulators,
// loads one 16-byte piece of a G row and the row's K values of X (the same addresses for every column lane of the row lane: L1
// hits) per trip, UR rows in flight; rows ascending per thread, then the row lanes in order through LDS - fixed order, exact fp32
// products (precision >= every split mode).  Partial results go to the usual slab layout ([Nc][Kq] + bias [Nc] per chunk), so
// k_reduce_slabs / k_wgrad_finalize fold them unchanged.
template <int KP>
__global__ __launch_bounds__(BLOCK) void k_gemm_tn_skinny(const float* __restrict__ G, int64_t ldg, const float* __restrict__ X, int64_t ldx,
                                                         int64_t M, int Nc, int K, int Kq, int has_bias,
                                                         const float* __restrict__ row_weight, int64_t ld_weight, int rows_per_chunk,
                                                         float* __restrict__ slab) {
    constexpr int UR = 8;
    const int CL = Nc / 4, RL = BLOCK / CL;                                   // column lanes per row, row lanes (BLOCK - CL RL threads idle)
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const bool live = rl < RL;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < M ? r0 + rows_per_chunk : M;
    const int nrows = (int)(r1 - r0);
    // the chunk's [X | w] rows once into LDS ((KP + 4) floats per row: K values, the row weight, padding to 16 bytes): the inner
    // loop then issues ONE global load per row (the G piece) and reads its X row as LDS broadcasts
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];           // [rows_per_chunk][KP + 4], later [Nc][KP + 1]
    constexpr int XP = KP + 4;
    for (int i = threadIdx.x; i < nrows * (KP / 4); i += BLOCK) {
        const int r = i / (KP / 4), q = i % (KP / 4);
        *reinterpret_cast<float4*>(sk_smem + r * XP + q * 4) = ld4(X + (r0 + r) * ldx + q * 4);
    }
    for (int r = threadIdx.x; r < nrows; r += BLOCK) sk_smem[r * XP + KP] = row_weight != nullptr ? row_weight[(r0 + r) * ld_weight] : 1.f;
    __syncthreads();
    float acc[4][KP + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k <= KP; ++k) acc[c][k] = 0.f;
    if (live) {
        const float* Gp = G + r0 * ldg + cl * 4;
        for (int rb = rl; rb < nrows; rb += UR * RL) {
            float4 g[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = rb + u * RL;
                g[u] = ld4(Gp + (int64_t)(r < nrows ? r : rb) * ldg);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = rb + u * RL;
                if (r >= nrows) continue;
                const float* xr = sk_smem + r * XP;
                const float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
                const float w = xr[KP];
#pragma unroll
                for (int q = 0; q < KP / 4; ++q) {
                    const float4 xv = *reinterpret_cast<const float4*>(xr + q * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c][4 * q + 0] += gv[c] * xv.x;
                        acc[c][4 * q + 1] += gv[c] * xv.y;
                        acc[c][4 * q + 2] += gv[c] * xv.z;
                        acc[c][4 * q + 3] += gv[c] * xv.w;
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c][KP] += gv[c] * w;
            }
        }
    }
    __syncthreads();                                                          // (the X image is dead: its space takes the column sums)
    // row lanes in order: row lane 0 stores its sums into the block's [Nc][KP + 1] image, 1 .. RL - 1 add in turn (fixed order)
    for (int t = 0; t < RL; ++t) {
        if (live && rl == t) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k <= KP; ++k) {
                    float* dst = sk_smem + (cl * 4 + c) * (KP + 1) + k;
                    *dst = t == 0 ? acc[c][k] : *dst + acc[c][k];
                }
        }
        __syncthreads();
    }
    float* out = slab + (int64_t)blockIdx.x * tn_chunk_stride(Nc, Kq);
    for (int i = threadIdx.x; i < Nc * Kq; i += BLOCK) {
        const int c = i / Kq, k = i % Kq;
        out[i] = k < K ? sk_smem[c * (KP + 1) + k] : 0.f;
    }
    if (has_bias)
        for (int c = threadIdx.x; c < Nc; c += BLOCK) out[(int64_t)Nc * Kq + c] = sk_smem[c * (KP + 1) + KP];
}
inline bool tn_skinny_shape(int storage, int Nc, int K, int64_t ldg, int64_t ldx, const void* G, const void* X) {
    return storage == 0 && K <= 16 && K % 4 == 0 && Nc % 4 == 0 && Nc >= 64 && Nc <= 4 * BLOCK && ldg % 4 == 0 && ldx % 4 == 0 &&
           stin_aligned16(G) && stin_aligned16(X);
}
inline int tn_skinny_rows(int64_t M) {                                         // ~1024 chunks, a multiple of 32 rows, at least 128
    constexpr int want = 1024;                                                  // measured: 256 / 512 / 1024 / 2048 chunks: 106 / 75 / 72 / 79 us at 200 704 x 320 x 12
    int64_t rows = (M + want - 1) / want;
    rows = (rows + 31) / 32 * 32;
    // the chunk's [X | w] image is rows x (KP + 4 <= 20) floats of dynamic LDS: capped at 512 rows = 40 KB (below the 64 KB a
    // launch gets without hipFuncSetAttribute) - beyond ~0.5 M rows the chunk COUNT grows instead of the chunk
    // (stin_gemm_tn_workspace_bytes sizes the slab from the same function)
    if (rows > 512) rows = 512;
    return (int)(rows < 128 ? 128 : rows);
}

// dW[row][col] = sum_c slab[c][row][col] (and the bias column from the chunk's bias block): 16 chunk-lanes x 16 float4
// groups per block, each chunk-lane walks the chunk list with stride 16 (4 x 16-byte loads in flight), then a
// fixed-order LDS reduction over the chunk-lanes -> deterministic.
constexpr int RS_COLS = 16, RS_KL = 16;
__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__global__ __launch_bounds__(BLOCK) void k_reduce_slabs(const float* __restrict__ slab, int64_t chunks, int Nc, int K, int Kq,
                                                        int has_bias, float* __restrict__ out, int64_t ldo,
                                                        float* __restrict__ bias_out) {
    __shared__ float4 sm[RS_KL][RS_COLS + 1];
    const int tx = threadIdx.x % RS_COLS, ty = threadIdx.x / RS_COLS;
    const int64_t cs = tn_chunk_stride(Nc, Kq);
    const int64_t nw4 = (int64_t)Nc * Kq / 4, nb4 = has_bias ? (Nc + 3) / 4 : 0;
    const int64_t g = (int64_t)blockIdx.x * RS_COLS + tx;          // float4 group: weight block first, then the bias block
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (g < nw4 + nb4) {
        const float* p = slab + 4 * g;
        int64_t c = ty;
        for (; c + 15 * RS_KL < chunks; c += 16 * RS_KL) {          // (sixteen loads in flight on long chunk lists; same addition order)
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = ld4(p + (c + u * RS_KL) * cs);
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                add4(s0, v[u]);
                add4(s1, v[u + 1]);
                add4(s2, v[u + 2]);
                add4(s3, v[u + 3]);
            }
        }
        for (; c + 3 * RS_KL < chunks; c += 4 * RS_KL) {
            add4(s0, ld4(p + c * cs));
            add4(s1, ld4(p + (c + RS_KL) * cs));
            add4(s2, ld4(p + (c + 2 * RS_KL) * cs));
            add4(s3, ld4(p + (c + 3 * RS_KL) * cs));
        }
        for (; c < chunks; c += RS_KL) add4(s0, ld4(p + c * cs));
    }
    add4(s0, s1);
    add4(s2, s3);
    add4(s0, s2);
    sm[ty][tx] = s0;
    __syncthreads();
    if (ty == 0 && g < nw4 + nb4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < RS_KL; ++k) add4(s, sm[k][tx]);
        const float v[4] = {s.x, s.y, s.z, s.w};
        if (g < nw4) {
            const int64_t row = (4 * g) / Kq;
            const int col = (int)((4 * g) % Kq);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < K) out[row * ldo + col + e] = v[e];
        } else {
            const int64_t i = 4 * (g - nw4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < Nc) {
                    if (bias_out != nullptr) bias_out[i + e] = v[e];          // (stin_gemm_tn_wb_*: db as its own vector)
                    else out[(i + e) * ldo + K] = v[e];
                }
        }
    }
}

inline int tn_tile(int n) {
    return n > 64 ? 128 : 64;    // (measured: 64x64 wgrad tiles are 2-5 % slower end to end at any slab count)
}

// (the STIN_NT_TILE sweep aid of rounds 1-5 is gone: 0 = the measured rule below)
inline int stin_nt_force_tile() { return 0; }

inline int stin_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

inline int tn_rows_per_chunk(int64_t M, int tiles, bool one_per_cu = false) {
    // ~384 blocks (up to 2 resident per CU, one round), chunks a multiple of the LDS slab.  Round 1 measured 512 best of
    // 256..1536; with loads two slabs ahead a block hides more latency by itself and fewer chunks mean fewer partial slabs to
    // store and reduce: 384 is 0.3-0.5 % faster on the step than 512 (256: 0.4 % slower)
    // one_per_cu (round 3: fp32 bf16x3 products on 128 x 128 tiles with M <= 32 k rows, the shapes of the bottleneck level):
    // 256 blocks - the producer / consumer kernel's fixed cost per block (first-load latency ~5 k cycles, slab store ~9 k) is
    // a sixth of a 750-row chunk; measured 18 063 x 1024 x 256 49.7 -> 45.7 us, block weight gradients 69 -> 62 us, while the
    // 60 k-row products lose 10 % with 256 blocks and keep 384
    const int want = one_per_cu ? 256 : 384;
    int64_t chunks = (want + tiles - 1) / tiles;
    if (chunks > 8) chunks = (chunks + 7) / 8 * 8;          // whole rounds of 8 chunks (one per XCD, see the kernels' block map)
    int64_t rows = (M + chunks - 1) / chunks;
    if (rows < 4 * TN_R) rows = 4 * TN_R;
    constexpr int max_rows = 128 * TN_R;
    if (rows > max_rows) rows = max_rows;
    rows = (rows + TN_R - 1) / TN_R * TN_R;
    return (int)rows;
}
// bf16-storage NT: when the 128 x 128 LDS-DMA kernel (k_gemm_nt_b16_glds) replaces the register-staged tiles.  It needs enough
// k-tiles to amortise its prologue and enough 128 x 128 tiles to fill the chip; the tall-skinny level-0 / level-1 shapes
// (K <= 128, bound by their output bytes) stay on the small tiles.  STIN_NT_GLDS = 0 | 1 forces (re-read per call).
inline bool nt_b16_glds_pays(int64_t M, int Nc, int K) {
    const char* e = getenv("STIN_NT_GLDS");
    if (e) return atoi(e) != 0;
    // measured (profiles/r03_nt_bf16_fat.md): +10..25 % from K = 1024 up (8 100 x 4096 x 1024: 124 -> 110 us, x 1024 x 4096: 104 -> 82),
    // a loss at K <= 512 where four to eight k-tiles do not amortise the two-buffer prologue and the output bytes dominate
    return K >= 1024 && Nc >= 256 && ((M + 127) / 128) * ((Nc + 127) / 128) >= 128;
}
// bf16-storage NT on the LDS-DMA kernel: 256 x 256 tiles when they still fill the chip (>= 200 tiles).  STIN_NT_BIG = 0 | 1 forces.
inline bool nt_b16_big_tile(int64_t M, int Nc, int K) {
    const char* e = getenv("STIN_NT_BIG");
    if (e) return atoi(e) != 0;
    (void)K;
    return Nc >= 512 && ((M + 255) / 256) * ((Nc + 255) / 256) >= 200;
}
// bf16-storage TN: 256 x 256 tiles (k_gemm_tn_b16_tr<2, 4, 4, 2>) for the fat products.  STIN_TN_BIG = 0 | 1 forces (re-read per call;
// the workspace bound covers both choices).
inline bool tn_b16_big_tile(int Nc, int K) {
    const char* e = getenv("STIN_TN_BIG");
    if (e) return atoi(e) != 0 && Nc >= 256 && K >= 256;
    return Nc >= 512 && K >= 512;
}
// bf16-storage TN, 128 x 128 tiles: the transposed-read kernel (k_gemm_tn_b16_tr); STIN_TN_TR=0 keeps the register-transpose
// kernel (A/B switch, re-read per call).
inline bool tn_b16_tr_enabled(int Nc, int K) {
    const char* e = getenv("STIN_TN_TR");
    if (e) return atoi(e) != 0;
    // whole tiles only: with a ragged third tile (161 362 x 320 x 128, the level-0 product of the crop batches) the 64 KB of LDS
    // (two blocks per CU instead of four) cost more than the transposes: 54.6 -> 68.6 us; whole-tile shapes gain 5-17 %
    return Nc % 128 == 0 && K % 128 == 0;
}
// Strip-kernel configuration (see k_gemm_nt_strip): 21 / 22 / 41.  STIN_STRIP_CFG overrides (tuning aid).
inline int strip_config(int64_t M, int Nc, int KC) {
    const char* e = getenv("STIN_STRIP_CFG");                         // re-read per call: profiles/gemm_shapes.py flips it
    const int forced = e ? atoi(e) : 0;
    if (forced == 21 || forced == 22 || forced == 41) return forced;
    // measured (profiles/r03_strip_cfg.md): two quads gain 2-5 % where a strip has many panels (18 063 x 1024 x 256: 46.6 -> 44.4 us,
    // x 1280 x 128 43.3 -> 42.1, 60 211 x 640 x 256 79.7 -> 77.8, 200 704 x 320 x 128 135.5 -> 133.2) and lose 3 % at Nc = 512
    // (30.2 -> 31.2); MT = 4 on one quad (one wave per SIMD) is 25-40 % slower everywhere and stays a tested tuning variant
    (void)M; (void)KC;
    return Nc >= 640 ? 22 : 21;
}
// all-columns NT kernel: waves along the rows of a block (see k_gemm_nt_wide).  8 waves per block (128 rows) for Nc = 256
// while the 128-row blocks fit the chip in one round: 18 063 x 256 x 1024 41.6 -> 36.7 us, x 512 29.3 -> 27.7; slower for 60 k
// rows (471 blocks: 59 -> 64 us) and for Nc = 128 (256-row blocks: 71 of them at 18 k rows, 34 -> 52 us), which keep 4 waves.
inline int wide_waves_m(int64_t M, int Nc) {
    const int nw = Nc / 64;
    return (Nc == 256 && (M + 127) / 128 <= (int64_t)stin_cu_count()) ? 2 : 4 / nw;
}
// Balanced column-panel kernel (k_gemm_nt_panel): 32-row tiles per block (MT0 + MT1, 2 .. 9) so that (row blocks) x (128-column
// panels) fills the chip in ONE round, or 0 = this shape stays on the strip / all-columns kernels.  Measured (MI355X, round 4,
// strip / all-columns -> panel, bf16x3): 18 063 x 256 x 1024 43.4 -> 36.7-38.8 us, x 256 x 512 26.4 -> 23.3, x 512 x 256 31.3 -> 24.2,
// x 128 x 1280 35.1 -> 28.8; grids of two and more rounds (one 147 KB block per CU at a time: nothing overlaps a block's
// prologue and its 5 us store phase) are no faster than the strip kernel (18 063 x 1024 x 256 44.0 -> 42.9-46.7, 60 211 x 640 x 256
// 80.6 -> 93.1) and stay there.  STIN_NT_PANEL = 0 disables, = 1 forces it for every fragment-order shape whatever the number
// of rounds (re-read per call: profiles/gemm_shapes.py flips it).
inline int panel_tiles(int64_t M, int Nc, int K) {
    const char* e = getenv("STIN_NT_PANEL");
    const int forced = e ? atoi(e) : -1;
    if (forced == 0 || Nc % 128 != 0 || K % WD_KC != 0 || M <= 0) return 0;
    const int P = Nc / 128;
    const int64_t rg = (M + 31) / 32, cu = stin_cu_count();
    const int max_rounds = forced == 1 ? 64 : 1;
    for (int rounds = 1; rounds <= max_rounds; ++rounds) {
        const int64_t slots = cu * rounds / P;
        if (slots < 1) continue;
        const int64_t mts = (rg + slots - 1) / slots;
        // (fewer row groups than slots - the coarse levels of a small crop, 1 806 x 256 x 1024: the finest blocks, 64 rows; the
        // all-columns kernel ran that shape on 15 workgroups, 41.8 us; STIN_NT_PANEL_SMALL=0 keeps it there)
        if (mts <= 9) return mts < 2 ? 2 : (int)mts;
    }
    return 0;
}
inline bool tn_one_per_cu(int storage, int precision, int TI, int TJ, int64_t M) {
    return storage == 0 && precision == STIN_GEMM_BF16X3 && TI == 128 && TJ == 128 && M <= 32768;
}

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
        static bool attr_set = false;                                                                                     \
        if (!attr_set) {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_panel<PT_, A_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                              \
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
        static bool attr_set = false;                                                                                     \
        if (!attr_set) {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_strip<PT_, MT_, QM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                              \
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
        static bool attr_set = false;                                                                                                \
        if (!attr_set) {                                                                                                             \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_stream<KC_, NT_, NS_, PT_, MODE, TF>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                                   \
            attr_set = true;                                                                                                         \
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
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)k_gemm_tn_b16_tr<WI_, WJ_, MT_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo_::LDS); \
            attr_set = true;                                                                                          \
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
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            (void)hipFuncSetAttribute((const void*)k_gemm_nt_b16_glds<OUT_, WM_, WN_, MT_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo_::LDS); \
            attr_set = true;                                                                                           \
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
