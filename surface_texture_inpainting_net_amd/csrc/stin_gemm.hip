// Per-vertex dense GEMMs of the STINet hot path on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf chain, 157 TFLOP/s chip peak).
//
//   stin_gemm_nt_f32 : C[M, Nc] = A[M, K] . W[Nc, K]^T (+ bias)      forward GEMMs and dgrad (with W^T)
//   stin_gemm_tn_f32 : dW[Nc, K(+1)] = G[M, Nc]^T . [X[M, K] | 1]    weight (+bias) gradients, split over M
//
// M is the vertex count (1e4..1e6), Nc and K are channel counts (3..2052): tall-skinny shapes where
// library GEMMs pick poor tiles.  Fragment maps (cdna_hip_programming.md §3): A operand lane l holds
// A[i = l&31][k = l>>5], B operand B[k = l>>5][j = l&31]; C/D reg r of lane l is
// row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
#include "stin_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BLOCK = 256;
constexpr int BK = 32;

// ----------------------------------------------------------------------------- NT
// Block tile BM x BN, 4 waves as WM x WN, each wave (BM/WM) x (BN/WN) = MT x NT MFMA tiles of 32x32.
// A and W tiles are staged K-major in LDS ([BK][rows + 1]: conflict-free transposed stores and
// stride-1 fragment reads); the next tile's global loads are issued before the current tile's MFMAs.
template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_gemm_nt(const float* __restrict__ A, int64_t lda,
                                                   const float* __restrict__ W, int64_t ldw,
                                                   const float* __restrict__ bias, int64_t M, int Nc, int K,
                                                   float* __restrict__ C, int64_t ldc) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int A_F4 = BM * BK / 4 / BLOCK, W_F4 = BN * BK / 4 / BLOCK;   // float4 per thread per tile
    static_assert(A_F4 >= 1 && W_F4 >= 1, "tile too small for 256 threads");
    __shared__ float As[BK][BM + 1];
    __shared__ float Ws[BK][BN + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
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
    auto load_tiles = [&](int k0) {
        const int k = k0 + kq * 4;
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
                if (row < M) C[row * ldc + col] = acc[i][j][r] + bv;
            }
        }
    }
}

// ----------------------------------------------------------------------------- TN
// One WAVE per (64-row i-tile of Nc, 64-col j-tile of K, row chunk): fragments come straight from
// global memory (for the reduction over rows m both operands are lane-contiguous: lane l reads
// G[m + (l>>5)][i0 + (l&31)] and X[m + (l>>5)][j0 + (l&31)], 2 x 128-byte segments per instruction).
// The bias gradient (column sums of G) rides along on the VALU in the j-tile-0 waves and lands in
// slab column K.  Partial tiles go to a slab [chunk][Nc][Kp]; k_reduce_slabs sums them in chunk order
// (deterministic).  Work items are numbered so that the waves of one row chunk share an XCD
// (blockIdx % 8 is the observed XCD round-robin): the chunk's G / X rows are then served by that L2.
constexpr int TN_UNROLL = 8;   // m-pairs in flight per wave (32 dword loads)

__global__ __launch_bounds__(BLOCK) void k_gemm_tn(const float* __restrict__ G, int64_t ldg,
                                                   const float* __restrict__ X, int64_t ldx, int64_t M, int Nc,
                                                   int K, int Kp, int rows_per_chunk, int tiles_i, int tiles_j,
                                                   int64_t chunks, float* __restrict__ slab) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // block b -> (xcd-affine chunk, tile group); 4 waves = 4 consecutive tiles of the same chunk
    const int tiles = tiles_i * tiles_j;
    const int groups = (tiles + 3) / 4;                        // blocks per chunk
    const int64_t b = blockIdx.x;
    const int64_t xcd = b % 8, q = b / 8;
    const int64_t chunk = (q / groups) * 8 + xcd;
    const int tile = (int)(q % groups) * 4 + wave;             // wave-uniform
    if (chunk >= chunks || tile >= tiles) return;
    const int tj = tile % tiles_j, ti = tile / tiles_j;
    const int kh = lane >> 5, li = lane & 31;
    const int i0 = ti * 64, j0 = tj * 64;
    const int64_t mb = chunk * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float gsum[2] = {0.f, 0.f};

    const int ci[2] = {i0 + li, i0 + 32 + li};
    const int cj[2] = {j0 + li, j0 + 32 + li};
    const bool gi[2] = {ci[0] < Nc, ci[1] < Nc};
    const bool xj[2] = {cj[0] < K, cj[1] < K};

    for (int64_t m = mb; m < me; m += 2 * TN_UNROLL) {
        float g[TN_UNROLL][2], x[TN_UNROLL][2];
#pragma unroll
        for (int u = 0; u < TN_UNROLL; ++u) {
            const int64_t row = m + 2 * u + kh;
            const bool ok = row < me;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                g[u][t] = (ok && gi[t]) ? G[row * ldg + ci[t]] : 0.f;
                x[u][t] = (ok && xj[t]) ? X[row * ldx + cj[t]] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < TN_UNROLL; ++u) {
            gsum[0] += g[u][0];
            gsum[1] += g[u][1];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[u][a], x[u][c], acc[a][c], 0, 0, 0);
        }
    }

    float* out = slab + chunk * (int64_t)Nc * Kp;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = j0 + c * 32 + li;
        if (col >= K) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < Nc) out[(int64_t)row * Kp + col] = acc[a][c][r];
            }
    }
    if (Kp > K && tj == 0) {   // bias-gradient column: even rows (lanes 0-31) + odd rows (lanes 32-63)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float other = __shfl_xor(gsum[t], 32);
            if (kh == 0 && gi[t]) out[(int64_t)ci[t] * Kp + K] = gsum[t] + other;
        }
    }
}

__global__ void k_reduce_slabs(const float* __restrict__ slab, int64_t chunks, int64_t n, int Kp,
                               float* __restrict__ out, int64_t ldo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t c = 0;
    for (; c + 3 < chunks; c += 4) {
        s0 += slab[c * n + t];
        s1 += slab[(c + 1) * n + t];
        s2 += slab[(c + 2) * n + t];
        s3 += slab[(c + 3) * n + t];
    }
    for (; c < chunks; ++c) s0 += slab[c * n + t];
    out[(t / Kp) * ldo + (t % Kp)] = (s0 + s1) + (s2 + s3);
}

inline int tn_rows_per_chunk(int64_t M, int tiles) {
    // aim for >= ~2048 waves in flight, at least 256 and at most 2048 rows per wave, even row counts
    int64_t chunks = (4096 + tiles - 1) / tiles;
    int64_t rows = (M + chunks - 1) / chunks;
    if (rows < 256) rows = 256;
    if (rows > 2048) rows = 2048;
    rows = (rows + 2 * TN_UNROLL - 1) / (2 * TN_UNROLL) * (2 * TN_UNROLL);
    return (int)rows;
}

}  // namespace

extern "C" int stin_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, int64_t M,
                                int Nc, int K, float* C, int64_t ldc, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(M >= 0 && Nc > 0 && K > 0 && lda >= K && ldw >= K && ldc >= Nc, STIN_E_SIZE);
    if (M == 0) return STIN_OK;
    STIN_REQUIRE(A && W && C, STIN_E_NULL);
    const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldw % 4 == 0) && stin_aligned16(A) && stin_aligned16(W);
#define STIN_NT(BM_, BN_, WM_, WN_)                                                                              \
    do {                                                                                                         \
        dim3 grid((unsigned)((M + BM_ - 1) / BM_), (unsigned)((Nc + BN_ - 1) / BN_));                            \
        if (vec) hipLaunchKernelGGL((k_gemm_nt<BM_, BN_, WM_, WN_, true>), grid, dim3(BLOCK), 0, stream, A, lda, W, ldw, bias, M, Nc, K, C, ldc); \
        else hipLaunchKernelGGL((k_gemm_nt<BM_, BN_, WM_, WN_, false>), grid, dim3(BLOCK), 0, stream, A, lda, W, ldw, bias, M, Nc, K, C, ldc);    \
    } while (0)
    // Tile choice: the largest tile that still leaves >= ~6 blocks per CU (256 CUs), so that the tail
    // wave of blocks does not idle half the chip on the M ~ 2e4 levels.
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Nc + bn - 1) / bn); };
    if (Nc <= 32) STIN_NT(128, 32, 4, 1);
    else if (Nc % 128 == 0 && blocks(128, 128) >= 1536) STIN_NT(128, 128, 2, 2);
    else if (blocks(128, 64) >= 1536) STIN_NT(128, 64, 2, 2);
    else STIN_NT(64, 64, 2, 2);
#undef STIN_NT
    return stin_launch_status();
}

extern "C" size_t stin_gemm_tn_workspace_bytes(int64_t M, int Nc, int K, int ones_column) {
    if (M < 0 || Nc <= 0 || K <= 0) return 0;
    const int Kp = K + (ones_column ? 1 : 0);
    const int tiles = ((Nc + 63) / 64) * ((K + 63) / 64);
    const int rows = tn_rows_per_chunk(M, tiles);
    const int64_t chunks = (M + rows - 1) / rows;
    return (size_t)(chunks > 0 ? chunks : 1) * Nc * Kp * sizeof(float) + 256;
}

extern "C" int stin_gemm_tn_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t M, int Nc, int K,
                                int ones_column, float* dW, int64_t lddw, void* workspace, size_t workspace_bytes,
                                stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    const int Kp = K + (ones_column ? 1 : 0);
    STIN_REQUIRE(M >= 0 && Nc > 0 && K > 0 && ldg >= Nc && ldx >= K && lddw >= Kp, STIN_E_SIZE);
    STIN_REQUIRE(dW && workspace && (M == 0 || (G && X)), STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_gemm_tn_workspace_bytes(M, Nc, K, ones_column), STIN_E_WORKSPACE);
    float* slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int tiles_i = (Nc + 63) / 64, tiles_j = (K + 63) / 64;
    const int rows = tn_rows_per_chunk(M, tiles_i * tiles_j);
    const int64_t chunks = M > 0 ? (M + rows - 1) / rows : 0;
    const int64_t n = (int64_t)Nc * Kp;
    if (chunks > 0) {
        const int64_t groups = (tiles_i * tiles_j + 3) / 4;
        const int64_t blocks = ((chunks + 7) / 8) * 8 * groups;      // 8 chunks (one per XCD) x groups per round
        hipLaunchKernelGGL(k_gemm_tn, dim3((unsigned)blocks), dim3(BLOCK), 0, stream, G, ldg, X, ldx, M, Nc, K, Kp, rows,
                           tiles_i, tiles_j, chunks, slab);
    }
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, slab, chunks, n,
                       Kp, dW, lddw);
    return stin_launch_status();
}
