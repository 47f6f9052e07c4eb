// Round 6 experiment (review item 2): the split-16-bit NT product with BOTH operands already split in memory ("split rows")
// and staged by LDS-DMA - no vector-ALU work, no register staging and no ds_write in the K loop.
//
// Split-row layout of a matrix X [R][K] (K % 32 == 0), the SAME bytes and row pitch as its fp32 rows: per row and 32-wide k chunk
// 128 bytes = [hi(k0 .. k0+31) 64 B | lo(k0 .. k0+31) 64 B], hi = PT(x * scale), lo = PT(x * scale - hi) - the arithmetic of
// stin_gemm.hip's split_store<2, PT>.  One (row, chunk) is one 128-byte line.
//
// k_nt_planes: C[M, Nc] = A W^T on a (32 (MT0 + MT1)) x (128 NTW) tile per 512-thread block: 8 waves = 2 row groups x 4 column
// groups, wave (q, wn) owns MTq row tiles x NTW column tiles.  Per 32-wide k chunk the (BM + BN) x 128 B operand image is copied
// global -> LDS by global_load_lds_dwordx4 (1 KB = 8 rows per wave-instruction, the XOR bank swizzle applied to the per-lane
// SOURCE address) into an S-deep ring; counted s_waitcnt vmcnt + ONE raw s_barrier per chunk; fragments by ds_read_b128.
// MFMA order per k-step (a_hi b_lo, a_lo b_hi, a_hi b_hi), k ascending, and the epilogue expression are those of the shipped split
// kernels: bit-identical results.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename PT> struct Piece;
template <> struct Piece<__bf16> { typedef bf16x8 vec8; };
template <> struct Piece<_Float16> { typedef f16x8 vec8; };
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------ split rows
template <typename PT>
__global__ __launch_bounds__(256) void k_split_rows(const float* __restrict__ X, int64_t ldx, int64_t M, int K, float scale,
                                                    PT* __restrict__ P) {
    typedef PT pt4 __attribute__((ext_vector_type(4)));
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k4 = K / 4;
    if (idx >= M * k4) return;
    const int64_t row = idx / k4;
    const int k = (int)(idx % k4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(X + row * ldx + k);
    float r[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    pt4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (PT)r[i];
        r[i] -= (float)h[i];
        l[i] = (PT)r[i];
    }
    PT* dst = P + row * (2 * (int64_t)K) + (k >> 5) * 64 + (k & 31);
    *reinterpret_cast<pt4*>(dst) = h;
    *reinterpret_cast<pt4*>(dst + 32) = l;
}

extern "C" int planes_split_rows(const float* X, int64_t ldx, int64_t M, int K, int f16, float scale, void* P, hipStream_t stream) {
    if (K % 32 != 0 || ldx % 4 != 0) return -1;
    const int64_t n = M * (K / 4);
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (f16) hipLaunchKernelGGL(k_split_rows<_Float16>, dim3(grid), dim3(256), 0, stream, X, ldx, M, K, scale, (_Float16*)P);
    else hipLaunchKernelGGL(k_split_rows<__bf16>, dim3(grid), dim3(256), 0, stream, X, ldx, M, K, scale, (__bf16*)P);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ NT on split rows
#define VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <typename PT, int MT0, int MT1, int NTW, int S, int EPI>
__global__ __launch_bounds__(512) void k_nt_planes(const unsigned char* __restrict__ Ap, int64_t lda_b,
                                                   const unsigned char* __restrict__ Wp, int64_t ldw_b,
                                                   const float* __restrict__ bias, const float* __restrict__ row_mask, int64_t ld_mask,
                                                   const float* __restrict__ res, int64_t ld_res, int64_t M, int Nc, int K,
                                                   float* __restrict__ C, int64_t ldc, int nrb, int P, int xcd_map, float sc,
                                                   unsigned long long* __restrict__ stamps) {
    typedef typename Piece<PT>::vec8 vec8;
#define STAMP(i)                                                                                                   \
    do {                                                                                                           \
        if (stamps != nullptr && lane == 0 && (i) < 64)                                                            \
            stamps[((size_t)blockIdx.x * 8 + wave) * 64 + (i)] = __builtin_amdgcn_s_memtime();                     \
    } while (0)
    constexpr int MTS = MT0 + MT1, BM = 32 * MTS, BN = 128 * NTW, ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;
    constexpr int NG = ROWS / 8;                       // 1 KB row groups per stage
    constexpr int NI = (NG + 7) / 8;                   // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave >> 2, wn = wave & 3;
    const int kh = lane >> 5, li = lane & 31;
    int rb, p;
    if (xcd_map) {
        const int slot = blockIdx.x >> 3;
        rb = (slot / P) * 8 + (blockIdx.x & 7);
        p = slot % P;
    } else {
        rb = blockIdx.x / P;
        p = blockIdx.x % P;
    }
    if (rb >= nrb) return;
    STAMP(0);
    const int64_t row0 = (int64_t)rb * BM;
    const int n0 = p * BN;
    const int nchunk = K / 32;

    // ---- DMA sources: instruction s of this wave copies row group g = wave + 8 s (clamped: the last groups are copied twice)
    const unsigned char* src[NI];
    int dst_off[NI];
#pragma unroll
    for (int s = 0; s < NI; ++s) {
        int g = wave + 8 * s;
        if (g >= NG) g = NG - 1;
        const int r = g * 8 + (lane >> 3);                 // row of the stage image: [0, BM) = A rows, [BM, ROWS) = W rows
        const int logical = (lane & 7) ^ ((r >> 1) & 7);
        if (r < BM) {
            int64_t gr = row0 + r;
            if (gr > M - 1) gr = M - 1;
            src[s] = Ap + gr * lda_b + logical * 16;
        } else {
            int wr = n0 + (r - BM);
            if (wr > Nc - 1) wr = Nc - 1;
            src[s] = Wp + (int64_t)wr * ldw_b + logical * 16;
        }
        dst_off[s] = g * 1024;
    }
    auto issue = [&](int slot, int chunk) {
        unsigned char* base = smem + slot * STAGE;
#pragma unroll
        for (int s = 0; s < NI; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[s] + (int64_t)chunk * 128),
                                             (__attribute__((address_space(3))) void*)(base + dst_off[s]), 16, 0, 0);
    };

    auto role = [&](auto MTc) {
        constexpr int MT = decltype(MTc)::value;
        const int trow = q * (MT0 * 32);
        f32x16 acc[MT][NTW];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // fragment offsets: row * 128 + ((plane * 4 + 2 ks + kh) ^ ((row >> 1) & 7)) * 16; (row >> 1) & 7 == (li >> 1) & 7 for every tile
        const int sw = (li >> 1) & 7;
        const int x0 = ((kh) ^ sw) << 4, x1 = ((2 + kh) ^ sw) << 4;
        const int a_base = (trow + li) * 128;
        const int b_base = (BM + wn * (32 * NTW) + li) * 128;

#pragma unroll
        for (int s = 0; s < S - 1; ++s) issue(s, s < nchunk ? s : nchunk - 1);
        STAMP(1);
        for (int t = 0; t < nchunk; ++t) {
            VMCNT((S - 2) * NI);                            // this wave's copies of chunk t have landed
            __builtin_amdgcn_s_barrier();                   // ... everybody's have, and everybody is done with chunk t - 1
            {
                const int nx = t + S - 1;
                issue(nx % S, nx < nchunk ? nx : nchunk - 1);   // (past the end: a copy nobody reads, keeps the counts uniform)
            }
            const unsigned char* tile = smem + (t % S) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int xo = ks ? x1 : x0;
                vec8 ah[MT], al[MT], bh[NTW], bl[NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    bh[j] = *reinterpret_cast<const vec8*>(tile + b_base + j * 4096 + xo);
                    bl[j] = *reinterpret_cast<const vec8*>(tile + b_base + j * 4096 + (xo ^ 64));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    ah[i] = *reinterpret_cast<const vec8*>(tile + a_base + i * 4096 + xo);
                    al[i] = *reinterpret_cast<const vec8*>(tile + a_base + i * 4096 + (xo ^ 64));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        acc[i][j] = mfma16(ah[i], bl[j], acc[i][j]);
                        acc[i][j] = mfma16(al[i], bh[j], acc[i][j]);
                        acc[i][j] = mfma16(ah[i], bh[j], acc[i][j]);
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of this chunk has returned before the next barrier
            STAMP(2 + (t < 56 ? t : 56));
        }
        STAMP(60);
        // ---- epilogue: v = acc * sc + bias [* row mask] (+ residual)
        if (EPI == 0) {                                     // direct dword stores: 2 rows x 128 B per instruction
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int col = n0 + wn * (32 * NTW) + j * 32 + li;
                const bool col_ok = col < Nc;
                const int colc = col_ok ? col : Nc - 1;
                const float bv = bias != nullptr ? bias[colc] : 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int64_t trow0 = row0 + trow + i * 32 + 4 * kh;
                    float v[16];
                    if (row_mask != nullptr) {              // (uniform branches: every load of a tile is issued before the first use)
                        float mk[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            int64_t row = trow0 + (r & 3) + 8 * (r >> 2);
                            if (row > M - 1) row = M - 1;
                            mk[r] = row_mask[row * ld_mask];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] * sc + bv * mk[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] * sc + bv;
                    }
                    if (res != nullptr) {
                        float rs[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            int64_t row = trow0 + (r & 3) + 8 * (r >> 2);
                            if (row > M - 1) row = M - 1;
                            rs[r] = res[row * ld_res + colc];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] += rs[r];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = trow0 + (r & 3) + 8 * (r >> 2);
                        if (row < M && col_ok) C[row * ldc + col] = v[r];
                    }
                }
            }
        } else {                                            // restaged through LDS: 16-byte row-contiguous stores
            __builtin_amdgcn_s_barrier();                   // the ring is idle (every wave is past its last fragment read)
            VMCNT(0);                                       // ... and no copy of this wave is still on its way into it
            __builtin_amdgcn_s_barrier();
            float* stage_f = reinterpret_cast<float*>(smem + wave * 4096);
            const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int colw = n0 + wn * (32 * NTW) + j * 32;
                const float bv = (bias != nullptr && colw + li < Nc) ? bias[colw + li] : 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int64_t trow0 = row0 + trow + i * 32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int tr = (r & 3) + 8 * (r >> 2) + 4 * kh;
                        int64_t mrow = trow0 + tr;
                        if (mrow > M - 1) mrow = M - 1;
                        stage_f[tr * 32 + li] = acc[i][j][r] * sc + (row_mask != nullptr ? bv * row_mask[mrow * ld_mask] : bv);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const int tr = t4 * 8 + r8;
                        const int64_t grow = trow0 + tr;
                        float4 v = *reinterpret_cast<const float4*>(stage_f + tr * 32 + c8 * 4);
                        if (grow < M && colw + c8 * 4 < Nc) {
                            if (res != nullptr) {
                                const float4 rv = *reinterpret_cast<const float4*>(res + grow * ld_res + colw + c8 * 4);
                                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                            }
                            *reinterpret_cast<float4*>(C + grow * ldc + colw + c8 * 4) = v;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
        STAMP(61);
        VMCNT(0);
        STAMP(62);
    };
    if (q == 0) role(std::integral_constant<int, MT0>());
    else role(std::integral_constant<int, MT1>());
#undef STAMP
}

template <typename PT, int MT0, int MT1, int NTW, int S, int EPI>
static int launch_nt(const void* Ap, int64_t lda_b, const void* Wp, int64_t ldw_b, const float* bias, const float* row_mask, int64_t ld_mask,
                     const float* res, int64_t ld_res, int64_t M, int Nc, int K, float* C, int64_t ldc, float sc, hipStream_t stream,
                     unsigned long long* stamps) {
    constexpr int BM = 32 * (MT0 + MT1), BN = 128 * NTW;
    const int nrb = (int)((M + BM - 1) / BM), P = (Nc + BN - 1) / BN;
    const int xmap = nrb >= 16 ? 1 : 0;
    const unsigned grid = (unsigned)((xmap ? ((nrb + 7) / 8) * 8 : nrb) * P);
    const size_t lds = (size_t)S * (BM + BN) * 128;
    auto kern = k_nt_planes<PT, MT0, MT1, NTW, S, EPI>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, (const unsigned char*)Ap, lda_b, (const unsigned char*)Wp, ldw_b, bias,
                       row_mask, ld_mask, res, ld_res, M, Nc, K, C, ldc, nrb, P, xmap, sc, stamps);
    return (int)hipGetLastError();
}

// cfg = MT0 * 1000 + MT1 * 100 + NTW * 10 + S; epi: 0 direct dword stores, 1 restaged float4 stores
extern "C" int planes_gemm_nt(const void* Ap, int64_t lda_b, const void* Wp, int64_t ldw_b, const float* bias, const float* row_mask,
                              int64_t ld_mask, const float* res, int64_t ld_res, int64_t M, int Nc, int K, float* C, int64_t ldc,
                              int f16, float sc, int cfg, int epi, hipStream_t stream, unsigned long long* stamps) {
    if (K % 32 != 0) return -1;
#define ARGS Ap, lda_b, Wp, ldw_b, bias, row_mask, ld_mask, res, ld_res, M, Nc, K, C, ldc, sc, stream, stamps
#define CASE(A_, B_, N_, S_)                                                                         \
    if (cfg == A_ * 1000 + B_ * 100 + N_ * 10 + S_) {                                                \
        if (f16) return epi ? launch_nt<_Float16, A_, B_, N_, S_, 1>(ARGS) : launch_nt<_Float16, A_, B_, N_, S_, 0>(ARGS); \
        return epi ? launch_nt<__bf16, A_, B_, N_, S_, 1>(ARGS) : launch_nt<__bf16, A_, B_, N_, S_, 0>(ARGS);            \
    }
    CASE(5, 4, 2, 2)
    CASE(3, 2, 1, 4)
    CASE(3, 2, 1, 3)
    CASE(3, 2, 2, 3)
    CASE(5, 4, 1, 2)
    CASE(5, 4, 1, 3)
    CASE(2, 1, 2, 3)
    CASE(3, 2, 2, 2)
#undef CASE
#undef ARGS
    return -2;
}
