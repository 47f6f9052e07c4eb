// The network's last layer - nn.Linear(ngf, output_nc) followed by tanh (reference models/surfacetextureinpaintingnet.py:461-471,
// `final_linear2` + `torch.tanh`) - as ONE launch per direction for gfx950.  Contract: include/stin_hip.h.
//
// The layer maps K = ngf (64 .. 256) channels to Nc <= 4 colour channels: as a matrix product it is a 200 704 x 3 x 64 problem
// that leaves every matrix-core tile 29 / 32 empty, and inside a training step it cost seven launches each way (product, tanh;
// tanh', transposed product + slab fold, weight transpose, input-gradient product, two slices of the weight | bias gradient).
// Both directions are bandwidth problems (one pass over the [N, K] activations), so they are written as such:
//   fwd: 16 lanes per row, 4 channels per lane and 64-channel chunk, the Nc weight rows live in registers; Nc dot products per
//        row, summed over the row's lanes by a butterfly of DPP shuffles, tanh, 4-byte stores.  Exact fp32 products.
//   bwd: dz = g (1 - y^2) per row; dx = dz W (each lane its own channels), dW += dz^T x and db += dz accumulated per lane over
//        the block's rows, folded over the block (shuffles across a wave's four rows, LDS across the 16 waves) into ONE partial
//        per block; the partials are published write-through (sc1) behind a ticket and the LAST block to arrive folds them in a
//        fixed order into dW / db (the scheme of k_colreduce_t, stin_norm.hip) - deterministic, no second launch, no atomics on
//        floating-point data.
#include <atomic>
#include <cstdlib>
#include "stin_common.h"

namespace {

constexpr int TL_LPR = 16;                       // lanes per row: 4 channels each, 64 channels per chunk
constexpr int TL_FWD_BLOCK = 256;                // 16 rows per block trip
constexpr int TL_BWD_BLOCK = 1024;               // 64 rows per block trip, 16 waves
constexpr int TL_SLOTS = 64;
__device__ unsigned int g_tail_tickets[TL_SLOTS];

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
__device__ __forceinline__ float group_sum16(float v) {   // sum over the 16 lanes of a row: every lane ends with the same value
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
}

template <typename T, int NC, int CH>
__global__ __launch_bounds__(TL_FWD_BLOCK) void k_linear_tanh_fwd(const T* __restrict__ x, int64_t ldx, const float* __restrict__ W,
                                                                  const float* __restrict__ b, int64_t N, int K,
                                                                  float* __restrict__ y) {
    constexpr int UR = 4, RPB = TL_FWD_BLOCK / TL_LPR;
    const int lg = threadIdx.x % TL_LPR, rg = threadIdx.x / TL_LPR;
    float4 w[NC][CH];
    bool on[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int k = (j * TL_LPR + lg) * 4;
        on[j] = k < K;
#pragma unroll
        for (int c = 0; c < NC; ++c) w[c][j] = on[j] ? ld4(W + (int64_t)c * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float bias[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) bias[c] = b != nullptr ? b[c] : 0.f;
    const int64_t step = (int64_t)gridDim.x * RPB;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB + rg; r0 < N; r0 += UR * step) {
        float4 xv[UR][CH];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t row = r0 + u * step;
            const int64_t rc = row < N ? row : r0;
#pragma unroll
            for (int j = 0; j < CH; ++j)
                xv[u][j] = on[j] ? ld4(x + rc * ldx + (j * TL_LPR + lg) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t row = r0 + u * step;
            float out = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < CH; ++j) s += dot4(xv[u][j], w[c][j]);
                s = group_sum16(s) + bias[c];
                if (lg == c) out = s;
            }
            if (row < N && lg < NC) y[row * NC + lg] = tanhf(out);
        }
    }
}

// partial [blocks][NC * K + NC] floats (weight block then bias block), dW [NC, K], db [NC] (may be NULL)
template <typename T, int NC, int CH>
__global__ __launch_bounds__(TL_BWD_BLOCK) void k_linear_tanh_bwd(const float* __restrict__ g, const float* __restrict__ y,
                                                                  const T* __restrict__ x, int64_t ldx,
                                                                  const float* __restrict__ W, int64_t N, int K,
                                                                  T* __restrict__ dx, int64_t lddx, float* partial, int slot,
                                                                  float* __restrict__ dW, float* __restrict__ db) {
    constexpr int UR = 2, RPB = TL_BWD_BLOCK / TL_LPR, WAVES = TL_BWD_BLOCK / 64;
    extern __shared__ __attribute__((aligned(16))) float tl_smem[];          // [WAVES][O], later [parts][O]
    __shared__ int last_s;
    const int O = NC * K + NC;
    const int lg = threadIdx.x % TL_LPR, rg = threadIdx.x / TL_LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 w[NC][CH], aw[NC][CH];
    float ab[NC];
    bool on[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int k = (j * TL_LPR + lg) * 4;
        on[j] = k < K;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            w[c][j] = on[j] ? ld4(W + (int64_t)c * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            aw[c][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) ab[c] = 0.f;
    const int64_t step = (int64_t)gridDim.x * RPB;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB + rg; r0 < N; r0 += UR * step) {
        float4 xv[UR][CH];
        float gv[UR][NC], yv[UR][NC];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t row = r0 + u * step;
            const int64_t rc = row < N ? row : r0;
#pragma unroll
            for (int j = 0; j < CH; ++j)
                xv[u][j] = on[j] ? ld4(x + rc * ldx + (j * TL_LPR + lg) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                gv[u][c] = g[rc * NC + c];
                yv[u][c] = y[rc * NC + c];
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t row = r0 + u * step;
            if (row >= N) continue;
            float dz[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                dz[c] = gv[u][c] * (1.f - yv[u][c] * yv[u][c]);
                ab[c] += dz[c];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    d.x += dz[c] * w[c][j].x;
                    d.y += dz[c] * w[c][j].y;
                    d.z += dz[c] * w[c][j].z;
                    d.w += dz[c] * w[c][j].w;
                    aw[c][j].x += dz[c] * xv[u][j].x;
                    aw[c][j].y += dz[c] * xv[u][j].y;
                    aw[c][j].z += dz[c] * xv[u][j].z;
                    aw[c][j].w += dz[c] * xv[u][j].w;
                }
                if (dx != nullptr && on[j]) st4(dx + row * lddx + (j * TL_LPR + lg) * 4, d);
            }
        }
    }
    // ---- the block's partial: the wave's four rows by shuffles (fixed order), the 16 waves through LDS in wave order
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float* v = reinterpret_cast<float*>(&aw[c][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += __shfl_xor(v[e], 16, 64);
                v[e] += __shfl_xor(v[e], 32, 64);
            }
            if (lane < TL_LPR && on[j]) st4(tl_smem + (int64_t)wave * O + (int64_t)c * K + (j * TL_LPR + lg) * 4, aw[c][j]);
        }
        ab[c] += __shfl_xor(ab[c], 16, 64);
        ab[c] += __shfl_xor(ab[c], 32, 64);
        if (lane == 0) tl_smem[(int64_t)wave * O + NC * K + c] = ab[c];
    }
    __syncthreads();
    float* mine = partial + (int64_t)blockIdx.x * O;
    for (int o = threadIdx.x; o < O; o += TL_BWD_BLOCK) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) t += tl_smem[k * O + o];
        __hip_atomic_store(mine + o, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);              // write-through (sc1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              // every storing wave drains
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int* word = &g_tail_tickets[slot];
        const unsigned int t = __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1) ? 1 : 0;
        if (last) __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the slot's next launch
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    // ---- the fold (sc1 loads only): thread (part, o) sums the partials p = part, part + parts, ...; then the parts in order
    const int P = (int)gridDim.x;
    // (round 5) at most WAVES parts: the parts' sums go into the [WAVES][O] floats of dynamic LDS this launch was given.  With
    // O < 64 (K <= 16) the unclamped TL_BWD_BLOCK / O exceeded that, the hardware dropped the out-of-range LDS writes, and on
    // grids of more than ~17 blocks the partials of the dropped parts were missing from dW (found by the 5-level fixture g13:
    // K = 4 on 3 025 vertices = 24 blocks; every earlier small-K test ran on fewer blocks than parts)
    const int parts_fit = TL_BWD_BLOCK / O > 0 ? TL_BWD_BLOCK / O : 1;
    const int parts = parts_fit < WAVES ? parts_fit : WAVES;
    constexpr int FU = 16;
    for (int t0 = threadIdx.x; t0 < parts * O; t0 += TL_BWD_BLOCK) {       // (one trip: parts * O <= TL_BWD_BLOCK unless O > it)
        const int o = t0 % O, part = t0 / O;
        float f = 0.f;
        for (int p0 = part; p0 < P; p0 += FU * parts) {
            float v[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int p = p0 + u * parts;
                v[u] = __hip_atomic_load(partial + (int64_t)(p < P ? p : part) * O + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < FU; ++u)
                if (p0 + u * parts < P) f += v[u];
        }
        tl_smem[part * O + o] = f;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < O; o += TL_BWD_BLOCK) {
        float t = 0.f;
        for (int k = 0; k < parts; ++k) t += tl_smem[k * O + o];
        if (o < NC * K) dW[o] = t;
        else if (db != nullptr) db[o - NC * K] = t;
    }
}

// out[n, 0:Cp) = [x[n, 0:Cin) | 0]: the network input (10 channels: reference datasets/scannetcolorgraph_dataloader.py x layout)
// padded to the 16-byte rows the first block's GEMM reads - one launch instead of the framework's fill + copy pair
template <typename T>
__global__ __launch_bounds__(256) void k_pad_rows(const T* __restrict__ x, int64_t ldx, int64_t N, int Cin, int Cp, T* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * Cp) return;
    const int64_t r = t / Cp;
    const int c = (int)(t % Cp);
    out[t] = c < Cin ? x[r * ldx + c] : (T)0.f;
}

// dst[n, c] += alpha * src[n, c] * [row n has an in-edge]  for c in [c0, c1): the translation-invariant SAGE message
// x_j[:, 3:9] - x_i[:, 3:9] (models/modules/sage_conv_filter.py:87-90) under the mean aggregation is
// mean_j x_j - x_i [deg_i > 0] on those columns - applied in place to the aggregated rows (alpha = -1), and to the input
// gradient in the backward pass.  One thread per (row, column) of the slice.
template <typename T>
__global__ __launch_bounds__(256) void k_cols_axpy_rowmask(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src, int64_t lds_,
                                                           const int32_t* __restrict__ rowptr, int64_t N, int c0, int c1, float alpha) {
    const int W = c1 - c0;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * W) return;
    const int64_t n = t / W;
    const int c = c0 + (int)(t % W);
    if (rowptr[n + 1] > rowptr[n]) st1(dst + n * ldd + c, ld1(dst + n * ldd + c) + alpha * ld1(src + n * lds_ + c));
}

inline int tail_cu_count() { return stin_cu_count_dev(); }          // (per device: stin_common.h)
inline int64_t tail_bwd_blocks(int64_t N) {
    constexpr int per_cu_x4 = 4;                                     // blocks = CUs x this / 4
    int64_t blocks = (N + 2 * (TL_BWD_BLOCK / TL_LPR) - 1) / (2 * (TL_BWD_BLOCK / TL_LPR));
    const int64_t cap = (int64_t)tail_cu_count() * per_cu_x4 / 4;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}
inline bool tail_shape_ok(int K, int Nc) { return K > 0 && K % 4 == 0 && K <= 256 && Nc >= 1 && Nc <= 4 && Nc * K + Nc <= TL_BWD_BLOCK; }

template <typename T>
int linear_tanh_fwd_impl(const T* x, int64_t ldx, const float* W, const float* b, int64_t N, int K, int Nc, float* y, hipStream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && ldx >= K, STIN_E_SIZE);
    STIN_REQUIRE(tail_shape_ok(K, Nc), STIN_E_UNSUPPORTED);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && W && y, STIN_E_NULL);
    STIN_REQUIRE(ldx % 4 == 0 && stin_aligned_vec4<T>(x) && stin_aligned16(W), STIN_E_ALIGN);
    const int rpb = TL_FWD_BLOCK / TL_LPR;
    int64_t blocks = (N + 4 * rpb - 1) / (4 * rpb);
    const int64_t cap = (int64_t)tail_cu_count() * 8;
    if (blocks > cap) blocks = cap;
    const int CH = K <= 64 ? 1 : (K <= 128 ? 2 : 4);
#define STIN_TL_F(NC_, CH_) hipLaunchKernelGGL((k_linear_tanh_fwd<T, NC_, CH_>), dim3((unsigned)blocks), dim3(TL_FWD_BLOCK), 0, stream, x, ldx, W, b, N, K, y)
#define STIN_TL_FC(NC_)                       \
    do {                                      \
        if (CH == 1) STIN_TL_F(NC_, 1);       \
        else if (CH == 2) STIN_TL_F(NC_, 2);  \
        else STIN_TL_F(NC_, 4);               \
    } while (0)
    switch (Nc) {
        case 1: STIN_TL_FC(1); break;
        case 2: STIN_TL_FC(2); break;
        case 3: STIN_TL_FC(3); break;
        default: STIN_TL_FC(4); break;
    }
#undef STIN_TL_FC
#undef STIN_TL_F
    return stin_launch_status();
}

template <typename T>
int linear_tanh_bwd_impl(const float* g, const float* y, const T* x, int64_t ldx, const float* W, int64_t N, int K, int Nc, T* dx,
                         int64_t lddx, float* dW, float* db, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && ldx >= K && (dx == nullptr || lddx >= K), STIN_E_SIZE);
    STIN_REQUIRE(tail_shape_ok(K, Nc), STIN_E_UNSUPPORTED);
    STIN_REQUIRE(dW && workspace && (N == 0 || (g && y && x && W)), STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_linear_tanh_bwd_workspace_bytes(N, K, Nc), STIN_E_WORKSPACE);
    STIN_REQUIRE(ldx % 4 == 0 && stin_aligned_vec4<T>(x) && stin_aligned16(W) && (dx == nullptr || (lddx % 4 == 0 && stin_aligned_vec4<T>(dx))),
                 STIN_E_ALIGN);
    const int O = Nc * K + Nc;
    if (N == 0) {
        hipError_t e = hipMemsetAsync(dW, 0, (size_t)Nc * K * sizeof(float), stream);
        if (e == hipSuccess && db != nullptr) e = hipMemsetAsync(db, 0, (size_t)Nc * sizeof(float), stream);
        return (int)e;
    }
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t blocks = tail_bwd_blocks(N);
    static std::atomic<unsigned> seq{0}, seq_cap{0};
    const int slot = stin_ticket_slot(seq, seq_cap, TL_SLOTS, stream);
    const size_t lds = (size_t)(TL_BWD_BLOCK / 64) * O * sizeof(float);
    const int CH = K <= 64 ? 1 : (K <= 128 ? 2 : 4);
#define STIN_TL_B(NC_, CH_)                                                                                                          \
    do {                                                                                                                             \
        static stin_once_per_device attr_once;                                                                                                \
        if (attr_once.first()) {                                                                                                             \
            (void)hipFuncSetAttribute((const void*)k_linear_tanh_bwd<T, NC_, CH_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
        }                                                                                                                            \
        hipLaunchKernelGGL((k_linear_tanh_bwd<T, NC_, CH_>), dim3((unsigned)blocks), dim3(TL_BWD_BLOCK), lds, stream, g, y, x, ldx, W, N, K, \
                           dx, lddx, partial, slot, dW, db);                                                                         \
    } while (0)
#define STIN_TL_BC(NC_)                       \
    do {                                      \
        if (CH == 1) STIN_TL_B(NC_, 1);       \
        else if (CH == 2) STIN_TL_B(NC_, 2);  \
        else STIN_TL_B(NC_, 4);               \
    } while (0)
    switch (Nc) {
        case 1: STIN_TL_BC(1); break;
        case 2: STIN_TL_BC(2); break;
        case 3: STIN_TL_BC(3); break;
        default: STIN_TL_BC(4); break;
    }
#undef STIN_TL_BC
#undef STIN_TL_B
    return stin_launch_status();
}

}  // namespace

extern "C" size_t stin_linear_tanh_bwd_workspace_bytes(int64_t N, int K, int Nc) {
    if (N < 0 || !tail_shape_ok(K, Nc)) return 0;
    // sized for the largest grid the tuning aid can ask for (8 blocks per CU), not only the default one
    const int64_t blocks = (int64_t)tail_cu_count() * 8;
    return (size_t)blocks * (size_t)(Nc * K + Nc) * sizeof(float) + 256;
}

extern "C" int stin_linear_tanh_fwd_f32(const float* x, int64_t ldx, const float* W, const float* b, int64_t N, int K, int Nc,
                                        float* y, stin_stream_t stream) {
    return linear_tanh_fwd_impl<float>(x, ldx, W, b, N, K, Nc, y, (hipStream_t)stream);
}
extern "C" int stin_linear_tanh_fwd_bf16(const stin_bf16_t* x, int64_t ldx, const float* W, const float* b, int64_t N, int K, int Nc,
                                         float* y, stin_stream_t stream) {
    return linear_tanh_fwd_impl<stin_bf16>(reinterpret_cast<const stin_bf16*>(x), ldx, W, b, N, K, Nc, y, (hipStream_t)stream);
}
extern "C" int stin_linear_tanh_bwd_f32(const float* g, const float* y, const float* x, int64_t ldx, const float* W, int64_t N, int K,
                                        int Nc, float* dx, int64_t lddx, float* dW, float* db, void* workspace,
                                        size_t workspace_bytes, stin_stream_t stream) {
    return linear_tanh_bwd_impl<float>(g, y, x, ldx, W, N, K, Nc, dx, lddx, dW, db, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int stin_linear_tanh_bwd_bf16(const float* g, const float* y, const stin_bf16_t* x, int64_t ldx, const float* W, int64_t N,
                                         int K, int Nc, stin_bf16_t* dx, int64_t lddx, float* dW, float* db, void* workspace,
                                         size_t workspace_bytes, stin_stream_t stream) {
    return linear_tanh_bwd_impl<stin_bf16>(g, y, reinterpret_cast<const stin_bf16*>(x), ldx, W, N, K, Nc,
                                           reinterpret_cast<stin_bf16*>(dx), lddx, dW, db, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int stin_pad_rows_f32(const float* x, int64_t ldx, int64_t N, int Cin, int Cp, float* out, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && Cin > 0 && Cp >= Cin && ldx >= Cin, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && out, STIN_E_NULL);
    hipLaunchKernelGGL(k_pad_rows<float>, dim3((unsigned)((N * Cp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, N, Cin, Cp, out);
    return stin_launch_status();
}
extern "C" int stin_pad_rows_bf16(const stin_bf16_t* x, int64_t ldx, int64_t N, int Cin, int Cp, stin_bf16_t* out, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && Cin > 0 && Cp >= Cin && ldx >= Cin, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && out, STIN_E_NULL);
    hipLaunchKernelGGL(k_pad_rows<stin_bf16>, dim3((unsigned)((N * Cp + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const stin_bf16*>(x), ldx, N, Cin, Cp, reinterpret_cast<stin_bf16*>(out));
    return stin_launch_status();
}

extern "C" int stin_cols_axpy_rowmask_f32(float* dst, int64_t ldd, const float* src, int64_t lds_, const int32_t* rowptr, int64_t N,
                                          int c0, int c1, float alpha, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && c0 >= 0 && c1 >= c0 && ldd >= c1 && lds_ >= c1, STIN_E_SIZE);
    if (N == 0 || c1 == c0) return STIN_OK;
    STIN_REQUIRE(dst && src && rowptr, STIN_E_NULL);
    hipLaunchKernelGGL(k_cols_axpy_rowmask<float>, dim3((unsigned)((N * (c1 - c0) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, ldd,
                       src, lds_, rowptr, N, c0, c1, alpha);
    return stin_launch_status();
}
extern "C" int stin_cols_axpy_rowmask_bf16(stin_bf16_t* dst, int64_t ldd, const stin_bf16_t* src, int64_t lds_, const int32_t* rowptr,
                                           int64_t N, int c0, int c1, float alpha, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && c0 >= 0 && c1 >= c0 && ldd >= c1 && lds_ >= c1, STIN_E_SIZE);
    if (N == 0 || c1 == c0) return STIN_OK;
    STIN_REQUIRE(dst && src && rowptr, STIN_E_NULL);
    hipLaunchKernelGGL(k_cols_axpy_rowmask<stin_bf16>, dim3((unsigned)((N * (c1 - c0) + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<stin_bf16*>(dst), ldd, reinterpret_cast<const stin_bf16*>(src), lds_, rowptr, N, c0, c1, alpha);
    return stin_launch_status();
}
