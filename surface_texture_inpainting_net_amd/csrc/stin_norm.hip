// Column statistics (fixed-order fp64 accumulation, two-stage, no atomics) and the fused
// normalise / ELU / residual epilogues of GraphResnetBlock for gfx950.
// Contract: include/stin_hip.h.  All of these are HBM-streaming kernels: 16-byte loads,
// one pass over [N, C] per call.
#include "stin_common.h"
#include <atomic>
#include <cstdlib>

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_SLABS = 1024;  // B * chunks-per-segment upper bound (workspace sizing)

template <int VW> struct V;
template <> struct V<4> {
    float v[4];
    template <typename T> __device__ __forceinline__ static V load(const T* p) { V r; float4 t = ld4(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r; }
    template <typename T> __device__ __forceinline__ void store(T* p) const { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
};
// 8 channels per lane: a full 16-byte load of bf16 rows (round 3: the 4-wide form moved 8 bytes per lane on bf16 storage)
template <> struct V<8> {
    float v[8];
    __device__ __forceinline__ static V load(const float* p) {
        V r;
        const float4 a = ld4(p), b = ld4(p + 4);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        return r;
    }
    __device__ __forceinline__ static V load(const stin_bf16* p) {
        V r;
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
        r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
        r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
        r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
        return r;
    }
    __device__ __forceinline__ void store(float* p) const {
        st4(p, make_float4(v[0], v[1], v[2], v[3]));
        st4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
    }
    __device__ __forceinline__ void store(stin_bf16* p) const {
        st4(p, make_float4(v[0], v[1], v[2], v[3]));            // (the compiler merges the two 8-byte stores)
        st4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
    }
};
template <> struct V<1> {
    float v[1];
    template <typename T> __device__ __forceinline__ static V load(const T* p) { V r; r.v[0] = ld1(p); return r; }
    template <typename T> __device__ __forceinline__ void store(T* p) const { st1(p, v[0]); }
};

// The statistics / normalisation kernels are short launches on the step's critical path that often run beside the long blocks
// of the weight-gradient stream: STIN_CRIT_PRIO > 0 (compile time) raises their waves' issue priority.  Measured same box,
// interleaved (alternative builds through STIN_LIB_PATH): 0 -> 7.52 / 7.53 ms per step, 1 -> 7.49 / 7.48, 3 -> 7.50 / 7.49.
#ifndef STIN_CRIT_PRIO
#define STIN_CRIT_PRIO 1
#endif
__device__ __forceinline__ void crit_prio() {
#if STIN_CRIT_PRIO > 0
    __builtin_amdgcn_s_setprio(STIN_CRIT_PRIO);
#endif
}

__device__ __forceinline__ float elu_grad_from_pre(float n) { return n > 0.f ? 1.f : __expf(n); }

// grid = (chunks, B). partial layout: [b][chunk][o][C] doubles, o in {0,1}.
template <typename T, int MODE, int VW>
__global__ __launch_bounds__(BLOCK) void k_colreduce(const T* __restrict__ x, int64_t ldx,
                                                     const T* __restrict__ gout, int64_t ldg, int64_t N, int C,
                                                     const int32_t* __restrict__ ptr, const int32_t* __restrict__ gid,
                                                     const int32_t* __restrict__ sid, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ coef,
                                                     double* __restrict__ partial) {
    constexpr bool DOT_BN = (MODE == STIN_RED_DOT_BN || MODE == STIN_RED_DOT_BN_RELU);
    constexpr int NOUT = (MODE == STIN_RED_DOT_ELU || MODE == STIN_RED_MOMENTS || DOT_BN) ? 2 : 1;
    __shared__ double sm[NOUT][BLOCK][VW];
    const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
    const int64_t r0 = ptr != nullptr ? ptr[b] : 0;
    const int64_t r1 = ptr != nullptr ? ptr[b + 1] : N;
    const int CV = C / VW;
    const int CG = CV < BLOCK ? CV : BLOCK;
    const int RL = BLOCK / CG;
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const bool live = rl < RL;
    for (int cv = cg; cv < CV; cv += CG) {   // uniform trip count across the block (cg < CG <= CV)
        const int c = cv * VW;
        double acc0[VW], acc1[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) { acc0[i] = 0.0; acc1[i] = 0.0; }
        if (live) {
            // UR rows per trip with every load issued before the first use (round 3: the one-row-per-trip loop waited a
            // full memory round trip per row - 22.6 us for the 18 k-row DOT_ELU reduction whose bytes take 6 us);
            // the accumulation order per thread is unchanged (rows ascending): bit-identical sums
            constexpr int UR = 4;
            const int64_t step = (int64_t)nch * RL;
            for (int64_t rb = r0 + (int64_t)chunk * RL + rl; rb < r1; rb += UR * step) {
                V<VW> xv[UR], go[UR];
                int gq[UR], sq[UR];
                bool ok[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int64_t r = rb + u * step;
                    ok[u] = r < r1;
                    const int64_t rc = ok[u] ? r : rb;                      // clamped: a valid row, its contribution is skipped
                    xv[u] = V<VW>::load(x + rc * ldx + c);
                    if (MODE == STIN_RED_DOT_ELU || DOT_BN) go[u] = V<VW>::load(gout + rc * ldg + c);
                    gq[u] = (MODE != STIN_RED_SUM && MODE != STIN_RED_MOMENTS && gid != nullptr) ? gid[rc] : 0;
                    sq[u] = (MODE == STIN_RED_COEF_XC && sid != nullptr) ? sid[rc] : 0;
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    if (!ok[u]) continue;
                    if (MODE == STIN_RED_SUM) {
#pragma unroll
                        for (int i = 0; i < VW; ++i) acc0[i] += (double)xv[u].v[i];
                    } else if (MODE == STIN_RED_MOMENTS) {
#pragma unroll
                        for (int i = 0; i < VW; ++i) {
                            const double d = (double)xv[u].v[i];
                            acc0[i] += d;
                            acc1[i] += d * d;
                        }
                    } else {
                        const int g = gq[u];
                        const V<VW> mu = V<VW>::load(mean + (int64_t)g * C + c);
                        if (DOT_BN) {
                            const V<VW> rs = V<VW>::load(rstd + c);
                            const V<VW> ga = V<VW>::load(coef + c);
                            const V<VW> be = V<VW>::load(coef + C + c);
#pragma unroll
                            for (int i = 0; i < VW; ++i) {
                                const float n = (xv[u].v[i] - mu.v[i]) * rs.v[i];
                                const float d = (MODE == STIN_RED_DOT_BN_RELU && !(ga.v[i] * n + be.v[i] > 0.f)) ? 0.f : go[u].v[i];
                                acc0[i] += (double)(d * n);
                                acc1[i] += (double)d;
                            }
                        } else if (MODE == STIN_RED_CSQ) {
#pragma unroll
                            for (int i = 0; i < VW; ++i) { const float d = xv[u].v[i] - mu.v[i]; acc0[i] += (double)(d * d); }
                        } else if (MODE == STIN_RED_DOT_ELU) {
                            const V<VW> rs = V<VW>::load(rstd + (int64_t)g * C + c);
#pragma unroll
                            for (int i = 0; i < VW; ++i) {
                                const float xc = xv[u].v[i] - mu.v[i];
                                const float dy = go[u].v[i] * elu_grad_from_pre(xc * rs.v[i]);
                                acc0[i] += (double)(dy * xc);
                                acc1[i] += (double)dy;
                            }
                        } else {  // STIN_RED_COEF_XC
                            const V<VW> cf = V<VW>::load(coef + (int64_t)sq[u] * C + c);
#pragma unroll
                            for (int i = 0; i < VW; ++i) acc0[i] += (double)(cf.v[i] * (xv[u].v[i] - mu.v[i]));
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            sm[0][threadIdx.x][i] = acc0[i];
            if (NOUT == 2) sm[NOUT - 1][threadIdx.x][i] = acc1[i];
        }
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                double t[VW];
#pragma unroll
                for (int i = 0; i < VW; ++i) t[i] = 0.0;
                for (int k = 0; k < RL; ++k)
#pragma unroll
                    for (int i = 0; i < VW; ++i) t[i] += sm[o][k * CG + cg][i];
                double* dst = partial + (((int64_t)b * nch + chunk) * NOUT + o) * C + c;
#pragma unroll
                for (int i = 0; i < VW; ++i) dst[i] = t[i];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: ONE launch per column reduction (k_colreduce_t).  The two-stage form above writes [chunks][2][C] fp64 partials and
// needs a second launch (k_colreduce_final / k_moments_final: 6-8 us each inside a step) to fold them.  Here the grid is
// (row chunks R) x (GC-column groups) x (ranges B): a block reduces its rows for ITS GC columns (GC / 4 column lanes x float4,
// 1024 / GC row lanes; rows ascending per thread, fp64), publishes one [NOUT][GC] partial and takes a ticket of its (range,
// column group); the block whose ticket is the last one folds that group's R partials in a fixed order (r ascending within
// 256 / GC interleaved sub-sums, then those in order) and applies the post-op - the final arithmetic of k_colreduce_final /
// k_moments_final.  Deterministic: which block arrives last changes who folds, never the order of the additions.
// Cross-block visibility without a release fence (a `fence(release, agent)` writes back the XCD's whole dirty L2 - the first
// version of this kernel paid 6-10 us for it, profiles/r04_ticket_colreduce.md): the partials are WRITE-THROUGH (`sc1`) stores
// -> every storing wave drains `s_waitcnt vmcnt(0)` -> barrier -> one lane's relaxed agent-scope atomic add (the ticket) ->
// the last arriver reads the partials with `sc1` loads only (MI355X_MICROARCH.md "Valid forms", row: one lane of each storing
// workgroup adds to ONE counter / the workgroup whose add came last loads after its add returned, the other waves after a barrier).
// The ticket words live in a zero-initialised __device__ array of the code object; the folding block resets its word, and every
// launch takes the next slot row of its population (eager launches the lower RED_SLOTS / 2 rows, launches captured into a
// hipGraph the upper half: stin_ticket_slot), so launches in flight on different streams never share a word.
constexpr int RED_SLOTS = 256, RED_WORDS = 512;
__device__ unsigned int g_red_tickets[RED_SLOTS][RED_WORDS];

template <typename T, int MODE, int GC>
__global__ __launch_bounds__(BLOCK) void k_colreduce_t(const T* __restrict__ x, int64_t ldx, const T* __restrict__ gout, int64_t ldg,
                                                       int64_t N, int C, const int32_t* __restrict__ ptr,
                                                       const int32_t* __restrict__ gid, const int32_t* __restrict__ sid,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ coef, double* partial, int slot,
                                                       int post, const float* __restrict__ inv_cnt, float eps,
                                                       float* __restrict__ out0, float* __restrict__ out1) {
    constexpr bool DOT_BN = (MODE == STIN_RED_DOT_BN || MODE == STIN_RED_DOT_BN_RELU);
    constexpr int NOUT = (MODE == STIN_RED_DOT_ELU || MODE == STIN_RED_MOMENTS || DOT_BN) ? 2 : 1;
    constexpr int VW = 4, CL = GC / VW, RLN = BLOCK / CL;                          // column lanes, row lanes
    crit_prio();
    __shared__ double sm[NOUT][RLN][GC + 1];
    __shared__ int last_s;
    const int r = blockIdx.x, R = gridDim.x, cg = blockIdx.y, ncg = gridDim.y, b = blockIdx.z;
    const int64_t r0 = ptr != nullptr ? ptr[b] : 0;
    const int64_t r1 = ptr != nullptr ? ptr[b + 1] : N;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = cg * GC + cl * VW;
    const bool live = c < C;                                                      // (C % 4 == 0: a float4 is in or out)
    double acc0[VW], acc1[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) { acc0[i] = 0.0; acc1[i] = 0.0; }
    if (live) {
        constexpr int UR = 4;                                                      // rows in flight per thread
        const int64_t step = (int64_t)R * RLN;
        for (int64_t rb = r0 + (int64_t)r * RLN + rl; rb < r1; rb += UR * step) {
            V<VW> xv[UR], go[UR];
            int gq[UR], sq[UR];
            bool ok[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int64_t row = rb + u * step;
                ok[u] = row < r1;
                const int64_t rc = ok[u] ? row : rb;
                xv[u] = V<VW>::load(x + rc * ldx + c);
                if (MODE == STIN_RED_DOT_ELU || DOT_BN) go[u] = V<VW>::load(gout + rc * ldg + c);
                gq[u] = (MODE != STIN_RED_SUM && MODE != STIN_RED_MOMENTS && gid != nullptr) ? gid[rc] : 0;
                sq[u] = (MODE == STIN_RED_COEF_XC && sid != nullptr) ? sid[rc] : 0;
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                if (!ok[u]) continue;
                if (MODE == STIN_RED_SUM) {
#pragma unroll
                    for (int i = 0; i < VW; ++i) acc0[i] += (double)xv[u].v[i];
                } else if (MODE == STIN_RED_MOMENTS) {
#pragma unroll
                    for (int i = 0; i < VW; ++i) {
                        const double d = (double)xv[u].v[i];
                        acc0[i] += d;
                        acc1[i] += d * d;
                    }
                } else {
                    const int g = gq[u];
                    const V<VW> mu = V<VW>::load(mean + (int64_t)g * C + c);
                    if (DOT_BN) {
                        const V<VW> rs = V<VW>::load(rstd + c);
                        const V<VW> ga = V<VW>::load(coef + c);
                        const V<VW> be = V<VW>::load(coef + C + c);
#pragma unroll
                        for (int i = 0; i < VW; ++i) {
                            const float n = (xv[u].v[i] - mu.v[i]) * rs.v[i];
                            const float d = (MODE == STIN_RED_DOT_BN_RELU && !(ga.v[i] * n + be.v[i] > 0.f)) ? 0.f : go[u].v[i];
                            acc0[i] += (double)(d * n);
                            acc1[i] += (double)d;
                        }
                    } else if (MODE == STIN_RED_CSQ) {
#pragma unroll
                        for (int i = 0; i < VW; ++i) { const float d = xv[u].v[i] - mu.v[i]; acc0[i] += (double)(d * d); }
                    } else if (MODE == STIN_RED_DOT_ELU) {
                        const V<VW> rs = V<VW>::load(rstd + (int64_t)g * C + c);
#pragma unroll
                        for (int i = 0; i < VW; ++i) {
                            const float xc = xv[u].v[i] - mu.v[i];
                            const float dy = go[u].v[i] * elu_grad_from_pre(xc * rs.v[i]);
                            acc0[i] += (double)(dy * xc);
                            acc1[i] += (double)dy;
                        }
                    } else {  // STIN_RED_COEF_XC
                        const V<VW> cf = V<VW>::load(coef + (int64_t)sq[u] * C + c);
#pragma unroll
                        for (int i = 0; i < VW; ++i) acc0[i] += (double)(cf.v[i] * (xv[u].v[i] - mu.v[i]));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        sm[0][rl][cl * VW + i] = acc0[i];
        if (NOUT == 2) sm[NOUT - 1][rl][cl * VW + i] = acc1[i];
    }
    __syncthreads();
    // partial of this block: [b][cg][r][o][GC], published write-through
    double* pg = partial + (((int64_t)b * ncg + cg) * R) * (NOUT * GC);
    if (threadIdx.x < NOUT * GC) {
        const int o = threadIdx.x / GC, cc = threadIdx.x % GC;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < RLN; ++k) t += sm[o][k][cc];
        __hip_atomic_store(pg + (int64_t)r * (NOUT * GC) + o * GC + cc, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              // every storing wave drains its sc1 stores
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int* word = &g_red_tickets[slot][b * ncg + cg];
        const unsigned int t = __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned int)(R - 1)) ? 1 : 0;
        if (last) __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the slot's next launch
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    // ---- the fold (sc1 loads only): thread (part, column) sums r = part, part + PARTS, ... ; then the parts in order
    constexpr int PARTS = BLOCK / GC;
    const int cc = threadIdx.x % GC, part = threadIdx.x / GC;
    double f[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) f[o] = 0.0;
    constexpr int FU = 16;                                                        // partials in flight per thread and output
    for (int k0 = part; k0 < R; k0 += FU * PARTS) {
        double v[NOUT][FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int k = k0 + u * PARTS;
            const int kc = k < R ? k : part;
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
                v[o][u] = __hip_atomic_load(pg + (int64_t)kc * (NOUT * GC) + o * GC + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < FU; ++u)
            if (k0 + u * PARTS < R) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) f[o] += v[o][u];
            }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) sm[o][part][cc] = f[o];
    __syncthreads();
    const int col = cg * GC + cc;
    if (part == 0 && col < C) {
        double t[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            t[o] = 0.0;
#pragma unroll
            for (int k = 0; k < PARTS; ++k) t[o] += sm[o][k][cc];
        }
        if (MODE == STIN_RED_MOMENTS) {                                           // as k_moments_final
            const double ic = (double)inv_cnt[b];
            const double mu = t[0] * ic;
            double var = t[NOUT - 1] * ic - mu * mu;
            if (var < 0.0) var = 0.0;
            out0[(int64_t)b * C + col] = (float)mu;
            out1[(int64_t)b * C + col] = (float)(1.0 / sqrt(var + (double)eps));
        } else {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {                                      // as k_colreduce_final
                float rr = (float)t[o];
                if (post == STIN_POST_SCALE) rr = rr * inv_cnt[b];
                else if (post == STIN_POST_RSTD) rr = 1.0f / sqrtf(rr * inv_cnt[b] + eps);
                else if (post == STIN_POST_NORM_COEF) {
                    const float rs = rstd[(int64_t)b * C + col], ic = inv_cnt[b];
                    rr = (o == 0) ? -(rs * rs * rs) * rr * ic : -(rs * rr) * ic;
                }
                (o == 0 ? out0 : out1)[(int64_t)b * C + col] = rr;
            }
        }
    }
}

// Second stage: out[b, o, c] = post(sum_k partial[b, k, o, c]).  One 256-thread block per 16 columns:
// 16 k-lanes x 16 columns, each k-lane walks the chunk list with stride 16 (coalesced 128-byte reads),
// then a fixed-order LDS reduction across the k-lanes -> deterministic.
constexpr int FIN_COLS = 16, FIN_KL = 16;
__global__ __launch_bounds__(BLOCK) void k_colreduce_final(const double* __restrict__ partial, int nch, int nout, int C,
                                                           int B, int post, const float* __restrict__ inv_cnt, float eps,
                                                           const float* __restrict__ aux, float* __restrict__ out0,
                                                           float* __restrict__ out1) {
    __shared__ double sm[FIN_KL][FIN_COLS + 1];
    const int tx = threadIdx.x % FIN_COLS, ty = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + tx, o = blockIdx.y, b = blockIdx.z;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < C) {
        const double* p = partial + ((int64_t)b * nch * nout + o) * C + c;
        const int64_t stride = (int64_t)nout * C;
        // Every load of a lane's chunk list in flight before the first add, 32 at a time (round 3).  The partials were written by
        // the previous kernel on other XCDs, so each dependent batch of loads is a trip to the fabric (~2 us under load): with
        // 4 in flight the 564-chunk list of the bottleneck level took nine trips (9.4 us for 2 MB), now two.  The adds keep the
        // order of the original loop - groups of four strides round-robin into s0..s3 while a whole group is in range, the
        // ragged tail into s0 - so the sums are bit-identical; out-of-range slots load a clamped address and add nothing.
        for (int k = ty; k < nch; k += 32 * FIN_KL) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int ku = k + u * FIN_KL;
                v[u] = p[(int64_t)(ku < nch ? ku : ty) * stride];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k0 = k + 4 * q * FIN_KL;
                if (k0 + 3 * FIN_KL < nch) {
                    s0 += v[4 * q];
                    s1 += v[4 * q + 1];
                    s2 += v[4 * q + 2];
                    s3 += v[4 * q + 3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k0 + j * FIN_KL < nch) s0 += v[4 * q + j];
                }
            }
        }
    }
    sm[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && c < C) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < FIN_KL; ++k) s += sm[k][tx];
        float r = (float)s;
        if (post == STIN_POST_SCALE) r = r * inv_cnt[b];
        else if (post == STIN_POST_RSTD) r = 1.0f / sqrtf(r * inv_cnt[b] + eps);
        else if (post == STIN_POST_NORM_COEF) {          // same float operations as k_norm_coef (stin_pack.hip)
            const float rs = aux[(int64_t)b * C + c], ic = inv_cnt[b];
            r = (o == 0) ? -(rs * rs * rs) * r * ic : -(rs * r) * ic;
        }
        (o == 0 ? out0 : out1)[(int64_t)b * C + c] = r;
    }
}

// (sum x, sum x^2) in fp64 -> mean and 1/sqrt(biased var + eps).  Exact to fp32 rounding: the fp64 sums carry
// ~1e-16 relative error, so E[x^2] - mean^2 loses nothing visible unless mean^2/var exceeds ~1e8.
__global__ void k_moments_final(const double* __restrict__ partial, int nch, int C, int B,
                                const float* __restrict__ inv_cnt, float eps, float* __restrict__ mean,
                                float* __restrict__ rstd) {
    __shared__ double sm[2][FIN_KL][FIN_COLS + 1];
    crit_prio();
    const int tx = threadIdx.x % FIN_COLS, ty = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + tx, b = blockIdx.z;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const double* p = partial + (int64_t)b * nch * 2 * C + c;
        const int64_t st = (int64_t)2 * C;
        double t1 = 0.0, t2 = 0.0;
        // (all loads of the lane's list in flight, 16 groups at a time; adds in the order of the original two-accumulator loop:
        // pairs of strides alternate (s1, s2) / (t1, t2) while a whole pair is in range, a ragged last stride goes to (s1, s2))
        for (int k = ty; k < nch; k += 16 * FIN_KL) {
            double a[16], q[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int ku = k + u * FIN_KL;
                const int kc = ku < nch ? ku : ty;
                a[u] = p[kc * st];
                q[u] = p[kc * st + C];
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int k0 = k + 2 * h * FIN_KL;
                if (k0 + FIN_KL < nch) {
                    s1 += a[2 * h];
                    s2 += q[2 * h];
                    t1 += a[2 * h + 1];
                    t2 += q[2 * h + 1];
                } else if (k0 < nch) {
                    s1 += a[2 * h];
                    s2 += q[2 * h];
                }
            }
        }
        s1 += t1;
        s2 += t2;
    }
    sm[0][ty][tx] = s1;
    sm[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        double a = 0.0, q = 0.0;
#pragma unroll
        for (int k = 0; k < FIN_KL; ++k) { a += sm[0][k][tx]; q += sm[1][k][tx]; }
        const double ic = (double)inv_cnt[b];
        const double mu = a * ic;
        double var = q * ic - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[(int64_t)b * C + c] = (float)mu;
        rstd[(int64_t)b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <typename T, int VW>
__global__ __launch_bounds__(BLOCK) void k_norm_fwd(const T* __restrict__ x, int64_t ldx,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const int32_t* __restrict__ gid, const T* __restrict__ res,
                                                    int64_t ldres, int64_t N, int C, int act, T* __restrict__ y,
                                                    int64_t ldy) {
    crit_prio();
    const int CV = C / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N * CV) return;
    const int64_t r = t / CV;
    const int c = (int)(t % CV) * VW;
    const int g = gid != nullptr ? gid[r] : 0;
    const V<VW> xv = V<VW>::load(x + r * ldx + c);
    const V<VW> mu = V<VW>::load(mean + (int64_t)g * C + c);
    const V<VW> rs = V<VW>::load(rstd + (int64_t)g * C + c);
    V<VW> o;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        float n = (xv.v[i] - mu.v[i]) * rs.v[i];
        if (act) n = n > 0.f ? n : expm1f(n);
        o.v[i] = n;
    }
    if (res != nullptr) {
        const V<VW> rv = V<VW>::load(res + r * ldres + c);
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] += rv.v[i];
    }
    o.store(y + r * ldy + c);
}

template <typename T, int VW>
__global__ __launch_bounds__(BLOCK) void k_norm_bwd(const T* __restrict__ x, int64_t ldx,
                                                    const T* __restrict__ gout, int64_t ldg,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ a, const float* __restrict__ kk,
                                                    const float* __restrict__ m, const int32_t* __restrict__ gid,
                                                    const int32_t* __restrict__ sid, int64_t N, int C, int act,
                                                    T* __restrict__ dx, int64_t lddx) {
    crit_prio();
    const int CV = C / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N * CV) return;
    const int64_t r = t / CV;
    const int c = (int)(t % CV) * VW;
    const int g = gid != nullptr ? gid[r] : 0;
    const int s = sid != nullptr ? sid[r] : 0;
    const V<VW> xv = V<VW>::load(x + r * ldx + c);
    const V<VW> go = V<VW>::load(gout + r * ldg + c);
    const V<VW> mu = V<VW>::load(mean + (int64_t)g * C + c);
    const V<VW> rs = V<VW>::load(rstd + (int64_t)g * C + c);
    const V<VW> av = V<VW>::load(a + (int64_t)g * C + c);
    const V<VW> kv = V<VW>::load(kk + (int64_t)s * C + c);
    const V<VW> mv = V<VW>::load(m + (int64_t)s * C + c);
    V<VW> o;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        const float xc = xv.v[i] - mu.v[i];
        const float dy = act ? go.v[i] * elu_grad_from_pre(xc * rs.v[i]) : go.v[i];
        o.v[i] = av.v[i] * dy + kv.v[i] * xc + mv.v[i];
    }
    o.store(dx + r * lddx + c);
}

// Round 5: fold + elementwise pass in ONE launch (k_norm_fold<BWD>), for the blocks whose column statistics arrive as per-row-group
// fp64 partials out of a GEMM epilogue (forward: stin_gemm_nt_colstats_f32 -> k_moments_final -> k_norm_fwd; backward: the hand-off
// statistics of stin_gemm_nt_dotelu_f32 -> k_colreduce_final -> k_norm_bwd).  The fold is a 5 us launch alone and 18-20 us beside
// the weight-gradient stream, on the critical path of every bottleneck block in both directions.  Round 4 tried ONE folding
// workgroup with the others waiting on a flag (three versions, all slower: every poll / atomic on a shared line serialises).  Here
// nobody waits for anybody: a workgroup owns NF_GC = 32 columns x a chunk of rows and folds ITS columns itself - the [groups][2][C]
// partials are L2 / Infinity-Cache resident (226 groups at 18 063 rows: 115 KB per workgroup, 29 MB over the grid beside 55 MB of
// elementwise traffic) - with exactly the per-lane sums, LDS reduction order and final float operations of k_moments_final /
// k_colreduce_final (16 k-lanes per column, the loops below are theirs), so mean / rstd / k / m and the outputs are BIT-IDENTICAL
// to the two-launch route.  The row-chunk-0 workgroups write mean / rstd (forward: saved for backward).  Host side: used when
// the fold traffic stays below half the elementwise traffic (norm_fold_rows()).
constexpr int NF_GC = 32, NF_BLOCK = 512, NF_RL = NF_BLOCK / (NF_GC / 4);      // 8 lanes per row, 64 rows per trip

// one k-lane's share of k_colreduce_final's sum over the chunk list (same loads, same add order)
__device__ __forceinline__ double fold_lane_sum(const double* __restrict__ p, int64_t stride, int nch, int ty) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int k = ty; k < nch; k += 32 * FIN_KL) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int ku = k + u * FIN_KL;
            v[u] = p[(int64_t)(ku < nch ? ku : ty) * stride];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k0 = k + 4 * q * FIN_KL;
            if (k0 + 3 * FIN_KL < nch) {
                s0 += v[4 * q];
                s1 += v[4 * q + 1];
                s2 += v[4 * q + 2];
                s3 += v[4 * q + 3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j * FIN_KL < nch) s0 += v[4 * q + j];
            }
        }
    }
    return (s0 + s1) + (s2 + s3);
}
// one k-lane's share of k_moments_final's two sums
__device__ __forceinline__ void fold_lane_moments(const double* __restrict__ p, int64_t st, int C, int nch, int ty, double& o1, double& o2) {
    double s1 = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int k = ty; k < nch; k += 16 * FIN_KL) {
        double a[16], q[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int ku = k + u * FIN_KL;
            const int kc = ku < nch ? ku : ty;
            a[u] = p[kc * st];
            q[u] = p[kc * st + C];
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int k0 = k + 2 * h * FIN_KL;
            if (k0 + FIN_KL < nch) {
                s1 += a[2 * h];
                s2 += q[2 * h];
                t1 += a[2 * h + 1];
                t2 += q[2 * h + 1];
            } else if (k0 < nch) {
                s1 += a[2 * h];
                s2 += q[2 * h];
            }
        }
    }
    o1 = s1 + t1;
    o2 = s2 + t2;
}

// BWD = false: y = ELU((x - mean) rstd) + res with (mean, rstd) folded from moment partials; mean_io / rstd_io are OUTPUTS.
// BWD = true:  dx = rstd dy + k xc + m, dy = g ELU'((x - mean) rstd), (k, m) folded from the (dy xc, dy) partials; mean_io /
//              rstd_io are INPUTS.  One graph (B = 1), fp32 rows, C % 32 == 0, 16-byte rows.
template <bool BWD>
__global__ __launch_bounds__(NF_BLOCK) void k_norm_fold(const double* __restrict__ partial, int nch, const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ other, int64_t ldo, float* __restrict__ mean_io,
                                                        float* __restrict__ rstd_io, const float* __restrict__ inv_cnt, float eps,
                                                        int64_t N, int C, int rows_per_block, float* __restrict__ y, int64_t ldy) {
    __shared__ double sm[2][FIN_KL][NF_GC + 1];
    __shared__ __attribute__((aligned(16))) float coef[4][NF_GC];                 // fwd: mean, rstd; bwd: mean, rstd, k, m
    crit_prio();
    const int c0 = blockIdx.x * NF_GC;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
    {
        const int tx = threadIdx.x % NF_GC, ty = threadIdx.x / NF_GC;             // 32 columns x 16 k-lanes
        const int c = c0 + tx;
        if (!BWD) {
            double a, q;
            fold_lane_moments(partial + c, (int64_t)2 * C, C, nch, ty, a, q);
            sm[0][ty][tx] = a;
            sm[1][ty][tx] = q;
        } else {
            sm[0][ty][tx] = fold_lane_sum(partial + c, (int64_t)2 * C, nch, ty);
            sm[1][ty][tx] = fold_lane_sum(partial + C + c, (int64_t)2 * C, nch, ty);
        }
        __syncthreads();
        if (ty == 0) {
            double a = 0.0, q = 0.0;
#pragma unroll
            for (int k = 0; k < FIN_KL; ++k) { a += sm[0][k][tx]; q += sm[1][k][tx]; }
            if (!BWD) {                                                           // (= k_moments_final)
                const double ic = (double)inv_cnt[0];
                const double mu = a * ic;
                double var = q * ic - mu * mu;
                if (var < 0.0) var = 0.0;
                const float mf = (float)mu, rf = (float)(1.0 / sqrt(var + (double)eps));
                coef[0][tx] = mf;
                coef[1][tx] = rf;
                if (blockIdx.y == 0) {
                    mean_io[c] = mf;
                    rstd_io[c] = rf;
                }
            } else {                                                              // (= k_colreduce_final, STIN_POST_NORM_COEF)
                const float rs = rstd_io[c], ic = inv_cnt[0];
                const float t1 = (float)a, s0 = (float)q;
                coef[0][tx] = mean_io[c];
                coef[1][tx] = rs;
                coef[2][tx] = -(rs * rs * rs) * t1 * ic;
                coef[3][tx] = -(rs * s0) * ic;
            }
        }
        __syncthreads();
    }
    const int cl = threadIdx.x % (NF_GC / 4), rl = threadIdx.x / (NF_GC / 4);
    const int c = c0 + cl * 4;
    const float4 mu4 = *reinterpret_cast<const float4*>(&coef[0][cl * 4]), rs4 = *reinterpret_cast<const float4*>(&coef[1][cl * 4]);
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
    float kv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f};
    if (BWD) {
        const float4 k4 = *reinterpret_cast<const float4*>(&coef[2][cl * 4]), m4 = *reinterpret_cast<const float4*>(&coef[3][cl * 4]);
        kv[0] = k4.x; kv[1] = k4.y; kv[2] = k4.z; kv[3] = k4.w;
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
    }
    constexpr int UR = 4;                                                         // rows in flight per thread
    for (int64_t rb = r0 + rl; rb < r1; rb += UR * NF_RL) {
        float4 xv[UR], ov[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t r = rb + u * NF_RL < r1 ? rb + u * NF_RL : rb;
            xv[u] = ld4(x + r * ldx + c);
            ov[u] = other != nullptr ? ld4(other + r * ldo + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t r = rb + u * NF_RL;
            if (r >= r1) break;
            const float xa[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, oa[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!BWD) {                                                       // (= k_norm_fwd, act = 1)
                    float n = (xa[i] - mu[i]) * rs[i];
                    n = n > 0.f ? n : expm1f(n);
                    o[i] = other != nullptr ? n + oa[i] : n;
                } else {                                                          // (= k_norm_bwd, act = 1, a = rstd)
                    const float xc = xa[i] - mu[i];
                    const float dy = oa[i] * elu_grad_from_pre(xc * rs[i]);
                    o[i] = rs[i] * dy + kv[i] * xc + mv[i];
                }
            }
            st4(y + r * ldy + c, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// BatchNorm1d-with-affine over all rows (+ ReLU): forward and backward elementwise passes (fp32, SingleConvMeshNet)
template <int VW>
__global__ __launch_bounds__(BLOCK) void k_bn_fwd(const float* __restrict__ x, int64_t ldx, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int64_t N, int C, int act,
                                                  float* __restrict__ y, int64_t ldy) {
    const int CV = C / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N * CV) return;
    const int64_t r = t / CV;
    const int c = (int)(t % CV) * VW;
    const V<VW> xv = V<VW>::load(x + r * ldx + c), mu = V<VW>::load(mean + c), rs = V<VW>::load(rstd + c);
    const V<VW> ga = V<VW>::load(gamma + c), be = V<VW>::load(beta + c);
    V<VW> o;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        const float z = ga.v[i] * ((xv.v[i] - mu.v[i]) * rs.v[i]) + be.v[i];
        o.v[i] = (act && !(z > 0.f)) ? 0.f : z;
    }
    o.store(y + r * ldy + c);
}

template <int VW>
__global__ __launch_bounds__(BLOCK) void k_bn_bwd(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gout,
                                                  int64_t ldg, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ P, const float* __restrict__ Q, float inv_n,
                                                  int64_t N, int C, int act, float* __restrict__ dx, int64_t lddx) {
    const int CV = C / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N * CV) return;
    const int64_t r = t / CV;
    const int c = (int)(t % CV) * VW;
    const V<VW> xv = V<VW>::load(x + r * ldx + c), go = V<VW>::load(gout + r * ldg + c);
    const V<VW> mu = V<VW>::load(mean + c), rs = V<VW>::load(rstd + c), ga = V<VW>::load(gamma + c), be = V<VW>::load(beta + c);
    const V<VW> pv = V<VW>::load(P + c), qv = V<VW>::load(Q + c);
    V<VW> o;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        const float n = (xv.v[i] - mu.v[i]) * rs.v[i];
        const float d = (act && !(ga.v[i] * n + be.v[i] > 0.f)) ? 0.f : go.v[i];
        o.v[i] = rs.v[i] * ga.v[i] * (d - qv.v[i] * inv_n - n * (pv.v[i] * inv_n));
    }
    o.store(dx + r * lddx + c);
}

// BatchNorm1d over the E edge rows FOLLOWED by the mean over each target's in-edges (SingleConvMeshNet's second edge norm,
// edge_conv_filter.py:34-44 + aggr='mean'): the affine map commutes with the mean, so forward normalises the N aggregated
// rows; this is the backward of both at once - the gradient of the raw edge rows m from the VERTEX gradient g:
//   dm[e] = gamma rstd ( g[dst e] / deg(dst e) - Q / E - nhat_e P / E ),   nhat_e = (m[e] - mean) rstd,
// P = sum_i g_i nhat(agg_i), Q = sum_i g_i over the vertices with in-edges (column sums over N rows, not E).
template <int VW>
__global__ __launch_bounds__(BLOCK) void k_bn_mean_bwd(const float* __restrict__ m, int64_t ldm, const float* __restrict__ g,
                                                       int64_t ldg, const int32_t* __restrict__ dst,
                                                       const float* __restrict__ inv_deg, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ P, const float* __restrict__ Q, float inv_e,
                                                       int64_t E, int C, float* __restrict__ dm, int64_t lddm) {
    const int CV = C / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= E * CV) return;
    const int64_t e = t / CV;
    const int c = (int)(t % CV) * VW;
    const int64_t i = dst[e];
    const float w = inv_deg[i];
    const V<VW> mv = V<VW>::load(m + e * ldm + c), gv = V<VW>::load(g + i * ldg + c);
    const V<VW> mu = V<VW>::load(mean + c), rs = V<VW>::load(rstd + c), ga = V<VW>::load(gamma + c);
    const V<VW> pv = V<VW>::load(P + c), qv = V<VW>::load(Q + c);
    V<VW> o;
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        const float n = (mv.v[k] - mu.v[k]) * rs.v[k];
        o.v[k] = rs.v[k] * ga.v[k] * (gv.v[k] * w - qv.v[k] * inv_e - n * (pv.v[k] * inv_e));
    }
    o.store(dm + e * lddm + c);
}

// nn.BatchNorm1d's running-statistics update from the batch (mean, rstd) of the kernels above, one launch:
//   running_mean <- (1 - m) running_mean + m mean ;  running_var <- (1 - m) running_var + m unbias max(1/rstd^2 - eps, 0)
__global__ void k_bn_running(const float* __restrict__ mean, const float* __restrict__ rstd, int C, float eps, float unbias,
                             float momentum, float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = rstd[c];
    const float var = fmaxf(1.0f / (r * r) - eps, 0.f);
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (var * unbias);
}

template <typename T>
inline bool vec4_ok(int C, std::initializer_list<const void*> data, std::initializer_list<const void*> stats,
                    std::initializer_list<int64_t> lds) {
    if (C % 4 != 0) return false;
    for (const void* p : data)
        if (p != nullptr && !stin_aligned_vec4<T>(p)) return false;
    for (const void* p : stats)
        if (p != nullptr && !stin_aligned16(p)) return false;
    for (int64_t ld : lds)
        if (ld % 4 != 0) return false;
    return true;
}

template <typename T>
inline bool vec8_ok(int C, std::initializer_list<const void*> data, std::initializer_list<const void*> stats,
                    std::initializer_list<int64_t> lds) {
    if (sizeof(T) != 2 || C % 8 != 0) return false;
    for (const void* p : data)
        if (p != nullptr && !stin_aligned16(p)) return false;
    for (const void* p : stats)
        if (p != nullptr && !stin_aligned16(p)) return false;
    for (int64_t ld : lds)
        if (ld % 8 != 0) return false;
    return true;
}

constexpr bool is_f32(const float*) { return true; }
constexpr bool is_f32(const stin_bf16*) { return false; }
inline const stin_bf16* b16(const stin_bf16_t* p) { return reinterpret_cast<const stin_bf16*>(p); }
inline stin_bf16* b16(stin_bf16_t* p) { return reinterpret_cast<stin_bf16*>(p); }

inline int norm_cu_count() { return stin_cu_count_dev(); }          // (per device: stin_common.h)

template <typename T>
int colreduce_impl(int mode, const T* x, int64_t ldx, const T* gout, int64_t ldg, int64_t N, int C, const int32_t* ptr,
                   int B, const int32_t* gid, const int32_t* sid, const float* mean, const float* rstd,
                   const float* coef, int post, const float* inv_cnt, float eps, float* out0, float* out1,
                   void* workspace, size_t workspace_bytes, hipStream_t stream) {
    STIN_REQUIRE(mode >= STIN_RED_SUM && mode <= STIN_RED_DOT_BN_RELU, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(N >= 0 && C > 0 && B > 0 && ldx >= C, STIN_E_SIZE);
    STIN_REQUIRE((ptr != nullptr) || B == 1, STIN_E_SIZE);
    STIN_REQUIRE(x && out0 && workspace, STIN_E_NULL);
    STIN_REQUIRE(post >= STIN_POST_NONE && post <= STIN_POST_NORM_COEF, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(post != STIN_POST_NORM_COEF || mode == STIN_RED_DOT_ELU, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(post == STIN_POST_NONE || inv_cnt != nullptr, STIN_E_NULL);
    if (mode != STIN_RED_SUM && mode != STIN_RED_MOMENTS) STIN_REQUIRE(mean != nullptr, STIN_E_NULL);
    if (mode == STIN_RED_MOMENTS) STIN_REQUIRE(out1 != nullptr && inv_cnt != nullptr, STIN_E_NULL);
    if (mode == STIN_RED_DOT_ELU) STIN_REQUIRE(gout && rstd && out1 && ldg >= C, STIN_E_NULL);
    if (mode == STIN_RED_COEF_XC) STIN_REQUIRE(coef != nullptr, STIN_E_NULL);
    if (mode == STIN_RED_DOT_BN || mode == STIN_RED_DOT_BN_RELU)
        STIN_REQUIRE(gout && rstd && coef && out1 && ldg >= C && B == 1 && gid == nullptr, STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_colreduce_workspace_bytes(C, B), STIN_E_WORKSPACE);
    double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);

    const bool vec = vec4_ok<T>(C, {x, gout}, {mean, rstd, coef}, {ldx, gout ? ldg : 0});
    if (!vec && !is_f32((const T*)nullptr)) return STIN_E_UNSUPPORTED;
    // one-launch form (k_colreduce_t): 16-byte fp32 rows, few enough (range, column group) pairs for one row of ticket words;
    // Column groups of 64 / 32 / 16 so that narrow
    // matrices still fold on four blocks side by side.
    if (vec && N > 0) {                                // (fp32 and bf16 rows: 4 channels per lane either way)
        const int GC = C >= 256 ? 64 : (C >= 128 ? 32 : 16);
        const int ncg = (C + GC - 1) / GC;
        if ((int64_t)B * ncg <= RED_WORDS) {
            static std::atomic<unsigned> seq{0}, seq_cap{0};
            const int slot = stin_ticket_slot(seq, seq_cap, RED_SLOTS, stream);
            const int RLN = BLOCK / (GC / 4);
            // row chunks: ~8 row trips per block, at most ~2 blocks per CU over all column groups and ranges, and within the workspace
            int64_t R = (N / B + (int64_t)RLN * 8 - 1) / ((int64_t)RLN * 8);
            constexpr int per_cu = 2;
            const int64_t cap = (per_cu * (int64_t)norm_cu_count() + (int64_t)B * ncg - 1) / ((int64_t)B * ncg);
            if (R > cap) R = cap;
            // partial [B][ncg][R][2][GC] doubles must fit the caller's workspace of max(B, MAX_SLABS) x 2 x C doubles
            const int64_t ws_cap = (int64_t)(B > MAX_SLABS ? B : MAX_SLABS) * C / ((int64_t)ncg * GC);
            if (R > ws_cap / B) R = ws_cap / B;
            if (R < 1) R = 1;
            dim3 grid((unsigned)R, (unsigned)ncg, (unsigned)B);
#define STIN_REDT_L(M, G)                                                                                                \
            hipLaunchKernelGGL((k_colreduce_t<T, M, G>), grid, dim3(BLOCK), 0, stream, x, ldx, gout, ldg, N, C, ptr, gid, sid, mean, rstd, coef, \
                               partial, slot, post, inv_cnt, eps, out0, out1)
#define STIN_REDT_LAUNCH(M)                                                                                              \
            do {                                                                                                         \
                if (GC == 64) STIN_REDT_L(M, 64);                                                                        \
                else if (GC == 32) STIN_REDT_L(M, 32);                                                                   \
                else STIN_REDT_L(M, 16);                                                                                 \
            } while (0)
            {
                switch (mode) {
                    case STIN_RED_SUM: STIN_REDT_LAUNCH(STIN_RED_SUM); break;
                    case STIN_RED_CSQ: STIN_REDT_LAUNCH(STIN_RED_CSQ); break;
                    case STIN_RED_DOT_ELU: STIN_REDT_LAUNCH(STIN_RED_DOT_ELU); break;
                    case STIN_RED_MOMENTS: STIN_REDT_LAUNCH(STIN_RED_MOMENTS); break;
                    case STIN_RED_DOT_BN: STIN_REDT_LAUNCH(STIN_RED_DOT_BN); break;
                    case STIN_RED_DOT_BN_RELU: STIN_REDT_LAUNCH(STIN_RED_DOT_BN_RELU); break;
                    default: STIN_REDT_LAUNCH(STIN_RED_COEF_XC); break;
                }
            }
#undef STIN_REDT_LAUNCH
#undef STIN_REDT_L
            return stin_launch_status();
        }
    }
    const int VW = vec ? 4 : 1;
    const int CV = C / VW;
    const int CG = CV < BLOCK ? CV : BLOCK;
    const int RL = BLOCK / CG;
    // row iterations per block: 8 (2 blocks per CU at the 18 k-row level) measured 0.3 % faster on the step than 16; 32 is 2 % slower
    constexpr int rows_mul = 8;
    int64_t want = (N / B + (int64_t)RL * rows_mul - 1) / ((int64_t)RL * rows_mul);
    int cap = MAX_SLABS / B;
    if (cap < 1) cap = 1;
    int nch = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    const int nout = (mode == STIN_RED_DOT_ELU || mode == STIN_RED_MOMENTS || mode >= STIN_RED_DOT_BN) ? 2 : 1;
    dim3 grid((unsigned)nch, (unsigned)B);
#define STIN_RED_LAUNCH(M)                                                                                          \
    do {                                                                                                            \
        if (vec) hipLaunchKernelGGL((k_colreduce<T, M, 4>), grid, dim3(BLOCK), 0, stream, x, ldx, gout, ldg, N, C, ptr, gid, sid, mean, rstd, coef, partial); \
        else if constexpr (is_f32((const T*)nullptr)) hipLaunchKernelGGL((k_colreduce<T, M, 1>), grid, dim3(BLOCK), 0, stream, x, ldx, gout, ldg, N, C, ptr, gid, sid, mean, rstd, coef, partial);     \
    } while (0)
    switch (mode) {
        case STIN_RED_SUM: STIN_RED_LAUNCH(STIN_RED_SUM); break;
        case STIN_RED_CSQ: STIN_RED_LAUNCH(STIN_RED_CSQ); break;
        case STIN_RED_DOT_ELU: STIN_RED_LAUNCH(STIN_RED_DOT_ELU); break;
        case STIN_RED_MOMENTS: STIN_RED_LAUNCH(STIN_RED_MOMENTS); break;
        case STIN_RED_DOT_BN: STIN_RED_LAUNCH(STIN_RED_DOT_BN); break;
        case STIN_RED_DOT_BN_RELU: STIN_RED_LAUNCH(STIN_RED_DOT_BN_RELU); break;
        default: STIN_RED_LAUNCH(STIN_RED_COEF_XC); break;
    }
#undef STIN_RED_LAUNCH
    if (mode == STIN_RED_MOMENTS) {
        hipLaunchKernelGGL(k_moments_final, dim3((unsigned)((C + FIN_COLS - 1) / FIN_COLS), 1u, (unsigned)B), dim3(BLOCK), 0,
                           stream, partial, nch, C, B, inv_cnt, eps, out0, out1);
        return stin_launch_status();
    }
    hipLaunchKernelGGL(k_colreduce_final, dim3((unsigned)((C + FIN_COLS - 1) / FIN_COLS), (unsigned)nout, (unsigned)B),
                       dim3(BLOCK), 0, stream, partial, nch, nout, C, B, post, inv_cnt, eps, rstd, out0, out1);
    return stin_launch_status();
}

template <typename T>
int norm_fwd_impl(const T* x, int64_t ldx, const float* mean, const float* rstd, const int32_t* gid, const T* res,
                  int64_t ldres, int64_t N, int C, int act, T* y, int64_t ldy, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && C > 0 && ldx >= C && ldy >= C && (res == nullptr || ldres >= C), STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && mean && rstd && y, STIN_E_NULL);
    if (vec8_ok<T>(C, {x, res, y}, {mean, rstd}, {ldx, ldy, res ? ldres : 0})) {
        const int64_t n = N * (C / 8);
        hipLaunchKernelGGL((k_norm_fwd<T, 8>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx,
                           mean, rstd, gid, res, ldres, N, C, act, y, ldy);
    } else if (vec4_ok<T>(C, {x, res, y}, {mean, rstd}, {ldx, ldy, res ? ldres : 0})) {
        const int64_t n = N * (C / 4);
        hipLaunchKernelGGL((k_norm_fwd<T, 4>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx,
                           mean, rstd, gid, res, ldres, N, C, act, y, ldy);
    } else if constexpr (is_f32((const T*)nullptr)) {
        const int64_t n = N * C;
        hipLaunchKernelGGL((k_norm_fwd<T, 1>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx,
                           mean, rstd, gid, res, ldres, N, C, act, y, ldy);
    } else {
        return STIN_E_UNSUPPORTED;
    }
    return stin_launch_status();
}

template <typename T>
int norm_bwd_impl(const T* x, int64_t ldx, const T* gout, int64_t ldg, const float* mean, const float* rstd,
                  const float* a, const float* k, const float* m, const int32_t* gid, const int32_t* sid, int64_t N, int C,
                  int act, T* dx, int64_t lddx, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && C > 0 && ldx >= C && ldg >= C && lddx >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && gout && mean && rstd && a && k && m && dx, STIN_E_NULL);
    if (vec8_ok<T>(C, {x, gout, dx}, {mean, rstd, a, k, m}, {ldx, ldg, lddx})) {
        const int64_t n = N * (C / 8);
        hipLaunchKernelGGL((k_norm_bwd<T, 8>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx,
                           gout, ldg, mean, rstd, a, k, m, gid, sid, N, C, act, dx, lddx);
    } else if (vec4_ok<T>(C, {x, gout, dx}, {mean, rstd, a, k, m}, {ldx, ldg, lddx})) {
        const int64_t n = N * (C / 4);
        hipLaunchKernelGGL((k_norm_bwd<T, 4>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx,
                           gout, ldg, mean, rstd, a, k, m, gid, sid, N, C, act, dx, lddx);
    } else if constexpr (is_f32((const T*)nullptr)) {
        const int64_t n = N * C;
        hipLaunchKernelGGL((k_norm_bwd<T, 1>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx,
                           gout, ldg, mean, rstd, a, k, m, gid, sid, N, C, act, dx, lddx);
    } else {
        return STIN_E_UNSUPPORTED;
    }
    return stin_launch_status();
}

// row chunk of a workgroup of k_norm_fold, or 0 when the one-launch form does not pay: the grid is (C / 32) x chunks with ~1
// workgroup (16 waves) per CU, and all of them together must not read more fold bytes than 3/4 of the elementwise traffic (12
// bytes per element; 18 063 x 256 with 226 row groups: 256 workgroups x 115 KB = 30 MB beside 55 MB)
inline int norm_fold_rows(int64_t N, int C, int64_t groups) {
    constexpr int per_cu = 1;
    if (C % NF_GC != 0 || N <= 0 || groups <= 0 || groups > INT32_MAX) return 0;
    const int cg = C / NF_GC;
    int64_t chunks = (per_cu * (int64_t)norm_cu_count() + cg - 1) / cg;
    int64_t rows = (N + chunks - 1) / chunks;
    rows = (rows + NF_RL - 1) / NF_RL * NF_RL;
    if (rows < 2 * NF_RL) rows = 2 * NF_RL;
    chunks = (N + rows - 1) / rows;
    const double fold_bytes = (double)chunks * cg * (double)groups * 2.0 * NF_GC * 8.0, elem_bytes = (double)N * C * 12.0;
    return (fold_bytes <= 0.75 * elem_bytes && chunks <= 65535 && rows <= INT32_MAX) ? (int)rows : 0;
}

}  // namespace

// One-launch "fold the GEMM epilogue's statistics partials + elementwise pass" forms of the instance norm (single graph, fp32
// rows; k_norm_fold).  stin_norm_fold_rows > 0 says the form applies to (N, C, groups) - the caller then checks 16-byte rows itself
// through the return code (STIN_E_UNSUPPORTED when a pointer / pitch does not allow float4 access).
extern "C" int stin_norm_fold_rows(int64_t N, int C, int64_t groups) { return norm_fold_rows(N, C, groups); }

extern "C" int stin_norm_act_res_fwd_fold_f32(const double* partial, int64_t groups, const float* x, int64_t ldx, const float* res,
                                              int64_t ldres, const float* inv_cnt, float eps, int64_t N, int C, float* mean,
                                              float* rstd, float* y, int64_t ldy, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N > 0 && C > 0 && ldx >= C && ldy >= C && (res == nullptr || ldres >= C), STIN_E_SIZE);
    STIN_REQUIRE(partial && x && inv_cnt && mean && rstd && y, STIN_E_NULL);
    const int rows = norm_fold_rows(N, C, groups);
    if (rows <= 0 || !vec4_ok<float>(C, {x, res, y}, {}, {ldx, ldy, res ? ldres : 0})) return STIN_E_UNSUPPORTED;
    hipLaunchKernelGGL((k_norm_fold<false>), dim3((unsigned)(C / NF_GC), (unsigned)((N + rows - 1) / rows)), dim3(NF_BLOCK), 0,
                       (hipStream_t)stream, partial, (int)groups, x, ldx, res, ldres, mean, rstd, inv_cnt, eps, N, C, rows, y, ldy);
    return stin_launch_status();
}

extern "C" int stin_norm_act_bwd_fold_f32(const double* partial, int64_t groups, const float* x, int64_t ldx, const float* gout,
                                          int64_t ldg, const float* mean, const float* rstd, const float* inv_cnt, int64_t N, int C,
                                          float* dx, int64_t lddx, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N > 0 && C > 0 && ldx >= C && ldg >= C && lddx >= C, STIN_E_SIZE);
    STIN_REQUIRE(partial && x && gout && inv_cnt && mean && rstd && dx, STIN_E_NULL);
    const int rows = norm_fold_rows(N, C, groups);
    if (rows <= 0 || !vec4_ok<float>(C, {x, gout, dx}, {}, {ldx, ldg, lddx})) return STIN_E_UNSUPPORTED;
    hipLaunchKernelGGL((k_norm_fold<true>), dim3((unsigned)(C / NF_GC), (unsigned)((N + rows - 1) / rows)), dim3(NF_BLOCK), 0,
                       (hipStream_t)stream, partial, (int)groups, x, ldx, gout, ldg, const_cast<float*>(mean), const_cast<float*>(rstd),
                       inv_cnt, 0.f, N, C, rows, dx, lddx);
    return stin_launch_status();
}

extern "C" size_t stin_colreduce_workspace_bytes(int C, int B) {
    if (C <= 0 || B <= 0) return 0;
    const size_t slabs = (size_t)(B > MAX_SLABS ? B : MAX_SLABS);
    return slabs * 2 * (size_t)C * sizeof(double) + 256;
}

extern "C" int stin_colreduce_f32(int mode, const float* x, int64_t ldx, const float* gout, int64_t ldg, int64_t N, int C,
                                  const int32_t* ptr, int B, const int32_t* gid, const int32_t* sid, const float* mean,
                                  const float* rstd, const float* coef, int post, const float* inv_cnt, float eps,
                                  float* out0, float* out1, void* workspace, size_t workspace_bytes, stin_stream_t stream) {
    stin_clear_stale_error();
    return colreduce_impl<float>(mode, x, ldx, gout, ldg, N, C, ptr, B, gid, sid, mean, rstd, coef, post, inv_cnt, eps, out0,
                                 out1, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int stin_colreduce_bf16(int mode, const stin_bf16_t* x, int64_t ldx, const stin_bf16_t* gout, int64_t ldg,
                                   int64_t N, int C, const int32_t* ptr, int B, const int32_t* gid, const int32_t* sid,
                                   const float* mean, const float* rstd, const float* coef, int post, const float* inv_cnt,
                                   float eps, float* out0, float* out1, void* workspace, size_t workspace_bytes,
                                   stin_stream_t stream) {
    stin_clear_stale_error();
    return colreduce_impl<stin_bf16>(mode, b16(x), ldx, b16(gout), ldg, N, C, ptr, B, gid, sid, mean, rstd, coef, post,
                                     inv_cnt, eps, out0, out1, workspace, workspace_bytes, (hipStream_t)stream);
}

// Second stage of the fused GEMM + backward statistics (stin_gemm_nt_dotelu_f32): partial [groups][2][C] doubles (sum of
// dy xc, sum of dy per row group) -> the instance-norm backward coefficients k = -rstd^3 T1 / n, m = -rstd S0 / n of ONE graph -
// what stin_colreduce_f32(STIN_RED_DOT_ELU, post = STIN_POST_NORM_COEF) ends with (same kernel, same float operations).
extern "C" int stin_norm_coef_from_partials_f32(const double* partial, int64_t groups, int C, const float* rstd,
                                                const float* inv_cnt, float* k, float* m, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(groups > 0 && groups < ((int64_t)1 << 30) && C > 0, STIN_E_SIZE);
    STIN_REQUIRE(partial && rstd && inv_cnt && k && m, STIN_E_NULL);
    hipLaunchKernelGGL(k_colreduce_final, dim3((unsigned)((C + FIN_COLS - 1) / FIN_COLS), 2u, 1u), dim3(BLOCK), 0, (hipStream_t)stream,
                       partial, (int)groups, 2, C, 1, (int)STIN_POST_NORM_COEF, inv_cnt, 0.f, rstd, k, m);
    return stin_launch_status();
}

// Second stage of the fused GEMM + statistics (stin_gemm_nt_colstats_f32): partial [groups][2][C] doubles (sum, sum of
// squares per row group) -> mean, rstd [C] of ONE instance of N rows (inv_cnt[0] = 1 / N), same final arithmetic as the
// MOMENTS mode of stin_colreduce_f32.
extern "C" int stin_moments_final_f32(const double* partial, int64_t groups, int C, const float* inv_cnt, float eps, float* mean,
                                      float* rstd, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(groups > 0 && groups <= INT32_MAX && C > 0, STIN_E_SIZE);
    STIN_REQUIRE(partial && inv_cnt && mean && rstd, STIN_E_NULL);
    hipLaunchKernelGGL(k_moments_final, dim3((unsigned)((C + FIN_COLS - 1) / FIN_COLS), 1u, 1u), dim3(BLOCK), 0, (hipStream_t)stream,
                       partial, (int)groups, C, 1, inv_cnt, eps, mean, rstd);
    return stin_launch_status();
}

extern "C" int stin_norm_act_res_fwd_f32(const float* x, int64_t ldx, const float* mean, const float* rstd,
                                         const int32_t* gid, const float* res, int64_t ldres, int64_t N, int C, int act,
                                         float* y, int64_t ldy, stin_stream_t stream) {
    stin_clear_stale_error();
    return norm_fwd_impl<float>(x, ldx, mean, rstd, gid, res, ldres, N, C, act, y, ldy, (hipStream_t)stream);
}
extern "C" int stin_norm_act_res_fwd_bf16(const stin_bf16_t* x, int64_t ldx, const float* mean, const float* rstd,
                                          const int32_t* gid, const stin_bf16_t* res, int64_t ldres, int64_t N, int C,
                                          int act, stin_bf16_t* y, int64_t ldy, stin_stream_t stream) {
    stin_clear_stale_error();
    return norm_fwd_impl<stin_bf16>(b16(x), ldx, mean, rstd, gid, b16(res), ldres, N, C, act, b16(y), ldy,
                                    (hipStream_t)stream);
}

extern "C" int stin_norm_act_bwd_f32(const float* x, int64_t ldx, const float* gout, int64_t ldg, const float* mean,
                                     const float* rstd, const float* a, const float* k, const float* m,
                                     const int32_t* gid, const int32_t* sid, int64_t N, int C, int act, float* dx,
                                     int64_t lddx, stin_stream_t stream) {
    stin_clear_stale_error();
    return norm_bwd_impl<float>(x, ldx, gout, ldg, mean, rstd, a, k, m, gid, sid, N, C, act, dx, lddx, (hipStream_t)stream);
}
extern "C" int stin_norm_act_bwd_bf16(const stin_bf16_t* x, int64_t ldx, const stin_bf16_t* gout, int64_t ldg,
                                      const float* mean, const float* rstd, const float* a, const float* k, const float* m,
                                      const int32_t* gid, const int32_t* sid, int64_t N, int C, int act, stin_bf16_t* dx,
                                      int64_t lddx, stin_stream_t stream) {
    stin_clear_stale_error();
    return norm_bwd_impl<stin_bf16>(b16(x), ldx, b16(gout), ldg, mean, rstd, a, k, m, gid, sid, N, C, act, b16(dx), lddx,
                                    (hipStream_t)stream);
}

extern "C" int stin_bn_act_fwd_f32(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, int64_t N, int C, int act, float* y, int64_t ldy, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(N >= 0 && C > 0 && ldx >= C && ldy >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && mean && rstd && gamma && beta && y, STIN_E_NULL);
    if (vec4_ok<float>(C, {x, y}, {mean, rstd, gamma, beta}, {ldx, ldy})) {
        const int64_t n = N * (C / 4);
        hipLaunchKernelGGL((k_bn_fwd<4>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx, mean, rstd, gamma,
                           beta, N, C, act, y, ldy);
    } else {
        const int64_t n = N * C;
        hipLaunchKernelGGL((k_bn_fwd<1>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx, mean, rstd, gamma,
                           beta, N, C, act, y, ldy);
    }
    return stin_launch_status();
}

extern "C" int stin_bn_act_bwd_f32(const float* x, int64_t ldx, const float* gout, int64_t ldg, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, const float* P, const float* Q,
                                   float inv_n, int64_t N, int C, int act, float* dx, int64_t lddx, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(N >= 0 && C > 0 && ldx >= C && ldg >= C && lddx >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && gout && mean && rstd && gamma && beta && P && Q && dx, STIN_E_NULL);
    if (vec4_ok<float>(C, {x, gout, dx}, {mean, rstd, gamma, beta, P, Q}, {ldx, ldg, lddx})) {
        const int64_t n = N * (C / 4);
        hipLaunchKernelGGL((k_bn_bwd<4>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx, gout, ldg, mean,
                           rstd, gamma, beta, P, Q, inv_n, N, C, act, dx, lddx);
    } else {
        const int64_t n = N * C;
        hipLaunchKernelGGL((k_bn_bwd<1>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, x, ldx, gout, ldg, mean,
                           rstd, gamma, beta, P, Q, inv_n, N, C, act, dx, lddx);
    }
    return stin_launch_status();
}

extern "C" int stin_bn_mean_bwd_f32(const float* m, int64_t ldm, const float* g, int64_t ldg, const int32_t* dst,
                                    const float* inv_deg, const float* mean, const float* rstd, const float* gamma,
                                    const float* P, const float* Q, float inv_e, int64_t E, int C, float* dm, int64_t lddm,
                                    stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(E >= 0 && C > 0 && ldm >= C && ldg >= C && lddm >= C, STIN_E_SIZE);
    if (E == 0) return STIN_OK;
    STIN_REQUIRE(m && g && dst && inv_deg && mean && rstd && gamma && P && Q && dm, STIN_E_NULL);
    if (vec4_ok<float>(C, {m, g, dm}, {mean, rstd, gamma, P, Q}, {ldm, ldg, lddm})) {
        const int64_t n = E * (C / 4);
        hipLaunchKernelGGL((k_bn_mean_bwd<4>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, m, ldm, g, ldg, dst,
                           inv_deg, mean, rstd, gamma, P, Q, inv_e, E, C, dm, lddm);
    } else {
        const int64_t n = E * C;
        hipLaunchKernelGGL((k_bn_mean_bwd<1>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, m, ldm, g, ldg, dst,
                           inv_deg, mean, rstd, gamma, P, Q, inv_e, E, C, dm, lddm);
    }
    return stin_launch_status();
}

extern "C" int stin_bn_running_stats_f32(const float* mean, const float* rstd, int C, float eps, float unbias, float momentum,
                                         float* running_mean, float* running_var, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(C > 0, STIN_E_SIZE);
    STIN_REQUIRE(mean && rstd && running_mean && running_var, STIN_E_NULL);
    hipLaunchKernelGGL(k_bn_running, dim3((unsigned)((C + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream_, mean, rstd, C,
                       eps, unbias, momentum, running_mean, running_var);
    return stin_launch_status();
}
