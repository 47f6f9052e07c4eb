// Graph kernels of the STINet hot path for gfx950: CSR gather + segmented reduction with a
// fixed summation order (no float atomics).  One GROUP of G lanes (G = 1..64, a power of two)
// owns one destination row; each lane holds VPL float4 channel chunks, so a wave64 covers
// 64/G rows and every neighbour-row gather is a run of coalesced 16-byte loads.  U neighbour
// rows are requested before the first is consumed (memory-level parallelism; mean mesh
// degree is ~6).  Contract: include/stin_hip.h.
#include <cstdlib>
#include "stin_common.h"

thread_local hipEvent_t stin_tl_stop_event = nullptr;   // (stin_common.h)

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// streamed (non-temporal) 16-byte stores for rows nobody re-reads while they are cache-resident (h rows, ReLU masks of the forward
// edge kernels): they should not displace the gathered operand from L2 / the Infinity Cache.  STIN_EDGE_FWD_NT=0 at compile time
// restores plain stores.
#ifndef STIN_EDGE_FWD_NT
#define STIN_EDGE_FWD_NT 1
#endif
typedef unsigned int u32x4_st __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_stream(void* p, uint4 v) {
#if STIN_EDGE_FWD_NT
    u32x4_st t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_st*>(p));
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
    st16_stream(p, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
}
__device__ __forceinline__ void st4_stream(stin_bf16* p, float4 v) { st4(p, v); }     // (8-byte bf16 rows: the 4-channel kernels keep plain stores)
// ... and the destination operand A[i, :] of the forward edge kernel is read exactly ONCE (by the lane group that owns row i), while
// every B row is gathered ~deg times: A goes through non-temporal loads so that it does not displace B either
__device__ __forceinline__ float4 ld4_stream(const float* p) {
#if STIN_EDGE_FWD_NT
    const u32x4_st t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_st*>(p));
    return make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
#else
    return ld4(p);
#endif
}
__device__ __forceinline__ float4 ld4_stream(const stin_bf16* p) { return ld4(p); }

template <typename T> struct is_f32_type { static constexpr bool value = false; };
template <> struct is_f32_type<float> { static constexpr bool value = true; };

// Block -> logical block.  Row kernels are launched as a plain 1-D grid (logical = blockIdx.x) or, as an experiment switch
// (xcd_rows_on below), as an (8, q) grid: workgroups are dealt round-robin to the 8 XCDs in linear order (x fastest), so
// blockIdx.x is the XCD and the logical block x * q + y gives every XCD one CONTIGUOUS range of rows.  One formula serves
// both launch shapes.
__device__ __forceinline__ unsigned vblock_id() { return blockIdx.x * gridDim.y + blockIdx.y; }
// two-role launches (role 0 first): plain = 2 nb blocks in x; XCD order = (8, 2 q) with the roles split along y
__device__ __forceinline__ unsigned vblock_role(unsigned nb, unsigned& role) {
    if (gridDim.y == 1) {
        role = blockIdx.x >= nb ? 1u : 0u;
        return blockIdx.x - role * nb;
    }
    const unsigned q = gridDim.y >> 1;
    role = blockIdx.y >= q ? 1u : 0u;
    return blockIdx.x * q + (blockIdx.y - role * q);
}

template <int G, int VPL>
struct Lane {
    int lg;        // lane within the group
    int64_t row;   // row owned by the group
    __device__ __forceinline__ Lane() : Lane(vblock_id()) {}
    __device__ __forceinline__ explicit Lane(unsigned vblock) {          // vblock: the block index the kernel body should see
        lg = threadIdx.x % G;
        row = (int64_t)vblock * (BLOCK / G) + threadIdx.x / G;
    }
    __device__ __forceinline__ int chan(int k) const { return (k * G + lg) * 4; }
};

// ------------------------------------------------------------------ edge stage, forward
// EXACT: H == 4 * G * VPL (every width the saved-mask path supports): no per-chunk predicates on the gathers
// TI (round 6, translation-invariant blocks in the compact layout): the row's own operand is not read from memory but formed
// as a = b1 - B_i (A points at the bias vector b1 [H], or is NULL for a filter without bias).  Against the A columns the Y product
// wrote when it multiplied by [-W1 ; W1] this differs by at most one unit in the last place of the matrix-core accumulator (the
// MFMA's internal adder is not symmetric under negation: measured 2.4e-7 .. 9.5e-7 absolute on unit-scale rows) - both are fp32
// evaluations of W1 (x_j - x_i) + b1.
template <typename T, int G, int VPL, int U, bool EXACT, bool TI = false>
__device__ __forceinline__ void edge_fwd_body(const T* __restrict__ A, int64_t lda,
                                              const T* __restrict__ B, int64_t ldb,
                                              const int32_t* __restrict__ rowptr,
                                              const int32_t* __restrict__ col, int64_t N, int H,
                                              T* __restrict__ out, int64_t ldo, int indicator,
                                              uint32_t* __restrict__ mask) {
    Lane<G, VPL> L;
    const bool row_ok = L.row < N;
    if (!row_ok) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    float4 a[VPL], acc[VPL];
    bool on[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        on[k] = EXACT || L.chan(k) < H;
        if (TI) {
            const float4 bo = ld4(B + L.row * ldb + L.chan(k));
            const float4 bi = A != nullptr ? ld4(A + L.chan(k)) : f4zero();
            a[k] = make_float4(bi.x - bo.x, bi.y - bo.y, bi.z - bo.z, bi.w - bo.w);
        } else {
            a[k] = on[k] ? ld4_stream(A + L.row * lda + L.chan(k)) : f4zero();
        }
        acc[k] = f4zero();
    }
    const int mwords = H >> 5;                            // mask words per edge slot
    for (int e = beg; e < end; e += U) {
        float4 b[U][VPL];
        uint4 mw[VPL];                                    // this lane's share of the U neighbours' ReLU masks
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);          // clamped: branch-free, always a valid row
            const int64_t j = col[ee];
#pragma unroll
            for (int k = 0; k < VPL; ++k) b[u][k] = on[k] ? ld4(B + j * ldb + L.chan(k)) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float w = (e + u < end) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const float tx = a[k].x + b[u][k].x, ty = a[k].y + b[u][k].y;
                const float tz = a[k].z + b[u][k].z, tw = a[k].w + b[u][k].w;
                acc[k].x += w * fmaxf(tx, 0.f);
                acc[k].y += w * fmaxf(ty, 0.f);
                acc[k].z += w * fmaxf(tz, 0.f);
                acc[k].w += w * fmaxf(tw, 0.f);
                if (G >= 32 && mask != nullptr) {
                    // Mask layout per destination-CSR slot (H bits): [k][component x,y,z,w][G lane bits]: exactly what
                    // the wave ballots of the four compares deliver (the lane masks are already in SGPRs).  Lane u of
                    // the row keeps neighbour u's words; after the u-loop lanes 0..U-1 store them with ONE instruction
                    // (U consecutive slots = U*16 contiguous bytes for G = 32).
                    const unsigned long long bx = __ballot(tx > 0.f), by = __ballot(ty > 0.f);
                    const unsigned long long bz = __ballot(tz > 0.f), bw = __ballot(tw > 0.f);
                    if (G == 32) {
                        const int sh = (threadIdx.x & 32);            // which half of the wave this row lives in
                        if (L.lg == u) mw[k] = make_uint4((uint32_t)(bx >> sh), (uint32_t)(by >> sh), (uint32_t)(bz >> sh), (uint32_t)(bw >> sh));
                    } else {
                        if (L.lg == 2 * u) mw[k] = make_uint4((uint32_t)bx, (uint32_t)(bx >> 32), (uint32_t)by, (uint32_t)(by >> 32));
                        if (L.lg == 2 * u + 1) mw[k] = make_uint4((uint32_t)bz, (uint32_t)(bz >> 32), (uint32_t)bw, (uint32_t)(bw >> 32));
                    }
                }
            }
        }
        if (G >= 32 && mask != nullptr) {
            // G = 32: lane u -> slot e+u, 16 bytes per k.  G = 64: lanes 2u, 2u+1 -> the two 16-byte halves of slot e+u.
            const int uu = (G == 32) ? L.lg : (L.lg >> 1);
            if (uu < U && e + uu < end) {
#pragma unroll
                for (int k = 0; k < VPL; ++k) {
                    uint32_t* m = mask + (int64_t)(e + uu) * mwords + k * (G / 8) + ((G == 32) ? 0 : (L.lg & 1) * 4);
                    st16_stream(m, mw[k]);
                }
            }
        }
    }
    const int deg = end - beg;
    const float s = (float)(deg > 0 ? deg : 1);      // true division, as torch_scatter's scatter_mean (sum / count)
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (on[k]) st4_stream(out + L.row * ldo + L.chan(k), make_float4(acc[k].x / s, acc[k].y / s, acc[k].z / s, acc[k].w / s));
    if (indicator && L.lg == 0) st4(out + L.row * ldo + H, make_float4(deg > 0 ? 1.f : 0.f, 0.f, 0.f, 0.f));
}
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_fwd(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    int64_t N, int H, T* __restrict__ out, int64_t ldo, int indicator,
                                                    uint32_t* __restrict__ mask) {
    edge_fwd_body<T, G, VPL, U, false>(A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
}
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_fwd_exact(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          int64_t N, int H, T* __restrict__ out, int64_t ldo, int indicator,
                                                          uint32_t* __restrict__ mask) {
    edge_fwd_body<T, G, VPL, U, true>(A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
}

template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_fwd_ti(const T* __restrict__ b1, int64_t, const T* __restrict__ B, int64_t ldb,
                                                       const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       int64_t N, int H, T* __restrict__ out, int64_t ldo, int indicator,
                                                       uint32_t* __restrict__ mask) {
    edge_fwd_body<T, G, VPL, U, true, true>(b1, 0, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
}

// ------------------------------------------ edge stage, backward w.r.t. A (destination CSR)
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_dst(const T* __restrict__ A, int64_t lda,
                                                        const T* __restrict__ B, int64_t ldb,
                                                        const T* __restrict__ Gr, int64_t ldg,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, int64_t N, int H,
                                                        T* __restrict__ dA, int64_t ldda) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    float4 a[VPL], cnt[VPL];
    bool on[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        on[k] = L.chan(k) < H;
        a[k] = on[k] ? ld4(A + L.row * lda + L.chan(k)) : f4zero();
        cnt[k] = f4zero();
    }
    for (int e = beg; e < end; e += U) {
        float4 b[U][VPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
            const int64_t j = col[ee];
#pragma unroll
            for (int k = 0; k < VPL; ++k) b[u][k] = on[k] ? ld4(B + j * ldb + L.chan(k)) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float w = (e + u < end) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                cnt[k].x += (a[k].x + b[u][k].x > 0.f) ? w : 0.f;
                cnt[k].y += (a[k].y + b[u][k].y > 0.f) ? w : 0.f;
                cnt[k].z += (a[k].z + b[u][k].z > 0.f) ? w : 0.f;
                cnt[k].w += (a[k].w + b[u][k].w > 0.f) ? w : 0.f;
            }
        }
    }
    const int deg = end - beg;
    const float s = 1.0f / (float)(deg > 0 ? deg : 1);
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (on[k]) {
            const float4 g = ld4(Gr + L.row * ldg + L.chan(k));
            st4(dA + L.row * ldda + L.chan(k),
                make_float4(g.x * s * cnt[k].x, g.y * s * cnt[k].y, g.z * s * cnt[k].z, g.w * s * cnt[k].w));
        }
}

// ----------------------------------------------- edge stage, backward w.r.t. B (source CSR)
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_src(const T* __restrict__ A, int64_t lda,
                                                        const T* __restrict__ B, int64_t ldb,
                                                        const T* __restrict__ Gr, int64_t ldg,
                                                        const float* __restrict__ inv_deg,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, int64_t N, int H,
                                                        T* __restrict__ dB, int64_t lddb) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    float4 b[VPL], acc[VPL];
    bool on[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        on[k] = L.chan(k) < H;
        b[k] = on[k] ? ld4(B + L.row * ldb + L.chan(k)) : f4zero();
        acc[k] = f4zero();
    }
    for (int e = beg; e < end; e += U) {
        float4 a[U][VPL], g[U][VPL];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
            const int64_t i = col[ee];
            const float wi = inv_deg[i];
            w[u] = (e + u < end) ? wi : 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                a[u][k] = on[k] ? ld4(A + i * lda + L.chan(k)) : f4zero();
                g[u][k] = on[k] ? ld4(Gr + i * ldg + L.chan(k)) : f4zero();
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                acc[k].x += (a[u][k].x + b[k].x > 0.f) ? w[u] * g[u][k].x : 0.f;
                acc[k].y += (a[u][k].y + b[k].y > 0.f) ? w[u] * g[u][k].y : 0.f;
                acc[k].z += (a[u][k].z + b[k].z > 0.f) ? w[u] * g[u][k].z : 0.f;
                acc[k].w += (a[u][k].w + b[k].w > 0.f) ? w[u] * g[u][k].w : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (on[k]) st4(dB + L.row * lddb + L.chan(k), acc[k]);
}

// ----------------------------- edge stage backward from the saved ReLU bit-mask (no recompute)
// dA[i,c] = G[i,c]/deg_i * popcount_e mask[e][c] over the in-edge slots e of i: a pure streaming kernel
// (reads H/8 bytes per edge instead of gathering a B row).
template <typename T, int G, int VPL, int U>
__device__ __forceinline__ void edge_bwd_dst_mask_body(unsigned vblock, const T* __restrict__ Gr, int64_t ldg,
                                                       const uint32_t* __restrict__ mask,
                                                       const int32_t* __restrict__ rowptr, int64_t N, int H,
                                                       T* __restrict__ dA, int64_t ldda) {
    Lane<G, VPL> L(vblock);
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    const int mwords = H >> 5;
    int cnt[VPL][4];
    bool on[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        on[k] = true;                                    // the mask path exists for H == 4 * G * VPL only
        cnt[k][0] = cnt[k][1] = cnt[k][2] = cnt[k][3] = 0;
    }
    constexpr int WPC = G / 32;                          // 32-bit words per component
    const int wsel = L.lg >> 5, bit = L.lg & 31;
    for (int e = beg; e < end; e += U) {
        uint32_t wv[U][VPL][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const uint32_t* m = mask + (int64_t)ee * mwords + k * (G / 8);
                if (WPC == 1) {
                    const uint4 q = *reinterpret_cast<const uint4*>(m);     // all lanes of the row: same address
                    wv[u][k][0] = q.x; wv[u][k][1] = q.y; wv[u][k][2] = q.z; wv[u][k][3] = q.w;
                } else {
                    const uint4 q0 = reinterpret_cast<const uint4*>(m)[0], q1 = reinterpret_cast<const uint4*>(m)[1];
                    wv[u][k][0] = wsel ? q0.y : q0.x; wv[u][k][1] = wsel ? q0.w : q0.z;
                    wv[u][k][2] = wsel ? q1.y : q1.x; wv[u][k][3] = wsel ? q1.w : q1.z;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u < end) {
#pragma unroll
                for (int k = 0; k < VPL; ++k) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) cnt[k][c] += (wv[u][k][c] >> bit) & 1u;
                }
            }
        }
    }
    const int deg = end - beg;
    const float s = 1.0f / (float)(deg > 0 ? deg : 1);
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (on[k]) {
            const float4 g = ld4(Gr + L.row * ldg + L.chan(k));
            st4(dA + L.row * ldda + L.chan(k), make_float4(g.x * s * (float)cnt[k][0], g.y * s * (float)cnt[k][1],
                                                           g.z * s * (float)cnt[k][2], g.w * s * (float)cnt[k][3]));
        }
}
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_dst_mask(const T* __restrict__ Gr, int64_t ldg,
                                                             const uint32_t* __restrict__ mask,
                                                             const int32_t* __restrict__ rowptr, int64_t N, int H,
                                                             T* __restrict__ dA, int64_t ldda) {
    edge_bwd_dst_mask_body<T, G, VPL, U>(vblock_id(), Gr, ldg, mask, rowptr, N, H, dA, ldda);
}

// dB[j,c] = sum over out-edges (j -> i) of inv_deg[i] * G[i,c] * mask[xslot][c]: gathers G rows and 32-bit mask
// words (xslot = destination-CSR slot of the same edge), half the bytes of the recompute form.
template <typename T, int G, int VPL, int U>
__device__ __forceinline__ void edge_bwd_src_mask_body(unsigned vblock, const T* __restrict__ Gr, int64_t ldg,
                                                       const float* __restrict__ w_slot,
                                                       const uint32_t* __restrict__ mask,
                                                       const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ xslot, int64_t N, int H,
                                                       T* __restrict__ dB, int64_t lddb) {
    Lane<G, VPL> L(vblock);
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    const int mwords = H >> 5;
    float4 acc[VPL];
    bool on[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        on[k] = true;                                    // the mask path exists for H == 4 * G * VPL only
        acc[k] = f4zero();
    }
    constexpr int WPC = G / 32;
    const int wsel = L.lg >> 5, bit = L.lg & 31;
    for (int e = beg; e < end; e += U) {
        float4 g[U][VPL];
        uint32_t wv[U][VPL][4];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
            const int64_t i = col[ee];
            const int64_t xs = xslot[ee];
            const float ws = w_slot[ee];                 // (unconditional: a load behind `e + u < end` is a branch, and hipcc then
            w[u] = (e + u < end) ? ws : 0.f;             //  drains the loads in flight before the next slot's gathers)
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                g[u][k] = ld4(Gr + i * ldg + L.chan(k));
                const uint32_t* m = mask + xs * mwords + k * (G / 8);
                if (WPC == 1) {
                    const uint4 q = *reinterpret_cast<const uint4*>(m);
                    wv[u][k][0] = q.x; wv[u][k][1] = q.y; wv[u][k][2] = q.z; wv[u][k][3] = q.w;
                } else {
                    const uint4 q0 = reinterpret_cast<const uint4*>(m)[0], q1 = reinterpret_cast<const uint4*>(m)[1];
                    wv[u][k][0] = wsel ? q0.y : q0.x; wv[u][k][1] = wsel ? q0.w : q0.z;
                    wv[u][k][2] = wsel ? q1.y : q1.x; wv[u][k][3] = wsel ? q1.w : q1.z;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                acc[k].x += ((wv[u][k][0] >> bit) & 1u) ? w[u] * g[u][k].x : 0.f;
                acc[k].y += ((wv[u][k][1] >> bit) & 1u) ? w[u] * g[u][k].y : 0.f;
                acc[k].z += ((wv[u][k][2] >> bit) & 1u) ? w[u] * g[u][k].z : 0.f;
                acc[k].w += ((wv[u][k][3] >> bit) & 1u) ? w[u] * g[u][k].w : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (on[k]) st4(dB + L.row * lddb + L.chan(k), acc[k]);
}
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_src_mask(const T* __restrict__ Gr, int64_t ldg,
                                                             const float* __restrict__ w_slot,
                                                             const uint32_t* __restrict__ mask,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ xslot, int64_t N, int H,
                                                             T* __restrict__ dB, int64_t lddb) {
    edge_bwd_src_mask_body<T, G, VPL, U>(vblock_id(), Gr, ldg, w_slot, mask, rowptr, col, xslot, N, H, dB, lddb);
}

// Both halves of the mask backward in ONE launch: blocks [0, nb) compute dB rows (the longer, gather-bound half first),
// blocks [nb, 2 nb) the dA rows - one launch overhead less per block backward, and the streaming half fills the tail of the
// gather half.  Same arithmetic per row as the two kernels (bit-identical).  (Interleaving the two roles block by block
// measured 13 % SLOWER than the two launches: it halves the locality of both.)
template <typename T, int G, int VPL, int UD, int US>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_mask_pair(const T* __restrict__ Gr, int64_t ldg,
                                                              const uint32_t* __restrict__ mask,
                                                              const int32_t* __restrict__ rowptr_dst,
                                                              const float* __restrict__ w_slot,
                                                              const int32_t* __restrict__ rowptr_src,
                                                              const int32_t* __restrict__ col_src,
                                                              const int32_t* __restrict__ xslot, int64_t N, int H,
                                                              T* __restrict__ dA, int64_t ldda, T* __restrict__ dB,
                                                              int64_t lddb, unsigned nb, const T* __restrict__ cp_src,
                                                              int64_t ld_cps, T* __restrict__ cp_dst, int64_t ld_cpd, int Ccp) {
    unsigned role;
    const unsigned vb = vblock_role(nb, role);
    if (role == 0) {
        edge_bwd_src_mask_body<T, G, VPL, US>(vb, Gr, ldg, w_slot, mask, rowptr_src, col_src, xslot, N, H, dB, lddb);
    } else {
        // optional row copy riding on the streaming role (the block backward's dY[:, 2H:] = g of a shortcut block: one
        // 2-D memcpy launch less); Ccp <= H channels, 4 per lane
        if (cp_src != nullptr) {
            Lane<G, VPL> L(vb);
            if (L.row < N) {
#pragma unroll
                for (int k = 0; k < VPL; ++k)
                    if (L.chan(k) < Ccp) st4(cp_dst + L.row * ld_cpd + L.chan(k), ld4(cp_src + L.row * ld_cps + L.chan(k)));
            }
        }
        edge_bwd_dst_mask_body<T, G, VPL, UD>(vb, Gr, ldg, mask, rowptr_dst, N, H, dA, ldda);
    }
}

// ---- translation-invariant blocks in the compact layout (round 6): A_i = b1 - B_i, so dL/dB_i collects BOTH roles of vertex i:
//   D_i = dB_i - dA_i,   dB_i = sum over out-edges (the gathering half above),  dA_i = G_i / deg_i * popcount of the in-edge masks
// (the streaming half).  One block computes both halves of its rows - same arithmetic per half as the two bodies above - and writes
// ONE row of H channels where the pair kernel wrote two; the first Linear's backward products (dx = D W1, dW1 = D^T x) then run
// over H columns instead of 2 H.  db1 = sum_i dA_i no longer falls out of the transposed product's ones column (sum_i D_i is ~ 0):
// every lane adds up dA over the TI_ITER rows it visits (a lane keeps its channels from row to row), the block folds its row slots
// in a fixed order through LDS and writes colsum[blockIdx][H]; stin_wgrad.hip's finalize adds the block rows in a fixed order.
constexpr int TI_ITER = 4;                                 // row groups per block (fixes the number of partial rows: see ti_colsum_rows)
template <typename T, int G, int VPL, int UD, int US>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_mask_ti(const T* __restrict__ Gr, int64_t ldg, const uint32_t* __restrict__ mask,
                                                            const int32_t* __restrict__ rowptr_dst, const float* __restrict__ w_slot,
                                                            const int32_t* __restrict__ rowptr_src, const int32_t* __restrict__ col_src,
                                                            const int32_t* __restrict__ xslot, int64_t N, int H, T* __restrict__ D,
                                                            int64_t ldd, const T* __restrict__ cp_src, int64_t ld_cps,
                                                            T* __restrict__ cp_dst, int64_t ld_cpd, int Ccp, float* __restrict__ colsum) {
    constexpr int RPB = BLOCK / G;                          // rows per block and iteration
    __shared__ float4 red[RPB][G * VPL];
    const int mwords = H >> 5;
    constexpr int WPC = G / 32;
    float4 csum[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) csum[k] = f4zero();
    for (int it = 0; it < TI_ITER; ++it) {
        Lane<G, VPL> L(blockIdx.x * TI_ITER + it);
        if (L.row >= N) break;                              // (rows ascend with `it`: nothing further for this row slot)
        const int wsel = L.lg >> 5, bit = L.lg & 31;
        // ---- gathering half: dB (edge_bwd_src_mask_body)
        float4 acc[VPL];
#pragma unroll
        for (int k = 0; k < VPL; ++k) acc[k] = f4zero();
        {
            const int beg = rowptr_src[L.row], end = rowptr_src[L.row + 1];
            for (int e = beg; e < end; e += US) {
                float4 g[US][VPL];
                uint32_t wv[US][VPL][4];
                float w[US];
#pragma unroll
                for (int u = 0; u < US; ++u) {
                    const int ee = min(e + u, end - 1);
                    const int64_t i = col_src[ee];
                    const int64_t xs = xslot[ee];
                    const float ws = w_slot[ee];
                    w[u] = (e + u < end) ? ws : 0.f;
#pragma unroll
                    for (int k = 0; k < VPL; ++k) {
                        g[u][k] = ld4(Gr + i * ldg + L.chan(k));
                        const uint32_t* m = mask + xs * mwords + k * (G / 8);
                        if (WPC == 1) {
                            const uint4 q = *reinterpret_cast<const uint4*>(m);
                            wv[u][k][0] = q.x; wv[u][k][1] = q.y; wv[u][k][2] = q.z; wv[u][k][3] = q.w;
                        } else {
                            const uint4 q0 = reinterpret_cast<const uint4*>(m)[0], q1 = reinterpret_cast<const uint4*>(m)[1];
                            wv[u][k][0] = wsel ? q0.y : q0.x; wv[u][k][1] = wsel ? q0.w : q0.z;
                            wv[u][k][2] = wsel ? q1.y : q1.x; wv[u][k][3] = wsel ? q1.w : q1.z;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < US; ++u) {
#pragma unroll
                    for (int k = 0; k < VPL; ++k) {
                        acc[k].x += ((wv[u][k][0] >> bit) & 1u) ? w[u] * g[u][k].x : 0.f;
                        acc[k].y += ((wv[u][k][1] >> bit) & 1u) ? w[u] * g[u][k].y : 0.f;
                        acc[k].z += ((wv[u][k][2] >> bit) & 1u) ? w[u] * g[u][k].z : 0.f;
                        acc[k].w += ((wv[u][k][3] >> bit) & 1u) ? w[u] * g[u][k].w : 0.f;
                    }
                }
            }
        }
        // ---- streaming half: dA (edge_bwd_dst_mask_body)
        int cnt[VPL][4];
#pragma unroll
        for (int k = 0; k < VPL; ++k) cnt[k][0] = cnt[k][1] = cnt[k][2] = cnt[k][3] = 0;
        const int beg = rowptr_dst[L.row], end = rowptr_dst[L.row + 1];
        for (int e = beg; e < end; e += UD) {
            uint32_t wv[UD][VPL][4];
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                const int ee = min(e + u, end - 1);
#pragma unroll
                for (int k = 0; k < VPL; ++k) {
                    const uint32_t* m = mask + (int64_t)ee * mwords + k * (G / 8);
                    if (WPC == 1) {
                        const uint4 q = *reinterpret_cast<const uint4*>(m);
                        wv[u][k][0] = q.x; wv[u][k][1] = q.y; wv[u][k][2] = q.z; wv[u][k][3] = q.w;
                    } else {
                        const uint4 q0 = reinterpret_cast<const uint4*>(m)[0], q1 = reinterpret_cast<const uint4*>(m)[1];
                        wv[u][k][0] = wsel ? q0.y : q0.x; wv[u][k][1] = wsel ? q0.w : q0.z;
                        wv[u][k][2] = wsel ? q1.y : q1.x; wv[u][k][3] = wsel ? q1.w : q1.z;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                if (e + u < end) {
#pragma unroll
                    for (int k = 0; k < VPL; ++k) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) cnt[k][c] += (wv[u][k][c] >> bit) & 1u;
                    }
                }
            }
        }
        const int deg = end - beg;
        const float s = 1.0f / (float)(deg > 0 ? deg : 1);
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const float4 g = ld4(Gr + L.row * ldg + L.chan(k));
            const float4 dA = make_float4(g.x * s * (float)cnt[k][0], g.y * s * (float)cnt[k][1], g.z * s * (float)cnt[k][2],
                                          g.w * s * (float)cnt[k][3]);
            st4(D + L.row * ldd + L.chan(k), make_float4(acc[k].x - dA.x, acc[k].y - dA.y, acc[k].z - dA.z, acc[k].w - dA.w));
            csum[k].x += dA.x; csum[k].y += dA.y; csum[k].z += dA.z; csum[k].w += dA.w;
            // optional row copy (the block backward's dY[:, H:] = g of a shortcut block)
            if (cp_src != nullptr && L.chan(k) < Ccp) st4(cp_dst + L.row * ld_cpd + L.chan(k), ld4(cp_src + L.row * ld_cps + L.chan(k)));
        }
    }
    // ---- column sums of dA over this block's rows: row slots folded in ascending order
    const int slot = threadIdx.x / G, lg = threadIdx.x % G;
#pragma unroll
    for (int k = 0; k < VPL; ++k) red[slot][k * G + lg] = csum[k];
    __syncthreads();
    if (slot == 0) {
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            float4 t = red[0][k * G + lg];
#pragma unroll
            for (int r = 1; r < RPB; ++r) {
                const float4 v = red[r][k * G + lg];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            *reinterpret_cast<float4*>(colsum + (int64_t)blockIdx.x * H + (k * G + lg) * 4) = t;
        }
    }
}

// ===================================================== bf16 rows, 8 channels (16 bytes) per lane
// The bf16-storage edge stage with the SAME bytes per lane and request as the fp32 kernels: a lane owns 8 channels, so a
// row of H channels takes H/8 lanes (G = 16 for H = 128 ... 64 for H >= 512) and a wave covers 64/G rows.  Arithmetic
// per channel is identical to the 4-channel kernels (same fp32 operations, same neighbour order) - only the lane
// geometry and therefore the bit layout of the saved ReLU mask differ:
//   slot (H bits) = [k][s][c][min(G, 32) bits]   k = lane chunk, s = 32-lane half of the row (G = 64 only),
//   c = channel 0..7 of the lane; for G = 16 the 16-bit pieces of channels 2w, 2w+1 share word w.
// Both backward kernels read exactly this layout; the forward / backward pair always runs on the same geometry because
// the bf16 mask path REQUIRES 16-byte aligned rows (it never falls back to the 8-byte kernels).
struct F8 { float v[8]; };
__device__ __forceinline__ F8 f8zero() { F8 o; 
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = 0.f;
    return o; }
__device__ __forceinline__ F8 ld8(const stin_bf16* p) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    F8 o;
    o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
    o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    o.v[4] = __uint_as_float(r.z << 16); o.v[5] = __uint_as_float(r.z & 0xffff0000u);
    o.v[6] = __uint_as_float(r.w << 16); o.v[7] = __uint_as_float(r.w & 0xffff0000u);
    return o;
}
__device__ __forceinline__ uint32_t pk2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 h = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<uint32_t*>(&h);
}
__device__ __forceinline__ void st8(stin_bf16* p, const F8& a) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pk2(a.v[0], a.v[1]), pk2(a.v[2], a.v[3]), pk2(a.v[4], a.v[5]), pk2(a.v[6], a.v[7]));
}
__device__ __forceinline__ void st8_stream(stin_bf16* p, const F8& a) {
    st16_stream(p, make_uint4(pk2(a.v[0], a.v[1]), pk2(a.v[2], a.v[3]), pk2(a.v[4], a.v[5]), pk2(a.v[6], a.v[7])));
}

template <int G>
struct Lane8 {
    int lg;
    int64_t row;
    __device__ __forceinline__ Lane8() : Lane8(vblock_id()) {}
    __device__ __forceinline__ explicit Lane8(unsigned vblock) {
        lg = threadIdx.x % G;
        row = (int64_t)vblock * (BLOCK / G) + threadIdx.x / G;
    }
    __device__ __forceinline__ int chan(int k) const { return (k * G + lg) * 8; }
};

// Every launch of the 8-channel kernels is EXACT: H == 8 G VPL (STIN_DISPATCH8 / wide8_ok admit H = 128 .. 2048 only), so no
// lane is idle and no load is predicated.  (Round 4: the per-chunk `chan < H` predicates of the first version put every
// neighbour-row load behind its own branch; hipcc then issued index load -> wait -> row load -> index load -> wait ..., i.e. ONE
// row in flight per lane group whatever U said - the kernel ran at 0.60-0.76 of peak where its fp32 twin reaches 0.88.)
template <int G, int VPL, int U, bool MASK>
__device__ __forceinline__ void edge_fwd8_body(const stin_bf16* __restrict__ A, int64_t lda, const stin_bf16* __restrict__ B,
                                               int64_t ldb, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                               int64_t N, int H, stin_bf16* __restrict__ out, int64_t ldo, int indicator,
                                               uint32_t* __restrict__ mask) {
    constexpr int WK = G / 4;                 // mask words per lane chunk k (8 channels x G lanes / 32)
    constexpr int LPS = WK / 4;               // lanes that store one neighbour's words of one k (16 bytes each)
    static_assert(U * LPS <= G, "not enough lanes to store the mask words of U neighbours");
    Lane8<G> L;
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    F8 a[VPL], acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        a[k] = ld8(A + L.row * lda + L.chan(k));
        acc[k] = f8zero();
    }
    const int mwords = H >> 5;
    const int sh = ((threadIdx.x & 63) / G) * G;          // bit position of this row inside the wave ballots
    const int su = L.lg / LPS, sp = L.lg % LPS;           // neighbour / 16-byte part this lane stores
    for (int e = beg; e < end; e += U) {
        // all U neighbour indices first, then all U rows: U rows in flight, kept PACKED (4 registers per 8 channels, widened at use)
        int64_t j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) j[u] = col[min(e + u, end - 1)];
        uint4 braw[U][VPL];
        uint4 mw[VPL];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < VPL; ++k) braw[u][k] = *reinterpret_cast<const uint4*>(B + j[u] * ldb + L.chan(k));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = e + u < end;
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                uint32_t wd[WK];
#pragma unroll
                for (int q = 0; q < WK; ++q) wd[q] = 0u;
                const uint32_t rw[4] = {braw[u][k].x, braw[u][k].y, braw[u][k].z, braw[u][k].w};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float bv = __uint_as_float((c & 1) ? (rw[c >> 1] & 0xffff0000u) : (rw[c >> 1] << 16));
                    const float t = a[k].v[c] + bv;
                    const bool pos = t > 0.f;
                    acc[k].v[c] += (valid && pos) ? t : 0.f;        // = w ReLU(t), w = [the slot is an edge of this row]
                    if (MASK) {
                        const unsigned long long bc = __ballot(pos);
                        if (G == 16) wd[c >> 1] |= ((uint32_t)(bc >> sh) & 0xffffu) << ((c & 1) * 16);
                        else if (G == 32) wd[c] = (uint32_t)(bc >> sh);
                        else { wd[c] = (uint32_t)bc; wd[8 + c] = (uint32_t)(bc >> 32); }
                    }
                }
                if (MASK && su == u) {
#pragma unroll
                    for (int q = 0; q < LPS; ++q)
                        if (sp == q) mw[k] = make_uint4(wd[4 * q], wd[4 * q + 1], wd[4 * q + 2], wd[4 * q + 3]);
                }
            }
        }
        if (MASK && su < U && e + su < end) {
#pragma unroll
            for (int k = 0; k < VPL; ++k)
                st16_stream(mask + (int64_t)(e + su) * mwords + k * WK + 4 * sp, mw[k]);
        }
    }
    const int deg = end - beg;
    const float s = (float)(deg > 0 ? deg : 1);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        F8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o.v[c] = acc[k].v[c] / s;
        st8_stream(out + L.row * ldo + L.chan(k), o);
    }
    if (indicator && L.lg == 0) st4(out + L.row * ldo + H, make_float4(deg > 0 ? 1.f : 0.f, 0.f, 0.f, 0.f));
}
template <int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_fwd8(const stin_bf16* __restrict__ A, int64_t lda,
                                                     const stin_bf16* __restrict__ B, int64_t ldb,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     int64_t N, int H, stin_bf16* __restrict__ out, int64_t ldo, int indicator,
                                                     uint32_t* __restrict__ mask) {
    if (mask != nullptr) edge_fwd8_body<G, VPL, U, true>(A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
    else edge_fwd8_body<G, VPL, U, false>(A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
}

// this lane's 8 mask words (one per channel) of slot `m` for chunk k, and the bit to test
template <int G>
__device__ __forceinline__ void mask_words8(const uint32_t* __restrict__ m, int k, int lg, uint32_t (&w)[8], int& bit) {
    constexpr int WK = G / 4;
    if (G == 16) {
        const uint4 q = *reinterpret_cast<const uint4*>(m + k * WK);
        w[0] = q.x; w[1] = q.x >> 16; w[2] = q.y; w[3] = q.y >> 16; w[4] = q.z; w[5] = q.z >> 16; w[6] = q.w; w[7] = q.w >> 16;
        bit = lg;
    } else {
        const uint32_t* p = m + k * WK + (G == 64 ? (lg >> 5) * 8 : 0);
        const uint4 q0 = reinterpret_cast<const uint4*>(p)[0], q1 = reinterpret_cast<const uint4*>(p)[1];
        w[0] = q0.x; w[1] = q0.y; w[2] = q0.z; w[3] = q0.w; w[4] = q1.x; w[5] = q1.y; w[6] = q1.z; w[7] = q1.w;
        bit = lg & 31;
    }
}

template <int G, int VPL, int U>
__device__ __forceinline__ void edge_bwd_dst_mask8_body(unsigned vblock, const stin_bf16* __restrict__ Gr, int64_t ldg,
                                                        const uint32_t* __restrict__ mask,
                                                        const int32_t* __restrict__ rowptr, int64_t N, int H,
                                                        stin_bf16* __restrict__ dA, int64_t ldda) {
    Lane8<G> L(vblock);
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    const int mwords = H >> 5;
    int cnt[VPL][8];
#pragma unroll
    for (int k = 0; k < VPL; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) cnt[k][c] = 0;
    for (int e = beg; e < end; e += U) {
        uint32_t wv[U][VPL][8];
        int bit = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
#pragma unroll
            for (int k = 0; k < VPL; ++k) mask_words8<G>(mask + (int64_t)ee * mwords, k, L.lg, wv[u][k], bit);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u < end) {
#pragma unroll
                for (int k = 0; k < VPL; ++k)
#pragma unroll
                    for (int c = 0; c < 8; ++c) cnt[k][c] += (wv[u][k][c] >> bit) & 1u;
            }
        }
    }
    const int deg = end - beg;
    const float s = 1.0f / (float)(deg > 0 ? deg : 1);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const F8 g = ld8(Gr + L.row * ldg + L.chan(k));
        F8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o.v[c] = g.v[c] * s * (float)cnt[k][c];
        st8(dA + L.row * ldda + L.chan(k), o);
    }
}
template <int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_dst_mask8(const stin_bf16* __restrict__ Gr, int64_t ldg,
                                                              const uint32_t* __restrict__ mask,
                                                              const int32_t* __restrict__ rowptr, int64_t N, int H,
                                                              stin_bf16* __restrict__ dA, int64_t ldda) {
    edge_bwd_dst_mask8_body<G, VPL, U>(vblock_id(), Gr, ldg, mask, rowptr, N, H, dA, ldda);
}

template <int G, int VPL, int U>
__device__ __forceinline__ void edge_bwd_src_mask8_body(unsigned vblock, const stin_bf16* __restrict__ Gr, int64_t ldg,
                                                        const float* __restrict__ w_slot,
                                                        const uint32_t* __restrict__ mask,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col,
                                                        const int32_t* __restrict__ xslot, int64_t N, int H,
                                                        stin_bf16* __restrict__ dB, int64_t lddb) {
    Lane8<G> L(vblock);
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    const int mwords = H >> 5;
    F8 acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = f8zero();
    for (int e = beg; e < end; e += U) {
        F8 g[U][VPL];
        uint32_t wv[U][VPL][8];
        float w[U];
        int64_t ii[U], xs[U];
        int bit = 0;
        // indices of all U slots first, then their rows and mask words: U gathers in flight (see edge_fwd8_body)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
            ii[u] = col[ee];
            xs[u] = xslot[ee];
            const float ws = w_slot[ee];
            w[u] = (e + u < end) ? ws : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                g[u][k] = ld8(Gr + ii[u] * ldg + L.chan(k));
                mask_words8<G>(mask + xs[u] * mwords, k, L.lg, wv[u][k], bit);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < VPL; ++k)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[k].v[c] += ((wv[u][k][c] >> bit) & 1u) ? w[u] * g[u][k].v[c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k) st8(dB + L.row * lddb + L.chan(k), acc[k]);
}
template <int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_src_mask8(const stin_bf16* __restrict__ Gr, int64_t ldg,
                                                              const float* __restrict__ w_slot,
                                                              const uint32_t* __restrict__ mask,
                                                              const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col,
                                                              const int32_t* __restrict__ xslot, int64_t N, int H,
                                                              stin_bf16* __restrict__ dB, int64_t lddb) {
    edge_bwd_src_mask8_body<G, VPL, U>(vblock_id(), Gr, ldg, w_slot, mask, rowptr, col, xslot, N, H, dB, lddb);
}
// dB blocks then dA blocks in one launch, as k_edge_bwd_mask_pair does for fp32 rows
template <int G, int VPL, int UD, int US>
__global__ __launch_bounds__(BLOCK) void k_edge_bwd_mask_pair8(const stin_bf16* __restrict__ Gr, int64_t ldg,
                                                               const uint32_t* __restrict__ mask,
                                                               const int32_t* __restrict__ rowptr_dst,
                                                               const float* __restrict__ w_slot,
                                                               const int32_t* __restrict__ rowptr_src,
                                                               const int32_t* __restrict__ col_src,
                                                               const int32_t* __restrict__ xslot, int64_t N, int H,
                                                               stin_bf16* __restrict__ dA, int64_t ldda,
                                                               stin_bf16* __restrict__ dB, int64_t lddb, unsigned nb,
                                                               const stin_bf16* __restrict__ cp_src, int64_t ld_cps,
                                                               stin_bf16* __restrict__ cp_dst, int64_t ld_cpd, int Ccp) {
    unsigned role;
    const unsigned vb = vblock_role(nb, role);
    if (role == 0) {
        edge_bwd_src_mask8_body<G, VPL, US>(vb, Gr, ldg, w_slot, mask, rowptr_src, col_src, xslot, N, H, dB, lddb);
    } else {
        if (cp_src != nullptr) {                           // optional row copy, 8 channels (16 bytes) per lane
            Lane8<G> L(vb);
            if (L.row < N) {
#pragma unroll
                for (int k = 0; k < VPL; ++k)
                    if (L.chan(k) < Ccp)
                        *reinterpret_cast<uint4*>(cp_dst + L.row * ld_cpd + L.chan(k)) =
                            *reinterpret_cast<const uint4*>(cp_src + L.row * ld_cps + L.chan(k));
            }
        }
        edge_bwd_dst_mask8_body<G, VPL, UD>(vb, Gr, ldg, mask, rowptr_dst, N, H, dA, ldda);
    }
}

// H in {128, 256, 512, 1024, 2048}: G = H/8 capped at 64, VPL = H / (8 G).
// U (neighbour rows in flight per lane group), MI355X, after the predicate-free rewrite (round 4, profiles/probes/edge8_sweep.py and
// _edge8_bwd_sweep.py; regular and Delaunay meshes): forward U = 2 up to H = 1024 (200 704 x 128: 63 us at U = 2, 65 / 68 / 66 at
// 3 / 4 / 6; the Delaunay mesh 71 / 71 / 73 / 97 - a row whose degree is not a multiple of U re-loads its last neighbour; 1 M
// vertices would take U = 6: 316 vs 342 us, not worth the irregular-mesh loss), U = 1 at H = 2048 (41 us vs 44 / 59 at 2 / 3);
// backward gathering role 2 / 2 / 2 / 1 / 1, streaming dA role 6 / 3 / 3 / 2 / 1.
#define STIN_BWD8_US_16 2
#define STIN_BWD8_US_32 2
#define STIN_BWD8_US_64 2
#define STIN_BWD8_US_64X2 1
#define STIN_BWD8_US_64X4 1
#define STIN_FWD8_U_SMALL 2      /* H <= 1024 (one or two 16-byte chunks per lane) */
#define STIN_FWD8_U_BIG 1        /* H = 2048 (4 chunks per lane) */
#define STIN_L8(KERNEL_, G_, V_, U_, grid_, ...) hipLaunchKernelGGL((KERNEL_<G_, V_, U_>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__)
#define STIN_DISPATCH8(H_, KERNEL, U16_, U32_, U64_, U64X2_, U64X4_, ...)                                            \
    do {                                                                                                             \
        if ((H_) == 128) STIN_L8(KERNEL, 16, 1, U16_, grid_rows(N, 16), __VA_ARGS__);                                \
        else if ((H_) == 256) STIN_L8(KERNEL, 32, 1, U32_, grid_rows(N, 32), __VA_ARGS__);                           \
        else if ((H_) == 512) STIN_L8(KERNEL, 64, 1, U64_, grid_rows(N, 64), __VA_ARGS__);                           \
        else if ((H_) == 1024) STIN_L8(KERNEL, 64, 2, U64X2_, grid_rows(N, 64), __VA_ARGS__);                        \
        else STIN_L8(KERNEL, 64, 4, U64X4_, grid_rows(N, 64), __VA_ARGS__);                                          \
    } while (0)

// --------------------------------------------------------------- segment sum / mean
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_segment_sum(const T* __restrict__ src, int64_t lds_,
                                                       const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, int64_t N, int C,
                                                       int mean, T* __restrict__ out, int64_t ldo) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    float4 acc[VPL];
    bool on[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        on[k] = L.chan(k) < C;
        acc[k] = f4zero();
    }
    for (int e = beg; e < end; e += U) {
        float4 v[U][VPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
            const int64_t j = col != nullptr ? (int64_t)col[ee] : (int64_t)ee;
#pragma unroll
            for (int k = 0; k < VPL; ++k) v[u][k] = on[k] ? ld4(src + j * lds_ + L.chan(k)) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float w = (e + u < end) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                acc[k].x += w * v[u][k].x;
                acc[k].y += w * v[u][k].y;
                acc[k].z += w * v[u][k].z;
                acc[k].w += w * v[u][k].w;
            }
        }
    }
    float s = 1.f;
    if (mean) {
        const int deg = end - beg;
        s = (float)(deg > 0 ? deg : 1);
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (on[k]) st4(out + L.row * ldo + L.chan(k), make_float4(acc[k].x / s, acc[k].y / s, acc[k].z / s, acc[k].w / s));
}

// Round 3 form for rows whose chunks divide evenly over the lanes (C / 4 = G * VPL): no per-chunk predicates, and FEWER
// lanes per row than chunks - each lane owns VPL = 2 chunks 16 G bytes apart, so a wave covers twice the rows and every
// load instruction touches twice as many independent rows (memory-level parallelism at the same register cost).  Measured on
// MI355X (profiles/probes/seg_tune.py): the standalone scatter-add (E = 1.2 M random 256-byte rows -> N = 200 k) 90.1 us with
// G = 16 / U = 4 -> 72.6 us with G = 8 / VPL = 2 / U = 2 -> 68.9 us with non-temporal loads on top (0.50 -> 0.66 of the HBM
// peak); the unpool backward of the step 28.5 -> 23.3 us (C = 128), 17.6 -> 15.5 us (C = 256).  NT (non-temporal loads of the
// gathered rows) only pays when the gathered source cannot stay in the 256 MB Infinity Cache anyway: cache-resident sources
// LOSE 25-40 % with it, so the host sets it by the source's size.  Same summation order per row as k_segment_sum
// (sequential over the row's entries): bit-identical results.
template <typename T, int G, int VPL, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void k_segment_sum_x(const T* __restrict__ src, int64_t lds_,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, int64_t N, int C,
                                                         int mean, T* __restrict__ out, int64_t ldo) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    float4 acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = f4zero();
    for (int e = beg; e < end; e += U) {
        float4 v[U][VPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
            const int64_t j = col != nullptr ? (int64_t)col[ee] : (int64_t)ee;
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const T* p = src + j * lds_ + L.chan(k);
                if constexpr (NT && is_f32_type<T>::value) {
                    const float* q = reinterpret_cast<const float*>(p);
                    v[u][k].x = __builtin_nontemporal_load(q);
                    v[u][k].y = __builtin_nontemporal_load(q + 1);
                    v[u][k].z = __builtin_nontemporal_load(q + 2);
                    v[u][k].w = __builtin_nontemporal_load(q + 3);
                } else {
                    v[u][k] = ld4(p);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float w = (e + u < end) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                acc[k].x += w * v[u][k].x;
                acc[k].y += w * v[u][k].y;
                acc[k].z += w * v[u][k].z;
                acc[k].w += w * v[u][k].w;
            }
        }
    }
    float s = 1.f;
    if (mean) {
        const int deg = end - beg;
        s = (float)(deg > 0 ? deg : 1);
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        st4(out + L.row * ldo + L.chan(k), make_float4(acc[k].x / s, acc[k].y / s, acc[k].z / s, acc[k].w / s));
}

// Segment MEAN over a CSR (out[i] = mean of src[col[e]] over the slots e of row i; 0 for an empty row) AND the first stage of the
// column moments of the SOURCE rows it visits: SingleConvMeshNet's scatter_mean(BatchNorm1d(m)) needs the batch statistics of all
// E edge rows m and their mean per target vertex - every edge row is a slot of exactly one target, so one pass over m gives both
// (the separate moments pass re-read the [E, cout] matrix).  Persistent grid: G = C / 4 lanes per row, a block walks the rows in
// passes, fp64 sums of v and v^2 per thread, block fold through LDS in lane order, ONE partial [2][C] per block (the layout
// stin_moments_final_f32 folds).  Per-row summation order = k_segment_sum_x's (sequential over the slots): same means bit for bit.
template <int U>
__global__ __launch_bounds__(BLOCK) void k_segment_mean_stats(const float* __restrict__ src, int64_t lds_, const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col, int64_t N, int c4,
                                                              float* __restrict__ out, int64_t ldo, double* __restrict__ partial) {
    extern __shared__ double sg_sm[];                                  // [2][BLOCK / c4][4 c4]
    const int lg = threadIdx.x % c4, rl = threadIdx.x / c4, nrl = BLOCK / c4, C = 4 * c4;
    const int ch = lg * 4;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t row = (int64_t)blockIdx.x * nrl + rl; row < N; row += (int64_t)gridDim.x * nrl) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = beg; e < end; e += U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ee = min(e + u, end - 1);
                v[u] = ld4(src + (int64_t)col[ee] * lds_ + ch);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = (e + u < end) ? 1.f : 0.f;
                acc.x += w * v[u].x;
                acc.y += w * v[u].y;
                acc.z += w * v[u].z;
                acc.w += w * v[u].w;
                if (e + u < end) {
                    const float q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double d = (double)q[i];
                        s1[i] += d;
                        s2[i] += d * d;
                    }
                }
            }
        }
        const int deg = end - beg;
        const float s = (float)(deg > 0 ? deg : 1);
        st4(out + row * ldo + ch, make_float4(acc.x / s, acc.y / s, acc.z / s, acc.w / s));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sg_sm[(0 * nrl + rl) * C + ch + i] = s1[i];
        sg_sm[(1 * nrl + rl) * C + ch + i] = s2[i];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += BLOCK) {
        const int o = t / C, c = t % C;
        double s = 0.0;
        for (int k = 0; k < nrl; ++k) s += sg_sm[(o * nrl + k) * C + c];
        partial[((int64_t)blockIdx.x * 2 + o) * C + c] = s;
    }
}

// ------------------------------------------------------------------------- max pool
template <typename T, int G, int VPL, int U>
__global__ __launch_bounds__(BLOCK) void k_pool_max_fwd(const T* __restrict__ x, int64_t ldx,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, int64_t N, int C,
                                                        T* __restrict__ out, int64_t ldo,
                                                        int32_t* __restrict__ arg) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int beg = rowptr[L.row], end = rowptr[L.row + 1];
    float4 best[VPL];
    int4 who[VPL];
    bool on[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        on[k] = L.chan(k) < C;
        best[k] = f4zero();
        who[k] = make_int4(-1, -1, -1, -1);
    }
    for (int e = beg; e < end; e += U) {
        float4 v[U][VPL];
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ee = min(e + u, end - 1);
            id[u] = col[ee];
#pragma unroll
            for (int k = 0; k < VPL; ++k) v[u][k] = on[k] ? ld4(x + (int64_t)id[u] * ldx + L.chan(k)) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u < end) {
#pragma unroll
                for (int k = 0; k < VPL; ++k) {
                    // first element initialises; afterwards strict '>' => first maximum wins ties
                    if (who[k].x < 0 || v[u][k].x > best[k].x) { best[k].x = v[u][k].x; who[k].x = id[u]; }
                    if (who[k].y < 0 || v[u][k].y > best[k].y) { best[k].y = v[u][k].y; who[k].y = id[u]; }
                    if (who[k].z < 0 || v[u][k].z > best[k].z) { best[k].z = v[u][k].z; who[k].z = id[u]; }
                    if (who[k].w < 0 || v[u][k].w > best[k].w) { best[k].w = v[u][k].w; who[k].w = id[u]; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (on[k]) {
            st4(out + L.row * ldo + L.chan(k), best[k]);
            *reinterpret_cast<int4*>(arg + L.row * (int64_t)C + L.chan(k)) = who[k];
        }
}

template <typename T, int G, int VPL>
__global__ __launch_bounds__(BLOCK) void k_pool_max_bwd(const T* __restrict__ g, int64_t ldg,
                                                        const int32_t* __restrict__ arg,
                                                        const int32_t* __restrict__ trace, int64_t N, int C,
                                                        T* __restrict__ gx, int64_t ldgx) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int64_t t = trace[L.row];
    const int v = (int)L.row;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (L.chan(k) < C) {
            const float4 gg = ld4(g + t * ldg + L.chan(k));
            const int4 w = *reinterpret_cast<const int4*>(arg + t * (int64_t)C + L.chan(k));
            st4(gx + L.row * ldgx + L.chan(k),
                make_float4(w.x == v ? gg.x : 0.f, w.y == v ? gg.y : 0.f, w.z == v ? gg.z : 0.f, w.w == v ? gg.w : 0.f));
        }
}

template <typename T, int G, int VPL>
__global__ __launch_bounds__(BLOCK) void k_gather_rows(const T* __restrict__ src, int64_t lds_,
                                                       const int32_t* __restrict__ idx,
                                                       const float* __restrict__ row_scale, int64_t N, int C,
                                                       T* __restrict__ out, int64_t ldo) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int64_t t = idx[L.row];
    const float s = row_scale != nullptr ? row_scale[t] : 1.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (L.chan(k) < C) {
            const float4 v = ld4(src + t * lds_ + L.chan(k));
            st4(out + L.row * ldo + L.chan(k), make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
        }
}

// out[e, :] = a[ia[e], :] + b[ib[e], :]: the per-edge pre-activation A_dst + B_src of SingleConvMeshNet's edge MLP
template <typename T, int G, int VPL>
__global__ __launch_bounds__(BLOCK) void k_gather_add_rows(const T* __restrict__ a, int64_t lda, const int32_t* __restrict__ ia,
                                                           const T* __restrict__ b, int64_t ldb, const int32_t* __restrict__ ib,
                                                           int64_t N, int C, T* __restrict__ out, int64_t ldo) {
    Lane<G, VPL> L;
    if (L.row >= N) return;
    const int64_t ta = ia[L.row], tb = ib[L.row];
#pragma unroll
    for (int k = 0; k < VPL; ++k)
        if (L.chan(k) < C) {
            const float4 u = ld4(a + ta * lda + L.chan(k));
            const float4 v = ld4(b + tb * ldb + L.chan(k));
            st4(out + L.row * ldo + L.chan(k), make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
        }
}

// (round 5) U rows per thread: the one-row form above issues index load -> two dependent row gathers -> store per thread with
// nothing else in flight and writes the 614 MB [E, 128] matrix of a level-0 SingleConvMeshNet layer at 2.4 TB/s (254 us);
// here a thread owns one 16-byte column chunk of U rows R apart (R = rows per pass of the grid): 2 U index loads, then 2 U row
// gathers in flight, then U stores.  C / 4 must divide the block.  Same sums: bit-identical.
template <int U>
__global__ __launch_bounds__(BLOCK) void k_gather_add_rows_u(const float* __restrict__ a, int64_t lda, const int32_t* __restrict__ ia,
                                                             const float* __restrict__ b, int64_t ldb, const int32_t* __restrict__ ib,
                                                             int64_t N, int c4, float* __restrict__ out, int64_t ldo) {
    const int64_t gid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t R = (int64_t)gridDim.x * BLOCK / c4;                 // rows per pass
    const int64_t r0 = gid / c4;
    const int ch = (int)(gid % c4) * 4;
    int32_t ta[U], tb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * R;
        const int64_t rc = r < N ? r : N - 1;
        ta[u] = ia[rc];
        tb[u] = ib[rc];
    }
    float4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        x[u] = ld4(a + (int64_t)ta[u] * lda + ch);
        y[u] = ld4(b + (int64_t)tb[u] * ldb + ch);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * R;
        if (r < N) st4(out + r * ldo + ch, make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w));
    }
}

// The same pass WITH the first stage of the column moments of its output (BatchNorm1d over the E edge rows that follows it in
// SingleConvMeshNet: the separate moments pass re-read the 614 MB it had just written).  Persistent grid: a block walks the rows in
// passes of R, every thread keeps fp64 sums of x and x^2 for its four columns, the row lanes of a block are folded through LDS in
// lane order and the block writes ONE partial [2][C] - the [groups][2][C] layout stin_moments_final_f32 folds (groups = gridDim).
// Fixed row -> thread map and fixed fold order: deterministic.  C / 4 divides the block.
template <int U>
__global__ __launch_bounds__(BLOCK) void k_gather_add_rows_stats(const float* __restrict__ a, int64_t lda, const int32_t* __restrict__ ia,
                                                                 const float* __restrict__ b, int64_t ldb, const int32_t* __restrict__ ib,
                                                                 int64_t N, int c4, float* __restrict__ out, int64_t ldo,
                                                                 double* __restrict__ partial) {
    extern __shared__ double ga_sm[];                                  // [2][BLOCK / c4][4 c4]
    const int64_t gid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t R = (int64_t)gridDim.x * BLOCK / c4;                 // rows per pass
    const int ch = (int)(gid % c4) * 4;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t r0 = gid / c4; r0 < N; r0 += U * R) {
        int32_t ta[U], tb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * R;
            const int64_t rc = r < N ? r : N - 1;
            ta[u] = ia[rc];
            tb[u] = ib[rc];
        }
        float4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = ld4(a + (int64_t)ta[u] * lda + ch);
            y[u] = ld4(b + (int64_t)tb[u] * ldb + ch);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * R;
            if (r < N) {
                const float4 v = make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w);
                st4(out + r * ldo + ch, v);
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double d = (double)e[i];
                    s1[i] += d;
                    s2[i] += d * d;
                }
            }
        }
    }
    const int rl = threadIdx.x / c4, nrl = BLOCK / c4, C = 4 * c4;
    const int cc = (threadIdx.x % c4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ga_sm[(0 * nrl + rl) * C + cc + i] = s1[i];
        ga_sm[(1 * nrl + rl) * C + cc + i] = s2[i];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += BLOCK) {
        const int o = t / C, c = t % C;
        double s = 0.0;
        for (int k = 0; k < nrl; ++k) s += ga_sm[(o * nrl + k) * C + c];
        partial[((int64_t)blockIdx.x * 2 + o) * C + c] = s;
    }
}

__global__ void k_gather_add_rows_scalar(const float* __restrict__ a, int64_t lda, const int32_t* __restrict__ ia,
                                         const float* __restrict__ b, int64_t ldb, const int32_t* __restrict__ ib, int64_t N,
                                         int C, float* __restrict__ out, int64_t ldo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * C) return;
    const int64_t r = t / C;
    const int c = (int)(t % C);
    out[r * ldo + c] = a[(int64_t)ia[r] * lda + c] + b[(int64_t)ib[r] * ldb + c];
}

// ------------------------------------------------- scalar fallbacks (C % 4 != 0 / unaligned)
enum ScalarOp { OP_EDGE_FWD, OP_EDGE_BWD_DST, OP_EDGE_BWD_SRC, OP_SEG_SUM, OP_POOL_MAX, OP_POOL_MAX_BWD, OP_GATHER };

template <int OP>
__global__ void k_scalar(const float* __restrict__ p0, int64_t ld0, const float* __restrict__ p1, int64_t ld1,
                         const float* __restrict__ p2, int64_t ld2, const float* __restrict__ vec,
                         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N, int C,
                         int flag, float* __restrict__ out, int64_t ldo, int32_t* __restrict__ iout) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * C) return;
    const int64_t r = t / C;
    const int c = (int)(t % C);
    if (OP == OP_POOL_MAX_BWD) {  // p0 = g, col = trace, iout = arg
        const int64_t k = col[r];
        out[r * ldo + c] = (iout[k * C + c] == (int)r) ? p0[k * ld0 + c] : 0.f;
        return;
    }
    if (OP == OP_GATHER) {  // p0 = src, col = idx, vec = row_scale
        const int64_t k = col[r];
        out[r * ldo + c] = p0[k * ld0 + c] * (vec != nullptr ? vec[k] : 1.f);
        return;
    }
    const int beg = rowptr[r], end = rowptr[r + 1];
    const int deg = end - beg;
    const float s = 1.0f / (float)(deg > 0 ? deg : 1);
    float acc = 0.f;
    if (OP == OP_EDGE_FWD) {  // p0 = A, p1 = B
        const float a = p0[r * ld0 + c];
        for (int e = beg; e < end; ++e) acc += fmaxf(a + p1[(int64_t)col[e] * ld1 + c], 0.f);
        out[r * ldo + c] = acc / (float)(deg > 0 ? deg : 1);
        if (flag && c < 4) out[r * ldo + C + c] = (c == 0 && deg > 0) ? 1.f : 0.f;
    } else if (OP == OP_EDGE_BWD_DST) {  // p0 = A, p1 = B, p2 = G
        const float a = p0[r * ld0 + c];
        for (int e = beg; e < end; ++e) acc += (a + p1[(int64_t)col[e] * ld1 + c] > 0.f) ? 1.f : 0.f;
        out[r * ldo + c] = p2[r * ld2 + c] * s * acc;
    } else if (OP == OP_EDGE_BWD_SRC) {  // p0 = A, p1 = B, p2 = G, vec = inv_deg
        const float b = p1[r * ld1 + c];
        for (int e = beg; e < end; ++e) {
            const int64_t i = col[e];
            acc += (p0[i * ld0 + c] + b > 0.f) ? vec[i] * p2[i * ld2 + c] : 0.f;
        }
        out[r * ldo + c] = acc;
    } else if (OP == OP_SEG_SUM) {  // p0 = src, flag = mean
        for (int e = beg; e < end; ++e) acc += p0[(col != nullptr ? (int64_t)col[e] : (int64_t)e) * ld0 + c];
        out[r * ldo + c] = flag ? acc / (float)(deg > 0 ? deg : 1) : acc;
    } else if (OP == OP_POOL_MAX) {  // p0 = x
        float best = 0.f;
        int who = -1;
        for (int e = beg; e < end; ++e) {
            const int id = col[e];
            const float v = p0[(int64_t)id * ld0 + c];
            if (who < 0 || v > best) { best = v; who = id; }
        }
        out[r * ldo + c] = best;
        iout[r * C + c] = who;
    }
}

__global__ void k_batch_pool(const int64_t* __restrict__ batch, const int32_t* __restrict__ rowptr,
                             const int32_t* __restrict__ col, int64_t N, int64_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    int64_t best = 0;  // torch_scatter: empty segment -> 0
    const int beg = rowptr[r], end = rowptr[r + 1];
    for (int e = beg; e < end; ++e) {
        const int64_t v = batch[col[e]];
        if (e == beg || v > best) best = v;
    }
    out[r] = best;
}

// Row ids of a batched instance norm in one pass: gid[r] = (int32) batch[r]; sid[r] = the number of slice boundaries
// ptr_sum[1 .. B] that are <= r (= torch.searchsorted(ptr_sum[1:], r, right=True): the reference's linspace slice of row r,
// fastinstancenorm.py:53-82).  B + 1 boundaries in LDS, binary search.
__global__ __launch_bounds__(BLOCK) void k_norm_group_ids(const int64_t* __restrict__ batch, const int32_t* __restrict__ ptr_sum, int B,
                                                         int64_t N, int32_t* __restrict__ gid, int32_t* __restrict__ sid) {
    extern __shared__ int32_t ids_ptr[];                                          // ptr_sum[1 .. B]
    if (sid != nullptr) {
        for (int i = threadIdx.x; i < B; i += BLOCK) ids_ptr[i] = ptr_sum[i + 1];
        __syncthreads();
    }
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= N) return;
    gid[r] = (int32_t)batch[r];
    if (sid != nullptr) {
        int lo = 0, hi = B;                                                       // first index with ids_ptr[i] > r
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int64_t)ids_ptr[mid] <= r) lo = mid + 1;
            else hi = mid;
        }
        sid[r] = lo;
    }
}

__global__ void k_gather_i64(const int64_t* __restrict__ src, const int32_t* __restrict__ idx, int64_t N,
                             int64_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < N) out[r] = src[idx[r]];
}

template <typename T>
inline bool vec_ok(int C, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
    if (C % 4 != 0 || C > 64 * 8 * 4) return false;
    for (const void* p : ptrs)
        if (p != nullptr && !stin_aligned_vec4<T>(p)) return false;
    for (int64_t ld : lds)
        if (ld % 4 != 0) return false;
    return true;
}

// full rows of 32 or 64 lanes (every lane live), so a slot is exactly H bits of ballot words
inline bool mask_shape_ok(int H) { return H == 128 || H == 256 || H == 512 || H == 1024 || H == 2048; }

inline unsigned grid_rows(int64_t N, int G) { return (unsigned)((N + (BLOCK / G) - 1) / (BLOCK / G)); }
// Launch shape of a row kernel (see vblock_id): plain 1-D order by default; STIN_XCD_ROWS=1 (re-read per call) selects (8, q) =
// one contiguous row range per XCD.  Measured in round 3 and left OFF: no gain on the randomly numbered benchmark meshes
// (nothing to reuse), and on coherently numbered ones the plain order is the better one (level-0 forward at 200 704 vertices,
// grid order: 82 us plain vs 96 us chunked; 1 M bf16: 344 vs 353) - with round-robin dealing the 8 XCDs sweep the SAME
// neighbourhood together and share its lines in the Infinity Cache, chunked they stream eight distant regions at once.
inline bool xcd_rows_on(unsigned) { return false; }   // (the (8, q) launch shape measured no gain - comment above - and stays off)
inline dim3 rows_grid(unsigned nwg) { return xcd_rows_on(nwg) ? dim3(8, (nwg + 7) / 8) : dim3(nwg); }
inline dim3 pair_grid(unsigned nb) { return xcd_rows_on(nb) ? dim3(8, 2 * ((nb + 7) / 8)) : dim3(2 * nb); }
inline unsigned grid_elems(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

inline bool wide8_ok(int H, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
    if (!mask_shape_ok(H)) return false;
    for (const void* p : ptrs)
        if (p != nullptr && !stin_aligned16(p)) return false;
    for (int64_t ld : lds)
        if (ld % 8 != 0) return false;
    return true;
}


// Dispatch on (G, VPL) for a channel count C (C % 4 == 0, C <= 2048).  U = neighbour rows requested per loop trip,
// tuned on MI355X at mean degree ~6 (level-0/1/2 edge kernels and the standalone scatter-add, U in {8, 6, 4}):
// rows of <= 256 B: 4; 512-B rows (G = 32): 6 - one trip for a typical mesh vertex; 1-KB rows (G = 64): 4; then 2, 2, 1
// as a lane holds 2, 4, 8 chunks.  DIV halves it for kernels that gather two rows per neighbour.
#define STIN_U(base, DIV) (((base) / (DIV)) > 0 ? ((base) / (DIV)) : 1)
#define STIN_DISPATCH(C_, KERNEL, DIV, ...)                                                                  \
    do {                                                                                                     \
        const int c4_ = (C_) / 4;                                                                            \
        const int g_ = stin_group_lanes(c4_);                                                                \
        const int vpl_ = (c4_ + g_ - 1) / g_;                                                                \
        const unsigned grid_ = grid_rows(N, g_);                                                             \
        if (g_ == 1) hipLaunchKernelGGL((KERNEL<T, 1, 1, STIN_U(4, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);        \
        else if (g_ == 2) hipLaunchKernelGGL((KERNEL<T, 2, 1, STIN_U(4, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);   \
        else if (g_ == 4) hipLaunchKernelGGL((KERNEL<T, 4, 1, STIN_U(4, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);   \
        else if (g_ == 8) hipLaunchKernelGGL((KERNEL<T, 8, 1, STIN_U(4, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);   \
        else if (g_ == 16) hipLaunchKernelGGL((KERNEL<T, 16, 1, STIN_U(4, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (g_ == 32) hipLaunchKernelGGL((KERNEL<T, 32, 1, STIN_U(6, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (vpl_ == 1) hipLaunchKernelGGL((KERNEL<T, 64, 1, STIN_U(4, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (vpl_ == 2) hipLaunchKernelGGL((KERNEL<T, 64, 2, STIN_U(2, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (vpl_ <= 4) hipLaunchKernelGGL((KERNEL<T, 64, 4, STIN_U(2, DIV)>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<T, 64, 8, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);       \
    } while (0)

#define STIN_DISPATCH_NOU(C_, KERNEL, ...)                                                                   \
    do {                                                                                                     \
        const int c4_ = (C_) / 4;                                                                            \
        const int g_ = stin_group_lanes(c4_);                                                                \
        const int vpl_ = (c4_ + g_ - 1) / g_;                                                                \
        const unsigned grid_ = grid_rows(N, g_);                                                             \
        if (g_ == 1) hipLaunchKernelGGL((KERNEL<T, 1, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);   \
        else if (g_ == 2) hipLaunchKernelGGL((KERNEL<T, 2, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);   \
        else if (g_ == 4) hipLaunchKernelGGL((KERNEL<T, 4, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);   \
        else if (g_ == 8) hipLaunchKernelGGL((KERNEL<T, 8, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);   \
        else if (g_ == 16) hipLaunchKernelGGL((KERNEL<T, 16, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (g_ == 32) hipLaunchKernelGGL((KERNEL<T, 32, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (vpl_ == 1) hipLaunchKernelGGL((KERNEL<T, 64, 1>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (vpl_ == 2) hipLaunchKernelGGL((KERNEL<T, 64, 2>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else if (vpl_ <= 4) hipLaunchKernelGGL((KERNEL<T, 64, 4>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<T, 64, 8>), rows_grid(grid_), dim3(BLOCK), 0, stream, __VA_ARGS__);          \
    } while (0)

constexpr bool is_f32(const float*) { return true; }
constexpr bool is_f32(const stin_bf16*) { return false; }

// ------------------------------------------------------------- host side, one implementation per element type
// (fp32 rows: every shape, scalar kernels when C % 4 != 0 or rows are not 16-byte aligned;
//  bf16 rows: the 4-channel vector kernels only -> STIN_E_UNSUPPORTED otherwise)
template <typename T>
int edge_fwd_impl(const T* A, int64_t lda, const T* B, int64_t ldb, const int32_t* rowptr, const int32_t* col, int64_t N,
                  int H, T* out, int64_t ldo, int indicator, uint32_t* mask, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && H > 0 && lda >= H && ldb >= H && ldo >= H + (indicator ? 4 : 0), STIN_E_SIZE);
    STIN_REQUIRE(!indicator || H >= 4, STIN_E_UNSUPPORTED);
    const bool vec = vec_ok<T>(H, {A, B, out}, {lda, ldb, ldo});
    STIN_REQUIRE(mask == nullptr || (mask_shape_ok(H) && vec), STIN_E_UNSUPPORTED);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(A && B && rowptr && out, STIN_E_NULL);
    if constexpr (!is_f32((const T*)nullptr)) {
        // bf16 rows: 16 bytes per lane whenever the rows allow it; with a mask this geometry is mandatory (it fixes the
        // mask's bit layout for the backward kernels)
        const bool wide = wide8_ok(H, {A, B, out}, {lda, ldb, ldo});
        STIN_REQUIRE(mask == nullptr || wide, STIN_E_ALIGN);
        if (wide) {
            // rows in flight per lane group: tuning aid STIN_EDGE8_U = "u16,u32,u64,u64x2,u64x4" digits, e.g. 44422 (re-read per call)
            const char* eu = getenv("STIN_EDGE8_U");
            const int cfg = eu ? atoi(eu) : 0;
            auto pick = [&](int pos, int dflt) { int d = cfg; for (int i = 0; i < 4 - pos; ++i) d /= 10; d %= 10; return (cfg > 0 && d > 0) ? d : dflt; };
            const int hsel = H == 128 ? 0 : H == 256 ? 1 : H == 512 ? 2 : H == 1024 ? 3 : 4;
            // (round 6, profiles/probes/edge8_u_sweep.py, random graph of mean degree 6: 1 M x 128 reads 549 / 407 / 416 / 393 / 439 us at
            // U = 1 / 2 / 3 / 4 / 6, 200 704 x 128 82 / 76 / 73 / 76 / 98; on config 5's 6-regular mesh U = 4 measured 383.0 us against
            // 383.7 at U = 2 - nothing: the rate of 256-byte random rows, not the rows in flight, bounds this kernel - U stays 2)
            const int u = pick(hsel, hsel <= 3 ? STIN_FWD8_U_SMALL : STIN_FWD8_U_BIG);
#define STIN_FWD8(U_) STIN_DISPATCH8(H, k_edge_fwd8, U_, U_, U_, U_, U_, A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask)
            if (u == 1) STIN_FWD8(1);
            else if (u == 2) STIN_FWD8(2);
            else if (u == 3) STIN_FWD8(3);
            else if (u == 4) STIN_FWD8(4);
            else STIN_FWD8(6);
#undef STIN_FWD8
            return stin_launch_status();
        }
    }
    // (round 6) mask == NULL at a mask shape - the forward of a no-grad / evaluation pass - runs the SAME kernel without the mask
    // stores (edge_fwd_body tests the pointer): same gathers, same summation order, bit-identical rows, E * H / 8 bytes less written
    if (vec && (mask != nullptr || mask_shape_ok(H))) {   // mask shapes are exact multiples of the lane geometry
        STIN_DISPATCH(H, k_edge_fwd_exact, 1, A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
    } else if (vec) {
        STIN_DISPATCH(H, k_edge_fwd, 1, A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
    } else if constexpr (is_f32((const T*)nullptr)) {
        hipLaunchKernelGGL((k_scalar<OP_EDGE_FWD>), dim3(grid_elems(N * H)), dim3(BLOCK), 0, stream, A, lda, B, ldb,
                           (const float*)nullptr, (int64_t)0, (const float*)nullptr, rowptr, col, N, H, indicator, out, ldo,
                           (int32_t*)nullptr);
    } else {
        return STIN_E_UNSUPPORTED;
    }
    return stin_launch_status();
}

template <typename T>
int edge_bwd_dst_mask_impl(const T* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr, int64_t N, int H, T* dA,
                           int64_t ldda, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && H > 0 && ldg >= H && ldda >= H, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(G && mask && rowptr && dA, STIN_E_NULL);
    STIN_REQUIRE(mask_shape_ok(H) && vec_ok<T>(H, {G, dA}, {ldg, ldda}), STIN_E_UNSUPPORTED);
    if constexpr (!is_f32((const T*)nullptr)) {
        STIN_REQUIRE(wide8_ok(H, {G, dA}, {ldg, ldda}), STIN_E_ALIGN);       // the geometry the forward kernel wrote the mask in
        STIN_DISPATCH8(H, k_edge_bwd_dst_mask8, 6, 3, 3, 2, 1, G, ldg, mask, rowptr, N, H, dA, ldda);
        return stin_launch_status();
    }
    STIN_DISPATCH(H, k_edge_bwd_dst_mask, 1, G, ldg, mask, rowptr, N, H, dA, ldda);
    return stin_launch_status();
}

template <typename T>
int edge_bwd_src_mask_impl(const T* G, int64_t ldg, const float* w_src, const uint32_t* mask, const int32_t* rowptr_src,
                           const int32_t* col_src, const int32_t* xslot, int64_t N, int H, T* dB, int64_t lddb,
                           hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && H > 0 && ldg >= H && lddb >= H, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(G && w_src && mask && rowptr_src && col_src && xslot && dB, STIN_E_NULL);
    STIN_REQUIRE(mask_shape_ok(H) && vec_ok<T>(H, {G, dB}, {ldg, lddb}), STIN_E_UNSUPPORTED);
    if constexpr (!is_f32((const T*)nullptr)) {
        STIN_REQUIRE(wide8_ok(H, {G, dB}, {ldg, lddb}), STIN_E_ALIGN);
        STIN_DISPATCH8(H, k_edge_bwd_src_mask8, 2, 2, 2, 1, 1, G, ldg, w_src, mask, rowptr_src, col_src, xslot, N, H, dB, lddb);
        return stin_launch_status();
    }
    STIN_DISPATCH(H, k_edge_bwd_src_mask, 2, G, ldg, w_src, mask, rowptr_src, col_src, xslot, N, H, dB, lddb);
    return stin_launch_status();
}

// fp32 rows only (the block backward's path); unroll factors per (G, VPL) are STIN_DISPATCH's with DIV 1 (dA) and 2 (dB)
int edge_bwd_mask_pair_impl(const float* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst, const float* w_src,
                            const int32_t* rowptr_src, const int32_t* col_src, const int32_t* xslot, int64_t N, int H,
                            float* dA, int64_t ldda, float* dB, int64_t lddb, const float* cp_src, int64_t ld_cps, float* cp_dst,
                            int64_t ld_cpd, int Ccp, hipStream_t stream) {
    using T = float;
    STIN_REQUIRE(N >= 0 && H > 0 && ldg >= H && ldda >= H && lddb >= H, STIN_E_SIZE);
    if (cp_src != nullptr) {
        STIN_REQUIRE(cp_dst != nullptr && Ccp > 0 && Ccp <= H && Ccp % 4 == 0 && ld_cps >= Ccp && ld_cpd >= Ccp, STIN_E_SIZE);
        STIN_REQUIRE(stin_aligned16(cp_src) && stin_aligned16(cp_dst) && ld_cps % 4 == 0 && ld_cpd % 4 == 0, STIN_E_ALIGN);
    }
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(G && mask && rowptr_dst && w_src && rowptr_src && col_src && xslot && dA && dB, STIN_E_NULL);
    STIN_REQUIRE(mask_shape_ok(H) && vec_ok<T>(H, {G, dA, dB}, {ldg, ldda, lddb}), STIN_E_UNSUPPORTED);
    const int c4 = H / 4, g = stin_group_lanes(c4), vpl = (c4 + g - 1) / g;
    const unsigned nb = grid_rows(N, g);
#define STIN_PAIR(G_, VPL_, BASE_)                                                                                       \
    STIN_LAUNCH_STOP((k_edge_bwd_mask_pair<T, G_, VPL_, STIN_U(BASE_, 1), STIN_U(BASE_, 2)>), pair_grid(nb), dim3(BLOCK),      \
                     stream, G, ldg, mask, rowptr_dst, w_src, rowptr_src, col_src, xslot, N, H, dA, ldda, dB, lddb, nb, cp_src,       \
                     ld_cps, cp_dst, ld_cpd, Ccp)
    if (g == 32) STIN_PAIR(32, 1, 6);                 // H = 128
    else if (vpl == 1) STIN_PAIR(64, 1, 4);           // 256
    else if (vpl == 2) {                              // 512: two rows in flight in the gathering role (18 063 x 512: 56.8 us at 1, 54.0 at 2, 56.1 / 58.8 at 3 / 4)
        STIN_LAUNCH_STOP((k_edge_bwd_mask_pair<T, 64, 2, 2, 2>), pair_grid(nb), dim3(BLOCK), stream, G, ldg, mask, rowptr_dst, w_src, rowptr_src, col_src, xslot, N, H, dA, ldda, dB, lddb, nb, cp_src, ld_cps, cp_dst, ld_cpd, Ccp);
    }
    else if (vpl <= 4) STIN_PAIR(64, 4, 2);           // 1024
    else STIN_PAIR(64, 8, 1);                         // 2048
#undef STIN_PAIR
    return stin_launch_status();
}

// translation-invariant compact layout, fp32 rows (see k_edge_bwd_mask_ti): D [N, H] = dB - dA, colsum [ti_colsum_rows(N, H)][H]
inline int64_t ti_colsum_rows(int64_t N, int H) {
    const int c4 = H / 4, g = stin_group_lanes(c4);
    const int64_t nb = (N + (BLOCK / g) - 1) / (BLOCK / g);
    return (nb + TI_ITER - 1) / TI_ITER;
}
int edge_bwd_mask_ti_impl(const float* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst, const float* w_src,
                          const int32_t* rowptr_src, const int32_t* col_src, const int32_t* xslot, int64_t N, int H, float* D,
                          int64_t ldd, const float* cp_src, int64_t ld_cps, float* cp_dst, int64_t ld_cpd, int Ccp, float* colsum,
                          int64_t colsum_rows, hipStream_t stream) {
    using T = float;
    STIN_REQUIRE(N >= 0 && H > 0 && ldg >= H && ldd >= H, STIN_E_SIZE);
    if (cp_src != nullptr) {
        STIN_REQUIRE(cp_dst != nullptr && Ccp > 0 && Ccp <= H && Ccp % 4 == 0 && ld_cps >= Ccp && ld_cpd >= Ccp, STIN_E_SIZE);
        STIN_REQUIRE(stin_aligned16(cp_src) && stin_aligned16(cp_dst) && ld_cps % 4 == 0 && ld_cpd % 4 == 0, STIN_E_ALIGN);
    }
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(G && mask && rowptr_dst && w_src && rowptr_src && col_src && xslot && D && colsum, STIN_E_NULL);
    STIN_REQUIRE(mask_shape_ok(H) && vec_ok<T>(H, {G, D}, {ldg, ldd}) && stin_aligned16(colsum), STIN_E_UNSUPPORTED);
    const int64_t nblk = ti_colsum_rows(N, H);
    STIN_REQUIRE(colsum_rows >= nblk, STIN_E_WORKSPACE);
    const int c4 = H / 4, g = stin_group_lanes(c4), vpl = (c4 + g - 1) / g;
    // rows in flight per role: the pair kernel's choices (UD = STIN_U(base, 1), US = STIN_U(base, 2))
#define STIN_TI(G_, VPL_, UD_, US_)                                                                                      \
    STIN_LAUNCH_STOP((k_edge_bwd_mask_ti<T, G_, VPL_, UD_, US_>), dim3((unsigned)nblk), dim3(BLOCK), stream, G, ldg, mask,  \
                     rowptr_dst, w_src, rowptr_src, col_src, xslot, N, H, D, ldd, cp_src, ld_cps, cp_dst, ld_cpd, Ccp, colsum)
    if (g == 32) STIN_TI(32, 1, STIN_U(6, 1), STIN_U(6, 2));          // H = 128
    else if (vpl == 1) STIN_TI(64, 1, STIN_U(4, 1), STIN_U(4, 2));    // 256
    else if (vpl == 2) STIN_TI(64, 2, 2, 2);                          // 512
    else if (vpl <= 4) STIN_TI(64, 4, STIN_U(2, 1), STIN_U(2, 2));    // 1024
    else STIN_TI(64, 8, 1, 1);                                        // 2048
#undef STIN_TI
    return stin_launch_status();
}

int edge_fwd_ti_impl(const float* b1, const float* B, int64_t ldb, const int32_t* rowptr, const int32_t* col, int64_t N, int H,
                     float* out, int64_t ldo, int indicator, uint32_t* mask, hipStream_t stream) {
    using T = float;
    STIN_REQUIRE(N >= 0 && H > 0 && ldb >= H && ldo >= H + (indicator ? 4 : 0), STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(B && rowptr && out, STIN_E_NULL);
    STIN_REQUIRE(mask_shape_ok(H) && vec_ok<T>(H, {B, out}, {ldb, ldo}) && (b1 == nullptr || stin_aligned16(b1)), STIN_E_UNSUPPORTED);
    const float* A = b1;
    const int64_t lda = 0;
    STIN_DISPATCH(H, k_edge_fwd_ti, 1, A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask);
    return stin_launch_status();
}

int edge_bwd_mask_pair8_impl(const stin_bf16* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst,
                             const float* w_src, const int32_t* rowptr_src, const int32_t* col_src, const int32_t* xslot,
                             int64_t N, int H, stin_bf16* dA, int64_t ldda, stin_bf16* dB, int64_t lddb, const stin_bf16* cp_src,
                             int64_t ld_cps, stin_bf16* cp_dst, int64_t ld_cpd, int Ccp, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && H > 0 && ldg >= H && ldda >= H && lddb >= H, STIN_E_SIZE);
    if (cp_src != nullptr) {
        STIN_REQUIRE(cp_dst != nullptr && Ccp > 0 && Ccp <= H && Ccp % 8 == 0 && ld_cps >= Ccp && ld_cpd >= Ccp, STIN_E_SIZE);
        STIN_REQUIRE(stin_aligned16(cp_src) && stin_aligned16(cp_dst) && ld_cps % 8 == 0 && ld_cpd % 8 == 0, STIN_E_ALIGN);
    }
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(G && mask && rowptr_dst && w_src && rowptr_src && col_src && xslot && dA && dB, STIN_E_NULL);
    STIN_REQUIRE(mask_shape_ok(H), STIN_E_UNSUPPORTED);
    STIN_REQUIRE(wide8_ok(H, {G, dA, dB}, {ldg, ldda, lddb}), STIN_E_ALIGN);    // the geometry the forward kernel wrote the mask in
#define STIN_PAIR8(G_, V_, UD_, US_)                                                                                          \
    do {                                                                                                                      \
        const unsigned nb = grid_rows(N, G_);                                                                                 \
        STIN_LAUNCH_STOP((k_edge_bwd_mask_pair8<G_, V_, UD_, US_>), pair_grid(nb), dim3(BLOCK), stream, G, ldg, mask,          \
                           rowptr_dst, w_src, rowptr_src, col_src, xslot, N, H, dA, ldda, dB, lddb, nb, cp_src, ld_cps,        \
                           cp_dst, ld_cpd, Ccp);                                                                               \
    } while (0)
    // (UD, US): rows in flight of the streaming / the gathering role.  Tuning aid STIN_EDGE8_US = digits "u16,u32,u64,u64x2,u64x4"
    // for the gathering role, as STIN_EDGE8_U for the forward kernel (re-read per call).
    const char* eu = getenv("STIN_EDGE8_US");
    const int cfg = eu ? atoi(eu) : 0;
    auto pick = [&](int pos, int dflt) { int d = cfg; for (int i = 0; i < 4 - pos; ++i) d /= 10; d %= 10; return (cfg > 0 && d > 0) ? d : dflt; };
#define STIN_PAIR8_U(G_, V_, UD_, POS_, DFLT_)                  \
    do {                                                        \
        const int us_ = pick(POS_, DFLT_);                      \
        if (us_ == 1) STIN_PAIR8(G_, V_, UD_, 1);               \
        else if (us_ == 2) STIN_PAIR8(G_, V_, UD_, 2);          \
        else if (us_ == 3) STIN_PAIR8(G_, V_, UD_, 3);          \
        else if (us_ == 4) STIN_PAIR8(G_, V_, UD_, 4);          \
        else STIN_PAIR8(G_, V_, UD_, 6);                        \
    } while (0)
    if (H == 128) STIN_PAIR8_U(16, 1, 6, 0, STIN_BWD8_US_16);
    else if (H == 256) STIN_PAIR8_U(32, 1, 3, 1, STIN_BWD8_US_32);
    else if (H == 512) STIN_PAIR8_U(64, 1, 3, 2, STIN_BWD8_US_64);
    else if (H == 1024) STIN_PAIR8_U(64, 2, 2, 3, STIN_BWD8_US_64X2);
    else STIN_PAIR8_U(64, 4, 1, 4, STIN_BWD8_US_64X4);
#undef STIN_PAIR8_U
#undef STIN_PAIR8
    return stin_launch_status();
}

template <typename T>
int segment_sum_impl(const T* src, int64_t ld_src, const int32_t* rowptr, const int32_t* col, int64_t N, int C, int mean,
                     T* out, int64_t ld_out, hipStream_t stream) {
    const bool want_nt = (mean & STIN_SEG_NONTEMPORAL) != 0;
    mean &= 1;
    STIN_REQUIRE(N >= 0 && C > 0 && ld_src >= C && ld_out >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(src && rowptr && out, STIN_E_NULL);
    {
        // even split of the row over G = C / 8 lanes with 2 chunks each (C / 4 a power of two, 16 <= C <= 512), wider rows on a
        // full wave; anything else (ragged channel counts) keeps the predicated kernel below
        const int c4 = C / 4;
        if (C % 4 == 0 && c4 >= 4 && (c4 & (c4 - 1)) == 0 && c4 <= 512 && vec_ok<T>(C, {src, out}, {ld_src, ld_out})) {
            // non-temporal loads only for a source that cannot be Infinity-Cache resident (see the kernel comment)
            const bool nt = is_f32((const T*)nullptr) && want_nt;
#define SEGX(G_, V_, U_)                                                                                                  \
    do {                                                                                                                  \
        if (nt) hipLaunchKernelGGL((k_segment_sum_x<T, G_, V_, U_, true>), rows_grid(grid_rows(N, G_)), dim3(BLOCK), 0, stream, src, ld_src, rowptr, col, N, C, mean, out, ld_out); \
        else hipLaunchKernelGGL((k_segment_sum_x<T, G_, V_, U_, false>), rows_grid(grid_rows(N, G_)), dim3(BLOCK), 0, stream, src, ld_src, rowptr, col, N, C, mean, out, ld_out);   \
    } while (0)
            if (c4 == 4) SEGX(2, 2, 2);
            else if (c4 == 8) SEGX(4, 2, 2);
            else if (c4 == 16) SEGX(8, 2, 2);
            else if (c4 == 32) SEGX(16, 2, 2);
            else if (c4 == 64) SEGX(32, 2, 2);
            else if (c4 == 128) SEGX(64, 2, 2);
            else if (c4 == 256) SEGX(64, 4, 2);
            else SEGX(64, 8, 1);
#undef SEGX
            return stin_launch_status();
        }
    }
    if (vec_ok<T>(C, {src, out}, {ld_src, ld_out})) {
        STIN_DISPATCH(C, k_segment_sum, 1, src, ld_src, rowptr, col, N, C, mean, out, ld_out);
    } else if constexpr (is_f32((const T*)nullptr)) {
        hipLaunchKernelGGL((k_scalar<OP_SEG_SUM>), dim3(grid_elems(N * C)), dim3(BLOCK), 0, stream, src, ld_src,
                           (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr,
                           rowptr, col, N, C, mean, out, ld_out, (int32_t*)nullptr);
    } else {
        return STIN_E_UNSUPPORTED;
    }
    return stin_launch_status();
}

template <typename T>
int pool_max_fwd_impl(const T* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, int64_t N, int C, T* out,
                      int64_t ldo, int32_t* arg, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && C > 0 && ldx >= C && ldo >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && rowptr && col && out && arg, STIN_E_NULL);
    if (vec_ok<T>(C, {x, out}, {ldx, ldo}) && stin_aligned16(arg)) {
        STIN_DISPATCH(C, k_pool_max_fwd, 1, x, ldx, rowptr, col, N, C, out, ldo, arg);
    } else if constexpr (is_f32((const T*)nullptr)) {
        hipLaunchKernelGGL((k_scalar<OP_POOL_MAX>), dim3(grid_elems(N * C)), dim3(BLOCK), 0, stream, x, ldx,
                           (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr,
                           rowptr, col, N, C, 0, out, ldo, arg);
    } else {
        return STIN_E_UNSUPPORTED;
    }
    return stin_launch_status();
}

template <typename T>
int pool_max_bwd_impl(const T* g, int64_t ldg, const int32_t* arg, const int32_t* trace, int64_t N, int C, T* gx,
                      int64_t ldgx, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && C > 0 && ldg >= C && ldgx >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(g && arg && trace && gx, STIN_E_NULL);
    if (vec_ok<T>(C, {g, gx}, {ldg, ldgx}) && stin_aligned16(arg)) {
        STIN_DISPATCH_NOU(C, k_pool_max_bwd, g, ldg, arg, trace, N, C, gx, ldgx);
    } else if constexpr (is_f32((const T*)nullptr)) {
        hipLaunchKernelGGL((k_scalar<OP_POOL_MAX_BWD>), dim3(grid_elems(N * C)), dim3(BLOCK), 0, stream, g, ldg,
                           (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr,
                           (const int32_t*)nullptr, trace, N, C, 0, gx, ldgx, const_cast<int32_t*>(arg));
    } else {
        return STIN_E_UNSUPPORTED;
    }
    return stin_launch_status();
}

template <typename T>
int gather_rows_impl(const T* src, int64_t ld_src, const int32_t* idx, const float* row_scale, int64_t N, int C, T* out,
                     int64_t ldo, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && C > 0 && ld_src >= C && ldo >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(src && idx && out, STIN_E_NULL);
    if (vec_ok<T>(C, {src, out}, {ld_src, ldo})) {
        STIN_DISPATCH_NOU(C, k_gather_rows, src, ld_src, idx, row_scale, N, C, out, ldo);
    } else if constexpr (is_f32((const T*)nullptr)) {
        hipLaunchKernelGGL((k_scalar<OP_GATHER>), dim3(grid_elems(N * C)), dim3(BLOCK), 0, stream, src, ld_src,
                           (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, row_scale,
                           (const int32_t*)nullptr, idx, N, C, 0, out, ldo, (int32_t*)nullptr);
    } else {
        return STIN_E_UNSUPPORTED;
    }
    return stin_launch_status();
}

inline const stin_bf16* b16(const stin_bf16_t* p) { return reinterpret_cast<const stin_bf16*>(p); }
inline stin_bf16* b16(stin_bf16_t* p) { return reinterpret_cast<stin_bf16*>(p); }

}  // namespace

extern "C" int stin_edge_relu_mean_fwd_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                           const int32_t* rowptr, const int32_t* col, int64_t N, int H, float* out,
                                           int64_t ldo, int indicator, uint32_t* mask, stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_fwd_impl<float>(A, lda, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask, (hipStream_t)stream);
}
extern "C" int stin_edge_relu_mean_fwd_bf16(const stin_bf16_t* A, int64_t lda, const stin_bf16_t* B, int64_t ldb,
                                            const int32_t* rowptr, const int32_t* col, int64_t N, int H,
                                            stin_bf16_t* out, int64_t ldo, int indicator, uint32_t* mask,
                                            stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_fwd_impl<stin_bf16>(b16(A), lda, b16(B), ldb, rowptr, col, N, H, b16(out), ldo, indicator, mask,
                                    (hipStream_t)stream);
}

extern "C" int stin_edge_relu_mean_fwd_ti_f32(const float* b1, const float* B, int64_t ldb, const int32_t* rowptr, const int32_t* col,
                                              int64_t N, int H, float* out, int64_t ldo, int indicator, uint32_t* mask,
                                              stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_fwd_ti_impl(b1, B, ldb, rowptr, col, N, H, out, ldo, indicator, mask, (hipStream_t)stream);
}
extern "C" int64_t stin_edge_bwd_ti_colsum_rows(int64_t N, int H) {
    if (N <= 0 || H <= 0 || !mask_shape_ok(H)) return 0;
    return ti_colsum_rows(N, H);
}
extern "C" int stin_edge_relu_mean_bwd_mask_ti_f32(const float* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst,
                                                   const float* w_src, const int32_t* rowptr_src, const int32_t* col_src,
                                                   const int32_t* xslot, int64_t N, int H, float* D, int64_t ldd,
                                                   const float* cp_src, int64_t ld_cps, float* cp_dst, int64_t ld_cpd, int Ccp,
                                                   float* colsum, int64_t colsum_rows, stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_bwd_mask_ti_impl(G, ldg, mask, rowptr_dst, w_src, rowptr_src, col_src, xslot, N, H, D, ldd, cp_src, ld_cps, cp_dst,
                                 ld_cpd, Ccp, colsum, colsum_rows, (hipStream_t)stream);
}

extern "C" int stin_edge_relu_mean_bwd_dst_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* G,
                                               int64_t ldg, const int32_t* rowptr, const int32_t* col, int64_t N, int H,
                                               float* dA, int64_t ldda, stin_stream_t stream_) {
    typedef float T;
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(N >= 0 && H > 0 && lda >= H && ldb >= H && ldg >= H && ldda >= H, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(A && B && G && rowptr && dA, STIN_E_NULL);
    if (vec_ok<T>(H, {A, B, G, dA}, {lda, ldb, ldg, ldda})) {
        STIN_DISPATCH(H, k_edge_bwd_dst, 1, A, lda, B, ldb, G, ldg, rowptr, col, N, H, dA, ldda);
    } else {
        hipLaunchKernelGGL((k_scalar<OP_EDGE_BWD_DST>), dim3(grid_elems(N * H)), dim3(BLOCK), 0, stream, A, lda, B, ldb,
                           G, ldg, (const float*)nullptr, rowptr, col, N, H, 0, dA, ldda, (int32_t*)nullptr);
    }
    return stin_launch_status();
}

extern "C" int stin_edge_relu_mean_bwd_src_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* G,
                                               int64_t ldg, const float* inv_deg, const int32_t* rowptr_src,
                                               const int32_t* col_src, int64_t N, int H, float* dB, int64_t lddb,
                                               stin_stream_t stream_) {
    typedef float T;
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(N >= 0 && H > 0 && lda >= H && ldb >= H && ldg >= H && lddb >= H, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(A && B && G && inv_deg && rowptr_src && dB, STIN_E_NULL);
    if (vec_ok<T>(H, {A, B, G, dB}, {lda, ldb, ldg, lddb})) {
        STIN_DISPATCH(H, k_edge_bwd_src, 2, A, lda, B, ldb, G, ldg, inv_deg, rowptr_src, col_src, N, H, dB, lddb);
    } else {
        hipLaunchKernelGGL((k_scalar<OP_EDGE_BWD_SRC>), dim3(grid_elems(N * H)), dim3(BLOCK), 0, stream, A, lda, B, ldb,
                           G, ldg, inv_deg, rowptr_src, col_src, N, H, 0, dB, lddb, (int32_t*)nullptr);
    }
    return stin_launch_status();
}

extern "C" int stin_edge_relu_mean_bwd_dst_mask_f32(const float* G, int64_t ldg, const uint32_t* mask,
                                                    const int32_t* rowptr, int64_t N, int H, float* dA, int64_t ldda,
                                                    stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_bwd_dst_mask_impl<float>(G, ldg, mask, rowptr, N, H, dA, ldda, (hipStream_t)stream);
}
extern "C" int stin_edge_relu_mean_bwd_dst_mask_bf16(const stin_bf16_t* G, int64_t ldg, const uint32_t* mask,
                                                     const int32_t* rowptr, int64_t N, int H, stin_bf16_t* dA,
                                                     int64_t ldda, stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_bwd_dst_mask_impl<stin_bf16>(b16(G), ldg, mask, rowptr, N, H, b16(dA), ldda, (hipStream_t)stream);
}

extern "C" int stin_edge_relu_mean_bwd_src_mask_f32(const float* G, int64_t ldg, const float* w_src,
                                                    const uint32_t* mask, const int32_t* rowptr_src,
                                                    const int32_t* col_src, const int32_t* xslot, int64_t N, int H,
                                                    float* dB, int64_t lddb, stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_bwd_src_mask_impl<float>(G, ldg, w_src, mask, rowptr_src, col_src, xslot, N, H, dB, lddb,
                                         (hipStream_t)stream);
}
extern "C" int stin_edge_relu_mean_bwd_src_mask_bf16(const stin_bf16_t* G, int64_t ldg, const float* w_src,
                                                     const uint32_t* mask, const int32_t* rowptr_src,
                                                     const int32_t* col_src, const int32_t* xslot, int64_t N, int H,
                                                     stin_bf16_t* dB, int64_t lddb, stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_bwd_src_mask_impl<stin_bf16>(b16(G), ldg, w_src, mask, rowptr_src, col_src, xslot, N, H, b16(dB), lddb,
                                             (hipStream_t)stream);
}

extern "C" int stin_edge_relu_mean_bwd_mask_f32(const float* G, int64_t ldg, const uint32_t* mask, const int32_t* rowptr_dst,
                                               const float* w_src, const int32_t* rowptr_src, const int32_t* col_src,
                                               const int32_t* xslot, int64_t N, int H, float* dA, int64_t ldda, float* dB,
                                               int64_t lddb, const float* copy_src, int64_t ld_copy_src, float* copy_dst,
                                               int64_t ld_copy_dst, int C_copy, stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_bwd_mask_pair_impl(G, ldg, mask, rowptr_dst, w_src, rowptr_src, col_src, xslot, N, H, dA, ldda, dB, lddb,
                                   copy_src, ld_copy_src, copy_dst, ld_copy_dst, C_copy, (hipStream_t)stream);
}

extern "C" int stin_edge_relu_mean_bwd_mask_bf16(const stin_bf16_t* G, int64_t ldg, const uint32_t* mask,
                                                const int32_t* rowptr_dst, const float* w_src, const int32_t* rowptr_src,
                                                const int32_t* col_src, const int32_t* xslot, int64_t N, int H,
                                                stin_bf16_t* dA, int64_t ldda, stin_bf16_t* dB, int64_t lddb,
                                                const stin_bf16_t* copy_src, int64_t ld_copy_src, stin_bf16_t* copy_dst,
                                                int64_t ld_copy_dst, int C_copy, stin_stream_t stream) {
    stin_clear_stale_error();
    return edge_bwd_mask_pair8_impl(b16(G), ldg, mask, rowptr_dst, w_src, rowptr_src, col_src, xslot, N, H, b16(dA), ldda,
                                    b16(dB), lddb, b16(copy_src), ld_copy_src, b16(copy_dst), ld_copy_dst, C_copy,
                                    (hipStream_t)stream);
}

extern "C" int stin_segment_sum_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* col,
                                    int64_t N, int C, int mean, float* out, int64_t ld_out, stin_stream_t stream) {
    stin_clear_stale_error();
    return segment_sum_impl<float>(src, ld_src, rowptr, col, N, C, mean, out, ld_out, (hipStream_t)stream);
}
extern "C" int stin_segment_sum_bf16(const stin_bf16_t* src, int64_t ld_src, const int32_t* rowptr, const int32_t* col,
                                     int64_t N, int C, int mean, stin_bf16_t* out, int64_t ld_out, stin_stream_t stream) {
    stin_clear_stale_error();
    return segment_sum_impl<stin_bf16>(b16(src), ld_src, rowptr, col, N, C, mean, b16(out), ld_out, (hipStream_t)stream);
}

extern "C" int stin_pool_max_fwd_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                                     int64_t N, int C, float* out, int64_t ldo, int32_t* arg, stin_stream_t stream) {
    stin_clear_stale_error();
    return pool_max_fwd_impl<float>(x, ldx, rowptr, col, N, C, out, ldo, arg, (hipStream_t)stream);
}
extern "C" int stin_pool_max_fwd_bf16(const stin_bf16_t* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                                      int64_t N, int C, stin_bf16_t* out, int64_t ldo, int32_t* arg,
                                      stin_stream_t stream) {
    stin_clear_stale_error();
    return pool_max_fwd_impl<stin_bf16>(b16(x), ldx, rowptr, col, N, C, b16(out), ldo, arg, (hipStream_t)stream);
}

extern "C" int stin_pool_max_bwd_f32(const float* g, int64_t ldg, const int32_t* arg, const int32_t* trace,
                                     int64_t N, int C, float* gx, int64_t ldgx, stin_stream_t stream) {
    stin_clear_stale_error();
    return pool_max_bwd_impl<float>(g, ldg, arg, trace, N, C, gx, ldgx, (hipStream_t)stream);
}
extern "C" int stin_pool_max_bwd_bf16(const stin_bf16_t* g, int64_t ldg, const int32_t* arg, const int32_t* trace,
                                      int64_t N, int C, stin_bf16_t* gx, int64_t ldgx, stin_stream_t stream) {
    stin_clear_stale_error();
    return pool_max_bwd_impl<stin_bf16>(b16(g), ldg, arg, trace, N, C, b16(gx), ldgx, (hipStream_t)stream);
}

extern "C" int stin_gather_add_rows_f32(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb,
                                        const int32_t* idx_b, int64_t N, int C, float* out, int64_t ldo,
                                        stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    using T = float;
    STIN_REQUIRE(N >= 0 && C > 0 && lda >= C && ldb >= C && ldo >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(a && b && idx_a && idx_b && out, STIN_E_NULL);
    if (vec_ok<T>(C, {a, b, out}, {lda, ldb, ldo}) && BLOCK % (C / 4) == 0 && N >= 4096) {
        const int c4 = C / 4;
        const int64_t threads = (N + 3) / 4 * c4;                                      // U = 4 rows per thread
        const unsigned grid = (unsigned)((threads + BLOCK - 1) / BLOCK);
        hipLaunchKernelGGL((k_gather_add_rows_u<4>), dim3(grid), dim3(BLOCK), 0, stream, a, lda, idx_a, b, ldb, idx_b, N, c4, out, ldo);
    } else if (vec_ok<T>(C, {a, b, out}, {lda, ldb, ldo})) {
        STIN_DISPATCH_NOU(C, k_gather_add_rows, a, lda, idx_a, b, ldb, idx_b, N, C, out, ldo);
    } else {
        hipLaunchKernelGGL(k_gather_add_rows_scalar, dim3(grid_elems(N * C)), dim3(BLOCK), 0, stream, a, lda, idx_a, b, ldb,
                           idx_b, N, C, out, ldo);
    }
    return stin_launch_status();
}

// out[e] = a[ia[e]] + b[ib[e]] AND partial [groups][2][C] doubles = per-block column sums of the output and of its squares
// (second stage: stin_moments_final_f32).  groups = stin_gather_add_rows_stats_groups(N, C) (0: shape not supported - the caller
// runs stin_gather_add_rows_f32 + stin_colreduce_f32(MOMENTS)).
extern "C" int64_t stin_gather_add_rows_stats_groups(int64_t N, int C) {
    if (N < 4096 || C <= 0 || C % 4 != 0 || BLOCK % (C / 4) != 0 || C > 1024) return 0;
    const int64_t need = ((N + 3) / 4 * (C / 4) + BLOCK - 1) / BLOCK;          // blocks of the one-pass form (4 rows per thread)
    return need < 1024 ? need : 1024;
}
extern "C" int stin_gather_add_rows_stats_f32(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb,
                                              const int32_t* idx_b, int64_t N, int C, float* out, int64_t ldo, double* partial,
                                              size_t partial_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t groups = stin_gather_add_rows_stats_groups(N, C);
    STIN_REQUIRE(groups > 0, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(lda >= C && ldb >= C && ldo >= C, STIN_E_SIZE);
    STIN_REQUIRE(a && b && idx_a && idx_b && out && partial, STIN_E_NULL);
    STIN_REQUIRE(partial_bytes >= (size_t)groups * 2 * (size_t)C * sizeof(double), STIN_E_WORKSPACE);
    if (!vec_ok<float>(C, {a, b, out}, {lda, ldb, ldo})) return STIN_E_ALIGN;
    const int c4 = C / 4;
    const size_t lds = (size_t)2 * (BLOCK / c4) * C * sizeof(double);
    hipLaunchKernelGGL((k_gather_add_rows_stats<4>), dim3((unsigned)groups), dim3(BLOCK), lds, stream, a, lda, idx_a, b, ldb, idx_b, N, c4,
                       out, ldo, partial);
    return stin_launch_status();
}

// out [N, C] = segment mean of src rows over the CSR (rowptr, col) AND partial [groups][2][C] doubles = per-block column sums of
// the visited source rows and of their squares (second stage: stin_moments_final_f32 with inv_cnt = 1 / #slots).
extern "C" int64_t stin_segment_mean_stats_groups(int64_t N, int C) {
    if (N < 1024 || C <= 0 || C % 4 != 0 || C > 1024 || BLOCK % (C / 4) != 0) return 0;
    const int64_t need = (N + BLOCK / (C / 4) - 1) / (BLOCK / (C / 4));
    return need < 2048 ? need : 2048;
}
extern "C" int stin_segment_mean_stats_f32(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* col, int64_t N, int C,
                                           float* out, int64_t ldo, double* partial, size_t partial_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    const int64_t groups = stin_segment_mean_stats_groups(N, C);
    STIN_REQUIRE(groups > 0, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(ld_src >= C && ldo >= C, STIN_E_SIZE);
    STIN_REQUIRE(src && rowptr && col && out && partial, STIN_E_NULL);
    STIN_REQUIRE(partial_bytes >= (size_t)groups * 2 * (size_t)C * sizeof(double), STIN_E_WORKSPACE);
    if (!vec_ok<float>(C, {src, out}, {ld_src, ldo})) return STIN_E_ALIGN;
    const int c4 = C / 4;
    const size_t lds = (size_t)2 * (BLOCK / c4) * C * sizeof(double);
    hipLaunchKernelGGL((k_segment_mean_stats<2>), dim3((unsigned)groups), dim3(BLOCK), lds, (hipStream_t)stream_, src, ld_src, rowptr, col, N,
                       c4, out, ldo, partial);
    return stin_launch_status();
}

extern "C" int stin_gather_rows_f32(const float* src, int64_t ld_src, const int32_t* idx, const float* row_scale,
                                    int64_t N, int C, float* out, int64_t ldo, stin_stream_t stream) {
    stin_clear_stale_error();
    return gather_rows_impl<float>(src, ld_src, idx, row_scale, N, C, out, ldo, (hipStream_t)stream);
}
extern "C" int stin_gather_rows_bf16(const stin_bf16_t* src, int64_t ld_src, const int32_t* idx, const float* row_scale,
                                     int64_t N, int C, stin_bf16_t* out, int64_t ldo, stin_stream_t stream) {
    stin_clear_stale_error();
    return gather_rows_impl<stin_bf16>(b16(src), ld_src, idx, row_scale, N, C, b16(out), ldo, (hipStream_t)stream);
}

extern "C" int stin_batch_pool_i64(const int64_t* batch, const int32_t* rowptr, const int32_t* col, int64_t N,
                                   int64_t* out, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(batch && rowptr && col && out, STIN_E_NULL);
    hipLaunchKernelGGL(k_batch_pool, dim3(grid_elems(N)), dim3(BLOCK), 0, (hipStream_t)stream_, batch, rowptr, col, N, out);
    return stin_launch_status();
}

extern "C" int stin_norm_group_ids_i64(const int64_t* batch, const int32_t* ptr_sum, int B, int64_t N, int32_t* gid, int32_t* sid,
                                       stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && B > 0 && B <= 8192, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(batch && gid && (sid == nullptr || ptr_sum != nullptr), STIN_E_NULL);
    hipLaunchKernelGGL(k_norm_group_ids, dim3(grid_elems(N)), dim3(BLOCK), (size_t)B * sizeof(int32_t), (hipStream_t)stream_, batch,
                       ptr_sum, B, N, gid, sid);
    return stin_launch_status();
}

extern "C" int stin_gather_i64(const int64_t* src, const int32_t* idx, int64_t N, int64_t* out, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(src && idx && out, STIN_E_NULL);
    hipLaunchKernelGGL(k_gather_i64, dim3(grid_elems(N)), dim3(BLOCK), 0, (hipStream_t)stream_, src, idx, N, out);
    return stin_launch_status();
}
