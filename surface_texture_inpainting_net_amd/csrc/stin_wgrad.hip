// Weight gradients of one fused EdgeConv block (reference models/surfacetextureinpaintingnet.py:507-521, the backward of
// its two Linears + shortcut) as TWO launches instead of five:
//
//   k_gemm_tn_ws        dW2|db2 = dagg^T [hE | w]  AND  [dW1|dWs|db] = dY^T [x | 1]  - both products in ONE grid
//   k_wgrad_finalize    fixed-order sum of the per-chunk partial slabs of both products, written straight into the
//                       reference-layout gradients dW1, db1, dW2, db2, dWs, dbs (k_reduce_slabs x 2 + k_unpack in one pass)
//
// k_gemm_tn_ws is the split-bf16 TN product of stin_gemm.hip (k_gemm_tn_bf16s<128, 128, 2>) re-cut for two waves per SIMD
// with FIXED ROLES: the reduction index of a weight gradient is the vertex row m, so every operand element has to be split
// into its two bf16 pieces AND transposed on its way into the MFMA fragments - per 32-row slab that is as much vector-ALU
// work (global loads, 2 x cvt + sub per element, ds_write_b64) as the 24 MFMAs it feeds are matrix-pipe work.  In the
// 4-wave kernel every wave alternates between the two (stage, barrier, multiply, barrier): the matrix pipe idles while the
// block stages and the vector ALU idles while it multiplies, and the second block on the CU only hides that when it
// happens to run in the opposite phase.  Here a block has 8 waves: waves 4-7 (producers) only load, split and store the
// NEXT slab into the other half of a double-buffered LDS image, waves 0-3 (consumers) only read fragments and issue MFMAs
// - one of each per SIMD (a workgroup's wave w and w + 4 share a SIMD), so the vector pipe and the matrix pipe of a SIMD
// run side by side by construction, with one barrier per slab.  The producers' global loads run two slabs ahead.
// Arithmetic, chunking, k order and MFMA order are those of k_gemm_tn_bf16s<128, 128, 2, true>: bit-identical slabs
// (tests/test_hip_parity.py::test_gemm_tn_ws_kernel_equals_four_wave_kernel).
#include <cstdlib>
#include <type_traits>
#include "stin_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int WS_T = 128;             // output tile (both directions)
constexpr int WS_R = 32;              // rows (m) per LDS slab = two MFMA k-steps of 16
constexpr int WS_THREADS = 512;
// LDS image of one operand slab, per bf16 piece: [column][32 m] = 64-byte rows without padding, addressed through
//   ws_lds(col, g) = (col ^ ((col >> 5) & 1)) * 64 + ((((g >> 1) ^ (col >> 2)) & 3) << 4) + ((g & 1) << 3)     (bytes)
// g = m / 4 = the 8-byte group of 4 consecutive rows; an MFMA fragment (8 consecutive m, g even) is ONE aligned 16-byte read.
//  * consumers (ds_read_b128, 16-lane bank groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of the 32 columns of a tile):
//    column bits 0, 1 choose one of four 64-byte rows modulo 256 bytes (bit 0 through the row flip), bits 2, 3 one of the four
//    swizzled 16-byte slots of the row - in each bank group the (bit 4, 3, 2) patterns are {000, 011, 101, 110} resp.
//    {001, 010, 100, 111}: 16 distinct slots = all 64 banks once.
//  * producers (ds_write_b64, 16 contiguous lanes): a lane owns a column group (4 columns), the lanes of a write differ in
//    column bits 2..5: bits 2, 3 -> slot, bit 5 -> row parity (the other 16 banks); bit 4 would collide, so lanes with
//    column bit 4 set work on the OTHER row group of the pair (rg = rb ^ bit 4, see the kernel) and hit the other 8-byte
//    half of the slot -> all 32 banks once.
constexpr int WS_PLANE = WS_T * WS_R;                  // bf16 elements per piece plane (8 KB)
__device__ __forceinline__ int ws_lds(int col, int g) {   // ELEMENT offset inside a piece plane
    return ((col ^ ((col >> 5) & 1)) << 5) + ((((g >> 1) ^ (col >> 2)) & 3) << 3) + ((g & 1) << 2);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// profiling build only (profiles/tn_stamps.hip compiles this file with -DSTIN_WS_STAMPS): s_memtime stamps of one wave per role
#ifdef STIN_WS_STAMPS
__device__ unsigned long long* stin_ws_stamp_buf = nullptr;
#define WS_STAMP(i)                                                                                                   \
    do {                                                                                                              \
        if (lane == 0 && stin_ws_stamp_buf != nullptr && (i) < 64)                                                    \
            stin_ws_stamp_buf[((size_t)blockIdx.x * 8 + wave) * 64 + (i)] = __builtin_amdgcn_s_memtime();             \
    } while (0)
#define WS_ABL(bit) (((prio >> 4) & (bit)) != 0)     /* profiling ablations: 1 no MFMA, 2 no split/store, 4 no global loads */
#else
#define WS_STAMP(i)
#define WS_ABL(bit) false
#endif

struct WsSlab {
    float4 g[4], x[4];                // this thread's 4 (rows) x 4 (cols) patch of G and of X
    float pw[4];
    bool pv[4];
};

__global__ __launch_bounds__(WS_THREADS) void k_gemm_tn_ws(const stin_tn_batch batch, const int prio) {
    __shared__ __attribute__((aligned(16))) __bf16 Gt[2][2][WS_PLANE];            // [buffer][piece][ws_lds(column, m / 4)]
    __shared__ __attribute__((aligned(16))) __bf16 Xt[2][2][WS_PLANE];
    __shared__ float bsum[8][WS_T + 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int pi = (batch.n > 1 && blockIdx.x >= batch.p[1].block0) ? 1 : 0;
    const float* __restrict__ G = batch.p[pi].G;
    const float* __restrict__ X = batch.p[pi].X;
    const float* __restrict__ row_w = batch.p[pi].row_w;
    const int64_t ldg = batch.p[pi].ldg, ldx = batch.p[pi].ldx, ld_w = batch.p[pi].ld_w, M = batch.p[pi].M;
    const int64_t chunks = batch.p[pi].chunks;
    const int Nc = batch.p[pi].Nc, K = batch.p[pi].K, Kq = batch.p[pi].Kq;
    const int tiles_j = batch.p[pi].tiles_j, tiles = batch.p[pi].tiles_i * tiles_j;
    const int rows_per_chunk = batch.p[pi].rows_per_chunk;

    // block -> (row chunk, output tile): the map of k_gemm_tn_bf16s (chunks of one XCD share its L2)
    const int64_t b = (int64_t)blockIdx.x - batch.p[pi].block0;
    const int64_t xcd = b % 8, q = b / 8;
    const int64_t chunk = chunks >= 8 ? (q / tiles) * 8 + xcd : b / tiles;
    const int tile = (int)(chunks >= 8 ? q % tiles : b % tiles);
    if (chunk >= chunks) return;                                  // (block-uniform; before any barrier)
    const int tj = tile % tiles_j, ti = tile / tiles_j;
    const int i0 = ti * WS_T, j0 = tj * WS_T;
    const int64_t mb = chunk * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;
    const bool want_bias = batch.p[pi].has_bias && (tj == 0);

    if (prio == 1 && !producer) __builtin_amdgcn_s_setprio(1);
    if (prio == 2 && producer) __builtin_amdgcn_s_setprio(1);

    float* out = batch.p[pi].slab + chunk * ((int64_t)Nc * Kq + ((Nc + 3) & ~3));

    // The two roles are two separate programs behind a wave-uniform (scalar) branch, so that the register allocation is the
    // maximum of the two and not their sum; both execute the same barrier sequence: one after the prologue, two per 64 rows.
    // Pipeline: slab s is multiplied out of buffer s & 1 while the producers split slab s + 1 into the other buffer and
    // request slab s + 3 into the register set that has just been stored (loads two slabs = two barriers ahead).  Rows past
    // the chunk load zeros, so the slab count is rounded up to an even number as in the four-wave kernel.
    if (producer) {
        // ---- producer: patch (row group rg, column group c4) of G and of X per slab
        const int ptid = tid & 255;
        // lane = column group: one wave-wide load instruction covers 2 rows x 512 contiguous bytes.  (The four-wave kernel's
        // map - 8 rows x 128 bytes per instruction - reads at HALF the rate: measured 3080 vs 1600 cycles per 32 KB slab with
        // every CU loading, profiles/tn_stamps.hip; the column-major LDS image above is what the faster map needs.)
        // A lane with column bit 4 set (c4 bit 2) takes the other row group of its pair: keeps the LDS stores conflict-free
        // (comment at ws_lds) and leaves the set of addresses of a load instruction - two whole 512-byte rows - unchanged.
        const int c4 = ptid % 32, rg = (ptid / 32) ^ ((c4 >> 2) & 1);
        // a column group past Nc / K reads column group 0 instead (valid memory): the tile columns it feeds are never stored,
        // so nothing has to be zeroed for them (Nc, K are multiples of 4: a float4 is wholly in or out)
        const float* gp = G + (i0 + c4 * 4 < Nc ? i0 + c4 * 4 : 0) + (mb + rg * 4) * ldg;     // row rg * 4 of the NEXT slab to load
        const int xcol = j0 + c4 * 4 < K ? j0 + c4 * 4 : 0;
        const float* xp = X + xcol + (mb + rg * 4) * ldx;
        const stin_bn_tf xtf = batch.p[pi].xtf;                   // (round 5) X rows read as relu(bn(.)) per column: block-uniform
        stin_bn_coef4 xq;                                           // this thread's four X columns are fixed: (s, t) once
        xq.s = xq.t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xtf.mean != nullptr) xq = stin_bn_coef4_load(xtf, xcol);
        const float* wp = (want_bias && row_w != nullptr) ? row_w + (mb + rg * 4) * ld_w : nullptr;
        int64_t m_next = mb + rg * 4;                                                          // its row index
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        // FULL = every row of the slab is inside the chunk (all but the last one or two slabs): no row test, no zero fill
        auto load_slab = [&](WsSlab& S, auto FULL) {
            if (WS_ABL(4)) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (decltype(FULL)::value) {
                    S.g[r] = ld4(gp + r * ldg);
                    S.x[r] = ld4(xp + r * ldx);
                    if (wp != nullptr) S.pw[r] = wp[r * ld_w];
                    S.pv[r] = true;
                } else {
                    const bool in = m_next + r < me;
                    float4 vg = make_float4(0.f, 0.f, 0.f, 0.f), vx = vg;
                    if (in) {
                        vg = ld4(gp + r * ldg);
                        vx = ld4(xp + r * ldx);
                    }
                    S.g[r] = vg;
                    S.x[r] = vx;
                    S.pw[r] = (in && wp != nullptr) ? wp[r * ld_w] : 0.f;
                    S.pv[r] = in;
                }
            }
            gp += WS_R * ldg;
            xp += WS_R * ldx;
            if (wp != nullptr) wp += WS_R * ld_w;
            m_next += WS_R;
        };
        // 4 consecutive rows of one column -> its two bf16 pieces, 8 bytes each (same roundings as the element-wise
        // (__bf16)x ; x -= (float)h of k_gemm_tn_bf16s, written so that hipcc emits 3 vector instructions per element:
        // v_cvt_pk_bf16_f32 for two elements at once, shift / mask to widen, v_sub, v_cvt_pk again)
        auto split_col = [&](float a, float b, float c, float d, __bf16* dst) {
            const unsigned h01 = pack_bf16(a, b), h23 = pack_bf16(c, d);
            const float r0 = a - __uint_as_float(h01 << 16), r1 = b - __uint_as_float(h01 & 0xffff0000u);
            const float r2 = c - __uint_as_float(h23 << 16), r3 = d - __uint_as_float(h23 & 0xffff0000u);
            *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(dst + WS_PLANE) = make_uint2(pack_bf16(r0, r1), pack_bf16(r2, r3));
        };
        // element offsets of this thread's four columns (row group rg) inside a piece plane
        const int o0 = ws_lds(c4 * 4, rg), o1 = ws_lds(c4 * 4 + 1, rg), o2 = ws_lds(c4 * 4 + 2, rg), o3 = ws_lds(c4 * 4 + 3, rg);
        auto split_store = [&](const float4 (&p)[4], __bf16* plane) {
            split_col(p[0].x, p[1].x, p[2].x, p[3].x, plane + o0);
            split_col(p[0].y, p[1].y, p[2].y, p[3].y, plane + o1);
            split_col(p[0].z, p[1].z, p[2].z, p[3].z, plane + o2);
            split_col(p[0].w, p[1].w, p[2].w, p[3].w, plane + o3);
        };
        auto store_slab = [&](const WsSlab& S, int buf) {
            if (WS_ABL(2)) return;
            if (want_bias) {                                       // block-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float w = S.pv[r] ? (row_w != nullptr ? S.pw[r] : 1.f) : 0.f;
                    bs.x += w * S.g[r].x;
                    bs.y += w * S.g[r].y;
                    bs.z += w * S.g[r].z;
                    bs.w += w * S.g[r].w;
                }
            }
            split_store(S.g, &Gt[buf][0][0]);
            if (xtf.mean != nullptr) {                             // (at STORE time: the loads stay in flight two slabs ahead)
                const float4 tx[4] = {stin_bn_relu4(S.x[0], xq), stin_bn_relu4(S.x[1], xq), stin_bn_relu4(S.x[2], xq), stin_bn_relu4(S.x[3], xq)};
                split_store(tx, &Xt[buf][0][0]);
            } else {
                split_store(S.x, &Xt[buf][0][0]);
            }
        };
        WsSlab S0, S1;
        WS_STAMP(0);
        load_slab(S0, std::false_type());
        load_slab(S1, std::false_type());
        store_slab(S0, 0);
        WS_STAMP(1);
        load_slab(S0, std::false_type());
        __syncthreads();
        WS_STAMP(2);
        int64_t m0 = mb;
        int st = 3;
        (void)st;
        for (; m0 + 5 * WS_R <= me; m0 += 2 * WS_R) {             // the slabs requested here (m0 + 96, m0 + 128) are whole
            store_slab(S1, 1);
            WS_STAMP(st);
            load_slab(S1, std::true_type());
            __syncthreads();
            WS_STAMP(st + 1);
            store_slab(S0, 0);
            load_slab(S0, std::true_type());
            __syncthreads();
            st += 2;
        }
        WS_STAMP(62);
        for (; m0 < me; m0 += 2 * WS_R) {
            store_slab(S1, 1);
            load_slab(S1, std::false_type());
            __syncthreads();
            store_slab(S0, 0);
            load_slab(S0, std::false_type());
            __syncthreads();
        }
        WS_STAMP(63);
        if (want_bias) st4(&bsum[rg][c4 * 4], bs);
    } else {
        // ---- consumer: wave (wi, wj) owns a 64 x 64 quarter of the tile
        const int wi = (wave >> 1) & 1, wj = wave & 1;
        const int kh = lane >> 5, li = lane & 31;
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
        // fragment (column tile t, k-step ks) of lane (li, kh): 8 consecutive m = two swizzled 8-byte groups
        int ga[2], gc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ga[t] = wi * 64 + t * 32 + li;
            gc[t] = wj * 64 + t * 32 + li;
        }
        auto frag = [&](const __bf16* plane, int col, int ks) {
            return *reinterpret_cast<const bf16x8*>(plane + ws_lds(col, (ks >> 2) + 2 * kh));
        };
        auto multiply = [&](int buf) {
            if (WS_ABL(1)) return;
#pragma unroll
            for (int ks = 0; ks < WS_R; ks += 16) {
                bf16x8 a[2][2], c[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        a[p][t] = frag(&Gt[buf][p][0], ga[t], ks);
                        c[p][t] = frag(&Xt[buf][p][0], gc[t], ks);
                    }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][t], c[1][u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][t], c[0][u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][t], c[0][u], acc[t][u], 0, 0, 0);
                    }
            }
        };
        WS_STAMP(0);
        __syncthreads();
        WS_STAMP(2);
        int st = 3;
        (void)st;
        for (int64_t m0 = mb; m0 < me; m0 += 2 * WS_R) {
            multiply(0);
            WS_STAMP(st);
            __syncthreads();
            WS_STAMP(st + 1);
            multiply(1);
            __syncthreads();
            st += 2;
        }
        WS_STAMP(62);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = j0 + wj * 64 + u * 32 + li;
            if (col >= K) continue;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wi * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (row < Nc) out[(int64_t)row * Kq + col] = acc[t][u][r];
                }
        }
        WS_STAMP(63);
    }
    if (want_bias) {                                              // block-uniform
        __syncthreads();
        if (tid < WS_T) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += bsum[r][tid];
            if (i0 + tid < Nc) out[(int64_t)Nc * Kq + i0 + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------- finalize
// Sums the partial slabs of both products in the fixed order of k_reduce_slabs (16 chunk-lanes x 16 float4 groups per block:
// each chunk-lane walks the chunk list with stride 16, four partial sums, then a fixed-order LDS reduction over the lanes)
// and writes the sums where k_unpack would have put them: bit-identical to k_reduce_slabs + k_unpack.
constexpr int FN_COLS = 16, FN_KL = 16, FN_BLOCK = 256;
struct WgFinal {
    const float* slabA;      // dW2 product: [chunksA][Cout][KqA] + bias [Cout]
    const float* slabB;      // packed product: [chunksB][Yw][KqB] + bias [Yw]
    int64_t chunksA, chunksB;
    int KqA, KqB, Cin, Cp, H, Cout, has_shortcut, trans_inv;
    float *dW1, *db1, *dWs, *dbs, *dW2, *db2;
    const float* ti_colsum;  // compact trans-inv layout: [ti_rows][H] column partials of dA (stin_graph.hip k_edge_bwd_mask_ti) -> db1
    int64_t ti_rows;
};
__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// One chunk-lane's share of a fixed-order fold over `chunks` rows of stride `cs` floats (lane ty of FN_KL walks rows ty, ty + FN_KL, ...;
// four running sums, merged (s0 + s1) + (s2 + s3)): shared by k_wgrad_finalize and k_colsum_fold so that both give the same bits.
__device__ __forceinline__ float4 fn_partial(const float* __restrict__ p, int64_t cs, int64_t chunks, int ty) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        if (p != nullptr) {
            int64_t c = ty;
            // (round 4) sixteen slab loads in flight per thread where the chunk list is long (the first block's skinny product
            // writes ~900 chunks: with four in flight this fold is 14-28 dependent trips to the fabric - 200 us at the very end of
            // the backward pass under rocprofv3, 11 us now; stand-alone and in the un-profiled step the difference is small); the additions keep the order of the four-at-a-time loop below: bit-identical sums
            for (; c + 15 * FN_KL < chunks; c += 16 * FN_KL) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = ld4(p + (c + u * FN_KL) * cs);
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    add4(s0, v[u]);
                    add4(s1, v[u + 1]);
                    add4(s2, v[u + 2]);
                    add4(s3, v[u + 3]);
                }
            }
            for (; c + 3 * FN_KL < chunks; c += 4 * FN_KL) {
                add4(s0, ld4(p + c * cs));
                add4(s1, ld4(p + (c + FN_KL) * cs));
                add4(s2, ld4(p + (c + 2 * FN_KL) * cs));
                add4(s3, ld4(p + (c + 3 * FN_KL) * cs));
            }
            for (; c < chunks; c += FN_KL) add4(s0, ld4(p + c * cs));
        }
        add4(s0, s1);
        add4(s2, s3);
        add4(s0, s2);
        return s0;
}

// db1 of a compact trans-inv block from the edge stage's column partials [rows][H] alone (the per-op path; the whole-block path folds
// them inside k_wgrad_finalize): same lanes, same order -> the same bits
__global__ __launch_bounds__(FN_BLOCK) void k_colsum_fold(const float* __restrict__ colsum, int64_t rows, int H, float* __restrict__ out) {
    __shared__ float4 sm[FN_KL][FN_COLS + 1];
    const int tx = threadIdx.x % FN_COLS, ty = threadIdx.x / FN_COLS;
    const int64_t g = (int64_t)blockIdx.x * FN_COLS + tx;
    const bool on = g < H / 4;
    sm[ty][tx] = fn_partial(on ? colsum + 4 * g : nullptr, H, rows, ty);
    __syncthreads();
    if (ty != 0 || !on) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < FN_KL; ++k) add4(a, sm[k][tx]);
    st4(out + 4 * g, a);
}

__global__ __launch_bounds__(FN_BLOCK) void k_wgrad_finalize(const WgFinal f) {
    __shared__ float4 sm[2][FN_KL][FN_COLS + 1];
    const int tx = threadIdx.x % FN_COLS, ty = threadIdx.x / FN_COLS;
    const bool compact = f.trans_inv == STIN_TI_COMPACT;
    const int Yw = stin_yw(f.H, f.Cout, f.has_shortcut, f.trans_inv);
    const int s0 = Yw - (f.has_shortcut ? f.Cout : 0);          // first shortcut row of the packed product (2 H, compact: H)
    const int64_t csA = (int64_t)f.Cout * f.KqA + ((f.Cout + 3) & ~3), csB = (int64_t)Yw * f.KqB + ((Yw + 3) & ~3);
    // sections of float4 groups: A weights | A bias | B rows [0, H) (paired with rows [H, 2H); compact: alone) | B shortcut rows |
    // B bias | (compact) db1 from the edge stage's column partials of dA
    const int64_t nAw = (int64_t)f.Cout * f.KqA / 4, nAb = (f.Cout + 3) / 4;
    const int64_t nB1 = (int64_t)f.H * f.KqB / 4, nB2 = f.has_shortcut ? (int64_t)f.Cout * f.KqB / 4 : 0, nBb = (Yw + 3) / 4;
    const int64_t nT = (compact && f.db1 != nullptr) ? f.H / 4 : 0;
    int64_t g = (int64_t)blockIdx.x * FN_COLS + tx;
    const float *p0 = nullptr, *p1 = nullptr;
    int64_t cs = 0, chunks = 0;
    int section = -1;
    if (g < nAw) section = 0, p0 = f.slabA + 4 * g, cs = csA, chunks = f.chunksA;
    else if ((g -= nAw) < nAb) section = 1, p0 = f.slabA + (int64_t)f.Cout * f.KqA + 4 * g, cs = csA, chunks = f.chunksA;
    else if ((g -= nAb) < nB1) section = 2, p0 = f.slabB + 4 * g, p1 = compact ? nullptr : p0 + (int64_t)f.H * f.KqB, cs = csB, chunks = f.chunksB;
    else if ((g -= nB1) < nB2) section = 3, p0 = f.slabB + (int64_t)s0 * f.KqB + 4 * g, cs = csB, chunks = f.chunksB;
    else if ((g -= nB2) < nBb) section = 4, p0 = f.slabB + (int64_t)Yw * f.KqB + 4 * g, cs = csB, chunks = f.chunksB;
    else if ((g -= nBb) < nT) section = 5, p0 = f.ti_colsum + 4 * g, cs = f.H, chunks = f.ti_rows;
    auto partial = [&](const float* p) { return fn_partial(p, cs, chunks, ty); };
    sm[0][ty][tx] = partial(p0);
    sm[1][ty][tx] = partial(p1);
    __syncthreads();
    if (ty != 0 || section < 0) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll
    for (int k = 0; k < FN_KL; ++k) add4(a, sm[0][k][tx]);
    if (section == 2) {
#pragma unroll
        for (int k = 0; k < FN_KL; ++k) add4(b, sm[1][k][tx]);
    }
    const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {b.x, b.y, b.z, b.w};
    if (section == 0) {                                   // dW2 [Cout, H]
        const int64_t row = (4 * g) / f.KqA;
        const int col = (int)((4 * g) % f.KqA);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < f.H) f.dW2[row * f.H + col + e] = va[e];
    } else if (section == 1) {                            // db2 [Cout]
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g + e < f.Cout && f.db2 != nullptr) f.db2[4 * g + e] = va[e];
    } else if (section == 2) {                            // dW1 [H, Cin] (trans-inv: d/dW1 of (-W1, W1)) or [H, 2 Cin] (Wa | Wb)
        const int64_t row = (4 * g) / f.KqB;
        const int col = (int)((4 * g) % f.KqB);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (col + e >= f.Cin) continue;
            if (compact) f.dW1[row * f.Cin + col + e] = va[e];                  // the operand is W1 itself
            else if (f.trans_inv) f.dW1[row * f.Cin + col + e] = vb[e] - va[e];
            else {
                f.dW1[row * 2 * f.Cin + col + e] = va[e];
                f.dW1[row * 2 * f.Cin + f.Cin + col + e] = vb[e] - va[e];
            }
        }
    } else if (section == 3) {                            // dWs [Cout, Cin]
        const int64_t row = (4 * g) / f.KqB;
        const int col = (int)((4 * g) % f.KqB);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < f.Cin) f.dWs[row * f.Cin + col + e] = va[e];
    } else if (section == 4) {                            // bias gradients of the packed operand: db1 = rows [0, H), dbs = rows [s0, ..)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = 4 * g + e;
            if (i < f.H && !compact) {
                if (f.db1 != nullptr) f.db1[i] = va[e];
            } else if (i >= s0 && i < Yw && f.dbs != nullptr) f.dbs[i - s0] = va[e];
        }
    } else {                                              // compact trans-inv: db1 = sum_i dA_i, the block rows of the edge stage's partials
#pragma unroll
        for (int e = 0; e < 4; ++e) f.db1[4 * g + e] = va[e];
    }
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// (re-read on every call: one process can A/B the four-wave and the producer / consumer kernel)
bool stin_tn_ws_enabled() {
    const char* e = getenv("STIN_TN_WS");
    return e == nullptr || atoi(e) != 0;
}

int stin_tn_ws_launch(stin_tn_batch batch, stin_stream_t stream) {
    const int prio = 1;                                            // consumers (the MFMA-issuing waves) at raised priority
    unsigned blocks = 0;
    for (int i = 0; i < batch.n; ++i) {
        stin_tn_problem& p = batch.p[i];
        blocks = (blocks + 7u) & ~7u;                              // every problem starts on an XCD round (blockIdx % 8 == XCD)
        p.block0 = blocks;
        blocks += (unsigned)((p.chunks >= 8 ? ((p.chunks + 7) / 8) * 8 : p.chunks) * (int64_t)p.tiles_i * p.tiles_j);
    }
    if (blocks == 0) return STIN_OK;
    hipLaunchKernelGGL(k_gemm_tn_ws, dim3(blocks), dim3(WS_THREADS), 0, (hipStream_t)stream, batch, prio);
    return stin_launch_status();
}

extern "C" size_t stin_edgeconv_wgrad_workspace_bytes(int64_t N, int Cp, int H, int Cout, int has_shortcut) {
    if (N < 0 || Cp <= 0 || H <= 0 || Cout <= 0) return 0;
    const int Yw = 2 * H + (has_shortcut ? Cout : 0);
    return up256(stin_gemm_tn_workspace_bytes(N, Cout, H, 1)) + up256(stin_gemm_tn_workspace_bytes(N, Yw, Cp, 1)) + 256;
}

extern "C" int stin_edgeconv_wgrad(int storage, const void* dagg, int64_t ld_dagg, const void* hE, int64_t ldh, const void* dY,
                                   int64_t ldy, const void* x, int64_t ldx, int64_t N, int Cin, int Cp, int H, int Cout,
                                   int has_shortcut, int trans_inv, int precision, float* dW1, float* db1, float* dW2, float* db2,
                                   float* dWs, float* dbs, void* workspace, size_t workspace_bytes, stin_stream_t stream) {
    STIN_REQUIRE(trans_inv != STIN_TI_COMPACT, STIN_E_UNSUPPORTED);        // (the compact layout needs the dA partials: _ti below)
    return stin_edgeconv_wgrad_ti(storage, dagg, ld_dagg, hE, ldh, dY, ldy, x, ldx, N, Cin, Cp, H, Cout, has_shortcut, trans_inv, precision,
                                  dW1, db1, dW2, db2, dWs, dbs, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int stin_edgeconv_wgrad_ti(int storage, const void* dagg, int64_t ld_dagg, const void* hE, int64_t ldh, const void* dY,
                                      int64_t ldy, const void* x, int64_t ldx, int64_t N, int Cin, int Cp, int H, int Cout,
                                      int has_shortcut, int trans_inv, int precision, float* dW1, float* db1, float* dW2, float* db2,
                                      float* dWs, float* dbs, const float* ti_colsum, int64_t ti_rows, void* workspace,
                                      size_t workspace_bytes, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(storage == 0 || storage == 1, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(N >= 0 && Cin > 0 && Cp >= Cin && H > 0 && Cout > 0, STIN_E_SIZE);
    STIN_REQUIRE(trans_inv >= 0 && trans_inv <= STIN_TI_COMPACT, STIN_E_UNSUPPORTED);
    const bool compact = trans_inv == STIN_TI_COMPACT;
    STIN_REQUIRE(!compact || (storage == 0 && H % 4 == 0 && (db1 == nullptr || N == 0 || (ti_colsum != nullptr && ti_rows > 0))), STIN_E_UNSUPPORTED);
    const int Yw = stin_yw(H, Cout, has_shortcut, trans_inv);
    STIN_REQUIRE(ld_dagg >= Cout && ldh >= H + 1 && ldy >= Yw && ldx >= Cp, STIN_E_SIZE);
    STIN_REQUIRE(dW1 && dW2 && workspace && (!has_shortcut || dWs) && (N == 0 || (dagg && hE && dY && x)), STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_edgeconv_wgrad_workspace_bytes(N, Cp, H, Cout, has_shortcut), STIN_E_WORKSPACE);
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float* slabA = reinterpret_cast<float*>(p);
    float* slabB = reinterpret_cast<float*>(p + up256(stin_gemm_tn_workspace_bytes(N, Cout, H, 1)));
    const size_t es = storage ? 2 : 4;
    const void* row_w = static_cast<const char*>(hE) + (size_t)H * es;          // the [deg > 0] column of hE weights db2
    stin_tn_problem pa, pb;
    int wsa = 0, wsb = 0;
    int rc = stin_tn_problem_init(&pa, storage, dagg, ld_dagg, hE, ldh, N, Cout, H, 1, row_w, ldh, precision, slabA, &wsa);
    if (rc != STIN_OK) return rc;
    rc = stin_tn_problem_init(&pb, storage, dY, ldy, x, ldx, N, Yw, Cp, 1, nullptr, 0, precision, slabB, &wsb);
    if (rc != STIN_OK) return rc;
    if (wsa && wsb) {                                             // both products on the producer / consumer kernel: ONE grid
        stin_tn_batch batch;
        batch.p[0] = pa;
        batch.p[1] = pb;
        batch.n = 2;
        rc = stin_tn_ws_launch(batch, stream);
    } else {
        rc = stin_tn_slabs(&pa, storage, precision, stream);
        if (rc == STIN_OK) rc = stin_tn_slabs(&pb, storage, precision, stream);
    }
    if (rc != STIN_OK) return rc;
    WgFinal f;
    f.slabA = slabA;
    f.slabB = slabB;
    f.chunksA = pa.chunks;
    f.chunksB = pb.chunks;
    f.KqA = pa.Kq;
    f.KqB = pb.Kq;
    f.Cin = Cin;
    f.Cp = Cp;
    f.H = H;
    f.Cout = Cout;
    f.has_shortcut = has_shortcut;
    f.trans_inv = trans_inv;
    f.dW1 = dW1;
    f.db1 = db1;
    f.dWs = dWs;
    f.dbs = dbs;
    f.dW2 = dW2;
    f.db2 = db2;
    f.ti_colsum = ti_colsum;
    f.ti_rows = ti_rows;
    const int64_t groups = (int64_t)Cout * pa.Kq / 4 + (Cout + 3) / 4 + (int64_t)H * pb.Kq / 4 +
                           (has_shortcut ? (int64_t)Cout * pb.Kq / 4 : 0) + (Yw + 3) / 4 + ((compact && db1 != nullptr) ? H / 4 : 0);
    hipLaunchKernelGGL(k_wgrad_finalize, dim3((unsigned)((groups + FN_COLS - 1) / FN_COLS)), dim3(FN_BLOCK), 0, (hipStream_t)stream, f);
    return stin_launch_status();
}

extern "C" int stin_edge_bwd_ti_colsum_fold_f32(const float* colsum, int64_t rows, int H, float* db1, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(rows >= 0 && H > 0 && H % 4 == 0, STIN_E_SIZE);
    STIN_REQUIRE(db1 && (rows == 0 || colsum) && stin_aligned16(db1) && stin_aligned16(colsum), STIN_E_NULL);
    const unsigned grid = (unsigned)((H / 4 + FN_COLS - 1) / FN_COLS);
    hipLaunchKernelGGL(k_colsum_fold, dim3(grid), dim3(FN_BLOCK), 0, (hipStream_t)stream, colsum, rows, H, db1);
    return stin_launch_status();
}
