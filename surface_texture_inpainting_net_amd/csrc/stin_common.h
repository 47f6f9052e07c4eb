// Shared helpers for the libstin_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/stin_hip.h"

#define STIN_WAVE 64

#define STIN_REQUIRE(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process
// (PyTorch included): drop whatever an earlier, unrelated call left behind before we launch.
static inline void stin_clear_stale_error() { (void)hipGetLastError(); }

// Fork without a marker packet (round 4).  hipEventRecord on the compute stream puts a barrier packet behind the kernel and the
// next kernel waits for the command processor to retire it: ~4 us of idle compute stream per fork (profiles/probes/fork_bind_probe.hip:
// record + wait + side kernel 16.1 us per link of an 11.0 us chain, event bound to the kernel's own completion signal 12.0).  A
// caller that wants "this launch is done" as an event sets stin_tl_stop_event; a launch site that supports it passes the event as
// hipExtLaunchKernelGGL's stopEvent and clears the variable (= "bound"); the caller records the event the ordinary way if the
// variable is still set afterwards.
extern thread_local hipEvent_t stin_tl_stop_event;
#define STIN_LAUNCH_STOP(KERNEL, GRID, BLOCK_, STREAM, ...)                                                          \
    do {                                                                                                             \
        if (stin_tl_stop_event != nullptr) {                                                                         \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK_, 0, STREAM, nullptr, stin_tl_stop_event, 0, __VA_ARGS__);     \
            stin_tl_stop_event = nullptr;                                                                            \
        } else {                                                                                                     \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK_, 0, STREAM, __VA_ARGS__);                                        \
        }                                                                                                            \
    } while (0)

static inline int stin_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STIN_OK : (int)e;
}

// Shapes whose pre-split NT weight operand is stored in MFMA fragment order under STIN_GEMM_W_FRAG (stin_hip.h), i.e. the
// shapes the resident-strip kernel takes.  Layout constraints: K % 64 == 0, Nc % 32 == 0.  Selection (measured per shape on
// MI355X, profiles/r02_gemm_shapes.md): one resident K chunk (K <= 256) with enough work per staged strip (K >= 128) and
// at least 2.5 column panels (Nc >= 320) - 5-15 % faster there; narrower outputs leave waves of the 128-column panel idle,
// longer K would add its chunks through memory, and both stay on the tiled kernel.
// Narrow outputs with a long reduction (Nc = 256 and K >= 512, Nc = 128 and K >= 256; K a multiple of 64) go to the
// all-columns kernel (k_gemm_nt_wide), which reads the same layout; shorter K measured slower there than on the 64x64 tiling.
__host__ __device__ static inline bool stin_w_frag_shape(int Nc, int K) {
    if (K % 64 != 0) return false;
    if ((Nc == 256 && K >= 512) || (Nc == 128 && K >= 256)) return true;
    return Nc % 32 == 0 && K >= 128 && K <= 256 && Nc >= 320;
}

// Width of the packed first-Linear operand / of Y and dY of a fused block.  trans_inv: 0 = EdgeConv ([Wa - Wb ; Wb ; Ws], A and B
// materialised), 1 = translation-invariant, both halves materialised ([-W1 ; W1 ; Ws]), 2 = translation-invariant COMPACT (round 6):
// the operand is [W1 ; Ws], only B = x W1^T exists - A_i = b1 - B_i is formed by the edge stage (the value the GEMM wrote for mode 1
// up to one ulp of the accumulator) - and the backward pass carries D = dB - dA (stin_hip.h, STIN_TI_COMPACT).
__host__ __device__ static inline int stin_yw(int H, int Cout, int has_shortcut, int trans_inv) {
    return (trans_inv == STIN_TI_COMPACT ? H : 2 * H) + (has_shortcut ? Cout : 0);
}

// Ticket words of the one-launch reductions (stin_norm.hip k_colreduce_t, stin_tail.hip k_linear_tanh_bwd) live in a global
// array of `slots` rows and every launch takes the next row, so launches in flight never share a word.  A launch that is being
// CAPTURED into a hipGraph bakes its row into the graph and will run at every replay - possibly beside eager launches whose
// rotating counter has come round to the same row.  Captured launches therefore rotate through the upper half of the rows and
// eager launches through the lower half: the two populations never meet.
#include <atomic>
static inline int stin_ticket_slot(std::atomic<unsigned>& seq_eager, std::atomic<unsigned>& seq_captured, int slots, hipStream_t stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    const unsigned half = (unsigned)slots / 2;
    if (capturing) return (int)(half + seq_captured.fetch_add(1, std::memory_order_relaxed) % half);
    return (int)(seq_eager.fetch_add(1, std::memory_order_relaxed) % half);
}

// Per-DEVICE caches (round 6; a process may drive more than one GPU): the CU count, and "this function's dynamic-LDS attribute has
// been raised on this device" (hipFuncSetAttribute is per device: a process-wide once-flag left the second GPU's launches of the
// > 64 KB kernels failing).
static inline int stin_device_slot() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : 0;
}
static inline int stin_cu_count_dev() {
    static std::atomic<int> n[64];
    const int d = stin_device_slot();
    int v = n[d].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
        n[d].store(v, std::memory_order_relaxed);
    }
    return v;
}
struct stin_once_per_device {
    std::atomic<unsigned long long> mask{0};
    bool first() {                                     // true exactly once per device
        const unsigned long long bit = 1ull << stin_device_slot();
        return (mask.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
    }
};

static inline bool stin_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Lanes that cooperate on one feature row: smallest power of two >= ceil(C/4), capped at a wave.
static inline int stin_group_lanes(int c4) {
    int g = 1;
    while (g < c4 && g < STIN_WAVE) g <<= 1;
    return g;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// bf16 STORAGE (the *_bf16 entry points): rows of __bf16 in HBM, every kernel widens to fp32 on load, computes and
// accumulates in fp32 and rounds to nearest-even on store.  A 4-channel chunk is 8 bytes.
typedef __bf16 stin_bf16;
typedef __bf16 stin_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const stin_bf16* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(stin_bf16* p, float4 v) {
    stin_bf16x4 h = {(stin_bf16)v.x, (stin_bf16)v.y, (stin_bf16)v.z, (stin_bf16)v.w};
    *reinterpret_cast<stin_bf16x4*>(p) = h;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const stin_bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(stin_bf16* p, float v) { *p = (stin_bf16)v; }

// 4-channel vector access needs 16-byte (fp32) / 8-byte (bf16) aligned rows
template <typename T> static inline bool stin_aligned_vec4(const void* p) {
    return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(T) - 1)) == 0;
}

// ---- weight-gradient (TN) products shared between stin_gemm.hip and stin_wgrad.hip ------------------------------------
// One dW[Nc, K (+1)] = G[M, Nc]^T [X[M, K] | w] product split over row chunks: its geometry and operands as the TN kernels
// take them.  Pointers are typed float* also for bf16 storage (only the fp32 kernels read them through this struct).
// Operand transform of the SingleConvMeshNet products (round 5): the operand rows are read as relu(bn(v)) per column - BatchNorm1d
// + ReLU of the E x 2 cout edge rows applied while the GEMM stages them, so that the normalised matrix never exists in memory
// (stin_gemm_nt_bn_f32: the A operand; stin_gemm_tn_bn_f32: the X operand).  mean == NULL: off.
// The staging threads of these kernels are bound by vector-ALU work (the 16-bit split), so the transform is the affine form
//   relu(v s + t),  s = gamma rstd,  t = beta - mean s        (mul, add, max: 3 operations per element)
// with (s, t) formed once per thread for its fixed columns - not stin_bn_act_fwd_f32's gamma ((v - mean) rstd) + beta (6
// operations; measured +80 % on the four-wave TN kernel, +50 % on the producer / consumer kernel).  The two agree to fp32 rounding
// (~1e-7 relative); forward product and weight gradient use the SAME form, so they stay consistent with each other.
struct stin_bn_tf {
    const float *mean, *rstd, *gamma, *beta;
};
__device__ __forceinline__ void stin_bn_st(const stin_bn_tf& tf, int c, float& s, float& t) {
    s = tf.gamma[c] * tf.rstd[c];
    t = tf.beta[c] - tf.mean[c] * s;
}
__device__ __forceinline__ float stin_bn_relu(float v, float s, float t) { return fmaxf(v * s + t, 0.f); }
// (s, t) of columns [c, c + 4) in registers (a staging thread's columns are fixed: formed once)
struct stin_bn_coef4 {
    float4 s, t;
};
__device__ __forceinline__ stin_bn_coef4 stin_bn_coef4_load(const stin_bn_tf& tf, int c) {
    stin_bn_coef4 q;
    stin_bn_st(tf, c, q.s.x, q.t.x);
    stin_bn_st(tf, c + 1, q.s.y, q.t.y);
    stin_bn_st(tf, c + 2, q.s.z, q.t.z);
    stin_bn_st(tf, c + 3, q.s.w, q.t.w);
    return q;
}
__device__ __forceinline__ float4 stin_bn_relu4(float4 v, const stin_bn_coef4& q) {
    return make_float4(stin_bn_relu(v.x, q.s.x, q.t.x), stin_bn_relu(v.y, q.s.y, q.t.y), stin_bn_relu(v.z, q.s.z, q.t.z),
                       stin_bn_relu(v.w, q.s.w, q.t.w));
}
// ragged / unaligned form: element e of the float4 is column c + e, valid while c + e < lim
__device__ __forceinline__ float4 stin_bn_relu4_ragged(float4 v, const stin_bn_tf& tf, int c, int lim) {
    float s, t;
    if (c + 0 < lim) { stin_bn_st(tf, c, s, t); v.x = stin_bn_relu(v.x, s, t); }
    if (c + 1 < lim) { stin_bn_st(tf, c + 1, s, t); v.y = stin_bn_relu(v.y, s, t); }
    if (c + 2 < lim) { stin_bn_st(tf, c + 2, s, t); v.z = stin_bn_relu(v.z, s, t); }
    if (c + 3 < lim) { stin_bn_st(tf, c + 3, s, t); v.w = stin_bn_relu(v.w, s, t); }
    return v;
}

struct stin_tn_problem {
    stin_bn_tf xtf;                    // transform of the X operand (mean == NULL: none)
    const float *G, *X, *row_w;
    float* slab;                       // [chunks][Nc * Kq + roundup4(Nc)] partial results
    int64_t ldg, ldx, ld_w, M, chunks;
    int Nc, K, Kq, has_bias, rows_per_chunk, tiles_i, tiles_j, TI, TJ, vec;
    unsigned block0;                   // first block of the problem in a merged grid (set by stin_tn_ws_launch)
};
struct stin_tn_batch {
    stin_tn_problem p[2];
    int n;
};
// stin_gemm.hip: geometry of one product (*ws_eligible = 1: the producer / consumer kernel of stin_wgrad.hip takes it);
// stin_tn_slabs launches the product's TN kernel only (partial slabs, no reduction)
int stin_tn_problem_init(stin_tn_problem* p, int storage, const void* G, int64_t ldg, const void* X, int64_t ldx, int64_t M,
                         int Nc, int K, int ones_column, const void* row_w, int64_t ld_w, int precision, float* slab,
                         int* ws_eligible);
int stin_tn_slabs(const stin_tn_problem* p, int storage, int precision, stin_stream_t stream);
// stin_wgrad.hip
bool stin_tn_ws_enabled();
int stin_tn_ws_launch(stin_tn_batch batch, stin_stream_t stream);
