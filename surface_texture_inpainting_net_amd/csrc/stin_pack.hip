// Small parameter-side kernels: (un)packing of the EdgeConv weights into the per-vertex GEMM operands of
// the restructured block and the instance-norm backward coefficients.  One launch each
// instead of dozens of tiny framework ops per block and step.
#include <cstdlib>
#include "stin_common.h"

namespace {
constexpr int BLOCK = 256;

// Weight operand of stin_gemm_nt_f32 in PRE-SPLIT form (precision | STIN_GEMM_W_PRESPLIT): the two 16-bit pieces the
// kernel would otherwise compute for every block that stages the tile.  Same footprint as fp32: the 16 bytes of the
// k-group 4g..4g+3 of a row hold [hi x 4 | lo x 4]; piece type / pre-scale as in stin_gemm.hip (fp16: x 64, bf16: x 1).
// mode 0 = plain fp32; STIN_GEMM_W_BF16 = plain bf16 (the weight operand of the bf16-storage GEMMs).
// With STIN_GEMM_W_FRAG in `mode` and a shape the strip kernel takes (nc rows, kk columns: stin_w_frag_shape) the two pieces go to
// the MFMA fragment order documented at STIN_GEMM_W_FRAG in stin_hip.h instead (r = the element's row).
__device__ __forceinline__ void put_weight(float* __restrict__ base, int64_t row_off, int c, float v, int mode, int r = 0,
                                           int nc = 0, int kk = 0) {
    if ((mode & STIN_GEMM_W_FRAG) && stin_w_frag_shape(nc, kk)) {
        const int prec = mode & ~STIN_GEMM_W_FRAG;
        const int64_t lane_slot = ((int64_t)(r >> 5) * (kk >> 4) + (c >> 4)) * 64 + ((c >> 3) & 1) * 32 + (r & 31);
        uint16_t* h = reinterpret_cast<uint16_t*>(base) + lane_slot * 16 + (c & 7);
        if (prec == STIN_GEMM_F16X3) {
            const float s = v * 64.f;
            const _Float16 hi = (_Float16)s;
            const _Float16 lo = (_Float16)(s - (float)hi);
            h[0] = *reinterpret_cast<const uint16_t*>(&hi);
            h[8] = *reinterpret_cast<const uint16_t*>(&lo);
        } else {
            const __bf16 hi = (__bf16)v;
            const __bf16 lo = (__bf16)(v - (float)hi);
            h[0] = *reinterpret_cast<const uint16_t*>(&hi);
            h[8] = *reinterpret_cast<const uint16_t*>(&lo);
        }
        return;
    }
    mode &= ~STIN_GEMM_W_FRAG;
    if (mode == 0) {
        base[row_off + c] = v;
        return;
    }
    if (mode == STIN_GEMM_W_BF16) {                      // plain bf16 row-major in the same buffer (bf16 element indices)
        reinterpret_cast<__bf16*>(base)[row_off + c] = (__bf16)v;
        return;
    }
    uint16_t* h = reinterpret_cast<uint16_t*>(base + row_off + (c & ~3)) + (c & 3);
    if (mode == STIN_GEMM_F16X3) {
        const float s = v * 64.f;
        const _Float16 hi = (_Float16)s;
        const _Float16 lo = (_Float16)(s - (float)hi);
        h[0] = *reinterpret_cast<const uint16_t*>(&hi);
        h[4] = *reinterpret_cast<const uint16_t*>(&lo);
    } else {
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        h[0] = *reinterpret_cast<const uint16_t*>(&hi);
        h[4] = *reinterpret_cast<const uint16_t*>(&lo);
    }
}

__global__ void k_split_weights(const float* __restrict__ W, int64_t ldw, int Nc, int K, int mode, float* __restrict__ out,
                                int64_t ldo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Nc * K) return;
    const int r = (int)(t / K), c = (int)(t % K);
    put_weight(out, (int64_t)r * ldo, c, W[(int64_t)r * ldw + c], mode, r, Nc, K);
}

// wcat [Yw, Cin] = [Wa - Wb ; Wb ; Ws]   (trans_inv: [-W1 ; W1 ; Ws]),  bcat [Yw] = [b1 ; 0 ; bs],
// wcatT [Cin, Yw] = wcat^T,  w2T [H, Cout] = W2^T,  w2s [Cout, H] = W2 (only when pre-split).  Yw = 2H (+ Cout with a
// shortcut).  fwd_mode / bwd_mode: the storage form (put_weight) of the forward operands (wcat, w2s) and of the
// backward operands (wcatT, w2T).
// element (r, c) of the packed operand wcat, c < Cin; and entry r of its bias bcat (compact trans-inv: b1 is NOT in bcat - the
// edge stage adds it when it forms A_i = b1 - B_i)
__device__ __forceinline__ float pack_value(const float* __restrict__ W1, const float* __restrict__ Ws, int Cin, int H, int ld1,
                                            int trans_inv, int r, int c) {
    if (trans_inv == STIN_TI_COMPACT) return r < H ? W1[(int64_t)r * ld1 + c] : Ws[(int64_t)(r - H) * Cin + c];
    if (r < H) return trans_inv ? -W1[(int64_t)r * ld1 + c] : W1[(int64_t)r * ld1 + c] - W1[(int64_t)r * ld1 + Cin + c];
    if (r < 2 * H) return trans_inv ? W1[(int64_t)(r - H) * ld1 + c] : W1[(int64_t)(r - H) * ld1 + Cin + c];
    return Ws[(int64_t)(r - 2 * H) * Cin + c];
}
__device__ __forceinline__ float pack_bias(const float* __restrict__ b1, const float* __restrict__ bs, int H, int trans_inv, int r) {
    if (trans_inv == STIN_TI_COMPACT) return r < H ? 0.f : (bs != nullptr ? bs[r - H] : 0.f);
    return r < H ? (b1 != nullptr ? b1[r] : 0.f) : (r < 2 * H ? 0.f : (bs != nullptr ? bs[r - 2 * H] : 0.f));
}

__device__ __forceinline__ void pack_body(int64_t t, const float* __restrict__ W1, const float* __restrict__ b1,
                                          const float* __restrict__ Ws, const float* __restrict__ bs,
                                          const float* __restrict__ W2, int Cin, int Cp, int H, int Cout, int has_shortcut,
                                          int trans_inv, float* __restrict__ wcat, float* __restrict__ bcat,
                                          float* __restrict__ wcatT, float* __restrict__ w2T, float* __restrict__ w2s,
                                          int fwd_mode, int bwd_mode) {
    const int Yw = stin_yw(H, Cout, has_shortcut, trans_inv);
    const int ld1 = trans_inv ? Cin : 2 * Cin;
    const int64_t n_w = (int64_t)Yw * Cp, n_2 = (int64_t)H * Cout;
    if (t < n_w) {
        const int r = (int)(t / Cp), c = (int)(t % Cp);
        float v = 0.f;                                   // zero padding columns c >= Cin (inner dimension padded to Cp)
        if (c < Cin) v = pack_value(W1, Ws, Cin, H, ld1, trans_inv, r, c);
        put_weight(wcat, (int64_t)r * Cp, c, v, fwd_mode, r, Yw, Cp);
        put_weight(wcatT, (int64_t)c * Yw, r, v, bwd_mode, c, Cp, Yw);
        if (c == 0) bcat[r] = pack_bias(b1, bs, H, trans_inv, r);
    } else if (t < n_w + n_2) {
        const int64_t u = t - n_w;
        const int k = (int)(u / Cout), o = (int)(u % Cout);      // w2T[k][o] = W2[o][k]
        const float v = W2[(int64_t)o * H + k];
        put_weight(w2T, (int64_t)k * Cout, o, v, bwd_mode, k, H, Cout);
        if (w2s != nullptr) put_weight(w2s, (int64_t)o * H, k, v, fwd_mode, o, Cout, H);
    }
}

// ---- round 4: the same pack with EIGHT consecutive k-elements of a destination row per thread.  pack_body writes one element per
// thread - 2-byte stores into the fragment / k-group layouts and 4-byte stores strided by a whole row into the transposed
// operands: ~80 us for the 32 MB of operands of the headline network at the head of every step, where the bytes take ~10.
// Here a thread forms 8 values and stores them as whole 16 / 32-byte pieces of the destination layout; the transposed operands
// read their sources coalesced ACROSS threads (consecutive threads = consecutive source columns).  Same values, same roundings,
// same layouts: bit-identical buffers (tests/test_hip_parity.py::test_pack8_equals_the_elementwise_pack).
__device__ __forceinline__ uint32_t pk16(uint16_t lo, uint16_t hi) { return (uint32_t)lo | ((uint32_t)hi << 16); }
__device__ __forceinline__ void split16(float v, int prec, uint16_t& hi, uint16_t& lo) {
    if (prec == STIN_GEMM_F16X3) {
        const float s = v * 64.f;
        const _Float16 h = (_Float16)s;
        const _Float16 l = (_Float16)(s - (float)h);
        hi = *reinterpret_cast<const uint16_t*>(&h);
        lo = *reinterpret_cast<const uint16_t*>(&l);
    } else {
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        hi = *reinterpret_cast<const uint16_t*>(&h);
        lo = *reinterpret_cast<const uint16_t*>(&l);
    }
}
// elements [c0, c0 + 8) of row r (c0 % 8 == 0) of an [nc, kk] operand, in the layout `mode` asks for (put_weight x 8)
__device__ __forceinline__ void put_weight8(float* __restrict__ base, int64_t row_off, int c0, const float (&v)[8], int mode, int r,
                                            int nc, int kk) {
    if ((mode & STIN_GEMM_W_FRAG) && stin_w_frag_shape(nc, kk)) {
        const int prec = mode & ~STIN_GEMM_W_FRAG;
        const int64_t lane_slot = ((int64_t)(r >> 5) * (kk >> 4) + (c0 >> 4)) * 64 + ((c0 >> 3) & 1) * 32 + (r & 31);
        uint16_t hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split16(v[e], prec, hi[e], lo[e]);
        uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + lane_slot * 16);
        dst[0] = make_uint4(pk16(hi[0], hi[1]), pk16(hi[2], hi[3]), pk16(hi[4], hi[5]), pk16(hi[6], hi[7]));
        dst[1] = make_uint4(pk16(lo[0], lo[1]), pk16(lo[2], lo[3]), pk16(lo[4], lo[5]), pk16(lo[6], lo[7]));
        return;
    }
    mode &= ~STIN_GEMM_W_FRAG;
    if (mode == 0) {
        float4* dst = reinterpret_cast<float4*>(base + row_off + c0);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        return;
    }
    if (mode == STIN_GEMM_W_BF16) {
        uint16_t h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 b = (__bf16)v[e];
            h[e] = *reinterpret_cast<const uint16_t*>(&b);
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(base) + row_off + c0) =
            make_uint4(pk16(h[0], h[1]), pk16(h[2], h[3]), pk16(h[4], h[5]), pk16(h[6], h[7]));
        return;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {                                   // k-group layout: 16 bytes = [hi x 4 | lo x 4] per 4 elements
        uint16_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split16(v[4 * g + e], mode, hi[e], lo[e]);
        *reinterpret_cast<uint4*>(base + row_off + c0 + 4 * g) =
            make_uint4(pk16(hi[0], hi[1]), pk16(hi[2], hi[3]), pk16(lo[0], lo[1]), pk16(lo[2], lo[3]));
    }
}

__device__ __forceinline__ bool pack8_shape(int Cp, int H, int Cout, int has_shortcut, int trans_inv) {
    const int Yw = stin_yw(H, Cout, has_shortcut, trans_inv);
    return Cp % 8 == 0 && Yw % 8 == 0 && H % 8 == 0 && Cout % 8 == 0;
}

__device__ __forceinline__ void pack_body8(int64_t t, const float* __restrict__ W1, const float* __restrict__ b1,
                                           const float* __restrict__ Ws, const float* __restrict__ bs,
                                           const float* __restrict__ W2, int Cin, int Cp, int H, int Cout, int has_shortcut,
                                           int trans_inv, float* __restrict__ wcat, float* __restrict__ bcat,
                                           float* __restrict__ wcatT, float* __restrict__ w2T, float* __restrict__ w2s,
                                           int fwd_mode, int bwd_mode) {
    const int Yw = stin_yw(H, Cout, has_shortcut, trans_inv);
    const int ld1 = trans_inv ? Cin : 2 * Cin;
    auto val = [&](int r, int c) -> float {                        // wcat[r][c] (pack_body's expression)
        if (c >= Cin) return 0.f;
        return pack_value(W1, Ws, Cin, H, ld1, trans_inv, r, c);
    };
    const int64_t nA = (int64_t)Yw * (Cp / 8), nB = (int64_t)Cp * (Yw / 8), nC = (int64_t)H * (Cout / 8),
                  nD = w2s != nullptr ? (int64_t)Cout * (H / 8) : 0;
    float v[8];
    if (t < nA) {                                                  // wcat rows: 8 consecutive input channels
        const int r = (int)(t / (Cp / 8)), c0 = (int)(t % (Cp / 8)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = val(r, c0 + e);
        put_weight8(wcat, (int64_t)r * Cp, c0, v, fwd_mode, r, Yw, Cp);
        if (c0 == 0) bcat[r] = pack_bias(b1, bs, H, trans_inv, r);
    } else if ((t -= nA) < nB) {                                   // wcatT rows (= input channel c): 8 consecutive output rows; c fastest
        const int c = (int)(t % Cp), r0 = (int)(t / Cp) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = val(r0 + e, c);
        put_weight8(wcatT, (int64_t)c * Yw, r0, v, bwd_mode, c, Cp, Yw);
    } else if ((t -= nB) < nC) {                                   // w2T rows (= hidden channel k): 8 consecutive outputs; k fastest
        const int k = (int)(t % H), o0 = (int)(t / H) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W2[(int64_t)(o0 + e) * H + k];
        put_weight8(w2T, (int64_t)k * Cout, o0, v, bwd_mode, k, H, Cout);
    } else if ((t -= nC) < nD) {                                   // w2s rows (= output o): 8 consecutive hidden channels
        const int o = (int)(t / (H / 8)), k0 = (int)(t % (H / 8)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W2[(int64_t)o * H + k0 + e];
        put_weight8(w2s, (int64_t)o * H, k0, v, fwd_mode, o, Cout, H);
    }
}

__global__ void k_pack(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ Ws,
                       const float* __restrict__ bs, const float* __restrict__ W2, int Cin, int Cp, int H, int Cout,
                       int has_shortcut, int trans_inv, float* __restrict__ wcat, float* __restrict__ bcat,
                       float* __restrict__ wcatT, float* __restrict__ w2T, float* __restrict__ w2s, int fwd_mode,
                       int bwd_mode, int pack8) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pack8 && pack8_shape(Cp, H, Cout, has_shortcut, trans_inv))
        pack_body8(t, W1, b1, Ws, bs, W2, Cin, Cp, H, Cout, has_shortcut, trans_inv, wcat, bcat, wcatT, w2T, w2s, fwd_mode, bwd_mode);
    else
        pack_body(t, W1, b1, Ws, bs, W2, Cin, Cp, H, Cout, has_shortcut, trans_inv, wcat, bcat, wcatT, w2T, w2s, fwd_mode, bwd_mode);
}

// Every block of a network in ONE launch: blockIdx.y picks the job (a table in device memory, written once per model).
__global__ void k_pack_many(const stin_pack_job_t* __restrict__ jobs, int pack8) {
    const stin_pack_job_t j = jobs[blockIdx.y];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pack8 && pack8_shape(j.Cp, j.H, j.Cout, j.has_shortcut, j.trans_inv))       // (block-uniform: one job per blockIdx.y)
        pack_body8(t, j.W1, j.b1, j.Ws, j.bs, j.W2, j.Cin, j.Cp, j.H, j.Cout, j.has_shortcut, j.trans_inv, j.wcat, j.bcat, j.wcatT, j.w2T,
                   j.w2s, j.fwd_split, j.bwd_split);
    else
        pack_body(t, j.W1, j.b1, j.Ws, j.bs, j.W2, j.Cin, j.Cp, j.H, j.Cout, j.has_shortcut, j.trans_inv, j.wcat, j.bcat, j.wcatT, j.w2T,
                  j.w2s, j.fwd_split, j.bwd_split);
}

// dwb [Yw, Cin + 1] (weight grad | bias grad of the packed operand) -> grads of the reference-layout
// parameters: dW1 [H, Cin or 2Cin], db1 [H], dWs [Cout, Cin], dbs [Cout].
__global__ void k_unpack(const float* __restrict__ dwb, const float* __restrict__ dw2b, int Cin, int Cp, int H,
                         int Cout, int has_shortcut, int trans_inv, float* __restrict__ dW1, float* __restrict__ db1,
                         float* __restrict__ dWs, float* __restrict__ dbs, float* __restrict__ dW2,
                         float* __restrict__ db2) {
    const int ld = Cp + 1;                                // dwb rows: Cp weight-gradient columns (Cin used) | bias gradient
    const int ld1 = trans_inv ? Cin : 2 * Cin;
    const int64_t n1 = (int64_t)H * ld1, ns = has_shortcut ? (int64_t)Cout * Cin : 0;
    const int64_t n2 = dw2b != nullptr ? (int64_t)Cout * H : 0;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n1 + ns && t < n1 + ns + n2) {          // dw2b [Cout, H + 1] = dW2 | db2 -> contiguous dW2, db2
        const int64_t u = t - n1 - ns;
        const int r = (int)(u / H), c = (int)(u % H);
        dW2[u] = dw2b[(int64_t)r * (H + 1) + c];
        if (c == 0 && db2 != nullptr) db2[r] = dw2b[(int64_t)r * (H + 1) + H];
        return;
    }
    const int s0 = trans_inv == STIN_TI_COMPACT ? H : 2 * H;   // first shortcut row of dwb
    if (t < n1) {
        const int r = (int)(t / ld1), c = (int)(t % ld1);
        float v;
        if (trans_inv == STIN_TI_COMPACT) {                    // the operand IS W1: its gradient rows as they are; db1 comes from the
            dW1[t] = dwb[(int64_t)r * ld + c];                 // edge stage's column sums of dA (not from this product)
            return;
        }
        if (trans_inv) v = dwb[(int64_t)(H + r) * ld + c] - dwb[(int64_t)r * ld + c];              // d/dW1 of (-W1, W1)
        else if (c < Cin) v = dwb[(int64_t)r * ld + c];                                            // Wa
        else v = dwb[(int64_t)(H + r) * ld + (c - Cin)] - dwb[(int64_t)r * ld + (c - Cin)];        // Wb
        dW1[t] = v;
        if (c == 0 && db1 != nullptr) db1[r] = dwb[(int64_t)r * ld + Cp];
    } else if (t < n1 + ns) {
        const int64_t u = t - n1;
        const int r = (int)(u / Cin), c = (int)(u % Cin);
        dWs[u] = dwb[(int64_t)(s0 + r) * ld + c];
        if (c == 0 && dbs != nullptr) dbs[r] = dwb[(int64_t)(s0 + r) * ld + Cp];
    }
}

// k = -rstd^3 T1 inv_cnt ; m = -rstd S0 inv_cnt   (instance-norm backward, slices == graphs)
__global__ void k_norm_coef(const float* __restrict__ T1, const float* __restrict__ S0, const float* __restrict__ rstd,
                            const float* __restrict__ inv_cnt, int B, int C, float* __restrict__ kk,
                            float* __restrict__ mm) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const float r = rstd[t], ic = inv_cnt[t / C];
    kk[t] = -(r * r * r) * T1[t] * ic;
    mm[t] = -(r * S0[t]) * ic;
}

// linspace-slice quirk: m = -(rstd S0 + U) inv_cnt with U = sum over the slice of k xc (separately rounded mul / add)
__global__ void k_norm_coef_m_quirk(const float* __restrict__ S0, const float* __restrict__ U, const float* __restrict__ rstd,
                                    const float* __restrict__ inv_cnt, int B, int C, float* __restrict__ mm) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    mm[t] = __fmul_rn(-__fadd_rn(__fmul_rn(rstd[t], S0[t]), U[t]), inv_cnt[t / C]);
}
}  // namespace

// STIN_PACK8=0 keeps the one-element-per-thread pack (A/B and test switch, re-read per call)
static inline int pack8_on() {
    const char* e = getenv("STIN_PACK8");
    return (e != nullptr && atoi(e) == 0) ? 0 : 1;
}

static inline bool split_mode_ok(int m) {
    if (m == 0) return true;
    m &= ~STIN_GEMM_W_FRAG;
    return m == STIN_GEMM_BF16X3 || m == STIN_GEMM_F16X3;
}

extern "C" int stin_gemm_w_is_frag(int Nc, int K) { return stin_w_frag_shape(Nc, K) ? 1 : 0; }

extern "C" int stin_gemm_split_weights_f32(const float* W, int64_t ldw, int Nc, int K, int precision, float* out, int64_t ldo,
                                           stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(Nc > 0 && K > 0 && ldw >= K && ldo >= K, STIN_E_SIZE);
    STIN_REQUIRE(W && out, STIN_E_NULL);
    STIN_REQUIRE(split_mode_ok(precision) && precision != 0, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(K % 4 == 0 && ldo % 4 == 0 && stin_aligned16(out), STIN_E_ALIGN);
    const int64_t n = (int64_t)Nc * K;
    hipLaunchKernelGGL(k_split_weights, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream_, W, ldw,
                       Nc, K, precision, out, ldo);
    return stin_launch_status();
}

extern "C" int stin_edgeconv_pack_f32(const float* W1, const float* b1, const float* Ws, const float* bs, const float* W2,
                                      int Cin, int Cp, int H, int Cout, int has_shortcut, int trans_inv, float* wcat,
                                      float* bcat, float* wcatT, float* w2T, float* w2s, int fwd_split, int bwd_split,
                                      stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(Cin > 0 && Cp >= Cin && H > 0 && Cout > 0, STIN_E_SIZE);
    STIN_REQUIRE(W1 && W2 && wcat && bcat && wcatT && w2T && (!has_shortcut || Ws), STIN_E_NULL);
    STIN_REQUIRE((split_mode_ok(fwd_split) || fwd_split == STIN_GEMM_W_BF16) && (split_mode_ok(bwd_split) || bwd_split == STIN_GEMM_W_BF16),
                 STIN_E_UNSUPPORTED);
    STIN_REQUIRE(fwd_split == 0 || (w2s != nullptr && Cp % 4 == 0 && H % 4 == 0), STIN_E_ALIGN);
    STIN_REQUIRE(trans_inv >= 0 && trans_inv <= STIN_TI_COMPACT, STIN_E_UNSUPPORTED);
    STIN_REQUIRE(bwd_split == 0 || (Cout % 4 == 0 && H % (trans_inv == STIN_TI_COMPACT ? 4 : 2) == 0), STIN_E_ALIGN);   // Yw a multiple of 4
    const int Yw = stin_yw(H, Cout, has_shortcut, trans_inv);
    const int64_t n = (int64_t)Yw * Cp + (int64_t)H * Cout;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream_, W1, b1, Ws,
                       bs, W2, Cin, Cp, H, Cout, has_shortcut, trans_inv, wcat, bcat, wcatT, w2T, w2s, fwd_split, bwd_split, pack8_on());
    return stin_launch_status();
}

extern "C" int stin_edgeconv_pack_many_f32(const stin_pack_job_t* jobs_device, int n_jobs, int64_t max_elems,
                                           stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(n_jobs >= 0 && max_elems >= 0 && n_jobs <= 65535, STIN_E_SIZE);
    if (n_jobs == 0 || max_elems == 0) return STIN_OK;
    STIN_REQUIRE(jobs_device != nullptr, STIN_E_NULL);
    hipLaunchKernelGGL(k_pack_many, dim3((unsigned)((max_elems + BLOCK - 1) / BLOCK), (unsigned)n_jobs), dim3(BLOCK), 0,
                       (hipStream_t)stream_, jobs_device, pack8_on());
    return stin_launch_status();
}

extern "C" int stin_edgeconv_unpack_grads_f32(const float* dwb, const float* dw2b, int Cin, int Cp, int H, int Cout,
                                              int has_shortcut, int trans_inv, float* dW1, float* db1, float* dWs,
                                              float* dbs, float* dW2, float* db2, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(Cin > 0 && Cp >= Cin && H > 0 && Cout > 0, STIN_E_SIZE);
    STIN_REQUIRE(dwb && dW1 && (!has_shortcut || dWs) && (dw2b == nullptr || dW2 != nullptr), STIN_E_NULL);
    const int64_t n = (int64_t)H * (trans_inv ? Cin : 2 * Cin) + (has_shortcut ? (int64_t)Cout * Cin : 0) +
                      (dw2b != nullptr ? (int64_t)Cout * H : 0);
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream_, dwb, dw2b,
                       Cin, Cp, H, Cout, has_shortcut, trans_inv, dW1, db1, dWs, dbs, dW2, db2);
    return stin_launch_status();
}

extern "C" int stin_norm_bwd_coef_f32(const float* T1, const float* S0, const float* rstd, const float* inv_cnt, int B,
                                      int C, float* k, float* m, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(B > 0 && C > 0, STIN_E_SIZE);
    STIN_REQUIRE(T1 && S0 && rstd && inv_cnt && k && m, STIN_E_NULL);
    hipLaunchKernelGGL(k_norm_coef, dim3((unsigned)((B * C + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream_, T1,
                       S0, rstd, inv_cnt, B, C, k, m);
    return stin_launch_status();
}

extern "C" int stin_norm_bwd_coef_m_quirk_f32(const float* S0, const float* U, const float* rstd, const float* inv_cnt, int B,
                                              int C, float* m, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(B > 0 && C > 0, STIN_E_SIZE);
    STIN_REQUIRE(S0 && U && rstd && inv_cnt && m, STIN_E_NULL);
    hipLaunchKernelGGL(k_norm_coef_m_quirk, dim3((unsigned)((B * C + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream_,
                       S0, U, rstd, inv_cnt, B, C, m);
    return stin_launch_status();
}

// ------------------------------------------------------------------ train-step epilogues
// Masked, distance-weighted L1 loss of the inpainting trainer and its gradient in one pass
// (reference trainers/inpainting3d_trainer.py:127-137):
//   pred = mask > 0 ? out : color ; loss = mean_{v,c} |pred - color| * 0.99^mask_v
//   dloss/dout[v,c] = mask_v > 0 ? sign(out - color) * 0.99^mask_v / (N*C) : 0
// partial[block] holds each block's fp64 sum (fixed order); k_loss_final adds them.
namespace {
__global__ __launch_bounds__(256) void k_masked_l1(const float* __restrict__ out, const float* __restrict__ color,
                                                   const int64_t* __restrict__ mask, int64_t N, int C, int use_weight,
                                                   float inv_count, float* __restrict__ grad,
                                                   double* __restrict__ partial) {
    __shared__ double sm[256];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (t < N * C) {
        const int64_t v = t / C;
        const int64_t m = mask[v];
        const float w = use_weight ? powf(0.99f, (float)m) : 1.f;
        float d = 0.f, g = 0.f;
        if (m > 0) {
            d = out[t] - color[t];
            g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w * inv_count;
        }
        grad[t] = g;
        acc = (double)(fabsf(d) * w);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void k_loss_final(const double* __restrict__ partial, int n, float inv_count,
                                                    float* __restrict__ loss) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sm[0] * (double)inv_count);
}

// graph total variation (utils/metrics/graph_metrics.py:34-38): sum over the directed edges of |x_src - x_dst| over all
// channels.  One thread per destination row walks its in-edges (C is 1 or 3 for the metric: a row is one 4..12-byte gather);
// fp64 block partials in a fixed order, k_loss_final sums them and applies 1 / (N * C).
__global__ __launch_bounds__(256) void k_total_variation(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, int64_t N, int C,
                                                         double* __restrict__ partial) {
    __shared__ double sm[256];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (i < N) {
        const float* xi = x + i * ldx;
        for (int32_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const float* xj = x + (int64_t)col[e] * ldx;
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += fabsf(xj[c] - xi[c]);
            acc += (double)s;
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

// Adam with amsgrad over one flat parameter buffer, the arithmetic of torch.optim.Adam's single-tensor path
// (weight_decay added to the gradient; exp_avg by lerp; the scalar factors come from the host in double):
//   m += (1-b1) (g - m) ; v = b2 v + (1-b2) g^2 ; vmax = max(vmax, v)
//   p += (-lr / (1 - b1^t)) * m / (sqrt(vmax) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, float* __restrict__ vmax, int64_t n, float neg_step_size,
                                              float one_minus_b1, float b2, float one_minus_b2, float eps, float wd,
                                              float bc2_sqrt, int amsgrad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float m0 = m[i];
    const float mi = m0 + one_minus_b1 * (gi - m0);
    const float vi = b2 * v[i] + one_minus_b2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float vh = vi;
    if (amsgrad) {
        vh = fmaxf(vmax[i], vi);
        vmax[i] = vh;
    }
    const float denom = sqrtf(vh) / bc2_sqrt + eps;
    p[i] = pi + neg_step_size * mi / denom;
}
// the same update on four elements per lane (16-byte loads / stores; round 3: the scalar form moved 151 MB in 64 us).  Identical
// arithmetic per element -> identical bits.
__global__ __launch_bounds__(256) void k_adam4(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                               float4* __restrict__ v, float4* __restrict__ vmax, int64_t n4, int tail,
                                               float neg_step_size, float one_minus_b1, float b2, float one_minus_b2, float eps,
                                               float wd, float bc2_sqrt, int amsgrad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tail) {                                   // the n % 4 ragged elements ride on the first lanes (no second launch)
        float* ps = reinterpret_cast<float*>(p) + 4 * n4 + i;
        const float* gs = reinterpret_cast<const float*>(g) + 4 * n4 + i;
        float* ms = reinterpret_cast<float*>(m) + 4 * n4 + i;
        float* vs = reinterpret_cast<float*>(v) + 4 * n4 + i;
        float gi = *gs;
        const float pi = *ps;
        if (wd != 0.f) gi += wd * pi;
        const float m0 = *ms;
        const float mi = m0 + one_minus_b1 * (gi - m0);
        const float vi = b2 * *vs + one_minus_b2 * gi * gi;
        *ms = mi;
        *vs = vi;
        float vh = vi;
        if (amsgrad) {
            float* xs = reinterpret_cast<float*>(vmax) + 4 * n4 + i;
            vh = fmaxf(*xs, vi);
            *xs = vh;
        }
        const float denom = sqrtf(vh) / bc2_sqrt + eps;
        *ps = pi + neg_step_size * mi / denom;
    }
    if (i >= n4) return;
    const float4 g4 = g[i], p4 = p[i], m4 = m[i], v4 = v[i];
    const float4 x4 = amsgrad ? vmax[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
    const float vv[4] = {v4.x, v4.y, v4.z, v4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w};
    float po[4], mo[4], vo[4], xo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float gi = gg[e];
        const float pi = pp[e];
        if (wd != 0.f) gi += wd * pi;
        const float m0 = mm[e];
        const float mi = m0 + one_minus_b1 * (gi - m0);
        const float vi = b2 * vv[e] + one_minus_b2 * gi * gi;
        float vh = vi;
        if (amsgrad) vh = fmaxf(xx[e], vi);
        const float denom = sqrtf(vh) / bc2_sqrt + eps;
        mo[e] = mi;
        vo[e] = vi;
        xo[e] = vh;
        po[e] = pi + neg_step_size * mi / denom;
    }
    m[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
    v[i] = make_float4(vo[0], vo[1], vo[2], vo[3]);
    if (amsgrad) vmax[i] = make_float4(xo[0], xo[1], xo[2], xo[3]);
    p[i] = make_float4(po[0], po[1], po[2], po[3]);
}
}  // namespace

extern "C" size_t stin_masked_l1_workspace_bytes(int64_t N, int C) {
    if (N < 0 || C <= 0) return 0;
    return (size_t)((N * C + 255) / 256 + 1) * sizeof(double) + 256;
}

extern "C" int stin_masked_l1_loss_f32(const float* out, const float* color, const int64_t* mask, int64_t N, int C,
                                       int use_weight, float* loss, float* grad, void* workspace, size_t workspace_bytes,
                                       stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(N > 0 && C > 0, STIN_E_SIZE);
    STIN_REQUIRE(out && color && mask && loss && grad && workspace, STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_masked_l1_workspace_bytes(N, C), STIN_E_WORKSPACE);
    double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t n = N * C;
    const int blocks = (int)((n + 255) / 256);
    const float inv = 1.0f / (float)n;
    hipLaunchKernelGGL(k_masked_l1, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, out, color, mask, N, C,
                       use_weight, inv, grad, partial);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, (hipStream_t)stream_, partial, blocks, inv, loss);
    return stin_launch_status();
}

// graph Laplace of the trainer's per-step metric (utils/metrics/graph_metrics.py:6-16): out[i, c] = sum_{j in N(i)} x[j, c] - deg_i x[i, c].
// The reference propagates [1 | x] with aggr = 'add' (a scatter-add in edge order) and subtracts prop[:, 0] * x: here one thread per
// (vertex, channel) walks the destination-CSR row in the same order (fp32 adds in edge order; the degree is exact) - bit for bit the
// same numbers as the torch.cat + segment-sum + elementwise composition it replaces, without the [N, C + 1] temporaries.
__global__ __launch_bounds__(256) void k_graph_laplace(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, int64_t N, int C, float* __restrict__ out,
                                                       int64_t ldo) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= N * C) return;
    const int64_t i = t / C;
    const int c = (int)(t % C);
    const int beg = rowptr[i], end = rowptr[i + 1];
    float s = 0.f;
    for (int e = beg; e < end; ++e) s += x[(int64_t)col[e] * ldx + c];
    out[i * ldo + c] = s - (float)(end - beg) * x[i * ldx + c];
}

extern "C" int stin_graph_laplace_f32(const float* x, int64_t ldx, const int32_t* rowptr_dst, const int32_t* col_dst, int64_t N, int C,
                                      float* out, int64_t ldo, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && C > 0 && ldx >= C && ldo >= C, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && rowptr_dst && col_dst && out, STIN_E_NULL);
    const int64_t blocks = (N * C + 255) / 256;
    hipLaunchKernelGGL(k_graph_laplace, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, x, ldx, rowptr_dst, col_dst, N, C, out, ldo);
    return stin_launch_status();
}

extern "C" size_t stin_total_variation_workspace_bytes(int64_t N) {
    if (N < 0) return 0;
    return (size_t)((N + 255) / 256 + 1) * sizeof(double) + 256;
}

extern "C" int stin_total_variation_f32(const float* x, int64_t ldx, const int32_t* rowptr_dst, const int32_t* col_dst, int64_t N,
                                        int C, float* out, void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(N > 0 && C > 0 && ldx >= C, STIN_E_SIZE);
    STIN_REQUIRE(x && rowptr_dst && out && workspace, STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_total_variation_workspace_bytes(N), STIN_E_WORKSPACE);
    double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int blocks = (int)((N + 255) / 256);
    hipLaunchKernelGGL(k_total_variation, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, x, ldx, rowptr_dst, col_dst, N,
                       C, partial);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, (hipStream_t)stream_, partial, blocks, 1.0f / ((float)N * (float)C), out);
    return stin_launch_status();
}

extern "C" int stin_adam_f32(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int step, int amsgrad, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(n >= 0 && step >= 1, STIN_E_SIZE);
    if (n == 0) return STIN_OK;
    STIN_REQUIRE(p && g && m && v && (!amsgrad || vmax), STIN_E_NULL);
    // bias corrections in double on the host, as torch computes them (python floats)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2s = sqrt(1.0 - pow(beta2, (double)step));
    const bool vec = stin_aligned16(p) && stin_aligned16(g) && stin_aligned16(m) && stin_aligned16(v) && (!amsgrad || stin_aligned16(vmax));
    const int64_t n4 = vec ? n / 4 : 0;
    if (n4 > 0) {
        hipLaunchKernelGGL(k_adam4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, reinterpret_cast<float4*>(p),
                           reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v),
                           reinterpret_cast<float4*>(vmax), n4, (int)(n - 4 * n4), (float)(-(lr / bc1)), (float)(1.0 - beta1),
                           (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)bc2s, amsgrad);
    } else {                                                            // (an unaligned buffer, or fewer than four elements)
        hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, p, g, m, v, vmax, n,
                           (float)(-(lr / bc1)), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                           (float)weight_decay, (float)bc2s, amsgrad);
    }
    return stin_launch_status();
}
