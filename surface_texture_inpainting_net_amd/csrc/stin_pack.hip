// Small parameter-side kernels: (un)packing of the EdgeConv weights into the per-vertex GEMM operands of
// the restructured block and the instance-norm backward coefficients.  One launch each
// instead of dozens of tiny framework ops per block and step.
#include "stin_common.h"

namespace {
constexpr int BLOCK = 256;

// wcat [Yw, Cin] = [Wa - Wb ; Wb ; Ws]   (trans_inv: [-W1 ; W1 ; Ws]),  bcat [Yw] = [b1 ; 0 ; bs],
// wcatT [Cin, Yw] = wcat^T,  w2T [H, Cout] = W2^T.   Yw = 2H (+ Cout with a shortcut).
__global__ void k_pack(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ Ws,
                       const float* __restrict__ bs, const float* __restrict__ W2, int Cin, int Cp, int H, int Cout,
                       int has_shortcut, int trans_inv, float* __restrict__ wcat, float* __restrict__ bcat,
                       float* __restrict__ wcatT, float* __restrict__ w2T) {
    const int Yw = 2 * H + (has_shortcut ? Cout : 0);
    const int ld1 = trans_inv ? Cin : 2 * Cin;
    const int64_t n_w = (int64_t)Yw * Cp, n_2 = (int64_t)H * Cout;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_w) {
        const int r = (int)(t / Cp), c = (int)(t % Cp);
        float v = 0.f;                                   // zero padding columns c >= Cin (inner dimension padded to Cp)
        if (c < Cin) {
            if (r < H) v = trans_inv ? -W1[(int64_t)r * ld1 + c] : W1[(int64_t)r * ld1 + c] - W1[(int64_t)r * ld1 + Cin + c];
            else if (r < 2 * H) v = trans_inv ? W1[(int64_t)(r - H) * ld1 + c] : W1[(int64_t)(r - H) * ld1 + Cin + c];
            else v = Ws[(int64_t)(r - 2 * H) * Cin + c];
        }
        wcat[t] = v;
        wcatT[(int64_t)c * Yw + r] = v;
        if (c == 0) bcat[r] = r < H ? (b1 != nullptr ? b1[r] : 0.f) : (r < 2 * H ? 0.f : (bs != nullptr ? bs[r - 2 * H] : 0.f));
    } else if (t < n_w + n_2) {
        const int64_t u = t - n_w;
        const int k = (int)(u / Cout), o = (int)(u % Cout);      // w2T[k][o] = W2[o][k]
        w2T[u] = W2[(int64_t)o * H + k];
    }
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
    if (t < n1) {
        const int r = (int)(t / ld1), c = (int)(t % ld1);
        float v;
        if (trans_inv) v = dwb[(int64_t)(H + r) * ld + c] - dwb[(int64_t)r * ld + c];              // d/dW1 of (-W1, W1)
        else if (c < Cin) v = dwb[(int64_t)r * ld + c];                                            // Wa
        else v = dwb[(int64_t)(H + r) * ld + (c - Cin)] - dwb[(int64_t)r * ld + (c - Cin)];        // Wb
        dW1[t] = v;
        if (c == 0 && db1 != nullptr) db1[r] = dwb[(int64_t)r * ld + Cp];
    } else if (t < n1 + ns) {
        const int64_t u = t - n1;
        const int r = (int)(u / Cin), c = (int)(u % Cin);
        dWs[u] = dwb[(int64_t)(2 * H + r) * ld + c];
        if (c == 0 && dbs != nullptr) dbs[r] = dwb[(int64_t)(2 * H + r) * ld + Cp];
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
}  // namespace

extern "C" int stin_edgeconv_pack_f32(const float* W1, const float* b1, const float* Ws, const float* bs, const float* W2,
                                      int Cin, int Cp, int H, int Cout, int has_shortcut, int trans_inv, float* wcat,
                                      float* bcat, float* wcatT, float* w2T, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(Cin > 0 && Cp >= Cin && H > 0 && Cout > 0, STIN_E_SIZE);
    STIN_REQUIRE(W1 && W2 && wcat && bcat && wcatT && w2T && (!has_shortcut || Ws), STIN_E_NULL);
    const int Yw = 2 * H + (has_shortcut ? Cout : 0);
    const int64_t n = (int64_t)Yw * Cp + (int64_t)H * Cout;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream_, W1, b1, Ws,
                       bs, W2, Cin, Cp, H, Cout, has_shortcut, trans_inv, wcat, bcat, wcatT, w2T);
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
