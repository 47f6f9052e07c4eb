// SingleConvMeshNet (reference models/singleconvmeshnet.py:10-156, models/modules/edge_conv_filter.py:34-44):
// the small per-layer passes of the fused EdgeConv(BN) layer that used to be framework elementwise kernels - operand packing of
// the first Linear, the N-row epilogue (BatchNorm affine of the aggregated rows + "has an in-edge" mask + residual + ReLU) and the
// head of its backward (ReLU mask of the layer output, the in-edge mask).  fp32, gfx950.  Contract: include/stin_hip.h.
#include "stin_common.h"

namespace {

constexpr int BLOCK = 256;

template <int VW> __device__ __forceinline__ void ldv(const float* p, float (&v)[VW]) {
    if (VW == 4) {
        const float4 t = ld4(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[VW - 1] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < VW; ++i) v[i] = p[i];
    }
}
template <int VW> __device__ __forceinline__ void stv(float* p, const float (&v)[VW]) {
    if (VW == 4) st4(p, make_float4(v[0], v[1], v[2], v[VW - 1]));
    else {
#pragma unroll
        for (int i = 0; i < VW; ++i) p[i] = v[i];
    }
}

// one thread per element of the largest output; every output is written by the thread whose index falls inside it
__global__ __launch_bounds__(BLOCK) void k_scmn_pack(const float* __restrict__ W1, const float* __restrict__ W2,
                                                     const float* __restrict__ g1, const float* __restrict__ b1,
                                                     const float* __restrict__ g2, const float* __restrict__ b2, int cin, int h2,
                                                     int cout, int trans_inv, float* __restrict__ wcat, float* __restrict__ wcatT,
                                                     float* __restrict__ w2T, float* __restrict__ gb1, float* __restrict__ gb2) {
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    const int n1 = 2 * h2 * cin, n2 = h2 * cout;
    if (t < n1) {                                       // wcat[r, c]: rows [0, h2) = A operand, [h2, 2 h2) = B operand
        const int r = t / cin, c = t % cin;
        float v;
        if (trans_inv) {                                // Lin1(x_j - x_i): A = -W1, B = W1
            const float w = W1[(r % h2) * cin + c];
            v = r < h2 ? -w : w;
        } else {                                        // Lin1([x_i ; x_j - x_i]) = (Wa - Wb) x_i + Wb x_j
            const float wb = W1[(r % h2) * 2 * cin + cin + c];
            v = r < h2 ? W1[r * 2 * cin + c] - wb : wb;
        }
        wcat[t] = v;
        wcatT[(int64_t)c * 2 * h2 + r] = v;
    }
    if (t < n2) {                                       // w2T[k, o] = W2[o, k]
        const int k = t / cout, o = t % cout;
        w2T[t] = W2[o * h2 + k];
    }
    if (t < h2) {
        gb1[t] = g1[t];
        gb1[h2 + t] = b1[t];
    }
    if (t < cout) {
        gb2[t] = g2[t];
        gb2[cout + t] = b2[t];
    }
}

// dW1 from the gradient of the packed operand dwcat [2 h2, cin]
__global__ __launch_bounds__(BLOCK) void k_scmn_unpack(const float* __restrict__ dwcat, int cin, int h2, int trans_inv,
                                                       float* __restrict__ dW1) {
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= h2 * cin) return;
    const int r = t / cin, c = t % cin;
    const float da = dwcat[r * cin + c], db = dwcat[(h2 + r) * cin + c];
    if (trans_inv) {
        dW1[t] = db - da;
    } else {
        dW1[r * 2 * cin + c] = da;                      // d/dWa
        dW1[r * 2 * cin + cin + c] = db - da;           // d/dWb
    }
}

// y = [relu]( res + [row has an in-edge] * (gamma ((x - mean) rstd) + beta) ): same float operations, in the same order, as the
// framework expression it replaces (stin_bn_act_fwd_f32, then `* has_in`, `res + .`, relu)
template <int VW>
__global__ __launch_bounds__(BLOCK) void k_bn_affine_res(const float* __restrict__ x, int64_t ldx, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const int32_t* __restrict__ rowptr,
                                                         const float* __restrict__ res, int64_t ldres, int64_t N, int C, int relu,
                                                         float* __restrict__ y, int64_t ldy) {
    const int CV = C / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N * CV) return;
    const int64_t r = t / CV;
    const int c = (int)(t % CV) * VW;
    const float hi = (rowptr == nullptr || rowptr[r + 1] > rowptr[r]) ? 1.f : 0.f;
    float xv[VW], rv[VW], o[VW];
    ldv<VW>(x + r * ldx + c, xv);
    if (res != nullptr) ldv<VW>(res + r * ldres + c, rv);
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        float z = gamma[c + i] * ((xv[i] - mean[c + i]) * rstd[c + i]) + beta[c + i];
        z = z * hi;
        if (res != nullptr) z = rv[i] + z;
        if (relu) z = z > 0.f ? z : 0.f;
        o[i] = z;
    }
    stv<VW>(y + r * ldy + c, o);
}

// head of the backward: g_eff = g [y > 0] (ReLU of the layer output; = g without it), g_in = g_eff [row has an in-edge]
template <int VW>
__global__ __launch_bounds__(BLOCK) void k_relu_mask_bwd(const float* __restrict__ g, int64_t ldg, const float* __restrict__ y,
                                                         int64_t ldy, const int32_t* __restrict__ rowptr, int64_t N, int C, int relu,
                                                         float* __restrict__ g_eff, float* __restrict__ g_in) {
    const int CV = C / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N * CV) return;
    const int64_t r = t / CV;
    const int c = (int)(t % CV) * VW;
    const bool has = rowptr == nullptr || rowptr[r + 1] > rowptr[r];
    float gv[VW], yv[VW], a[VW], b[VW];
    ldv<VW>(g + r * ldg + c, gv);
    if (relu) ldv<VW>(y + r * ldy + c, yv);
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        float v = gv[i];
        if (relu && !(yv[i] > 0.f)) v = 0.f;
        a[i] = v;
        b[i] = has ? v : 0.f;
    }
    if (g_eff != nullptr) stv<VW>(g_eff + r * (int64_t)C + c, a);
    stv<VW>(g_in + r * (int64_t)C + c, b);
}

// out[r] = [ skip[r, :cs] | coarse[trace[r], :cu] ]: the decoder's `torch.cat((skip, unpooled), -1)` with the unpool gather writing
// straight into its half (one thread per 4 (VW) output columns)
template <int VW>
__global__ __launch_bounds__(BLOCK) void k_concat_unpool(const float* __restrict__ skip, int64_t ld_skip, const float* __restrict__ coarse,
                                                         int64_t ld_c, const int32_t* __restrict__ trace, int64_t N, int cs, int cu,
                                                         float* __restrict__ out, int64_t ldo) {
    const int CV = (cs + cu) / VW;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N * CV) return;
    const int64_t r = t / CV;
    const int c = (int)(t % CV) * VW;
    float v[VW];
    if (c < cs) ldv<VW>(skip + r * ld_skip + c, v);
    else ldv<VW>(coarse + (int64_t)trace[r] * ld_c + (c - cs), v);
    stv<VW>(out + r * ldo + c, v);
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

}  // namespace

extern "C" int stin_scmn_pack_f32(const float* W1, const float* W2, const float* gamma1, const float* beta1, const float* gamma2,
                                  const float* beta2, int cin, int h2, int cout, int trans_inv, float* wcat, float* wcatT, float* w2T,
                                  float* gb1, float* gb2, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(cin > 0 && h2 > 0 && cout > 0 && (int64_t)2 * h2 * cin < ((int64_t)1 << 30) && (int64_t)h2 * cout < ((int64_t)1 << 30),
                 STIN_E_SIZE);
    STIN_REQUIRE(W1 && W2 && gamma1 && beta1 && gamma2 && beta2 && wcat && wcatT && w2T && gb1 && gb2, STIN_E_NULL);
    int n = 2 * h2 * cin;
    if (h2 * cout > n) n = h2 * cout;
    if (h2 > n) n = h2;
    if (cout > n) n = cout;
    hipLaunchKernelGGL(k_scmn_pack, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, W1, W2, gamma1, beta1, gamma2, beta2, cin, h2,
                       cout, trans_inv, wcat, wcatT, w2T, gb1, gb2);
    return stin_launch_status();
}

extern "C" int stin_scmn_unpack_f32(const float* dwcat, int cin, int h2, int trans_inv, float* dW1, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(cin > 0 && h2 > 0 && (int64_t)h2 * cin < ((int64_t)1 << 30), STIN_E_SIZE);
    STIN_REQUIRE(dwcat && dW1, STIN_E_NULL);
    hipLaunchKernelGGL(k_scmn_unpack, dim3(grid_for((int64_t)h2 * cin)), dim3(BLOCK), 0, (hipStream_t)stream, dwcat, cin, h2, trans_inv, dW1);
    return stin_launch_status();
}

extern "C" int stin_bn_affine_res_fwd_f32(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta, const int32_t* rowptr, const float* res, int64_t ldres, int64_t N, int C,
                                          int relu, float* y, int64_t ldy, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && C > 0 && ldx >= C && ldy >= C && (res == nullptr || ldres >= C), STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(x && mean && rstd && gamma && beta && y, STIN_E_NULL);
    if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && stin_aligned16(x) && stin_aligned16(y) &&
        (res == nullptr || (ldres % 4 == 0 && stin_aligned16(res))))
        hipLaunchKernelGGL((k_bn_affine_res<4>), dim3(grid_for(N * (C / 4))), dim3(BLOCK), 0, (hipStream_t)stream, x, ldx, mean, rstd, gamma,
                           beta, rowptr, res, ldres, N, C, relu, y, ldy);
    else
        hipLaunchKernelGGL((k_bn_affine_res<1>), dim3(grid_for(N * C)), dim3(BLOCK), 0, (hipStream_t)stream, x, ldx, mean, rstd, gamma, beta,
                           rowptr, res, ldres, N, C, relu, y, ldy);
    return stin_launch_status();
}

extern "C" int stin_relu_mask_bwd_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const int32_t* rowptr, int64_t N, int C,
                                      int relu, float* g_eff, float* g_in, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && C > 0 && ldg >= C && (!relu || ldy >= C), STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(g && g_in && (!relu || y), STIN_E_NULL);
    if (C % 4 == 0 && ldg % 4 == 0 && stin_aligned16(g) && stin_aligned16(g_in) && (g_eff == nullptr || stin_aligned16(g_eff)) &&
        (!relu || (ldy % 4 == 0 && stin_aligned16(y))))
        hipLaunchKernelGGL((k_relu_mask_bwd<4>), dim3(grid_for(N * (C / 4))), dim3(BLOCK), 0, (hipStream_t)stream, g, ldg, y, ldy, rowptr, N, C,
                           relu, g_eff, g_in);
    else
        hipLaunchKernelGGL((k_relu_mask_bwd<1>), dim3(grid_for(N * C)), dim3(BLOCK), 0, (hipStream_t)stream, g, ldg, y, ldy, rowptr, N, C, relu,
                           g_eff, g_in);
    return stin_launch_status();
}

extern "C" int stin_concat_unpool_f32(const float* skip, int64_t ld_skip, const float* coarse, int64_t ld_coarse, const int32_t* trace,
                                      int64_t N, int cs, int cu, float* out, int64_t ldo, stin_stream_t stream) {
    stin_clear_stale_error();
    STIN_REQUIRE(N >= 0 && cs > 0 && cu > 0 && ld_skip >= cs && ld_coarse >= cu && ldo >= cs + cu, STIN_E_SIZE);
    if (N == 0) return STIN_OK;
    STIN_REQUIRE(skip && coarse && trace && out, STIN_E_NULL);
    if (cs % 4 == 0 && cu % 4 == 0 && ld_skip % 4 == 0 && ld_coarse % 4 == 0 && ldo % 4 == 0 && stin_aligned16(skip) && stin_aligned16(coarse) &&
        stin_aligned16(out))
        hipLaunchKernelGGL((k_concat_unpool<4>), dim3(grid_for(N * ((cs + cu) / 4))), dim3(BLOCK), 0, (hipStream_t)stream, skip, ld_skip, coarse,
                           ld_coarse, trace, N, cs, cu, out, ldo);
    else
        hipLaunchKernelGGL((k_concat_unpool<1>), dim3(grid_for(N * (int64_t)(cs + cu))), dim3(BLOCK), 0, (hipStream_t)stream, skip, ld_skip, coarse,
                           ld_coarse, trace, N, cs, cu, out, ldo);
    return stin_launch_status();
}
