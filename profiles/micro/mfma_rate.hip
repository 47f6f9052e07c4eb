// Micro-benchmark: what one wave per SIMD sustains on v_mfma_f32_32x32x16_bf16 under the dependency / LDS-read patterns of
// the STINet GEMM kernels (cycles per MFMA from s_memtime, one 256-thread block per CU).
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma_rate.hip -o profiles/micro/_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2][2][128][40];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 2 * 128 * 40 / 8; i += 256) reinterpret_cast<bf16x8*>(&lds[0][0][0][0])[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 a0 = in[lane], a1 = in[lane + 64], b0 = in[lane + 128], b1 = in[lane + 192];
    const int li = lane & 31, kh = lane >> 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 4 independent accumulators, operands in registers, 12 MFMAs
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[a], 0, 0, 0);
        } else if (MODE == 1) {     // chains of 3 on one accumulator (the split product a0b1 + a1b0 + a0b0), 4 accumulators in turn
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[a], 0, 0, 0);
            }
        } else if (MODE == 2) {     // one accumulator only
#pragma unroll
            for (int r = 0; r < 12; ++r) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
        } else {                    // MODE 3: fragments re-read from LDS every k-step (8 ds_read_b128 per 12 MFMAs), chains of 3
            bf16x8 a[2][2], c[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[p][t] = *reinterpret_cast<const bf16x8*>(&lds[it & 1][p][t * 32 + li][8 * kh]);
                    c[p][t] = *reinterpret_cast<const bf16x8*>(&lds[it & 1][p][64 + t * 32 + li][8 * kh]);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[t * 2 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][t], c[1][u], acc[t * 2 + u], 0, 0, 0);
                    acc[t * 2 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][t], c[0][u], acc[t * 2 + u], 0, 0, 0);
                    acc[t * 2 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][t], c[0][u], acc[t * 2 + u], 0, 0, 0);
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE> void run(const char* what, const bf16x8* in, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= h.size();
    const double mf = (double)iters * 12;
    printf("%-70s blocks %4d: %.1f cycles / MFMA, %.2f ms, %.0f TFLOP/s, clock ~%.2f GHz\n", what, blocks, avg / mf, ms,
           mf * 32768.0 * 4 * blocks / (ms * 1e-3) / 1e12, avg / (ms * 1e-3) / 1e9);
}

int main() {
    bf16x8* in;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&in, 4096 * 16);
    std::vector<unsigned short> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f80 + (i * 7 % 64);      // bf16 values around 1.0
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&cyc, 1024 * 4 * 8);
    for (int blocks : {1, 256, 512}) {
        run<0>("4 independent accumulators, register operands", in, out, cyc, blocks);
        run<1>("chains of 3 per accumulator (split product), register operands", in, out, cyc, blocks);
        run<2>("ONE accumulator, register operands", in, out, cyc, blocks);
        run<3>("chains of 3, fragments from LDS (8 ds_read_b128 / 12 MFMA)", in, out, cyc, blocks);
    }
    return 0;
}
