// Micro-benchmark: global-load rate of a 256-thread block streaming a [64 rows x K] fp32 panel (row pitch K * 4 bytes) as a
// function of how one wave-wide 16-byte load instruction is laid over the panel: RUN contiguous bytes per row x (1024 / RUN)
// rows.  The GEMM staging loops of round 1-2 use 128-byte runs (8 rows per instruction); csrc/stin_wgrad.hip uses 512.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/load_pattern.hip -o profiles/micro/_load_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int RUN>
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, int64_t M, int K, float* __restrict__ out) {
    constexpr int LPR = RUN / 16;                 // lanes per row
    constexpr int RPB = 256 / LPR;                // rows per block-wide load
    const int tid = threadIdx.x;
    const int c = tid % LPR, r = tid / LPR;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K; k0 += RUN / 4 * ((RPB >= 64) ? 1 : 1)) {
        // one "chunk": 64 rows x RUN bytes -> 64 / RPB loads per thread, all in flight together
        float4 v[64 / RPB > 0 ? 64 / RPB : 1];
#pragma unroll
        for (int p = 0; p < 64 / RPB; ++p) {
            int64_t row = row0 + r + p * RPB;
            if (row >= M) row = M - 1;
            v[p] = *reinterpret_cast<const float4*>(A + row * K + k0 + c * 4);
        }
#pragma unroll
        for (int p = 0; p < 64 / RPB; ++p) {
            acc.x += v[p].x;
            acc.y += v[p].y;
            acc.z += v[p].z;
            acc.w += v[p].w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

template <int RUN> void run(const float* A, int64_t M, int K, float* out) {
    const int blocks = (int)((M + 63) / 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<RUN>, dim3(blocks), dim3(256), 0, 0, A, M, K, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<RUN>, dim3(blocks), dim3(256), 0, 0, A, M, K, out);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 100.0;
    printf("  run %4d B x %2d rows per load instruction: %7.1f us  %6.2f TB/s\n", RUN, 1024 / RUN, us, (double)M * K * 4 / us / 1e6);
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 18063;
    const int K = argc > 2 ? atoi(argv[2]) : 1024;
    float *A, *out;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMemset(A, 0, (size_t)M * K * 4);
    hipMalloc(&out, 1 << 20);
    printf("panel stream of [%lld x %d] fp32 (%.0f MB), one 256-thread block per 64 rows:\n", (long long)M, K, (double)M * K * 4 / 1e6);
    run<128>(A, M, K, out);
    run<256>(A, M, K, out);
    run<512>(A, M, K, out);
    run<1024>(A, M, K, out);
    return 0;
}
