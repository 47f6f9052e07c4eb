// Profiling harness (not part of the library): the all-columns NT kernel compiled with in-kernel s_memtime stamps.
//   (add -DSTIN_NT_ABLATE_MASK=<bits> for a compile-time ablation of the panel kernel, see stin_gemm.hip)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DSTIN_NT_STAMPS -I surface_texture_inpainting_net_amd/csrc \
//       -c profiles/nt_stamps.hip -o /tmp/nt_stamps.o && hipcc --offload-arch=gfx950 /tmp/nt_stamps.o <stin_wgrad.o stin_pack.o stin_api.o> -o profiles/probes/nt_stamps
#include "../surface_texture_inpainting_net_amd/csrc/stin_gemm.hip"
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 18063;
    const int Nc = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int prec = STIN_GEMM_BF16X3;
    float *A, *W, *Wf, *C;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&W, (size_t)Nc * K * 4);
    hipMalloc(&Wf, (size_t)Nc * K * 4);
    hipMalloc(&C, (size_t)M * Nc * 4);
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)Nc * K * 4, hipMemcpyHostToDevice);
    stin_gemm_split_weights_f32(W, K, Nc, K, prec | STIN_GEMM_W_FRAG, Wf, K, nullptr);
    const int pf = prec | STIN_GEMM_W_PRESPLIT | STIN_GEMM_W_FRAG;
    for (int it = 0; it < 3; ++it) stin_gemm_nt_f32(A, K, Wf, K, nullptr, nullptr, 0, nullptr, 0, M, Nc, K, C, Nc, pf, nullptr);
    hipDeviceSynchronize();
    unsigned long long* stamps;
    const size_t nblk = 4096;
    hipMalloc(&stamps, nblk * 8 * 32 * 8);
    hipMemset(stamps, 0, nblk * 8 * 32 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(stin_nt_stamp_buf), &stamps, sizeof(stamps));
    printf("ablation mask %d\n", (int)STIN_NT_ABLATE_MASK);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    stin_gemm_nt_f32(A, K, Wf, K, nullptr, nullptr, 0, nullptr, 0, M, Nc, K, C, Nc, pf, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("kernel %.1f us (M %lld Nc %d K %d)\n", ms * 1e3, (long long)M, Nc, K);
    std::vector<unsigned long long> hs(nblk * 8 * 32);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t i = 0; i < hs.size(); ++i)
        if (hs[i]) {
            if (hs[i] < t0) t0 = hs[i];
            if (hs[i] > t1) t1 = hs[i];
        }
    printf("first stamp -> last stamp: %llu cycles\n", t1 - t0);
    const int blocks[] = {0, 100};
    const int wv = argc > 4 ? atoi(argv[4]) : 0;                 // which wave of the block to print
    for (int bi : blocks) {
        const unsigned long long* s = &hs[((size_t)bi * 8 + wv) * 32];
        if (!s[0]) continue;
        printf("block %3d start %7llu:", bi, s[0] - t0);
        unsigned long long prev = s[0];
        for (int i = 1; i < 32; ++i) {
            if (!s[i]) continue;
            printf(" [%d]+%llu", i, s[i] - prev);
            prev = s[i];
        }
        printf("  total %llu\n", prev - s[0]);
    }
    return 0;
}
