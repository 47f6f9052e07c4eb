// Profiling harness (not part of the library): the producer / consumer TN kernel compiled with in-kernel s_memtime stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DSTIN_WS_STAMPS -I surface_texture_inpainting_net_amd/csrc \
//       profiles/tn_stamps.hip surface_texture_inpainting_net_amd/csrc/{stin_gemm,stin_api}.o -o profiles/probes/tn_stamps
// Prints, for a few blocks, the cycle stamps of producer wave 4 and consumer wave 0 (deltas between consecutive stamps).
#include "../surface_texture_inpainting_net_amd/csrc/stin_wgrad.hip"
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 18063;
    const int Nc = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
    float *G, *X, *slab;
    hipMalloc(&G, (size_t)M * Nc * 4);
    hipMalloc(&X, (size_t)M * K * 4);
    std::vector<float> h((size_t)M * (Nc > K ? Nc : K));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(G, h.data(), (size_t)M * Nc * 4, hipMemcpyHostToDevice);
    hipMemcpy(X, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    const size_t wsb = stin_gemm_tn_workspace_bytes(M, Nc, K, 1);
    hipMalloc(&slab, wsb);
    stin_tn_problem p;
    int ok = 0;
    stin_tn_problem_init(&p, 0, G, Nc, X, K, M, Nc, K, 1, nullptr, 0, STIN_GEMM_BF16X3, slab, &ok);
    printf("chunks %lld rows/chunk %d tiles %d x %d eligible %d\n", (long long)p.chunks, p.rows_per_chunk, p.tiles_i, p.tiles_j, ok);
    stin_tn_batch b;
    b.p[0] = p;
    b.p[1] = p;
    b.n = 1;
    unsigned long long* stamps;
    const size_t nblk = 4096;
    hipMalloc(&stamps, nblk * 8 * 64 * 8);
    hipMemset(stamps, 0, nblk * 8 * 64 * 8);
    for (int it = 0; it < 3; ++it) stin_tn_ws_launch(b, nullptr);      // warm (stamps off)
    hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(stin_ws_stamp_buf), &stamps, sizeof(stamps));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    if (argc > 4) setenv("STIN_TN_WS_PRIO", argv[4], 1);      // prio | ablation bits << 4
    hipEventRecord(e0, nullptr);
    stin_tn_ws_launch(b, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("kernel %.1f us\n", ms * 1e3);
    std::vector<unsigned long long> hs(nblk * 8 * 64);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < hs.size(); ++i)
        if (hs[i] && hs[i] < t0) t0 = hs[i];
    const int blocks[] = {0, 100, 300};
    for (int bi : blocks) {
        for (int w : {4, 0}) {
            const unsigned long long* s = &hs[((size_t)bi * 8 + w) * 64];
            if (!s[0]) continue;
            printf("block %3d %s start %7llu:", bi, w == 4 ? "producer" : "consumer", s[0] - t0);
            unsigned long long prev = s[0];
            for (int i = 1; i < 64; ++i) {
                if (!s[i]) continue;
                printf(" [%d]+%llu", i, s[i] - prev);
                prev = s[i];
            }
            printf("  total %llu\n", prev - s[0]);
        }
    }
    return 0;
}
