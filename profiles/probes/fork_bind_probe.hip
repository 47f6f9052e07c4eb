// What a fork of the compute stream costs it: hipEventRecord + hipStreamWaitEvent vs the event bound to the kernel's own
// completion signal (hipExtLaunchKernelGGL's stopEvent) + hipStreamWaitEvent.  Also checks that the bound event orders the side
// stream's kernel behind the compute stream's.  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/fork_bind profiles/probes/fork_bind_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_main(float* x, int n, float v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = v;
}
__global__ void k_side(const float* x, int n, float v, int* bad) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (x[i] != v) atomicAdd(bad, 1);
}
int main() {
    const int n = 16 << 20, links = 200;
    float* x[2]; int* bad;
    hipMalloc(&x[0], n * 4); hipMalloc(&x[1], n * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipStream_t ms, ss; hipStreamCreateWithFlags(&ms, hipStreamNonBlocking); hipStreamCreateWithFlags(&ss, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(links); for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t join; hipEventCreateWithFlags(&join, hipEventDisableTiming);
    auto run = [&](int mode, bool check = false) {
        for (int i = 0; i < links; ++i) {
            float* xi = x[i & 1];
            const float v = (float)(i + 1 + mode * 1000);
            if (mode == 2) hipExtLaunchKernelGGL(k_main, dim3(2048), dim3(256), 0, ms, nullptr, ev[i], 0, xi, n, v);
            else hipLaunchKernelGGL(k_main, dim3(2048), dim3(256), 0, ms, xi, n, v);
            if (mode == 1) hipEventRecord(ev[i], ms);
            if (mode >= 1) {
                hipStreamWaitEvent(ss, ev[i], 0);
                hipLaunchKernelGGL(k_side, dim3(256), dim3(256), 0, ss, xi, check ? n : 0, v, bad);
                // the compute stream must not overwrite xi before the side kernel has read it: it writes the OTHER buffer next,
                // and the side stream is in order, so two links later the reader of xi is done only if the side keeps up - join
                // every second link
                if (check && (i & 1)) { hipEventRecord(join, ss); hipStreamWaitEvent(ms, join, 0); }
            }
        }
        hipStreamSynchronize(ms); hipStreamSynchronize(ss);
    };
    const char* names[3] = {"plain chain", "record + wait + side kernel", "bound stop event + wait + side kernel"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            run(mode, true);
            int hb = 0; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            (void)hipMemset(bad, 0, 4);
            double best = 1e9;
            for (int r = 0; r < 5; ++r) {
                auto a = std::chrono::steady_clock::now();
                run(mode);
                best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / links);
            }
            printf("%-40s %.2f us per link   (order violations in the checked run: %d)\n", names[mode], best, hb);
        }
    return 0;
}
