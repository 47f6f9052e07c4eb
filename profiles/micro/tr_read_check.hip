// Check of the ds_read_b64_tr_b16 addressing used by k_gemm_tn_b16_tr (csrc/stin_gemm.hip): a [64 rows (m)][128 cols] 16-bit tile in
// LDS, 256-byte rows, 16-byte chunk ch of row r stored at chunk position ch ^ (((r & 3) << 2) | ((r >> 2) & 3)); two transposed reads
// must give lane l the 8 values tile[ks * 16 + 8 * (l >> 5) + e][cb + (l & 31)], e = 0..7 - the 32x32x16 MFMA operand of the
// TRANSPOSED tile.   hipcc --offload-arch=gfx950 -O2 profiles/micro/tr_read_check.hip -o profiles/micro/_tr_read_check && ./profiles/micro/_tr_read_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tile_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__global__ void k(uint16_t* out, int ks, int cb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 256];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 128; i += 64) {
        const int row = i / 128, col = i % 128;
        *reinterpret_cast<uint16_t*>(smem + tile_off(row, col / 8) + (col % 8) * 2) = (uint16_t)(row * 128 + col);
    }
    __syncthreads();
    const int g = l >> 4, tl = l & 15, q = tl >> 2, p = tl & 3;
    const int c0 = (cb + 16 * (g & 1)) / 8;
    for (int h = 0; h < 2; ++h) {
        const int row = ks * 16 + 8 * (g >> 1) + 4 * h + q;
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + tile_off(row, c0 + (p >> 1)) + 8 * (p & 1)));
        for (int e = 0; e < 4; ++e) out[l * 8 + 4 * h + e] = (uint16_t)v[e];
    }
}
int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 8 * 2);
    int bad = 0;
    for (int ks = 0; ks < 4; ++ks)
        for (int cb = 0; cb < 128; cb += 32) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ks, cb);
            std::vector<uint16_t> h(64 * 8);
            hipMemcpy(h.data(), d, 64 * 8 * 2, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int want = (ks * 16 + 8 * (l >> 5) + e) * 128 + cb + (l & 31);
                    if (h[l * 8 + e] != want) {
                        if (bad < 8) printf("ks %d cb %d lane %d e %d: got %d (row %d col %d) want row %d col %d\n", ks, cb, l, e, h[l * 8 + e],
                                            h[l * 8 + e] / 128, h[l * 8 + e] % 128, want / 128, want % 128);
                        ++bad;
                    }
                }
        }
    printf("tr_read_check: %d mismatches\n", bad);
    return bad != 0;
}
