// COO -> CSR graph-plan construction on the GPU (one-off per sample, not a hot op):
// stable LSD radix sort of (key, pair-id) with rocPRIM, binary-searched row pointers,
// gathered column ids.  See include/stin_hip.h for the contract.
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include "stin_common.h"

namespace {

__global__ void k_prepare(const int64_t* __restrict__ key, const int64_t* __restrict__ val, int64_t E, int64_t N,
                          int64_t val_limit, int32_t* __restrict__ key32, int32_t* __restrict__ iota,
                          int32_t* __restrict__ bad) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t k = key[e];
    bool oob = (k < 0) | (k >= N);
    if (val != nullptr) {
        int64_t v = val[e];
        oob |= (v < 0) | (v >= val_limit);
    }
    if (oob) {
        if (bad != nullptr) atomicOr(bad, 1);
        k = 0;  // keep every later kernel in bounds; the host raises before using the plan
    }
    key32[e] = (int32_t)k;
    iota[e] = (int32_t)e;
}

__global__ void k_rowptr(const int32_t* __restrict__ sorted_key, int64_t E, int64_t N, int32_t* __restrict__ rowptr,
                         float* __restrict__ inv_deg) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n > N) return;
    // lower_bound(sorted_key, n)
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (sorted_key[mid] < (int32_t)n) lo = mid + 1; else hi = mid;
    }
    rowptr[n] = (int32_t)lo;
    if (inv_deg != nullptr && n < N) {
        int64_t lo2 = lo, hi2 = E;
        while (lo2 < hi2) {
            int64_t mid = (lo2 + hi2) >> 1;
            if (sorted_key[mid] < (int32_t)(n + 1)) lo2 = mid + 1; else hi2 = mid;
        }
        int64_t deg = lo2 - lo;
        inv_deg[n] = 1.0f / (float)(deg > 0 ? deg : 1);
    }
}

__global__ void k_col(const int64_t* __restrict__ val, const int32_t* __restrict__ perm, int64_t E, int64_t val_limit,
                      int32_t* __restrict__ col) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int32_t p = perm[e];
    if (val == nullptr) {
        col[e] = p;
    } else {
        int64_t v = val[p];
        col[e] = (v < 0 || v >= val_limit) ? 0 : (int32_t)v;
    }
}

__global__ void k_narrow(const int64_t* __restrict__ src, int64_t n, int64_t limit, int32_t* __restrict__ dst,
                         int32_t* __restrict__ bad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t v = src[i];
    if (v < 0 || v >= limit) {
        if (bad != nullptr) atomicOr(bad, 1);
        v = 0;
    }
    dst[i] = (int32_t)v;
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

inline int key_bits(int64_t N) {
    int b = 1;
    while (((int64_t)1 << b) < N && b < 31) ++b;
    return b;
}

size_t sort_temp_bytes(int64_t E, int64_t N) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr,
                                    (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)(E > 0 ? E : 1), 0,
                                    key_bits(N), (hipStream_t)0);
    return bytes;
}

}  // namespace

extern "C" size_t stin_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    size_t e = (size_t)(E > 0 ? E : 1);
    return 4 * align_up(e * sizeof(int32_t)) + align_up(sort_temp_bytes(E, N)) + 256;
}

extern "C" int stin_csr_from_coo_i64(const int64_t* key, const int64_t* val, int64_t E, int64_t N, int64_t val_limit,
                                     int32_t* rowptr, int32_t* col, int32_t* perm, float* inv_deg, int32_t* bad,
                                     void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(E >= 0 && N >= 0 && N < ((int64_t)1 << 31) && E < ((int64_t)1 << 31), STIN_E_SIZE);
    STIN_REQUIRE(rowptr != nullptr && (E == 0 || (key != nullptr && col != nullptr)), STIN_E_NULL);
    STIN_REQUIRE(workspace != nullptr, STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_csr_workspace_bytes(E, N), STIN_E_WORKSPACE);
    if (val != nullptr) STIN_REQUIRE(val_limit >= 0 && val_limit < ((int64_t)1 << 31), STIN_E_SIZE);

    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    size_t e = (size_t)(E > 0 ? E : 1);
    size_t slab = align_up(e * sizeof(int32_t));
    int32_t* key_in = reinterpret_cast<int32_t*>(ws);
    int32_t* key_out = reinterpret_cast<int32_t*>(ws + slab);
    int32_t* iota = reinterpret_cast<int32_t*>(ws + 2 * slab);
    int32_t* perm_ws = reinterpret_cast<int32_t*>(ws + 3 * slab);
    void* sort_tmp = ws + 4 * slab;
    size_t sort_bytes = sort_temp_bytes(E, N);
    int32_t* perm_out = perm != nullptr ? perm : perm_ws;

    const int T = 256;
    if (E > 0) {
        hipLaunchKernelGGL(k_prepare, dim3((unsigned)((E + T - 1) / T)), dim3(T), 0, stream, key, val, E, N, val_limit,
                           key_in, iota, bad);
        hipError_t err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, (const int32_t*)key_in, key_out,
                                                   (const int32_t*)iota, perm_out, (size_t)E, 0, key_bits(N), stream);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL(k_col, dim3((unsigned)((E + T - 1) / T)), dim3(T), 0, stream, val, perm_out, E, val_limit, col);
    }
    hipLaunchKernelGGL(k_rowptr, dim3((unsigned)((N + 1 + T - 1) / T)), dim3(T), 0, stream, key_out, E, N, rowptr,
                       inv_deg);
    return stin_launch_status();
}

extern "C" int stin_narrow_i64_to_i32(const int64_t* src, int64_t n, int64_t limit, int32_t* dst, int32_t* bad,
                                      stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(n >= 0, STIN_E_SIZE);
    if (n == 0) return STIN_OK;
    STIN_REQUIRE(src != nullptr && dst != nullptr, STIN_E_NULL);
    const int T = 256;
    hipLaunchKernelGGL(k_narrow, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, (hipStream_t)stream_, src, n, limit, dst,
                       bad);
    return stin_launch_status();
}
