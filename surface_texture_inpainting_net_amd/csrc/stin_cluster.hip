// Voxel (Rossignac) vertex clustering and edge-list coalescing on the GPU - the hierarchy-generation alternative to QEM of the
// reference's preprocessing/graph_level_generation.py:193-244 (`vertex_clustering`) and the pyg.utils.coalesce calls around it
// (preprocessing/graph_dilation.py:53-56).  Contract: include/stin_hip.h.
//
//   cluster:  bin = coords // voxel (numpy's floor-division algorithm, per component, fp64) -> one 63-bit key per vertex
//             (three 21-bit fields relative to the per-axis minimum: lexicographic order of the bins = np.unique(axis=0)'s order)
//             -> stable rocPRIM radix sort of (key, vertex) -> head flags + scan = cluster ids in bin order -> trace[vertex],
//             and per cluster the fp64 mean of its members' coordinates IN VERTEX ORDER (the stable sort keeps it), cast to fp32.
//             Deterministic (the framework formulation summed with fp64 atomics in arrival order).
//   coalesce: keys = a * n + b of the (a, b) pairs with a != b (or of all pairs) -> radix sort -> unique, count on the device.
// Integer / byte work bound by the sort passes; nothing here is on the training step.
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include "stin_common.h"

namespace {

constexpr int T = 256;
typedef unsigned long long u64;
constexpr u64 KEY_DROP = ~0ull;               // sorts last; removed after the unique pass

// numpy's npy_floor_divide / CPython float floor division (what `coords // voxel_size` evaluates)
__device__ __forceinline__ double floor_div(double a, double b) {
    if (b == 0.0) return a / b;
    const double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0.0) != (mod < 0.0))) div -= 1.0;
    if (div != 0.0) {
        double f = floor(div);
        if (div - f > 0.5) f += 1.0;
        return f;
    }
    return copysign(0.0, a / b);
}

__global__ void k_state_init(long long* __restrict__ state) {
    if (threadIdx.x < 5) state[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffffffffffffll : 0ll;
}

// state: [0..2] per-axis minimum bin (int64), [3] flags (bit 0: a bin range does not fit 21 bits / a non-finite coordinate),
// [4] number of clusters
__global__ __launch_bounds__(T) void k_bin_min(const double* __restrict__ coords, int64_t N, double voxel, long long* __restrict__ state) {
    const int64_t n = (int64_t)blockIdx.x * T + threadIdx.x;
    if (n >= N) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double b = floor_div(coords[n * 3 + d], voxel);
        if (!(fabs(b) < 4.0e18)) {
            atomicOr(reinterpret_cast<u64*>(state + 3), 1ull);
            continue;
        }
        atomicMin(state + d, (long long)b);
    }
}

__global__ __launch_bounds__(T) void k_bin_keys(const double* __restrict__ coords, int64_t N, double voxel, long long* __restrict__ state,
                                                u64* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t n = (int64_t)blockIdx.x * T + threadIdx.x;
    if (n >= N) return;
    u64 key = 0;
    bool bad = false;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double b = floor_div(coords[n * 3 + d], voxel);
        const long long r = fabs(b) < 4.0e18 ? (long long)b - state[d] : -1;
        bad |= (r < 0) | (r >= (1ll << 21));
        key = (key << 21) | (u64)(r & ((1ll << 21) - 1));
    }
    if (bad) atomicOr(reinterpret_cast<u64*>(state + 3), 1ull);
    keys[n] = key;
    ids[n] = (int32_t)n;
}

__global__ __launch_bounds__(T) void k_heads(const u64* __restrict__ keys, int64_t n, int32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// sorted position i: cluster id = scan[i] - 1; heads also reduce their cluster (members are consecutive, in vertex order)
__global__ __launch_bounds__(T) void k_cluster_out(const double* __restrict__ coords, const int32_t* __restrict__ ids,
                                                   const int32_t* __restrict__ head, const int32_t* __restrict__ scan, int64_t N,
                                                   int64_t* __restrict__ trace, float* __restrict__ new_coords,
                                                   long long* __restrict__ state) {
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i >= N) return;
    const int32_t c = scan[i] - 1;
    trace[ids[i]] = c;
    if (i == N - 1) state[4] = (long long)c + 1;
    if (!head[i]) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int64_t k = i;
    do {
        const int64_t v = ids[k];
        s0 += coords[v * 3 + 0];
        s1 += coords[v * 3 + 1];
        s2 += coords[v * 3 + 2];
        ++k;
    } while (k < N && !head[k]);
    const double cnt = (double)(k - i);
    new_coords[(int64_t)c * 3 + 0] = (float)(s0 / cnt);
    new_coords[(int64_t)c * 3 + 1] = (float)(s1 / cnt);
    new_coords[(int64_t)c * 3 + 2] = (float)(s2 / cnt);
}

__global__ __launch_bounds__(T) void k_pair_keys(const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                                                 const int64_t* __restrict__ map, int64_t map_n, int64_t E, int64_t n, int drop_loops,
                                                 u64* __restrict__ keys, long long* __restrict__ state) {
    const int64_t e = (int64_t)blockIdx.x * T + threadIdx.x;
    if (e >= E) return;
    int64_t x = a[e], y = b[e];
    if (map != nullptr) {
        if (x < 0 || y < 0 || x >= map_n || y >= map_n) {     // the raw endpoints index map[]: checked before they do
            atomicOr(reinterpret_cast<u64*>(state + 3), 1ull);
            keys[e] = KEY_DROP;
            return;
        }
        x = map[x];
        y = map[y];
    }
    if (x < 0 || y < 0 || x >= n || y >= n) {
        atomicOr(reinterpret_cast<u64*>(state + 3), 1ull);
        keys[e] = KEY_DROP;
        return;
    }
    keys[e] = (drop_loops && x == y) ? KEY_DROP : (u64)x * (u64)n + (u64)y;
}

// unique of the sorted keys: head flags -> scan -> scatter (KEY_DROP entries are left out); state[4] = count
__global__ __launch_bounds__(T) void k_heads_keep(const u64* __restrict__ keys, int64_t n, int32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i >= n) return;
    head[i] = (keys[i] != KEY_DROP && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}
__global__ __launch_bounds__(T) void k_unique_out(const u64* __restrict__ keys, const int32_t* __restrict__ head,
                                                  const int32_t* __restrict__ scan, int64_t E, int64_t n, int64_t* __restrict__ out_a,
                                                  int64_t* __restrict__ out_b, long long* __restrict__ state) {
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i >= E) return;
    if (i == E - 1) state[4] = scan[i];
    if (!head[i]) return;
    const int64_t o = scan[i] - 1;
    out_a[o] = (int64_t)(keys[i] / (u64)n);
    out_b[o] = (int64_t)(keys[i] % (u64)n);
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
inline unsigned grid_for(int64_t n) { return (unsigned)((n + T - 1) / T); }

size_t sort_temp_bytes(int64_t n, bool pairs) {
    size_t bytes = 0;
    const size_t m = (size_t)(n > 0 ? n : 1);
    if (pairs)
        (void)rocprim::radix_sort_pairs(nullptr, bytes, (u64*)nullptr, (u64*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, m, 0, 64, (hipStream_t)0);
    else
        (void)rocprim::radix_sort_keys(nullptr, bytes, (u64*)nullptr, (u64*)nullptr, m, 0, 64, (hipStream_t)0);
    return bytes;
}
size_t scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)(n > 0 ? n : 1), rocprim::plus<int32_t>(),
                                  (hipStream_t)0);
    return bytes;
}

struct Layout {
    size_t keys0, keys1, ids0, ids1, head, scan, temp, total;
};
Layout layout(int64_t n, bool pairs) {
    Layout L;
    const size_t m = (size_t)(n > 0 ? n : 1);
    size_t off = 0;
    L.keys0 = off; off += up256(m * 8);
    L.keys1 = off; off += up256(m * 8);
    L.ids0 = off;  off += pairs ? up256(m * 4) : 0;
    L.ids1 = off;  off += pairs ? up256(m * 4) : 0;
    L.head = off;  off += up256(m * 4);
    L.scan = off;  off += up256(m * 4);
    L.temp = off;
    const size_t a = sort_temp_bytes(n, pairs), b = scan_temp_bytes(n);
    off += up256(a > b ? a : b);
    L.total = off + 256;
    return L;
}

}  // namespace

extern "C" size_t stin_voxel_cluster_workspace_bytes(int64_t N) { return N < 0 ? 0 : layout(N, true).total; }

// state (device, 5 x int64, written here): [0..2] scratch, [3] != 0: unsupported input (a bin range >= 2^21 or a non-finite
// coordinate: the caller falls back), [4] number of clusters Nc.  trace [N] int64, new_coords [>= Nc rows, 3] fp32 (N rows suffice).
extern "C" int stin_voxel_cluster_f64(const double* coords, int64_t N, double voxel, int64_t* trace, float* new_coords,
                                      int64_t* state, void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(N >= 0 && N < ((int64_t)1 << 31), STIN_E_SIZE);
    STIN_REQUIRE(state != nullptr, STIN_E_NULL);
    long long* st = reinterpret_cast<long long*>(state);
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, stream, st);
    hipError_t e = hipSuccess;
    if (N == 0) return stin_launch_status();
    STIN_REQUIRE(coords && trace && new_coords && workspace, STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_voxel_cluster_workspace_bytes(N), STIN_E_WORKSPACE);
    const Layout L = layout(N, true);
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    u64 *k0 = reinterpret_cast<u64*>(w + L.keys0), *k1 = reinterpret_cast<u64*>(w + L.keys1);
    int32_t *i0 = reinterpret_cast<int32_t*>(w + L.ids0), *i1 = reinterpret_cast<int32_t*>(w + L.ids1);
    int32_t *head = reinterpret_cast<int32_t*>(w + L.head), *scan = reinterpret_cast<int32_t*>(w + L.scan);
    hipLaunchKernelGGL(k_bin_min, dim3(grid_for(N)), dim3(T), 0, stream, coords, N, voxel, st);
    hipLaunchKernelGGL(k_bin_keys, dim3(grid_for(N)), dim3(T), 0, stream, coords, N, voxel, st, k0, i0);
    size_t tb = sort_temp_bytes(N, true);
    e = rocprim::radix_sort_pairs(w + L.temp, tb, k0, k1, i0, i1, (size_t)N, 0, 63, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_heads, dim3(grid_for(N)), dim3(T), 0, stream, k1, N, head);
    tb = scan_temp_bytes(N);
    e = rocprim::inclusive_scan(w + L.temp, tb, head, scan, (size_t)N, rocprim::plus<int32_t>(), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_cluster_out, dim3(grid_for(N)), dim3(T), 0, stream, coords, i1, head, scan, N, trace, new_coords, st);
    return stin_launch_status();
}

extern "C" size_t stin_coalesce_workspace_bytes(int64_t E) { return E < 0 ? 0 : layout(E, false).total; }

// Unique (a, b) pairs sorted by (a, b): a / b [E] int64, optionally mapped through map[] first (the coarse ids of the two
// endpoints), pairs with a == b left out when drop_loops; values must lie in [0, n), n < 2^31.  out_a / out_b [>= count] int64
// (E entries suffice); state as above: [3] != 0 an index was out of range, [4] = count.
extern "C" int stin_coalesce_pairs_i64(const int64_t* a, const int64_t* b, const int64_t* map, int64_t map_n, int64_t E, int64_t n, int drop_loops,
                                       int64_t* out_a, int64_t* out_b, int64_t* state, void* workspace, size_t workspace_bytes,
                                       stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(E >= 0 && E < ((int64_t)1 << 31) && n >= 0 && n < ((int64_t)1 << 31), STIN_E_SIZE);
    STIN_REQUIRE(state != nullptr, STIN_E_NULL);
    STIN_REQUIRE(map == nullptr || map_n >= 0, STIN_E_SIZE);
    long long* st = reinterpret_cast<long long*>(state);
    hipError_t e = hipMemsetAsync(st, 0, 5 * sizeof(long long), stream);
    if (e != hipSuccess) return (int)e;
    if (E == 0) return STIN_OK;
    STIN_REQUIRE(a && b && out_a && out_b && workspace, STIN_E_NULL);
    STIN_REQUIRE(workspace_bytes >= stin_coalesce_workspace_bytes(E), STIN_E_WORKSPACE);
    const Layout L = layout(E, false);
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    u64 *k0 = reinterpret_cast<u64*>(w + L.keys0), *k1 = reinterpret_cast<u64*>(w + L.keys1);
    int32_t *head = reinterpret_cast<int32_t*>(w + L.head), *scan = reinterpret_cast<int32_t*>(w + L.scan);
    hipLaunchKernelGGL(k_pair_keys, dim3(grid_for(E)), dim3(T), 0, stream, a, b, map, map_n, E, n, drop_loops, k0, st);
    size_t tb = sort_temp_bytes(E, false);
    e = rocprim::radix_sort_keys(w + L.temp, tb, k0, k1, (size_t)E, 0, 64, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_heads_keep, dim3(grid_for(E)), dim3(T), 0, stream, k1, E, head);
    tb = scan_temp_bytes(E);
    e = rocprim::inclusive_scan(w + L.temp, tb, head, scan, (size_t)E, rocprim::plus<int32_t>(), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_unique_out, dim3(grid_for(E)), dim3(T), 0, stream, k1, head, scan, E, n, out_a, out_b, st);
    return stin_launch_status();
}
