// COO -> CSR graph-plan construction on the GPU (once per sample): a counting sort.
//   1. count    : integer atomicAdd histogram of the keys (result independent of arrival order)
//   2. scan     : one rocPRIM inclusive scan -> row pointers
//   3. fill     : atomic cursor per row drops each pair id into its row (arrival order arbitrary)
//   4. rank     : every slot finds its rank among the pair ids of its row and moves there
//                 => within a row entries are in ORIGINAL pair order, exactly a stable sort,
//                 deterministic although steps 1 and 3 use atomics.
// Both CSRs of an edge set (by destination and by source) are built by the same launches.
// Contract: include/stin_hip.h.
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include "stin_common.h"

namespace {

constexpr int T = 256;

// sides: 0 = group by a[e] with value b[e]; 1 (pair mode only) = group by b[e] with value a[e].
__global__ void k_count(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t E, int64_t N,
                        int64_t b_limit, int pair, int32_t* __restrict__ cnt, int32_t* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t ka = a[e];
    bool oob = (ka < 0) | (ka >= N);
    int64_t kb = 0;
    if (b != nullptr) {
        kb = b[e];
        oob |= (kb < 0) | (kb >= b_limit);
    }
    if (oob) {
        if (bad != nullptr) atomicOr(bad, 1);
        return;                      // dropped from the plan; the host raises IndexError before using it
    }
    atomicAdd(&cnt[1 + ka], 1);
    if (pair) atomicAdd(&cnt[1 + N + kb], 1);
}

// cnt (inclusive-scanned, cnt[0] = 0) -> rowptr(s), cursors, inv_deg
__global__ void k_rows(const int32_t* __restrict__ scanned, int64_t N, int pair, int32_t* __restrict__ rowptr0,
                       int32_t* __restrict__ rowptr1, int32_t* __restrict__ cursor, float* __restrict__ inv_deg0,
                       float* __restrict__ inv_deg1) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n > N) return;
    const int32_t p0 = scanned[n];
    rowptr0[n] = p0;
    if (n < N) {
        cursor[n] = p0;
        if (inv_deg0 != nullptr) {
            const int32_t d = scanned[n + 1] - p0;
            inv_deg0[n] = 1.0f / (float)(d > 0 ? d : 1);
        }
    }
    if (pair) {
        const int32_t total0 = scanned[N];
        const int32_t p1 = scanned[N + n] - total0;
        rowptr1[n] = p1;
        if (n < N) {
            cursor[N + n] = p1;
            if (inv_deg1 != nullptr) {
                const int32_t d = scanned[N + n + 1] - total0 - p1;
                inv_deg1[n] = 1.0f / (float)(d > 0 ? d : 1);
            }
        }
    }
}

__global__ void k_fill(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t E, int64_t N,
                       int64_t b_limit, int pair, int32_t* __restrict__ cursor, int32_t* __restrict__ tmp_id0,
                       int32_t* __restrict__ tmp_key0, int32_t* __restrict__ tmp_id1, int32_t* __restrict__ tmp_key1) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t ka = a[e];
    const int64_t kb = b != nullptr ? b[e] : 0;
    if (ka < 0 || ka >= N || (b != nullptr && (kb < 0 || kb >= b_limit))) return;
    const int32_t s0 = atomicAdd(&cursor[ka], 1);
    tmp_id0[s0] = (int32_t)e;
    tmp_key0[s0] = (int32_t)ka;
    if (pair) {
        const int32_t s1 = atomicAdd(&cursor[N + kb], 1);
        tmp_id1[s1] = (int32_t)e;
        tmp_key1[s1] = (int32_t)kb;
    }
}

// slot t of side `side`: rank of its pair id within its row -> final position
__global__ void k_rank(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t N, int64_t n_slots0,
                       int64_t n_slots1, const int32_t* __restrict__ rowptr0, const int32_t* __restrict__ rowptr1,
                       const int32_t* __restrict__ tmp_id0, const int32_t* __restrict__ tmp_key0,
                       const int32_t* __restrict__ tmp_id1, const int32_t* __restrict__ tmp_key1,
                       int32_t* __restrict__ col0, int32_t* __restrict__ perm0, int32_t* __restrict__ col1,
                       int32_t* __restrict__ perm1, int32_t* __restrict__ slot_of_edge) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int side = 0;
    if (t >= n_slots0) {
        t -= n_slots0;
        side = 1;
        if (t >= n_slots1) return;
    }
    const int32_t* ids = side ? tmp_id1 : tmp_id0;
    const int32_t* keys = side ? tmp_key1 : tmp_key0;
    const int32_t* rowptr = side ? rowptr1 : rowptr0;
    if (t >= rowptr[N]) return;      // fewer slots than pairs when out-of-range pairs were dropped
    const int32_t row = keys[t];
    const int32_t beg = rowptr[row], end = rowptr[row + 1];
    const int32_t mine = ids[t];
    int32_t rank = 0;
    for (int32_t u = beg; u < end; ++u) rank += (ids[u] < mine) ? 1 : 0;
    const int32_t pos = beg + rank;
    int32_t* col = side ? col1 : col0;
    int32_t* perm = side ? perm1 : perm0;
    // value stored with the entry: the OTHER endpoint (or the pair id itself when there is no value array)
    const int64_t* other = side ? a : b;
    col[pos] = other != nullptr ? (int32_t)other[mine] : mine;
    if (perm != nullptr) perm[pos] = mine;
    if (side == 0 && slot_of_edge != nullptr) slot_of_edge[mine] = pos;
}

// xslot[src-CSR slot] = dst-CSR slot of the same edge (slot_of_edge is the inverse of the dst-side perm)
// and w_src[src-CSR slot] = 1 / max(1, in-degree of that edge's target): sequential weights for the dB pass
__global__ void k_xslot(const int32_t* __restrict__ perm_src, const int32_t* __restrict__ slot_of_edge, int64_t E,
                        const int32_t* __restrict__ rowptr_src, const int32_t* __restrict__ col_src,
                        const float* __restrict__ inv_deg_dst, int64_t N, int32_t* __restrict__ xslot,
                        float* __restrict__ w_src) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E || t >= rowptr_src[N]) return;
    xslot[t] = slot_of_edge[perm_src[t]];
    if (w_src != nullptr) w_src[t] = inv_deg_dst[col_src[t]];
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
inline unsigned grid_for(int64_t n) { return (unsigned)((n + T - 1) / T); }

size_t scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)(n > 0 ? n : 1),
                                  rocprim::plus<int32_t>(), (hipStream_t)0);
    return bytes;
}

struct Layout {
    size_t cnt, cursor, id0, key0, id1, key1, x0, x1, scan, total;
};

Layout layout(int64_t E, int64_t N, int pair) {
    Layout L;
    const size_t sides = pair ? 2 : 1;
    const size_t e = (size_t)(E > 0 ? E : 1);
    size_t off = 0;
    L.cnt = off;    off += align_up((sides * N + 2) * sizeof(int32_t));
    L.cursor = off; off += align_up((sides * N + 2) * sizeof(int32_t));
    L.id0 = off;    off += align_up(e * sizeof(int32_t));
    L.key0 = off;   off += align_up(e * sizeof(int32_t));
    L.id1 = off;    off += pair ? align_up(e * sizeof(int32_t)) : 0;
    L.key1 = off;   off += pair ? align_up(e * sizeof(int32_t)) : 0;
    L.x0 = off;     off += pair ? align_up(e * sizeof(int32_t)) : 0;
    L.x1 = off;     off += pair ? align_up(e * sizeof(int32_t)) : 0;
    L.scan = off;   off += align_up(scan_temp_bytes((int64_t)(sides * N + 1)));
    L.total = off + 256;
    return L;
}

int build(const int64_t* a, const int64_t* b, int64_t E, int64_t N, int64_t b_limit, int pair, int32_t* rowptr0,
          int32_t* col0, int32_t* perm0, float* inv_deg0, int32_t* rowptr1, int32_t* col1, int32_t* perm1,
          float* inv_deg1, int32_t* xslot, float* w_src, int32_t* bad, void* workspace, size_t workspace_bytes,
          hipStream_t stream) {
    const Layout L = layout(E, N, pair);
    STIN_REQUIRE(workspace_bytes >= L.total, STIN_E_WORKSPACE);
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws + L.cnt);
    int32_t* cursor = reinterpret_cast<int32_t*>(ws + L.cursor);
    int32_t* id0 = reinterpret_cast<int32_t*>(ws + L.id0);
    int32_t* key0 = reinterpret_cast<int32_t*>(ws + L.key0);
    int32_t* id1 = reinterpret_cast<int32_t*>(ws + L.id1);
    int32_t* key1 = reinterpret_cast<int32_t*>(ws + L.key1);
    const int64_t sides = pair ? 2 : 1;
    const int64_t n_cnt = sides * N + 1;

    hipError_t err = hipMemsetAsync(cnt, 0, (size_t)n_cnt * sizeof(int32_t), stream);
    if (err != hipSuccess) return (int)err;
    if (E > 0) hipLaunchKernelGGL(k_count, dim3(grid_for(E)), dim3(T), 0, stream, a, b, E, N, b_limit, pair, cnt, bad);
    size_t scan_bytes = scan_temp_bytes(n_cnt);
    err = rocprim::inclusive_scan(ws + L.scan, scan_bytes, cnt, cnt, (size_t)n_cnt, rocprim::plus<int32_t>(), stream);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(k_rows, dim3(grid_for(N + 1)), dim3(T), 0, stream, cnt, N, pair, rowptr0, rowptr1, cursor,
                       inv_deg0, inv_deg1);
    if (E > 0) {
        hipLaunchKernelGGL(k_fill, dim3(grid_for(E)), dim3(T), 0, stream, a, b, E, N, b_limit, pair, cursor, id0, key0, id1,
                           key1);
        // out-of-range pairs were dropped, so the slot counts are the scanned totals; over-launch with E per side and
        // let the kernel stop at the true totals (read from rowptr[N] would need a sync): slots beyond the total
        // hold stale ids, so bound the grid by E and guard by row ranges instead.
        // cross-slot map for the masked backward: reuse the (now dead) fill buffers key0 / key1 as
        // slot_of_edge / perm_src
        int32_t* slot_of_edge = xslot != nullptr ? key0 + 0 : nullptr;
        int32_t* perm_src = xslot != nullptr ? key1 : perm1;
        if (xslot != nullptr) {
            // key0/key1 are still read by k_rank: use id-free scratch instead -> the cursor array (2N+2 ints) is too
            // small, so take dedicated space appended to the workspace layout
            slot_of_edge = reinterpret_cast<int32_t*>(ws + L.x0);
            perm_src = reinterpret_cast<int32_t*>(ws + L.x1);
        }
        hipLaunchKernelGGL(k_rank, dim3(grid_for(sides * E)), dim3(T), 0, stream, a, b, N, E, pair ? E : 0, rowptr0, rowptr1,
                           id0, key0, id1, key1, col0, perm0, col1, perm_src, slot_of_edge);
        if (xslot != nullptr)
            hipLaunchKernelGGL(k_xslot, dim3(grid_for(E)), dim3(T), 0, stream, perm_src, slot_of_edge, E, rowptr1, col1, inv_deg0, N,
                               xslot, w_src);
    }
    return stin_launch_status();
}

}  // namespace

extern "C" size_t stin_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    return layout(E, N, 1).total;   // sized for the pair build (covers the single build)
}

extern "C" int stin_csr_from_coo_i64(const int64_t* key, const int64_t* val, int64_t E, int64_t N, int64_t val_limit,
                                     int32_t* rowptr, int32_t* col, int32_t* perm, float* inv_deg, int32_t* bad,
                                     void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(E >= 0 && N >= 0 && N < ((int64_t)1 << 30) && E < ((int64_t)1 << 30), STIN_E_SIZE);
    STIN_REQUIRE(rowptr != nullptr && workspace != nullptr && (E == 0 || (key != nullptr && col != nullptr)), STIN_E_NULL);
    if (val != nullptr) STIN_REQUIRE(val_limit >= 0 && val_limit < ((int64_t)1 << 31), STIN_E_SIZE);
    return build(key, val, E, N, val_limit, 0, rowptr, col, perm, inv_deg, nullptr, nullptr, nullptr, nullptr, nullptr,
                 nullptr, bad, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int stin_csr_pair_from_edges_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N,
                                            int32_t* rowptr_dst, int32_t* col_dst, float* inv_deg_dst,
                                            int32_t* rowptr_src, int32_t* col_src, int32_t* xslot, float* w_src,
                                            int32_t* bad, void* workspace, size_t workspace_bytes,
                                            stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(E >= 0 && N >= 0 && N < ((int64_t)1 << 30) && E < ((int64_t)1 << 30), STIN_E_SIZE);
    STIN_REQUIRE(rowptr_dst && rowptr_src && workspace && (E == 0 || (src && dst && col_dst && col_src)), STIN_E_NULL);
    STIN_REQUIRE(w_src == nullptr || (xslot != nullptr && inv_deg_dst != nullptr), STIN_E_NULL);
    // side 0 groups by dst (value = src), side 1 groups by src (value = dst)
    return build(dst, src, E, N, N, 1, rowptr_dst, col_dst, nullptr, inv_deg_dst, rowptr_src, col_src, nullptr, nullptr,
                 xslot, w_src, bad, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int stin_narrow_i64_to_i32(const int64_t* src, int64_t n, int64_t limit, int32_t* dst, int32_t* bad,
                                      stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(n >= 0, STIN_E_SIZE);
    if (n == 0) return STIN_OK;
    STIN_REQUIRE(src != nullptr && dst != nullptr, STIN_E_NULL);
    hipLaunchKernelGGL(k_narrow, dim3(grid_for(n)), dim3(T), 0, (hipStream_t)stream_, src, n, limit, dst, bad);
    return stin_launch_status();
}
