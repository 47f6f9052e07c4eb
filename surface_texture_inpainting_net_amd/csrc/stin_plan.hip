// COO -> CSR graph-plan construction on the GPU (once per sample): a counting sort, for ALL structures of a sample at once.
//   1. count : one returning integer atomicAdd per (edge, side): the arrival index of the edge in its row + the histogram
//   2. scan  : one rocPRIM inclusive scan over the concatenated histograms of every job -> row pointers
//   3. rows  : row pointers / 1 / max(1, degree) of every job
//   4. fill  : every edge drops its id at rowptr[row] + arrival index (plain stores, no second pass of atomics)
//   5. rank  : every edge finds the rank of its id among the ids of its row and writes the final entry there
//              => within a row entries are in ORIGINAL pair order, exactly a stable sort, deterministic although step 1
//              uses atomics; for an edge set the same thread places the edge in BOTH CSRs, so the cross map
//              xslot[source-CSR slot] = destination-CSR slot and w_src come out of the same kernel.
// A JOB is one CSR (pool map: children of each coarse vertex, optionally with the int64 -> int32 narrowing of the trace) or
// both CSRs of one directed edge set.  Round 3: the 7 edge sets and 2 pool maps of a 3-level sample are ONE batch = 7
// launches (memset, count, scan x 2, rows, fill, rank) where rounds 1-2 ran ~8 launches per structure (72 per sample),
// the atomic passes are halved (one returning atomic per edge and side instead of a histogram pass plus a cursor pass),
// and the small coarse-level builds fill the chip together instead of one after the other.
// Contract: include/stin_hip.h.
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include "stin_common.h"

namespace {

constexpr int T = 256;

struct Batch {
    stin_plan_job_t j[STIN_PLAN_MAX_JOBS];
    int64_t e_off[STIN_PLAN_MAX_JOBS + 1];        // first global edge index of job i (prefix sums of E): index into the scratch arrays
    int64_t c_off[STIN_PLAN_MAX_JOBS + 1];        // first counter of job i: the job owns sides * N + 1 counters, [0] stays 0
    unsigned e_blk[STIN_PLAN_MAX_JOBS + 1];       // first BLOCK of job i in the per-edge launches (every block serves one job:
    unsigned c_blk[STIN_PLAN_MAX_JOBS + 1];       //   the job lookup is wave-uniform = scalar loads), and in the per-counter launch
    int n;
};

// block -> job (n <= 16: a scan over kernel-argument scalars, uniform for the whole block)
__device__ __forceinline__ int job_of(const unsigned* blk, int n) {
    int i = 0;
#pragma unroll 1
    while (i + 1 < n && blockIdx.x >= blk[i + 1]) ++i;
    return i;
}

__global__ __launch_bounds__(T) void k_count(const Batch b, int32_t* __restrict__ cnt, int32_t* __restrict__ pos0,
                                             int32_t* __restrict__ pos1, int32_t* __restrict__ bad) {
    const int ji = job_of(b.e_blk, b.n);
    const stin_plan_job_t& J = b.j[ji];
    const int64_t e = (int64_t)(blockIdx.x - b.e_blk[ji]) * T + threadIdx.x;
    if (e >= J.E) return;
    const int64_t g = b.e_off[ji] + e;
    const int64_t ka = J.a[e];
    bool oob = (ka < 0) | (ka >= J.N);
    int64_t kb = 0;
    if (J.b != nullptr) {
        kb = J.b[e];
        oob |= (kb < 0) | (kb >= J.b_limit);
    }
    if (J.narrow_out != nullptr) J.narrow_out[e] = oob ? 0 : (int32_t)ka;
    if (oob) {
        if (bad != nullptr) atomicOr(bad, 1);
        pos0[g] = -1;                 // dropped from the plan; the host raises IndexError before using it
        return;
    }
    int32_t* c = cnt + b.c_off[ji];
    pos0[g] = atomicAdd(&c[1 + ka], 1);
    if (J.pair) pos1[g] = atomicAdd(&c[1 + J.N + kb], 1);
}

// scanned histograms -> rowptr(s), inv_deg
__global__ __launch_bounds__(T) void k_rows(const Batch b, const int32_t* __restrict__ scanned) {
    const int ji = job_of(b.c_blk, b.n);
    const stin_plan_job_t& J = b.j[ji];
    int64_t n = (int64_t)(blockIdx.x - b.c_blk[ji]) * T + threadIdx.x;      // 0 .. sides * N
    if (n > (J.pair ? 2 : 1) * J.N) return;
    const int32_t* s = scanned + b.c_off[ji];
    const int32_t base = s[0];                         // everything counted by earlier jobs
    if (n <= J.N) {
        const int32_t p0 = s[n] - base;
        J.rowptr0[n] = p0;
        if (n < J.N && J.inv_deg0 != nullptr) {
            const int32_t d = s[n + 1] - base - p0;
            J.inv_deg0[n] = 1.0f / (float)(d > 0 ? d : 1);
        }
    }
    if (J.pair && n >= J.N) {                          // side 1 shares counter N (= total of side 0) as its zero
        n -= J.N;
        J.rowptr1[n] = s[J.N + n] - s[J.N];
    }
}

__global__ __launch_bounds__(T) void k_fill(const Batch b, const int32_t* __restrict__ pos0, const int32_t* __restrict__ pos1,
                                            int32_t* __restrict__ id0, int32_t* __restrict__ id1) {
    const int ji = job_of(b.e_blk, b.n);
    const stin_plan_job_t& J = b.j[ji];
    const int64_t e = (int64_t)(blockIdx.x - b.e_blk[ji]) * T + threadIdx.x;
    if (e >= J.E) return;
    const int64_t eo = b.e_off[ji], g = eo + e;
    const int32_t p0 = pos0[g];
    if (p0 < 0) return;
    id0[eo + J.rowptr0[J.a[e]] + p0] = (int32_t)e;
    if (J.pair) id1[eo + J.rowptr1[J.b[e]] + pos1[g]] = (int32_t)e;
}

__device__ __forceinline__ int32_t rank_in_row(const int32_t* __restrict__ ids, int32_t beg, int32_t end, int32_t mine) {
    int32_t rank = 0;
    for (int32_t u = beg; u < end; ++u) rank += (ids[u] < mine) ? 1 : 0;
    return rank;
}

__global__ __launch_bounds__(T) void k_rank(const Batch b, const int32_t* __restrict__ pos0, const int32_t* __restrict__ id0,
                                            const int32_t* __restrict__ id1) {
    const int ji = job_of(b.e_blk, b.n);
    const stin_plan_job_t& J = b.j[ji];
    const int64_t e64 = (int64_t)(blockIdx.x - b.e_blk[ji]) * T + threadIdx.x;
    if (e64 >= J.E) return;
    const int64_t eo = b.e_off[ji];
    if (pos0[eo + e64] < 0) return;
    const int32_t e = (int32_t)e64;
    const int64_t ka = J.a[e];
    const int32_t beg0 = J.rowptr0[ka];
    const int32_t slot0 = beg0 + rank_in_row(id0 + eo, beg0, J.rowptr0[ka + 1], e);
    // value stored with the entry: the OTHER endpoint (or the pair id itself when there is no value array)
    const int64_t kb = J.b != nullptr ? J.b[e] : 0;
    J.col0[slot0] = J.b != nullptr ? (int32_t)kb : e;
    if (J.perm0 != nullptr) J.perm0[slot0] = e;
    if (J.pair) {
        const int32_t beg1 = J.rowptr1[kb];
        const int32_t slot1 = beg1 + rank_in_row(id1 + eo, beg1, J.rowptr1[kb + 1], e);
        J.col1[slot1] = (int32_t)ka;
        // cross map for the masked backward: the destination-CSR slot of the edge at this source-CSR slot, and the mean
        // weight of the edge's target laid out sequentially for the dB pass
        if (J.xslot != nullptr) J.xslot[slot1] = slot0;
        if (J.w_src != nullptr) J.w_src[slot1] = J.inv_deg0[ka];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the same plans from ONE stable radix sort instead of returning atomics.  The counting sort above issues two
// returning device-scope atomics per edge on random rows (~21 G atomics/s: k_count alone 200 us for the headline scene) beside
// the first kernels of the step it is prefetched for; here every (edge, side) becomes an item with key = the global index of its
// row's counter (c_off[job] + side * N + row; out-of-range pairs get the key past the last row) and value = its item index, one
// rocPRIM radix sort over the bits that are needed groups the items by row IN ORIGINAL ORDER (stable) - rowptr is a binary
// search per row, every output a gather.  No atomics except the out-of-range flag; identical results (k_rank's order is the
// stable order): tests/test_hip_parity.py::test_plan_build_by_sort_equals_the_counting_sort.
//   items [0, total_E): side 0 of global edge g;  items [total_E, 2 total_E): side 1 of global edge g - total_E (pair jobs only;
//   the slots of non-pair jobs carry the invalid key)
typedef unsigned long long u64;
__global__ __launch_bounds__(T) void k_sort_keys(const Batch b, int64_t total_E, uint32_t invalid, u64* __restrict__ items,
                                                 int32_t* __restrict__ bad) {
    const int ji = job_of(b.e_blk, b.n);
    const stin_plan_job_t& J = b.j[ji];
    const int64_t e = (int64_t)(blockIdx.x - b.e_blk[ji]) * T + threadIdx.x;
    if (e >= J.E) return;
    const int64_t g = b.e_off[ji] + e;
    const int64_t ka = J.a[e];
    bool oob = (ka < 0) | (ka >= J.N);
    int64_t kb = 0;
    if (J.b != nullptr) {
        kb = J.b[e];
        oob |= (kb < 0) | (kb >= J.b_limit);
    }
    if (J.narrow_out != nullptr) J.narrow_out[e] = oob ? 0 : (int32_t)ka;
    if (oob && bad != nullptr) atomicOr(bad, 1);
    // counter layout of the counting sort: [0] = 0, [1 + n] = row n of side 0, [1 + N + n] = row n of side 1
    // one 64-bit item = key in the high word (the sort looks at those bits only), item id in the low word: the id rides along
    items[g] = ((u64)(oob ? invalid : (uint32_t)(b.c_off[ji] + ka)) << 32) | (u64)(uint32_t)g;
    items[total_E + g] = ((u64)((oob || !J.pair) ? invalid : (uint32_t)(b.c_off[ji] + J.N + kb)) << 32) | (u64)(uint32_t)(total_E + g);
}

__device__ __forceinline__ int32_t lower_bound_key(const u64* __restrict__ a, int64_t n, uint32_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((uint32_t)(a[mid] >> 32) < key) lo = mid + 1;
        else hi = mid;
    }
    return (int32_t)lo;
}

// start[k] = first sorted position with key >= k, for every counter index k in [0, total_cnt]: exactly the exclusive prefix
// sums the counting sort's scan produces (start[c_off + n] = first entry of row n of side 0, start[c_off + N + n] of side 1), so
// k_rows turns them into rowptr0 / rowptr1 / inv_deg0 unchanged
__global__ __launch_bounds__(T) void k_sort_start(const u64* __restrict__ sorted, int64_t items, int64_t total_cnt,
                                                  int32_t* __restrict__ start) {
    const int64_t k = (int64_t)blockIdx.x * T + threadIdx.x;
    if (k > total_cnt) return;
    start[k] = lower_bound_key(sorted, items, (uint32_t)k);
}

// sorted position p -> the CSR entry of its item (after k_rows: w_src reads inv_deg0).  side 0 (phase 0): col0 / perm0 and the
// edge's destination-CSR slot; side 1 (phase 1, after phase 0): col1, xslot, w_src.  (One pass that finds the destination slot by
// a binary search in the edge's destination row instead of the scratch array measured slower: 84 us vs 2 x 38.)
__global__ __launch_bounds__(T) void k_sort_out(const Batch b, int64_t total_E, int64_t items, uint32_t invalid,
                                                const u64* __restrict__ sorted, const int32_t* __restrict__ start,
                                                int32_t* __restrict__ slot_of_edge, int phase) {
    const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
    if (p >= items) return;
    const u64 it = sorted[p];
    if ((uint32_t)(it >> 32) == invalid) return;
    const uint32_t item = (uint32_t)it;
    const int side = item >= (uint32_t)total_E ? 1 : 0;
    if (side != phase) return;
    const int64_t g = (int64_t)item - (side ? total_E : 0);
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < b.n && g >= b.e_off[ji + 1]) ++ji;
    const stin_plan_job_t& J = b.j[ji];
    const int32_t e = (int32_t)(g - b.e_off[ji]);
    if (side == 0) {
        const int32_t slot0 = (int32_t)p - start[b.c_off[ji]];
        J.col0[slot0] = J.b != nullptr ? (int32_t)J.b[e] : e;
        if (J.perm0 != nullptr) J.perm0[slot0] = e;
        slot_of_edge[g] = slot0;
    } else {
        const int32_t slot1 = (int32_t)p - start[b.c_off[ji] + J.N];
        const int64_t ka = J.a[e];
        J.col1[slot1] = (int32_t)ka;
        if (J.xslot != nullptr) J.xslot[slot1] = slot_of_edge[g];
        if (J.w_src != nullptr) J.w_src[slot1] = J.inv_deg0[ka];
    }
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
inline unsigned grid_for(int64_t n) { return (unsigned)((n + T - 1) / T); }

size_t scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)(n > 0 ? n : 1),
                                  rocprim::plus<int32_t>(), (hipStream_t)0);
    return bytes;
}

size_t sort_temp_bytes(int64_t items) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                   (size_t)(items > 0 ? items : 1), 32, 64, (hipStream_t)0);
    return bytes;
}

struct Layout {
    size_t cnt, pos0, pos1, id0, id1, scan, total;
    size_t keys0, keys1, vals0, vals1, slot, bases, sort;      // the sort-based build (the same workspace serves either)
};
Layout layout(int64_t total_E, int64_t total_cnt) {
    Layout L;
    const size_t e = (size_t)(total_E > 0 ? total_E : 1), c = (size_t)(total_cnt > 0 ? total_cnt : 1);
    size_t off = 0;
    L.cnt = off;  off += align_up(c * sizeof(int32_t));
    L.pos0 = off; off += align_up(e * sizeof(int32_t));
    L.pos1 = off; off += align_up(e * sizeof(int32_t));
    L.id0 = off;  off += align_up(e * sizeof(int32_t));
    L.id1 = off;  off += align_up(e * sizeof(int32_t));
    L.scan = off; off += align_up(scan_temp_bytes((int64_t)c));
    const size_t count_total = off;
    off = 0;
    L.keys0 = off; off += align_up(2 * e * sizeof(uint32_t));
    L.keys1 = off; off += align_up(2 * e * sizeof(uint32_t));
    L.vals0 = off; off += align_up(2 * e * sizeof(uint32_t));
    L.vals1 = off; off += align_up(2 * e * sizeof(uint32_t));
    L.slot = off;  off += align_up(e * sizeof(int32_t));
    L.bases = off; off += align_up((c + 1) * sizeof(int32_t));                  // start[0 .. total_cnt]
    L.sort = off;  off += align_up(sort_temp_bytes((int64_t)(2 * e)));
    L.total = (off > count_total ? off : count_total) + 256;
    return L;
}

// STIN_PLAN_SORT = 0 keeps the counting sort, 1 forces the sort form (A/B and test switch, re-read per call); default: the sort
// form from 400 k edges per batch up - below that the atomics are few and the sort's extra launches (five rocPRIM kernels and
// seven memsets) cost a launch-bound step more than they save (20 k-vertex crop under HIP-graph replay: 2.56 vs 2.65 ms)
inline bool plan_sort_on(int64_t total_E) {
    const char* e = getenv("STIN_PLAN_SORT");
    if (e != nullptr) return atoi(e) != 0;
    return total_E >= 400000;
}

int check_job(const stin_plan_job_t& J) {
    STIN_REQUIRE(J.E >= 0 && J.N >= 0 && J.N < ((int64_t)1 << 30) && J.E < ((int64_t)1 << 30), STIN_E_SIZE);
    STIN_REQUIRE(J.rowptr0 != nullptr && (J.E == 0 || (J.a != nullptr && J.col0 != nullptr)), STIN_E_NULL);
    if (J.b != nullptr) STIN_REQUIRE(J.b_limit >= 0 && J.b_limit < ((int64_t)1 << 31), STIN_E_SIZE);
    if (J.pair) {
        STIN_REQUIRE(J.rowptr1 != nullptr && (J.E == 0 || (J.b != nullptr && J.col1 != nullptr)), STIN_E_NULL);
        STIN_REQUIRE(J.w_src == nullptr || (J.xslot != nullptr && J.inv_deg0 != nullptr), STIN_E_NULL);
    }
    return STIN_OK;
}

int build(const stin_plan_job_t* jobs, int n_jobs, int32_t* bad, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    STIN_REQUIRE(n_jobs >= 0 && n_jobs <= STIN_PLAN_MAX_JOBS, STIN_E_SIZE);
    if (n_jobs == 0) return STIN_OK;
    STIN_REQUIRE(jobs != nullptr && workspace != nullptr, STIN_E_NULL);
    Batch b;
    b.n = n_jobs;
    b.e_off[0] = b.c_off[0] = 0;
    b.e_blk[0] = b.c_blk[0] = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const int rc = check_job(jobs[i]);
        if (rc != STIN_OK) return rc;
        b.j[i] = jobs[i];
        const int64_t counters = (jobs[i].pair ? 2 : 1) * jobs[i].N + 1;
        b.e_off[i + 1] = b.e_off[i] + jobs[i].E;
        b.c_off[i + 1] = b.c_off[i] + counters;
        b.e_blk[i + 1] = b.e_blk[i] + grid_for(jobs[i].E);
        b.c_blk[i + 1] = b.c_blk[i] + grid_for(counters);
    }
    for (int i = n_jobs; i < STIN_PLAN_MAX_JOBS; ++i) {
        memset(&b.j[i], 0, sizeof(stin_plan_job_t));
        b.e_off[i + 1] = b.e_off[n_jobs];
        b.c_off[i + 1] = b.c_off[n_jobs];
        b.e_blk[i + 1] = b.e_blk[n_jobs];
        b.c_blk[i + 1] = b.c_blk[n_jobs];
    }
    const int64_t total_E = b.e_off[n_jobs], total_cnt = b.c_off[n_jobs];
    STIN_REQUIRE(total_E < ((int64_t)1 << 31) && total_cnt < ((int64_t)1 << 31), STIN_E_SIZE);
    const Layout L = layout(total_E, total_cnt);
    STIN_REQUIRE(workspace_bytes >= L.total, STIN_E_WORKSPACE);
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws + L.cnt);
    int32_t* pos0 = reinterpret_cast<int32_t*>(ws + L.pos0);
    int32_t* pos1 = reinterpret_cast<int32_t*>(ws + L.pos1);
    int32_t* id0 = reinterpret_cast<int32_t*>(ws + L.id0);
    int32_t* id1 = reinterpret_cast<int32_t*>(ws + L.id1);

    const unsigned e_blocks = b.e_blk[n_jobs], c_blocks = b.c_blk[n_jobs];
    if (plan_sort_on(total_E) && total_cnt < ((int64_t)1 << 31) - 2) {
        u64* it0 = reinterpret_cast<u64*>(ws + L.keys0);                        // (keys0 | keys1 and vals0 | vals1 are adjacent: 2 x 16 e bytes)
        u64* it1 = reinterpret_cast<u64*>(ws + L.vals0);
        int32_t* slot = reinterpret_cast<int32_t*>(ws + L.slot);
        int32_t* start = reinterpret_cast<int32_t*>(ws + L.bases);
        const int64_t items = 2 * total_E;
        const uint32_t invalid = (uint32_t)total_cnt;                          // one past the last counter: sorts behind every row
        int bits = 1;
        while (bits < 32 && ((uint64_t)1 << bits) <= (uint64_t)invalid) ++bits;
        if (e_blocks > 0) {
            hipLaunchKernelGGL(k_sort_keys, dim3(e_blocks), dim3(T), 0, stream, b, total_E, invalid, it0, bad);
            size_t sb = sort_temp_bytes(items);
            const hipError_t es = rocprim::radix_sort_keys(ws + L.sort, sb, it0, it1, (size_t)items, 32u, 32u + (unsigned)bits, stream);
            if (es != hipSuccess) return (int)es;
        }
        // (no edges at all: every probe finds position 0 in an empty list - rowptr all zero)
        hipLaunchKernelGGL(k_sort_start, dim3(grid_for(total_cnt + 1)), dim3(T), 0, stream, it1, e_blocks > 0 ? items : 0, total_cnt, start);
        hipLaunchKernelGGL(k_rows, dim3(c_blocks), dim3(T), 0, stream, b, start);
        if (e_blocks > 0) {
            const unsigned ib = grid_for(items);
            hipLaunchKernelGGL(k_sort_out, dim3(ib), dim3(T), 0, stream, b, total_E, items, invalid, it1, start, slot, 0);
            bool any_pair = false;
            for (int i = 0; i < n_jobs; ++i) any_pair |= jobs[i].pair != 0;
            if (any_pair)
                hipLaunchKernelGGL(k_sort_out, dim3(ib), dim3(T), 0, stream, b, total_E, items, invalid, it1, start, slot, 1);
        }
        return stin_launch_status();
    }
    hipError_t err = hipMemsetAsync(cnt, 0, (size_t)total_cnt * sizeof(int32_t), stream);
    if (err != hipSuccess) return (int)err;
    if (e_blocks > 0) hipLaunchKernelGGL(k_count, dim3(e_blocks), dim3(T), 0, stream, b, cnt, pos0, pos1, bad);
    size_t scan_bytes = scan_temp_bytes(total_cnt);
    err = rocprim::inclusive_scan(ws + L.scan, scan_bytes, cnt, cnt, (size_t)total_cnt, rocprim::plus<int32_t>(), stream);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(k_rows, dim3(c_blocks), dim3(T), 0, stream, b, cnt);
    if (e_blocks > 0) {
        hipLaunchKernelGGL(k_fill, dim3(e_blocks), dim3(T), 0, stream, b, pos0, pos1, id0, id1);
        hipLaunchKernelGGL(k_rank, dim3(e_blocks), dim3(T), 0, stream, b, pos0, id0, id1);
    }
    return stin_launch_status();
}

}  // namespace

extern "C" size_t stin_plan_build_workspace_bytes(int64_t total_E, int64_t total_counters) {
    if (total_E < 0 || total_counters < 0) return 0;
    return layout(total_E, total_counters).total;
}

extern "C" int stin_plan_build_many(const stin_plan_job_t* jobs, int n_jobs, int32_t* bad, void* workspace, size_t workspace_bytes,
                                    stin_stream_t stream) {
    stin_clear_stale_error();
    return build(jobs, n_jobs, bad, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t stin_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    return layout(E, 2 * N + 1).total;   // sized for the pair build (covers the single build)
}

extern "C" int stin_csr_from_coo_i64(const int64_t* key, const int64_t* val, int64_t E, int64_t N, int64_t val_limit,
                                     int32_t* rowptr, int32_t* col, int32_t* perm, float* inv_deg, int32_t* bad,
                                     void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    stin_plan_job_t J;
    memset(&J, 0, sizeof(J));
    J.a = key;
    J.b = val;
    J.E = E;
    J.N = N;
    J.b_limit = val_limit;
    J.rowptr0 = rowptr;
    J.col0 = col;
    J.perm0 = perm;
    J.inv_deg0 = inv_deg;
    return build(&J, 1, bad, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int stin_csr_pair_from_edges_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N,
                                            int32_t* rowptr_dst, int32_t* col_dst, float* inv_deg_dst,
                                            int32_t* rowptr_src, int32_t* col_src, int32_t* xslot, float* w_src,
                                            int32_t* bad, void* workspace, size_t workspace_bytes,
                                            stin_stream_t stream_) {
    stin_clear_stale_error();
    // side 0 groups by dst (value = src), side 1 groups by src (value = dst)
    stin_plan_job_t J;
    memset(&J, 0, sizeof(J));
    J.a = dst;
    J.b = src;
    J.E = E;
    J.N = N;
    J.b_limit = N;
    J.pair = 1;
    J.rowptr0 = rowptr_dst;
    J.col0 = col_dst;
    J.inv_deg0 = inv_deg_dst;
    J.rowptr1 = rowptr_src;
    J.col1 = col_src;
    J.xslot = xslot;
    J.w_src = w_src;
    return build(&J, 1, bad, workspace, workspace_bytes, (hipStream_t)stream_);
}

namespace {
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
}  // namespace

extern "C" int stin_narrow_i64_to_i32(const int64_t* src, int64_t n, int64_t limit, int32_t* dst, int32_t* bad,
                                      stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(n >= 0, STIN_E_SIZE);
    if (n == 0) return STIN_OK;
    STIN_REQUIRE(src != nullptr && dst != nullptr, STIN_E_NULL);
    hipLaunchKernelGGL(k_narrow, dim3(grid_for(n)), dim3(T), 0, (hipStream_t)stream_, src, n, limit, dst, bad);
    return stin_launch_status();
}
