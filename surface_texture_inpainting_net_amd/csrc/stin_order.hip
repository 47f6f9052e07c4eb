// Vertex renumbering by locality for one hierarchical sample, on the GPU (round 3; SURVEY 7.3: "optional vertex reordering,
// inverted at the boundary").  The edge kernels gather neighbour rows; on a mesh whose numbering has no locality (ScanNet order, the
// benchmark's random permutation) every gathered row crosses the fabric, with a space-following numbering most are L2 hits.
//   level 0    : 30-bit Morton code of the vertex positions (10 bits per axis on the bounding box), stable radix sort
//   level l + 1: key = the smallest NEW id among a vertex's children (integer atomicMin: deterministic), stable radix sort
//   rank[old] = new (int32 [n + 1], rank[n] = n: the out-of-range sentinel the relabelling maps bad ids to), order[new] = old
// and one batched kernel that relabels the index arrays (edge lists keep their ORDER, only the values change: every CSR row keeps
// the neighbour order of the reference's sequential scatter, every children list the order the arg-first max rule needs).
// The Morton quantisation repeats, operation by operation, the fp32 arithmetic of the torch formulation in plan.py
// (`((pos - lo) / (hi - lo).clamp_min(1e-20) * 1024).to(int64).clamp(0, 1023)`), so both give the same permutation (tested).
// Contract: include/stin_hip.h.
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include "stin_common.h"

namespace {

constexpr int T = 256;

// order-preserving float <-> uint (for integer atomicMin / atomicMax on floats)
__device__ __forceinline__ unsigned f2o(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// mm[0..2] = min x, y, z ; mm[3..5] = max (ordered-uint encoding; initialised to 0xffffffff / 0 by the caller's memsets)
__global__ __launch_bounds__(T) void k_bbox(const float* __restrict__ pos, int64_t ld, int64_t n, unsigned* __restrict__ mm) {
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int64_t v = (int64_t)blockIdx.x * T + threadIdx.x; v < n; v += (int64_t)gridDim.x * T) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float p = pos[v * ld + a];
            if (p == p) {                                   // (NaN positions do not move the box; torch's amin / amax would propagate them)
                const unsigned o = f2o(p);
                lo[a] = o < lo[a] ? o : lo[a];
                hi[a] = o > hi[a] ? o : hi[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned l2 = __shfl_xor(lo[a], off), h2 = __shfl_xor(hi[a], off);
            lo[a] = l2 < lo[a] ? l2 : lo[a];
            hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&mm[a], lo[a]);
            atomicMax(&mm[3 + a], hi[a]);
        }
    }
}

__device__ __forceinline__ unsigned spread10(unsigned v) {     // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    return (v | (v << 2)) & 0x09249249u;
}

__global__ __launch_bounds__(T) void k_morton(const float* __restrict__ pos, int64_t ld, int64_t n, const unsigned* __restrict__ mm,
                                              unsigned* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t v = (int64_t)blockIdx.x * T + threadIdx.x;
    if (v >= n) return;
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = o2f(mm[a]), hi = o2f(mm[3 + a]);
        float ext = hi - lo;
        ext = ext < 1e-20f ? 1e-20f : ext;                    // clamp_min(1e-20)
        const float t = (pos[v * ld + a] - lo) / ext * 1024.0f;
        long long qi = (t >= 0.f) ? (long long)t : 0;          // trunc toward zero; NaN / negative -> 0 (torch: clamp(0, 1023))
        q[a] = (unsigned)(qi > 1023 ? 1023 : qi);
    }
    keys[v] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    vals[v] = (int32_t)v;
}

__global__ __launch_bounds__(T) void k_iota(int64_t n, int32_t* __restrict__ vals) {
    const int64_t v = (int64_t)blockIdx.x * T + threadIdx.x;
    if (v < n) vals[v] = (int32_t)v;
}

// rank[order[i]] = i ; rank[n] = n ; order_out (optional) = order
__global__ __launch_bounds__(T) void k_rank_scatter(const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ rank,
                                                    int32_t* __restrict__ order_out) {
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i == 0) rank[n] = (int32_t)n;
    if (i >= n) return;
    const int32_t v = order[i];
    rank[v] = (int32_t)i;
    if (order_out != nullptr) order_out[i] = v;
}

// first[c] = min over the children v of c of rank_fine[v]  (first initialised to 0xffffffff; out-of-range traces are skipped)
__global__ __launch_bounds__(T) void k_first_child(const int64_t* __restrict__ trace, int64_t n_fine, int64_t n_coarse,
                                                   const int32_t* __restrict__ rank_fine, unsigned* __restrict__ first) {
    const int64_t v = (int64_t)blockIdx.x * T + threadIdx.x;
    if (v >= n_fine) return;
    const int64_t c = trace[v];
    if (c >= 0 && c < n_coarse) atomicMin(&first[c], (unsigned)rank_fine[v]);
}

struct RelabelBatch {
    stin_relabel_job_t j[STIN_RELABEL_MAX_JOBS];
    unsigned blk[STIN_RELABEL_MAX_JOBS + 1];
    int n;
};

__global__ __launch_bounds__(T) void k_relabel(const RelabelBatch b) {
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < b.n && blockIdx.x >= b.blk[ji + 1]) ++ji;
    const stin_relabel_job_t& J = b.j[ji];
    const int64_t i = (int64_t)(blockIdx.x - b.blk[ji]) * T + threadIdx.x;
    if (i >= J.n) return;
    const int64_t old = J.in[i];
    const int64_t neu = (old >= 0 && old < J.limit) ? (int64_t)J.rank[old] : J.limit;
    J.out[i] = neu;
    if (J.rank_fine != nullptr) {                             // a trace: also the pair's second member and the new fine -> coarse map
        const int32_t f = J.rank_fine[i];
        J.fine_out[i] = (int64_t)f;
        J.trace_out[f] = neu < J.limit ? (int32_t)neu : 0;
    }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + T - 1) / T); }

size_t sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned*)nullptr, (unsigned*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                    (size_t)(n > 0 ? n : 1), 0, 32, (hipStream_t)0);
    return bytes;
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t stin_vertex_order_workspace_bytes(int64_t n_max) {
    if (n_max <= 0) return 0;
    return 256 + 4 * up256((size_t)n_max * 4) + 256 + up256(sort_temp_bytes(n_max)) + 256;
}

extern "C" int stin_vertex_order_f32(const float* pos, int64_t ld_pos, const stin_order_level_t* levels, int n_levels,
                                     void* workspace, size_t workspace_bytes, stin_stream_t stream_) {
    stin_clear_stale_error();
    hipStream_t stream = (hipStream_t)stream_;
    STIN_REQUIRE(n_levels >= 1 && levels != nullptr && pos != nullptr && ld_pos >= 3, STIN_E_NULL);
    int64_t n_max = 0;
    for (int l = 0; l < n_levels; ++l) {
        STIN_REQUIRE(levels[l].n >= 1 && levels[l].n < ((int64_t)1 << 31) - 1 && levels[l].rank != nullptr, STIN_E_SIZE);
        STIN_REQUIRE(l == 0 || levels[l].trace != nullptr, STIN_E_NULL);
        if (levels[l].n > n_max) n_max = levels[l].n;
    }
    STIN_REQUIRE(workspace != nullptr && workspace_bytes >= stin_vertex_order_workspace_bytes(n_max), STIN_E_WORKSPACE);
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const size_t arr = up256((size_t)n_max * 4);
    unsigned* keys_in = reinterpret_cast<unsigned*>(p);
    unsigned* keys_out = reinterpret_cast<unsigned*>(p + arr);
    int32_t* vals_in = reinterpret_cast<int32_t*>(p + 2 * arr);
    int32_t* vals_out = reinterpret_cast<int32_t*>(p + 3 * arr);
    unsigned* mm = reinterpret_cast<unsigned*>(p + 4 * arr);
    void* temp = p + 4 * arr + 256;
    size_t temp_bytes = up256(sort_temp_bytes(n_max));

    // ---- level 0: bounding box, Morton keys, stable sort
    const int64_t n0 = levels[0].n;
    hipError_t e = hipMemsetAsync(mm, 0xff, 3 * sizeof(unsigned), stream);
    if (e == hipSuccess) e = hipMemsetAsync(mm + 3, 0, 3 * sizeof(unsigned), stream);
    if (e != hipSuccess) return (int)e;
    unsigned bb = blocks_for(n0);
    if (bb > 1024) bb = 1024;
    hipLaunchKernelGGL(k_bbox, dim3(bb), dim3(T), 0, stream, pos, ld_pos, n0, mm);
    hipLaunchKernelGGL(k_morton, dim3(blocks_for(n0)), dim3(T), 0, stream, pos, ld_pos, n0, mm, keys_in, vals_in);
    size_t tb = temp_bytes;
    e = rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n0, 0, 30, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_rank_scatter, dim3(blocks_for(n0)), dim3(T), 0, stream, vals_out, n0, levels[0].rank, levels[0].order);

    // ---- coarser levels: key = first (smallest new id) child
    for (int l = 1; l < n_levels; ++l) {
        const int64_t nf = levels[l - 1].n, nc = levels[l].n;
        e = hipMemsetAsync(keys_in, 0xff, (size_t)nc * sizeof(unsigned), stream);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k_first_child, dim3(blocks_for(nf)), dim3(T), 0, stream, levels[l].trace, nf, nc, levels[l - 1].rank, keys_in);
        hipLaunchKernelGGL(k_iota, dim3(blocks_for(nc)), dim3(T), 0, stream, nc, vals_in);
        tb = temp_bytes;
        e = rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)nc, 0, 32, stream);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k_rank_scatter, dim3(blocks_for(nc)), dim3(T), 0, stream, vals_out, nc, levels[l].rank, levels[l].order);
    }
    return stin_launch_status();
}

extern "C" int stin_relabel_many_i64(const stin_relabel_job_t* jobs, int n_jobs, stin_stream_t stream_) {
    stin_clear_stale_error();
    STIN_REQUIRE(n_jobs >= 0 && n_jobs <= STIN_RELABEL_MAX_JOBS && (n_jobs == 0 || jobs != nullptr), STIN_E_SIZE);
    RelabelBatch b;
    b.n = 0;
    unsigned blk = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const stin_relabel_job_t& J = jobs[i];
        STIN_REQUIRE(J.n >= 0 && J.limit >= 0, STIN_E_SIZE);
        if (J.n == 0) continue;
        STIN_REQUIRE(J.in && J.rank && J.out && (J.rank_fine == nullptr || (J.fine_out && J.trace_out)), STIN_E_NULL);
        b.j[b.n] = J;
        b.blk[b.n] = blk;
        blk += blocks_for(J.n);
        ++b.n;
    }
    b.blk[b.n] = blk;
    if (b.n == 0) return STIN_OK;
    hipLaunchKernelGGL(k_relabel, dim3(blk), dim3(T), 0, (hipStream_t)stream_, b);
    return stin_launch_status();
}
