// Offline graph preprocessing that feeds the hot path, on the GPU: the dilated-edge walk of
// the reference's preprocessing/graph_dilation.py:85-137 (python loops there: ~30 min per ScanNet scene).
// One thread per directed adjacency entry (centre c -> one-hop h); every walker is independent.  The arithmetic
// is written out operation by operation in the order of oracle/dilation_oracle.py (no FMA contraction: this
// library is built with -ffp-contract=off; IEEE division and square root) so the selected vertices are
// bit-identical to the CPU restatement.
#include "stin_common.h"

namespace {

constexpr int BLOCK = 256;

template <typename T> struct V3 { T x, y, z; };
template <typename T> __device__ __forceinline__ V3<T> ld3(const T* p, int64_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
template <typename T> __device__ __forceinline__ V3<T> sub(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float root(float v) { return sqrtf(v); }
__device__ __forceinline__ double root(double v) { return sqrt(v); }
template <typename T> __device__ __forceinline__ T norm(V3<T> a) { return root(dot(a, a)); }
// the reference's plane_projection (graph_dilation.py:27-28): u - n * dot(u, n) / (|n| |u|), evaluated left to right
template <typename T> __device__ __forceinline__ V3<T> plane_projection(V3<T> n, V3<T> u) {
    const T d = dot(u, n);
    const T den = norm(n) * norm(u);
    return {u.x - (n.x * d) / den, u.y - (n.y * d) / den, u.z - (n.z * d) / den};
}
template <typename T> __device__ __forceinline__ T cosine(V3<T> a, V3<T> b) { return dot(a, b) / (norm(a) * norm(b)); }

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dilated_walk(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const int32_t* __restrict__ row_of, const T* __restrict__ pos,
                                                        const T* __restrict__ nrm, int64_t E, uint64_t want, int max_d,
                                                        int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= E) return;
    const int c = row_of[e], h = col[e];
    if (h == c) return;                                            // graph_dilation.py:96
    const int cb = rowptr[c], ce = rowptr[c + 1];
    int last = c, cur = h;
    V3<T> cur_n = ld3(nrm, cur);
    V3<T> dir = sub(ld3(pos, cur), ld3(pos, last));
    int di = 0;
    for (int d = 2; d <= max_d; ++d) {
        int best = -1;
        T best_sim = (T)0;                                         // > 90 degrees from the running direction never qualifies (:104-107)
        const V3<T> pc = ld3(pos, cur);
        const V3<T> pd = plane_projection(cur_n, dir);
        for (int a = rowptr[cur]; a < rowptr[cur + 1]; ++a) {
            const int nb = col[a];
            if (nb == last) continue;
            bool in_hood = false;                                  // `neighbor_idx not in one_hop_inds` (:111)
            for (int q = cb; q < ce; ++q) in_hood |= (col[q] == nb);
            if (in_hood) continue;
            const T sim = cosine(pd, plane_projection(cur_n, sub(ld3(pos, nb), pc)));
            if (sim >= best_sim) {                                 // NaN never qualifies; the LAST maximum wins (:115)
                best_sim = sim;
                best = nb;
            }
        }
        if (best < 0) break;
        if ((want >> d) & 1ull) out[(int64_t)di++ * E + e] = best;   // edge [far, centre] (:125-128)
        last = cur;
        cur = best;
        cur_n = ld3(nrm, cur);
        dir = plane_projection(cur_n, dir);
        const T nn = norm(dir);
        dir = {dir.x / nn, dir.y / nn, dir.z / nn};
    }
}

template <typename T>
int walk_impl(const int32_t* rowptr, const int32_t* col, const int32_t* row_of, const T* pos, const T* nrm, int64_t N, int64_t E,
              const int32_t* dilations, int n_dil, int32_t* out, hipStream_t stream) {
    STIN_REQUIRE(N >= 0 && E >= 0 && n_dil > 0, STIN_E_SIZE);
    STIN_REQUIRE(dilations != nullptr, STIN_E_NULL);
    uint64_t want = 0;
    int max_d = 0;
    for (int i = 0; i < n_dil; ++i) {
        STIN_REQUIRE(dilations[i] >= 2 && dilations[i] < 64, STIN_E_UNSUPPORTED);
        STIN_REQUIRE(i == 0 || dilations[i] > dilations[i - 1], STIN_E_UNSUPPORTED);   // ascending, as the reference consumes them
        want |= 1ull << dilations[i];
        max_d = dilations[i];
    }
    if (E == 0) return STIN_OK;
    STIN_REQUIRE(rowptr && col && row_of && pos && nrm && out, STIN_E_NULL);
    hipError_t err = hipMemsetAsync(out, 0xff, sizeof(int32_t) * (size_t)n_dil * (size_t)E, stream);   // -1 = walker stopped earlier
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL((k_dilated_walk<T>), dim3((unsigned)((E + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, rowptr, col, row_of,
                       pos, nrm, E, want, max_d, out);
    return stin_launch_status();
}

}  // namespace

extern "C" int stin_dilated_walk_f32(const int32_t* rowptr, const int32_t* col, const int32_t* row_of, const float* pos,
                                     const float* nrm, int64_t N, int64_t E, const int32_t* dilations, int n_dil, int32_t* out,
                                     stin_stream_t stream) {
    stin_clear_stale_error();
    return walk_impl<float>(rowptr, col, row_of, pos, nrm, N, E, dilations, n_dil, out, (hipStream_t)stream);
}

extern "C" int stin_dilated_walk_f64(const int32_t* rowptr, const int32_t* col, const int32_t* row_of, const double* pos,
                                     const double* nrm, int64_t N, int64_t E, const int32_t* dilations, int n_dil, int32_t* out,
                                     stin_stream_t stream) {
    stin_clear_stale_error();
    return walk_impl<double>(rowptr, col, row_of, pos, nrm, N, E, dilations, n_dil, out, (hipStream_t)stream);
}
