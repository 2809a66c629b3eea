// CSR staging of COO edge lists and of sorted graph-id vectors.
// Reference semantics replaced: PyG MessagePassing.propagate's per-call gather/scatter bookkeeping
// over edge_index (src_1gp/layer.py:40, :86) and the `batch` vector consumed by the global pools
// (src_1gp/layer.py:202).  Stable within a segment => downstream float sums are bit-reproducible.
#include "common.h"

namespace glam {

constexpr int kScanItems = 2048;  // items per scan block (256 threads x 8)

__global__ void __launch_bounds__(kBlock) k_csr_count(const int64_t* key, const int64_t* val, int E, int N,
                                                     int* count, int* err) {
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < E; e += gridDim.x * kBlock) {
        const int64_t k = key[e], v = val[e];
        if (k < 0 || k >= N || v < 0 || v >= N) { *err = 1; continue; }
        atomicAdd(&count[k], 1);   // integer: order-independent
    }
}

// block-local sums of count[] chunks
__global__ void __launch_bounds__(kBlock) k_scan_block_sums(const int* count, int n, int* block_sums) {
    __shared__ int s_part[kBlock];
    const int base = blockIdx.x * kScanItems;
    int t = 0;
    for (int i = threadIdx.x; i < kScanItems; i += kBlock) {
        const int idx = base + i;
        if (idx < n) t += count[idx];
    }
    s_part[threadIdx.x] = t;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0];
}

// exclusive scan of block_sums in place (single block, sequential over chunks)
__global__ void __launch_bounds__(kBlock) k_scan_top(int* block_sums, int nb) {
    __shared__ int s_val[kBlock];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += kBlock) {
        const int idx = base + threadIdx.x;
        const int v = idx < nb ? block_sums[idx] : 0;
        s_val[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < kBlock; o <<= 1) {
            const int add = threadIdx.x >= o ? s_val[threadIdx.x - o] : 0;
            __syncthreads();
            s_val[threadIdx.x] += add;
            __syncthreads();
        }
        const int incl = s_val[threadIdx.x], carry = s_carry;
        if (idx < nb) block_sums[idx] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) s_carry = carry + incl;
        __syncthreads();
    }
}

// rowptr[i] = exclusive prefix of count; each thread owns 8 consecutive items
__global__ void __launch_bounds__(kBlock) k_scan_apply(const int* count, int n, const int* block_offs, int* rowptr) {
    __shared__ int s_val[kBlock];
    const int base = blockIdx.x * kScanItems + threadIdx.x * 8;
    int v[8], t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (base + i < n) ? count[base + i] : 0; t += v[i]; }
    s_val[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {
        const int add = threadIdx.x >= o ? s_val[threadIdx.x - o] : 0;
        __syncthreads();
        s_val[threadIdx.x] += add;
        __syncthreads();
    }
    int run = block_offs[blockIdx.x] + s_val[threadIdx.x] - t;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (base + i <= n) rowptr[base + i] = run;   // writes rowptr[n] too (count[n] is 0)
        run += v[i];
    }
}

__global__ void __launch_bounds__(kBlock) k_csr_fill(const int64_t* key, const int64_t* val, int E, int N,
                                                    const int* rowptr, int* cursor, int* nbr, int* eid) {
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < E; e += gridDim.x * kBlock) {
        const int64_t k = key[e], v = val[e];
        if (k < 0 || k >= N || v < 0 || v >= N) continue;
        const int pos = rowptr[k] + atomicAdd(&cursor[k], 1);
        nbr[pos] = (int)v;
        eid[pos] = e;
    }
}

// restore original edge order inside every segment (the atomic cursor above is unordered)
__global__ void __launch_bounds__(kBlock) k_csr_sort_segments(const int* rowptr, int N, int* nbr, int* eid) {
    for (int n = blockIdx.x * kBlock + threadIdx.x; n < N; n += gridDim.x * kBlock) {
        const int beg = rowptr[n], len = rowptr[n + 1] - beg;
        int* e = eid + beg;
        int* v = nbr + beg;
        if (len <= 32) {
            for (int i = 1; i < len; ++i) {
                const int ke = e[i], kv = v[i];
                int j = i - 1;
                while (j >= 0 && e[j] > ke) { e[j + 1] = e[j]; v[j + 1] = v[j]; --j; }
                e[j + 1] = ke; v[j + 1] = kv;
            }
        } else {  // heap sort, in place, O(len log len)
            auto sift = [&](int root, int end) {
                for (;;) {
                    int child = 2 * root + 1;
                    if (child >= end) break;
                    if (child + 1 < end && e[child] < e[child + 1]) ++child;
                    if (e[root] >= e[child]) break;
                    int te = e[root]; e[root] = e[child]; e[child] = te;
                    int tv = v[root]; v[root] = v[child]; v[child] = tv;
                    root = child;
                }
            };
            for (int i = len / 2 - 1; i >= 0; --i) sift(i, len);
            for (int end = len - 1; end > 0; --end) {
                int te = e[0]; e[0] = e[end]; e[end] = te;
                int tv = v[0]; v[0] = v[end]; v[end] = tv;
                sift(0, end);
            }
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_batch_ptr(const int64_t* batch, int N, int B, int* ptr, int* err) {
    for (int n = blockIdx.x * kBlock + threadIdx.x; n <= N; n += gridDim.x * kBlock) {
        const int64_t prev = n > 0 ? batch[n - 1] : -1;
        const int64_t cur = n < N ? batch[n] : B;
        // prev is checked too: thread n-1 flags a bad batch[n-1] as ITS cur, but this thread would still run the fill loop
        // from prev + 1 — below ptr[0] for a negative id, past ptr[B] for one >= B
        if (cur < prev || cur < 0 || cur > B || (n < N && cur >= B) || prev < -1 || prev >= B) { *err = 1; continue; }
        for (int64_t g = prev + 1; g <= cur; ++g) ptr[g] = n;
    }
}

}  // namespace glam

using namespace glam;

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t glam_csr_workspace_bytes(int64_t N, int64_t E) {
    (void)E;
    const size_t nb = (size_t)(N + 1 + kScanItems - 1) / kScanItems;
    return align_up((size_t)(N + 1) * 4, 256) + align_up((size_t)(N + 1) * 4, 256) + align_up((nb + 1) * 4, 256) + 256;
}

extern "C" int glam_csr_build(const int64_t* edge_index, int64_t N, int64_t E, int by, int32_t* rowptr, int32_t* nbr,
                              int32_t* eid, int32_t* err_flag, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(N >= 0 && E >= 0 && N < INT32_MAX && E < INT32_MAX, "glam_csr_build: N/E out of int32 range");
    GLAM_REQUIRE(by == 0 || by == 1, "glam_csr_build: by must be 0 (target) or 1 (source)");
    GLAM_REQUIRE(rowptr && err_flag && ws && (E == 0 || (edge_index && nbr && eid)), "glam_csr_build: null pointer");
    GLAM_REQUIRE(ws_bytes >= glam_csr_workspace_bytes(N, E), "glam_csr_build: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int n1 = (int)N + 1;
    const int nb = (n1 + kScanItems - 1) / kScanItems;
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255;
    int* count = reinterpret_cast<int*>(base);
    int* cursor = reinterpret_cast<int*>(base + align_up((size_t)n1 * 4, 256));
    int* bsums = reinterpret_cast<int*>(base + 2 * align_up((size_t)n1 * 4, 256));
    (void)hipMemsetAsync(count, 0, (size_t)n1 * 4, s);
    (void)hipMemsetAsync(cursor, 0, (size_t)n1 * 4, s);
    const int64_t* src_row = edge_index;        // edge_index[0] = source j
    const int64_t* dst_row = edge_index + E;    // edge_index[1] = target i
    const int64_t* key = by == 0 ? dst_row : src_row;
    const int64_t* val = by == 0 ? src_row : dst_row;
    if (E > 0) {
        hipLaunchKernelGGL(k_csr_count, dim3(grid_for(E, kBlock)), dim3(kBlock), 0, s, key, val, (int)E, (int)N, count, err_flag);
        GLAM_LAUNCH_CHECK("glam_csr_build(count)");
    }
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(kBlock), 0, s, count, n1, bsums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, s, bsums, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kBlock), 0, s, count, (int)N, bsums, rowptr);
    GLAM_LAUNCH_CHECK("glam_csr_build(scan)");
    if (E > 0) {
        hipLaunchKernelGGL(k_csr_fill, dim3(grid_for(E, kBlock)), dim3(kBlock), 0, s, key, val, (int)E, (int)N, rowptr, cursor, nbr, eid);
        hipLaunchKernelGGL(k_csr_sort_segments, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, s, rowptr, (int)N, nbr, eid);
        GLAM_LAUNCH_CHECK("glam_csr_build(fill)");
    }
    return GLAM_OK;
}

extern "C" int glam_batch_ptr(const int64_t* batch, int64_t N, int64_t B, int32_t* ptr, int32_t* err_flag, void* stream) {
    GLAM_REQUIRE(N >= 0 && B >= 0 && N < INT32_MAX && B < INT32_MAX, "glam_batch_ptr: N/B out of int32 range");
    GLAM_REQUIRE(ptr && err_flag && (N == 0 || batch), "glam_batch_ptr: null pointer");
    hipLaunchKernelGGL(k_batch_ptr, dim3(grid_for(N + 1, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, batch, (int)N, (int)B, ptr, err_flag);
    GLAM_LAUNCH_CHECK("glam_batch_ptr");
    return GLAM_OK;
}
