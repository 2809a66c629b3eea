// Linear layers with a handful of outputs: y[N, M] = x[N, K] @ w[M, K]^T + b, M <= 16 (the model's output head
// `lin_out1 = LinearBlock(e_dim, out_dim)`, reference src_1gp/model.py:47,61 — out_dim = 1 for the regression sets, 2 / 12 for
// the multi-task ones).  A GEMM library treats this as a matrix product with one 16- or 256-wide tile column: 33 us forward and
// 11 + 7 + 4 us backward for [1024, 1024] x [1024, 1] on this stack (profiles/r2d_kernel_stats_model_*.txt) — it is a row dot
// product, i.e. 4 MB of reads.
//   forward   one wave per row: lanes stride the row in float4, M accumulators, DPP / shuffle sum; w rows come from L1/L2.
//   backward  block = (64-column chunk, row split): 16 column lanes x 16 row lanes; every thread walks its rows once, writing
//             d_x[n, chunk] = sum_m dy[n, m] w[m, chunk] and accumulating d_w[m, chunk] += dy[n, m] x[n, chunk]; the 16 row lanes
//             are summed through LDS in lane order, the row splits by a second launch in split order (no atomics: bit-reproducible).
#include "common.h"
#include "rng.h"

namespace glam {

// ACT (round 6): x is the PRE-activation of the hidden layer in front of the head (mol_flat, src_1gp/model.py:43-45, :60) and the head
// applies that layer's training-mode RReLU(lo, hi) and its own Dropout(p) (layer.py:232-236) to every element it reads:
//   y = Linear(Dropout(RReLU(x))).  Neither the activated matrix nor its dropped twin is ever written, and the two elementwise launches
// (and their two backward launches) are gone; the words are the ones glam_bias_res_act_rng_fwd draws for the same elements at the same
// stream position, so y equals the three-launch pipeline's bit for bit.  The backward regenerates them from the recorded pair.
struct NarrowAct { long long* state; long long* eff; float lo, hi, p; };
__device__ __forceinline__ float4 narrow_act(float4 v, const uint4& w4, const NarrowAct& a, float4* factor) {
    float4 o, f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned wd = philox_word(w4, j);
        const float vj = f4get(v, j), sl = vj > 0.f ? 1.f : rrelu_slope_w(wd, a.lo, a.hi), sc = drop_scale_w(wd, a.p);
        const float act = vj > 0.f ? vj : vj * rrelu_slope_w(wd, a.lo, a.hi);
        (&o.x)[j] = act * sc;
        (&f.x)[j] = sl;
        (void)sc;
    }
    if (factor) *factor = f;
    return o;
}

constexpr int kNarrowMaxM = 16;
constexpr int kNarrowSplits = 16;      // row splits of the backward (partials [splits][M + 1][K])

template <int MT, bool ACT = false>
__global__ void __launch_bounds__(kBlock) k_linear_narrow_fwd(const float* x, const float* w, const float* b, int N, int K, int M, float* y,
                                                             NarrowAct ra) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Philox ph{};
    if constexpr (ACT) ph = rng_begin(ra.state, ra.eff);
    for (int n = blockIdx.x * (kBlock / 64) + wave; n < N; n += gridDim.x * (kBlock / 64)) {
        float acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = 0.f;
        const float* xr = x + (size_t)n * K;
        for (int k = 4 * lane; k < K; k += 256) {
            float4 xv = ld4(xr + k);
            if constexpr (ACT) xv = narrow_act(xv, philox4(ph, ((size_t)n * K + k) >> 2), ra, nullptr);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < M) {
                    const float4 wv = ld4(w + (size_t)m * K + k);
                    acc[m] = fmaf(xv.x, wv.x, acc[m]); acc[m] = fmaf(xv.y, wv.y, acc[m]);
                    acc[m] = fmaf(xv.z, wv.z, acc[m]); acc[m] = fmaf(xv.w, wv.w, acc[m]);
                }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = group_sum<64>(acc[m]);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < M) y[(size_t)n * M + m] = acc[m] + (b ? b[m] : 0.f);
        }
    }
    if constexpr (ACT) rng_end(ra.state, ph);
}

// partial[split][m][k] (m < M: d_w, m == M: column 0 holds d_b's partial, written by the blocks of column chunk 0)
template <int MT, bool ACT = false>
__global__ void __launch_bounds__(kBlock) k_linear_narrow_bwd(const float* x, const float* w, const float* dy, int N, int K, int M,
                                                             float* dx, float* partial, float* dw_direct, float* db_direct,
                                                             NarrowAct ra) {
    Philox ph{};
    if constexpr (ACT) ph = philox_init(ra.eff);
    __shared__ float4 s_red[16][17];            // [row lane][column lane] (+1: bank spread), one output row at a time
    __shared__ float s_db[16][MT];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int chunk = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int k = chunk * 64 + 4 * cl;
    const bool kok = k < K;
    const int rows_per = (N + nsplit - 1) / nsplit;
    const int r0 = split * rows_per, r1 = min(r0 + rows_per, N);
    float4 wv[MT], acc[MT];
    float dbs[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        wv[m] = (kok && m < M && dx) ? ld4(w + (size_t)m * K + k) : f4zero();
        acc[m] = f4zero();
        dbs[m] = 0.f;
    }
    for (int n = r0 + rl; n < r1; n += 16) {
        float4 xv = kok ? ld4(x + (size_t)n * K + k) : f4zero();
        float4 f0 = f4zero();
        uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (ACT) {      // what the forward multiplied with: Dropout(RReLU(x)), on the regenerated words
            w4 = philox4(ph, ((size_t)n * K + (kok ? k : 0)) >> 2);
            xv = narrow_act(xv, w4, ra, &f0);
        }
        float4 dxv = f4zero();
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (m < M) {
                const float g = dy[(size_t)n * M + m];
                fma4(acc[m], g, xv);
                fma4(dxv, g, wv[m]);
                dbs[m] += g;
            }
        if constexpr (ACT) {      // ... and back through both: the chain of glam_bias_res_act_rng_bwd, in its order of operations
#pragma unroll
            for (int j = 0; j < 4; ++j) (&dxv.x)[j] = fmaf(f4get(dxv, j), drop_scale_w(philox_word(w4, j), ra.p), 0.f) * f4get(f0, j);
        }
        if (dx && kok) st4(dx + (size_t)n * K + k, dxv);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m < M) {                            // uniform: the barriers inside are reached by the whole block
            s_red[rl][cl] = acc[m];
            if (cl == 0) s_db[rl][m] = dbs[m];
            __syncthreads();
            if (rl == 0 && kok) {               // the 16 row lanes in lane order
                float4 s = f4zero();
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float4 v = s_red[r][cl]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
                // one row split (small batches): the block's sums ARE the result — no partial set, no reduction launch
                if (nsplit == 1) st4(dw_direct + (size_t)m * K + k, s);
                else st4(partial + ((size_t)split * (M + 1) + m) * K + k, s);
            }
            __syncthreads();
        }
    }
    if (chunk == 0 && threadIdx.x < M) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += s_db[r][threadIdx.x];
        if (nsplit == 1) { if (db_direct) db_direct[threadIdx.x] = s; }
        else partial[((size_t)split * (M + 1) + M) * K + threadIdx.x] = s;
    }
}

__global__ void __launch_bounds__(kBlock) k_linear_narrow_reduce(const float* partial, int nsplit, int K, int M, float* dw, float* db) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx < M * K) {
        const int m = idx / K, k = idx - m * K;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += partial[((size_t)sp * (M + 1) + m) * K + k];
        dw[idx] = s;
    } else if (db && idx < M * K + M) {
        const int m = idx - M * K;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += partial[((size_t)sp * (M + 1) + M) * K + m];
        db[m] = s;
    }
}

// Column sums out[d] = sum_n x[n, d] (the bias gradient of a library-GEMM linear: torch's generic reduction takes 14 us for
// [1024, 1024]).  Block = (64-column chunk, row split): 16 column lanes x 16 row lanes, LDS sum in lane order; the row splits (large
// N only) are summed in split order by a second launch.
__global__ void __launch_bounds__(kBlock) k_colsum(const float* x, int N, int D, int ld, float* out, int to_partial) {
    __shared__ float4 s_red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int k = blockIdx.x * 64 + 4 * cl, split = blockIdx.y, nsplit = gridDim.y;
    const bool kok = k < D;
    const int rows_per = (N + nsplit - 1) / nsplit;
    const int r0 = split * rows_per, r1 = min(r0 + rows_per, N);
    float4 a0 = f4zero(), a1 = f4zero();
    int n = r0 + rl;
    if (kok) {
        // eight row loads in flight per thread (two were not enough: 16 blocks x 64 dependent iterations took 12 us for [1024, 1024])
        for (; n + 7 * 16 < r1; n += 8 * 16) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld4(x + (size_t)(n + 16 * u) * ld + k);
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w;
                a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
            }
        }
        for (; n < r1; n += 16) { const float4 v0 = ld4(x + (size_t)n * ld + k); a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; }
    }
    s_red[rl][cl] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (rl == 0 && kok) {
        float4 s = f4zero();
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float4 v = s_red[r][cl]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        st4(out + (to_partial ? (size_t)split * D : 0) + k, s);
    }
}

__global__ void __launch_bounds__(kBlock) k_colsum_reduce(const float* partial, int nsplit, int D, float* out) {
    const int d = blockIdx.x * kBlock + threadIdx.x;
    if (d >= D) return;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += partial[(size_t)sp * D + d];
    out[d] = s;
}

}  // namespace glam

using namespace glam;

extern "C" size_t glam_colsum_workspace_bytes(int D) { return (size_t)kNarrowSplits * (size_t)D * sizeof(float); }

extern "C" int glam_colsum(const float* x, int64_t N, int D, int ld, float* out, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX && D > 0 && (D & 3) == 0 && ld >= D && (ld & 3) == 0, "glam_colsum: N / D / ld out of range (D, ld multiples of 4)");
    GLAM_REQUIRE(out && aligned16(out), "glam_colsum: null / misaligned output");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) { (void)hipMemsetAsync(out, 0, (size_t)D * sizeof(float), s); return GLAM_OK; }
    GLAM_REQUIRE(x && aligned16(x), "glam_colsum: null / misaligned input");
    const int nsplit = (int)(N <= 2048 ? 1 : (N + 2047) / 2048 > kNarrowSplits ? kNarrowSplits : (N + 2047) / 2048);   // (keep in step with ops._LinearLib)
    if (nsplit > 1) GLAM_REQUIRE(ws && aligned16(ws) && ws_bytes >= glam_colsum_workspace_bytes(D), "glam_colsum: workspace too small");
    float* dst = nsplit > 1 ? static_cast<float*>(ws) : out;
    hipLaunchKernelGGL(k_colsum, dim3((D + 63) / 64, nsplit), dim3(kBlock), 0, s, x, (int)N, D, ld, dst, nsplit > 1 ? 1 : 0);
    GLAM_LAUNCH_CHECK("glam_colsum");
    if (nsplit > 1) {
        hipLaunchKernelGGL(k_colsum_reduce, dim3((D + kBlock - 1) / kBlock), dim3(kBlock), 0, s, dst, nsplit, D, out);
        GLAM_LAUNCH_CHECK("glam_colsum(reduce)");
    }
    return GLAM_OK;
}


static int narrow_dims(const char* fn, int64_t N, int K, int M) {
    if (N < 0 || N > INT32_MAX) return fail(GLAM_E_INVALID, "%s: N out of range", fn);
    if (K < 16 || (K & 3)) return fail(GLAM_E_UNSUPPORTED, "%s: K=%d must be a multiple of 4, at least 16", fn, K);
    if (M < 1 || M > kNarrowMaxM) return fail(GLAM_E_UNSUPPORTED, "%s: M=%d not in 1..%d", fn, M, kNarrowMaxM);
    return GLAM_OK;
}

extern "C" int glam_linear_narrow_supported(int K, int M) { return K >= 16 && (K & 3) == 0 && M >= 1 && M <= kNarrowMaxM; }

extern "C" int glam_linear_narrow_fwd(const float* x, const float* w, const float* b, int64_t N, int K, int M, float* y, void* stream) {
    if (int rc = narrow_dims("glam_linear_narrow_fwd", N, K, M)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(x && w && y, "glam_linear_narrow_fwd: null pointer");
    GLAM_REQUIRE(aligned16(x) && aligned16(w), "glam_linear_narrow_fwd: x and w must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(N, kBlock / 64);
    if (M == 1) hipLaunchKernelGGL((k_linear_narrow_fwd<1>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, NarrowAct{});
    else if (M <= 2) hipLaunchKernelGGL((k_linear_narrow_fwd<2>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, NarrowAct{});
    else if (M <= 4) hipLaunchKernelGGL((k_linear_narrow_fwd<4>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, NarrowAct{});
    else if (M <= 8) hipLaunchKernelGGL((k_linear_narrow_fwd<8>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, NarrowAct{});
    else hipLaunchKernelGGL((k_linear_narrow_fwd<16>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, NarrowAct{});
    GLAM_LAUNCH_CHECK("glam_linear_narrow_fwd");
    return GLAM_OK;
}

// y = Linear(Dropout(p)(RReLU(lower, upper)(x))) in training mode: the output head reading the hidden layer's PRE-activation (see
// NarrowAct above).  rng_eff receives the (seed, offset) pair glam_linear_narrow_act_bwd regenerates the words from.
extern "C" int glam_linear_narrow_act_fwd(const float* x, const float* w, const float* b, int64_t N, int K, int M, float rr_lower,
                                          float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* y, void* stream) {
    if (int rc = narrow_dims("glam_linear_narrow_act_fwd", N, K, M)) return rc;
    GLAM_REQUIRE(rr_lower > 0.f && rr_lower <= rr_upper && drop_p >= 0.f && drop_p < 1.f, "glam_linear_narrow_act_fwd: needs 0 < lower <= upper and "
                 "0 <= p < 1 (got %g, %g, %g)", rr_lower, rr_upper, drop_p);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(x && w && y && rng_state && rng_eff, "glam_linear_narrow_act_fwd: null pointer");
    GLAM_REQUIRE(aligned16(x) && aligned16(w), "glam_linear_narrow_act_fwd: x and w must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(N, kBlock / 64);
    const NarrowAct ra{reinterpret_cast<long long*>(rng_state), reinterpret_cast<long long*>(rng_eff), rr_lower, rr_upper, drop_p};
    GLAM_PROF_LABEL("k_linear_narrow_fwd<rrelu+dropout>");
    if (M == 1) hipLaunchKernelGGL((k_linear_narrow_fwd<1, true>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, ra);
    else if (M <= 2) hipLaunchKernelGGL((k_linear_narrow_fwd<2, true>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, ra);
    else if (M <= 4) hipLaunchKernelGGL((k_linear_narrow_fwd<4, true>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, ra);
    else if (M <= 8) hipLaunchKernelGGL((k_linear_narrow_fwd<8, true>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, ra);
    else hipLaunchKernelGGL((k_linear_narrow_fwd<16, true>), dim3(grid), dim3(kBlock), 0, s, x, w, b, (int)N, K, M, y, ra);
    GLAM_LAUNCH_CHECK("glam_linear_narrow_act_fwd");
    return GLAM_OK;
}

extern "C" size_t glam_linear_narrow_bwd_workspace_bytes(int K, int M) {
    return (size_t)kNarrowSplits * (size_t)(M + 1) * (size_t)K * sizeof(float);
}

static int narrow_bwd_impl(const char* fn, const float* x, const float* w, const float* dy, int64_t N, int K, int M, float* dx, float* dw,
                           float* db, void* ws, size_t ws_bytes, void* stream, const NarrowAct* act);
extern "C" int glam_linear_narrow_bwd(const float* x, const float* w, const float* dy, int64_t N, int K, int M, float* dx, float* dw,
                                      float* db, void* ws, size_t ws_bytes, void* stream) {
    return narrow_bwd_impl("glam_linear_narrow_bwd", x, w, dy, N, K, M, dx, dw, db, ws, ws_bytes, stream, nullptr);
}
// backward of glam_linear_narrow_act_fwd: x is the same pre-activation; dx = the gradient of THAT (through the Dropout and the RReLU, on
// the words regenerated from rng_eff), dw / db as the head's (its input Dropout(RReLU(x)) is recomputed, never read)
extern "C" int glam_linear_narrow_act_bwd(const float* x, const float* w, const float* dy, int64_t N, int K, int M, float rr_lower,
                                          float rr_upper, float drop_p, const int64_t* rng_eff, float* dx, float* dw, float* db, void* ws,
                                          size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(rr_lower > 0.f && rr_lower <= rr_upper && drop_p >= 0.f && drop_p < 1.f && (rng_eff || N == 0), "glam_linear_narrow_act_bwd: needs "
                 "0 < lower <= upper, 0 <= p < 1 and the recorded stream position");
    const NarrowAct ra{nullptr, const_cast<long long*>(reinterpret_cast<const long long*>(rng_eff)), rr_lower, rr_upper, drop_p};
    return narrow_bwd_impl("glam_linear_narrow_act_bwd", x, w, dy, N, K, M, dx, dw, db, ws, ws_bytes, stream, &ra);
}
static int narrow_bwd_impl(const char* fn0, const float* x, const float* w, const float* dy, int64_t N, int K, int M, float* dx, float* dw,
                           float* db, void* ws, size_t ws_bytes, void* stream, const NarrowAct* act) {
    (void)fn0;
    if (int rc = narrow_dims("glam_linear_narrow_bwd", N, K, M)) return rc;
    GLAM_REQUIRE(dw && ws && ws_bytes >= glam_linear_narrow_bwd_workspace_bytes(K, M), "glam_linear_narrow_bwd: null output / workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        (void)hipMemsetAsync(dw, 0, (size_t)M * K * sizeof(float), s);
        if (db) (void)hipMemsetAsync(db, 0, (size_t)M * sizeof(float), s);
        return GLAM_OK;
    }
    GLAM_REQUIRE(x && w && dy, "glam_linear_narrow_bwd: null pointer");
    GLAM_REQUIRE(aligned16(x) && aligned16(w) && aligned16(dx) && aligned16(ws), "glam_linear_narrow_bwd: pointers must be 16-byte aligned");
    float* partial = static_cast<float*>(ws);
    // up to 256 rows (the reference's batch of 32): one row split, sums written straight to dw / db — ONE launch, 2.3 us at N = 32;
    // beyond: row splits + the fixed-order reduction (one split walks its rows serially: 15.8 us at N = 1 024 against 9.7 for the pair)
    const int nsplit = N <= 256 ? 1 : (int)(N < kNarrowSplits * 16 ? (N + 15) / 16 : kNarrowSplits);
    const dim3 grid((K + 63) / 64, nsplit);
    if (act) {
        const NarrowAct ra = *act;
        GLAM_PROF_LABEL("k_linear_narrow_bwd<rrelu+dropout>");
        if (M == 1) hipLaunchKernelGGL((k_linear_narrow_bwd<1, true>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, ra);
        else if (M <= 2) hipLaunchKernelGGL((k_linear_narrow_bwd<2, true>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, ra);
        else if (M <= 4) hipLaunchKernelGGL((k_linear_narrow_bwd<4, true>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, ra);
        else if (M <= 8) hipLaunchKernelGGL((k_linear_narrow_bwd<8, true>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, ra);
        else hipLaunchKernelGGL((k_linear_narrow_bwd<16, true>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, ra);
    } else
    if (M == 1) hipLaunchKernelGGL((k_linear_narrow_bwd<1>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, NarrowAct{});
    else if (M <= 2) hipLaunchKernelGGL((k_linear_narrow_bwd<2>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, NarrowAct{});
    else if (M <= 4) hipLaunchKernelGGL((k_linear_narrow_bwd<4>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, NarrowAct{});
    else if (M <= 8) hipLaunchKernelGGL((k_linear_narrow_bwd<8>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, NarrowAct{});
    else hipLaunchKernelGGL((k_linear_narrow_bwd<16>), grid, dim3(kBlock), 0, s, x, w, dy, (int)N, K, M, dx, partial, dw, db, NarrowAct{});
    GLAM_LAUNCH_CHECK("glam_linear_narrow_bwd");
    if (nsplit == 1) return GLAM_OK;
    hipLaunchKernelGGL(k_linear_narrow_reduce, dim3((M * K + M + kBlock - 1) / kBlock), dim3(kBlock), 0, s, partial, nsplit, K, M, dw, db);
    GLAM_LAUNCH_CHECK("glam_linear_narrow_bwd(reduce)");
    return GLAM_OK;
}
