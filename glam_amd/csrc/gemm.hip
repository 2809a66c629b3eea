// Dense ends of the TripletMessage layer on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact
// fp32, bit-equal to an fmaf chain, so the 1e-5 parity bar is untouched).
//
// The layer's linears (reference: src_1gp/layer.py:37 `x @ weight_node`, :58-60 `aggr @ weight_scale
// + bias`, and what autograd derives for them) are all "tall-skinny": N rows (atoms, 2e4..3e5) against
// weights of at most 192 x 64 / 64 x 192.  Library GEMMs pick tiles for square problems and ran 12-73 us
// per call here (profiles/r1a_*); two purpose-built kernels replace them:
//
//   k_ts_gemm   C[N, M] = [A1 | A2][N, K] @ W[K, M] (+ bias); (K <= 64, M <= 192), (K <= 192, M <= 64) and two wide variants
//               (ts_variant).
//               W (zero padded, pre-permuted "image") lives in LDS for the whole 8-wave block; a wave owns
//               16 rows and ALL M columns (MT accumulator tiles, <= 48 AGPRs); its whole A fragment is
//               loaded up front (one float4 per 16 k-values: lane (r, kq) supplies k = 16g + 4kq + j at
//               MFMA step j) and the B operands of a 16-k group are one ds_read_b128 per tile.  Two waves
//               per SIMD overlap one wave's loads / stores with the other's MFMAs.
//   k_wgrad     G[I, J] = [P1 | P2 | 1]^T[I, N] @ Q[N, J], I <= 192, J <= 64, reduction over rows.
//               Block = (64-column slab of P, row split); its 8 waves take disjoint row ranges, each holding a
//               64 x 64 accumulator slab, and are summed through LDS; block partials (16 KB) are combined in a
//               fixed order by k_final_reduce (deterministic, no atomics), which also folds the aggregate
//               kernel's d_W_edge partials.
#include "dense.h"
#include "bf16x3.h"

#include <stdlib.h>
#include <string.h>

namespace glam {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kTsBlock = 512;   // 8 waves: 2 per SIMD

// Image layout: img[k / 4][p][k % 4] (Kp/4 x MP x 4 floats, zero padded): the B operands a lane needs for the
// 4 MFMA steps of one 16-k group are ONE 16-byte LDS read (ds_read_b128, conflict free: 16 lanes x 16 B
// cover all 64 banks and MP*4 words == 0 mod 64 keeps the two k-halves of a read group on disjoint banks).
// Column order: position p = cg*64 + t*16 + c holds logical column m = cg*64 + 4c + t (ts_col_of_pos), so
// lane c of a wave owns 4 CONSECUTIVE output columns across the 4 tiles of a 64-column group and the
// epilogue stores float4 (256 contiguous bytes per 16 lanes).
// logical W[k][m] = transW ? W[m*ldw + k] : W[k*ldw + m]
__global__ void __launch_bounds__(kBlock) k_ts_make_image(const float* W, int ldw, int transW, int K, int M, int MT,
                                                         float* img) {
    const int MP = MT * 16, Kp = (K + 15) & ~15;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < Kp * MP; idx += gridDim.x * kBlock) {
        const int j = idx & 3, p = (idx >> 2) % MP, k = (idx >> 2) / MP * 4 + j;
        const int m = ts_col_of_pos(p);
        float v = 0.f;
        if (k < K && m < M) v = transW ? W[(size_t)m * ldw + k] : W[(size_t)k * ldw + m];
        img[idx] = v;
    }
}

// (ImageJob / ImageJobs / make_images_block: dense.h — the same block body serves glam_prestage's launch in layer.hip)
__global__ void __launch_bounds__(kBlock) k_ts_make_images(ImageJobs js) { make_images_block(js, (int)blockIdx.x); }

// Work item = (16-row tile, column split): TPI of the MT column tiles.  A 16 x 192 x 64 row tile is 192 MFMAs
// (6.1k cycles on one SIMD): whole row tiles leave some SIMDs with two and others with none at N ~ 2e4, so the
// tiles are cut into MT/TPI column items that are dealt round-robin to ALL waves of the grid (consecutive items,
// i.e. the splits of one row tile, land on neighbouring waves of one CU and share the A rows through L1).
#ifdef GLAM_TS_PROF   // developer aid (tools/ts_prof.py)
__device__ long long g_ts_prof[1024 * 8];
#define TS_STAMP(k) do { if (tid == 0 && blockIdx.x < 1024) g_ts_prof[blockIdx.x * 8 + (k)] = clock64(); } while (0)
#define TS_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define TS_STAMP(k) do { } while (0)
#define TS_DRAIN() do { } while (0)
#endif

// Two independent products of the same kernel variant may share one launch (the GRU's two gate linears, and the two input-gradient
// products of its backward): blocks [0, first_b) work on job a, the rest on job b, each with its own weight image.  Besides the
// dispatch it saves, the 48 KB-image variants then have two blocks (16 waves) per CU: one product's loads and stores run under
// the other's MFMAs.
struct TsArgs2 { TsArgs a, b; int first_b; };

// RB ("register B", the 192-column variant): a wave is BOUND to one column split for the whole launch and keeps that 64 x 64 slice of
// the weight image in 64 registers; it walks the row tiles.  No LDS image, no staging, no block barrier (the staging was 3.8 k of the
// 12.5 k cycles of the launch at B = 1 024) and no B-operand LDS reads in the loop; the grid is a multiple of the column-split count.
// RBLK: 0 = the LDS-image form (512 threads); 512 / 768 = the register-B form with 8 / 12 waves per block (two / three per SIMD: three
// pay once the launch streams — 131 vs 141 us at B = 16 384 — and cost at B = 1 024, 14.5 vs 11.6 us).  The register-B form is the PLAIN
// product only (one A source, no folded CELU, no gradient epilogue, fp32 out: 118 registers instead of 167 with those paths compiled in —
// 111 -> 103 us at B = 16 384, 11.3 -> 10.7 at B = 1 024); a 16-wave block (four per SIMD) measured 112.7 us.
template <int MT, int GMAX, int TPI, int RBLK = 0>
__global__ void __launch_bounds__(RBLK ? RBLK : kTsBlock) k_ts_gemm(TsArgs2 two) {
    constexpr bool RB = RBLK != 0;
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    const bool second = (int)blockIdx.x >= two.first_b;
    const TsArgs a = second ? two.b : two.a;      // by value: every field a scalar select (a reference made the compiler re-read the
    //                                               argument block and cost the single-product launch 1.5 us)
    const int bid = second ? (int)blockIdx.x - two.first_b : (int)blockIdx.x;          // block index / count inside the job
    const int nblk = second ? (int)gridDim.x - two.first_b : two.first_b;
    constexpr int MP = MT * 16;        // padded column count
    constexpr int CS = MT / TPI;       // column splits per row tile
    constexpr int kMaxStage = 16 * GMAX * MP / 4 / kTsBlock;   // float4 per thread for the largest image
    static_assert(kMaxStage * kTsBlock * 4 == 16 * GMAX * MP, "image must tile the block");
    static_assert(TPI == 4 || TPI == 2, "a column item is one or half a 64-column group");
    const int tid = threadIdx.x;
    const int K = a.K1 + a.K2, G = (K + 15) >> 4, M = a.M1 + a.M2;
    const int wave = tid >> 6, lane = tid & 63, c = lane & 15, kq = lane >> 4;
    const int ntiles = (a.N + 15) >> 4;
    const int nitems = ntiles * CS;
    constexpr int WPB = (RB ? RBLK : kTsBlock) / 64;
    const int stride = nblk * WPB;

    // A fragment of a tile: one float4 per 16-k group, every load in flight at once
    auto load_afrag = [&](int item, float4 (&af)[GMAX]) {
        const int row = (item / CS) * 16 + c;
        const bool rok = item < nitems && row < a.N;
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            const int k0 = 16 * g + 4 * kq;
            af[g] = f4zero();
            if (rok) {
                if (k0 < a.K1) af[g] = ld4(a.A1 + (size_t)row * a.lda1 + k0);
                else if (!RB && k0 < K) af[g] = ld4(a.A2 + (size_t)row * a.lda2 + (k0 - a.K1));
            }
        }
        if (!RB && a.a_celu) {
#pragma unroll
            for (int g = 0; g < GMAX; ++g) af[g] = celu4(af[g]);      // celu(0) = 0: the zero padding stays zero
        }
    };
    // Fewer items than wave slots (small batches; the 64-column variant at any batch): deal them block-minor, so that they
    // spread over all blocks / CUs first and a SIMD runs one MFMA stream instead of two back to back.
    const bool spread = nitems < nblk * WPB;
    int item = spread ? bid + wave * nblk : bid * WPB + wave;
    int stride_items = stride;
    float4 breg[RB ? GMAX : 1][RB ? TPI : 1];
    if constexpr (RB) {
        // wave gw of the grid: column split gw % CS, row tiles gw / CS, gw / CS + GW / CS, ... (the host makes GW a multiple of CS)
        const int gw = bid * WPB + wave;
        const int cs = gw % CS;
        item = (gw / CS) * CS + cs;
        stride_items = stride;          // GW waves: GW / CS row tiles per step = GW items
#pragma unroll
        for (int g = 0; g < GMAX; ++g)
#pragma unroll
            for (int t = 0; t < TPI; ++t)
                breg[g][t] = g < G ? ld4(a.Wimg + ((size_t)(4 * g + kq) * MP + (cs * TPI + t) * 16 + c) * 4) : f4zero();
    }
    float4 af[GMAX];
    TS_STAMP(0);
    load_afrag(item, af);              // flies while the weight image is staged

    // ---- stage the W image into LDS: all loads in flight first, then the LDS stores ----
    if constexpr (!RB) {
        const int n4 = G * 4 * MP;     // float4 count
        float4 buf[kMaxStage];
#pragma unroll
        for (int i = 0; i < kMaxStage; ++i) {
            const int idx = tid + i * kTsBlock;
            if (idx < n4) buf[i] = ld4(a.Wimg + 4 * idx);
        }
#pragma unroll
        for (int i = 0; i < kMaxStage; ++i) {
            const int idx = tid + i * kTsBlock;
            if (idx < n4) st4(s_w + 4 * idx, buf[i]);
        }
        __syncthreads();
    }
    TS_STAMP(1);
    TS_DRAIN();
    TS_STAMP(2);

    const float* wlane = s_w + (kq * MP + c) * 4;
    int pass = 0;
    for (; item < nitems; item += stride_items, ++pass) {
        const int tile = item / CS, cs = item - tile * CS;
        const int t0 = cs * TPI;       // first column tile of the item
        v4f acc[TPI];
#pragma unroll
        for (int t = 0; t < TPI; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            if (g < G) {
                float4 bv[TPI];
#pragma unroll
                for (int t = 0; t < TPI; ++t) {
                    if constexpr (RB) bv[t] = breg[g][t];
                    else bv[t] = ld4(wlane + g * 16 * MP + (t0 + t) * 64);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float aj = f4get(af[g], j);
#pragma unroll
                    for (int t = 0; t < TPI; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, f4get(bv[t], j), acc[t], 0, 0, 0);
                }
            }
        }
        if (pass == 0) TS_STAMP(3);
        load_afrag(item + stride_items, af);   // next item's A fragment flies under the epilogue
        // C layout: tile column = lane & 15 (-> logical columns cg*64 + 4c + t), row = (lane >> 4) * 4 + i
        const int cg = t0 >> 2, tq = t0 & 3;       // TPI == 4: tq = 0; TPI == 2: tq in {0, 2}
        const int m0 = cg * 64 + 4 * c + tq;
        if (m0 < M) {
            if constexpr (TPI == 4) {
                float4 b = f4zero();
                if (a.bias && m0 < a.M1) b = ld4(a.bias + m0);
                const bool use_cg = a.cgrad_src && m0 < a.M1, use_ad = a.addend && m0 < a.M1;
                if (!RB && (a.cgrad_src || a.addend)) {     // wave-uniform: the plain product keeps its lean epilogue (and its registers)
                    // every load of the epilogue before its first store (a load issued between stores waits for vmcnt(0), i.e. for
                    // the previous row's store to land)
                    float4 xs[4], ad[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = min(tile * 16 + kq * 4 + i, a.N - 1);
                        xs[i] = use_cg ? ld4(a.cgrad_src + (size_t)rr * a.ld_cgrad + m0) : f4zero();
                        ad[i] = use_ad ? ld4(a.addend + (size_t)rr * a.ld_add + m0) : f4zero();
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = tile * 16 + kq * 4 + i;
                        if (rr >= a.N) continue;
                        float4 v = make_float4(acc[0][i] + b.x, acc[1][i] + b.y, acc[2][i] + b.z, acc[3][i] + b.w);
                        if (use_cg) {
                            v.x *= celu1_grad(xs[i].x); v.y *= celu1_grad(xs[i].y); v.z *= celu1_grad(xs[i].z); v.w *= celu1_grad(xs[i].w);
                        }
                        if (use_ad) { v.x += ad[i].x; v.y += ad[i].y; v.z += ad[i].z; v.w += ad[i].w; }
                        if (m0 < a.M1) st4(a.out1 + (size_t)rr * a.ldo1 + m0, v);
                        else st4(a.out2 + (size_t)rr * a.ldo2 + (m0 - a.M1), v);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = tile * 16 + kq * 4 + i;
                        if (rr >= a.N) continue;
                        const float4 v = make_float4(acc[0][i] + b.x, acc[1][i] + b.y, acc[2][i] + b.z, acc[3][i] + b.w);
                        if (m0 < a.M1) st4(a.out1 + (size_t)rr * a.ldo1 + m0, v);
                        else st4(a.out2 + (size_t)rr * a.ldo2 + (m0 - a.M1), v);
                    }
                }
            } else {
                float2 b = make_float2(0.f, 0.f);
                if (a.bias && m0 < a.M1) b = *reinterpret_cast<const float2*>(a.bias + m0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = tile * 16 + kq * 4 + i;
                    if (rr >= a.N) continue;
                    float2 v = make_float2(acc[0][i] + b.x, acc[1][i] + b.y);
                    if (a.cgrad_src && m0 < a.M1) {
                        const float2 xs = *reinterpret_cast<const float2*>(a.cgrad_src + (size_t)rr * a.ld_cgrad + m0);
                        v.x *= celu1_grad(xs.x); v.y *= celu1_grad(xs.y);
                    }
                    if (a.addend && m0 < a.M1) {
                        const float2 ad = *reinterpret_cast<const float2*>(a.addend + (size_t)rr * a.ld_add + m0);
                        v.x += ad.x; v.y += ad.y;
                    }
                    if (m0 < a.M1) *reinterpret_cast<float2*>(a.out1 + (size_t)rr * a.ldo1 + m0) = v;
                    else *reinterpret_cast<float2*>(a.out2 + (size_t)rr * a.ldo2 + (m0 - a.M1)) = v;
                }
            }
        }
    }
    TS_STAMP(4);
    TS_DRAIN();
    TS_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// k_ts_gemm_x3: the register-B form of the wide products (K <= 64 x 192 columns: the node GEMM x @ [W_node | Wa_i | Wa_j]; K <= 96 x
// 320 columns: the same product, the GRU gate linears and the update GEMM's input gradient at hid_dim_alpha = 6) on the bf16 matrix
// cores in 3 x bf16 form (bf16x3.h: fp32 accuracy; 48 v_mfma_f32_16x16x32_bf16 per 16 x 64 item instead of 60 fp32 MFMAs at 1.65 x the
// cycles each, and off the fp32 datapath).  A wave is bound to one 64-column split and keeps that slice of W — split once, in the
// prologue — in 96 registers (two 32-k steps x four column tiles x three terms); it walks the row tiles, splitting each A fragment
// (16 floats per lane) on the fly.  Same image, same column permutation and same epilogue as k_ts_gemm<12, 4, 4, RBLK>.
// Lane (r = lane & 15, kb = lane >> 4): A row r / W column, k = 32 s + 8 kb .. + 7 of step s (the two operands share the k set of a lane,
// which is all the contraction needs).
// ------------------------------------------------------------------------------------------------
template <int RBLK, int KS = 2, int CS = 3>      // KS 32-k steps (K <= 32 KS), CS 64-column splits (image rows of 64 CS positions)
__global__ void __launch_bounds__(RBLK) k_ts_gemm_x3(TsArgs a, int nblk) {
    constexpr int MP = 64 * CS, WPB = RBLK / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, kb = lane >> 4;
    const int K = a.K1, Kp = (K + 15) & ~15, M = a.M1 + a.M2;
    const int ntiles = (a.N + 15) >> 4;
    const int gw = (int)blockIdx.x * WPB + wave;
    const int cs = gw % CS;
    // W slice of this wave: column tile t holds the logical columns cs * 64 + 4 c + t (image position cs * 64 + 16 t + c)
    Bf16x3 wreg[KS][4];
    {
        WRaw8 raw[KS][4];                                     // rows 32 s + 8 kb .. + 7 of the image (zero beyond K; the image ends at Kp)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) raw[s][t] = w_load8(a.Wimg, MP, cs * 64 + 16 * t + c, 32 * s + 8 * kb, Kp);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) wreg[s][t] = w_split8(raw[s][t], 32 * s + 8 * kb, Kp);
    }
    auto load_a = [&](int tile, float4 (&af)[KS][2]) {
        const int row = tile * 16 + c;
        const bool rok = tile < ntiles && row < a.N;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k0 = 32 * s + 8 * kb + 4 * u;
                af[s][u] = (rok && k0 < K) ? ld4(a.A1 + (size_t)row * a.lda1 + k0) : f4zero();
            }
    };
    const int tstride = nblk * WPB / CS;                       // row tiles per step of this wave (the host makes the wave count a multiple of CS)
    int tile = gw / CS;
    // the wave's four output columns never change: the bias is read ONCE, here.  Read inside the loop (under its condition) the compiler
    // re-waited for it — vmcnt(0) — in each of the four row-store blocks, i.e. for the previous row's STORE: three store round trips
    // per item in series (the launch ran at 3.7 us per item and wave)
    const int m0 = cs * 64 + 4 * c;
    float4 bias4 = f4zero();
    if (a.bias && m0 < a.M1) bias4 = ld4(a.bias + m0);
    float4 af[KS][2];
    load_a(tile, af);
    for (; tile < ntiles; tile += tstride) {
        Bf16x3 as[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) as[s] = split8(af[s][0], af[s][1]);
        load_a(tile + tstride, af);                            // next tile's rows fly under this tile's MFMAs and stores
        v4f_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (v4f_t){0.f, 0.f, 0.f, 0.f};
        // four independent accumulator chains (column tiles); within a chain: small partial products of both k steps, then the middle
        // ones, then hi x hi
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_x3_small(as[s], wreg[s][t], acc[t]);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_x3_mid(as[s], wreg[s][t], acc[t]);
        // (the large products in a chain of their own, added to the small + middle sums once at the end: the matrix instruction aligns
        //  a small accumulator to the large products by truncation — towards -inf, a bias — where the final fp32 add rounds to nearest)
        v4f_t accb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) accb[t] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) accb[t] = mfma_x3_big(as[s], wreg[s][t], accb[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += accb[t];
        // C layout: tile column = lane & 15 (-> logical columns cs * 64 + 4 c + t), row = (lane >> 4) * 4 + i
        if (m0 < M) {
            const float4 b = bias4;
            if constexpr (RBLK == 768) {      // (the 12-wave form only runs over >= 131 072 rows: plain stores)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = tile * 16 + kb * 4 + i;
                    if (rr >= a.N) continue;
                    const float4 v = make_float4(acc[0][i] + b.x, acc[1][i] + b.y, acc[2][i] + b.z, acc[3][i] + b.w);
                    if (m0 < a.M1) st4(a.out1 + (size_t)rr * a.ldo1 + m0, v);
                    else st4(a.out2 + (size_t)rr * a.ldo2 + (m0 - a.M1), v);
                }
            } else {
                // written through the L2 when the launch is small (st4o_sel): wave-uniform base of the tile's rows + the lane's offset inside
                const size_t trow = (size_t)__builtin_amdgcn_readfirstlane(tile) * 16;
                float* o1 = a.out1 + trow * a.ldo1;
                float* o2 = a.out2 + trow * a.ldo2;
                const bool first = m0 < a.M1, wt = a.N <= kWtMaxRows;
                const unsigned lane_off = first ? (unsigned)(kb * 4 * a.ldo1 + m0) * 4u : (unsigned)(kb * 4 * a.ldo2 + (m0 - a.M1)) * 4u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (tile * 16 + kb * 4 + i >= a.N) continue;
                    const float4 v = make_float4(acc[0][i] + b.x, acc[1][i] + b.y, acc[2][i] + b.z, acc[3][i] + b.w);
                    if (first) st4o_sel(wt, o1, lane_off, v, (unsigned)(i * a.ldo1) * 4u);
                    else st4o_sel(wt, o2, lane_off, v, (unsigned)(i * a.ldo2) * 4u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // one split's 96 operand registers at a time (hoisted, the three sets spill)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_ts_gemm_x3_sw: the same product, W split once per block (the default at every size).  At B = 1 024 a wave of k_ts_gemm_x3 sees two
// items.  There 2 040 waves each fetch and split their own 15 KB slice of W, and every row tile is fetched by three waves (one per column
// split): 181 KB come into a CU for the 19 KB of x it works on, and at the ~11 B/clk a CU takes in that IS the launch (7.8 of its 8.8-9.4 us).
// Here the block splits W ONCE — wave w loads and splits three of the 24 (column split, k step, column tile) fragments — into LDS in
// operand layout (72 KB), and a wave takes a row tile through all three column splits, its operands of W out of LDS (24 conflict-free
// 16-byte reads per split): 65 KB per CU.  No handshakes (one barrier behind the prologue), unlike the producer / consumer form of
// tall_x3.hip that was tried for this shape (profiles/NEGATIVES.md).  Same arithmetic in the same order as k_ts_gemm_x3: bit-identical.
// Beyond the LLC it wins for another reason: a row of the output leaves ONE wave as 752 contiguous bytes within a few hundred cycles,
// where the three column-split waves of k_ts_gemm_x3 wrote 256-byte pieces of it at unrelated times (N = 326 400: 114 -> 72 us alone,
// 79-82 -> 62 us inside the B = 16 384 step = 0.66 of the HBM roofline).
// ------------------------------------------------------------------------------------------------
template <int KS, int CS>
__global__ void __launch_bounds__(512) k_ts_gemm_x3_sw(TsArgs a) {
    constexpr int MP = 64 * CS, NF = CS * KS * 4, FPW = NF / 8;
    static_assert(NF % 8 == 0, "fragments per wave");
    extern __shared__ __attribute__((aligned(16))) char s_wf[];      // [fragment (cs, s, t)][term][lane] x 16 bytes
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, kb = lane >> 4;
    const int K = a.K1, Kp = (K + 15) & ~15, M = a.M1 + a.M2;
    const int ntiles = (a.N + 15) >> 4;
    auto load_a = [&](int tile, float4 (&af)[KS][2]) {
        const int row = tile * 16 + c;
        const bool rok = tile < ntiles && row < a.N;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k0 = 32 * s + 8 * kb + 4 * u;
                af[s][u] = (rok && k0 < K) ? ld4(a.A1 + (size_t)row * a.lda1 + k0) : f4zero();
            }
    };
    // Block b works on row tiles b, b + blocks, ... (nt_b of them).  A wave's work is a list of units (tile, column splits cs0 .. cs1 - 1):
    //   - whole tiles (all splits: one fetch and one split of the rows) round the 8 waves while 8 tiles are left;
    //   - of the r < 8 tiles left over, four go whole to waves 0-3 (one per SIMD) if r >= 4, the others split by split to waves 4-7
    //     (r >= 4) or to all 8 waves (r < 4): at B = 1 024 (5 tiles per block) no SIMD runs two whole tiles while another runs one, and a
    //     small launch (one tile per block) puts its three splits on three SIMDs.
    // (Measured at B = 1 024: whole tiles only 7.6-8.0 us; every tile split by split — its rows fetched and split three times — 8.3-8.5.)
    const int nblk = (int)gridDim.x, bid = (int)blockIdx.x;
    const int nt_b = bid < ntiles ? (ntiles - bid + nblk - 1) / nblk : 0, nfull = nt_b & ~7, r = nt_b - nfull;
    const int nA = nfull >> 3;                                 // whole tiles of this wave in the first phase
    // unit n of this wave -> its tile (ntiles = none) and column splits
    auto unit = [&](int n, int& cs0, int& cs1) -> int {
        cs0 = 0; cs1 = CS;
        if (n < nA) return bid + (wave + 8 * n) * nblk;
        const int m = n - nA;
        if (r >= 4) {
            if (wave < 4) return m == 0 ? bid + (nfull + wave) * nblk : ntiles;
            const int jx = (wave - 4) + 4 * m;
            if (jx >= CS * (r - 4)) return ntiles;
            cs0 = jx % CS; cs1 = cs0 + 1;
            return bid + (nfull + 4 + jx / CS) * nblk;
        }
        const int jx = wave + 8 * m;
        if (jx >= CS * r) return ntiles;
        cs0 = jx % CS; cs1 = cs0 + 1;
        return bid + (nfull + jx / CS) * nblk;
    };
    int n = 0, cs0, cs1, cs0n, cs1n;
    int tile = unit(0, cs0, cs1);
    WRaw8 raw[FPW];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = wave * FPW + i, cs = f / (KS * 4), s = (f >> 2) % KS, t = f & 3;
        raw[i] = w_load8(a.Wimg, MP, cs * 64 + 16 * t + c, 32 * s + 8 * kb, Kp);
    }
    float4 af[KS][2];
    load_a(tile, af);
    // the bias goes to LDS (one float per column position; zero where there is none): a split reads its four with its operands
    float* s_bias = reinterpret_cast<float*>(s_wf + (size_t)NF * 3 * 1024);
    if (tid < MP) s_bias[tid] = (a.bias && tid < a.M1) ? a.bias[tid] : 0.f;
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = wave * FPW + i, s = (f >> 2) % KS;
        const Bf16x3 w = w_split8(raw[i], 32 * s + 8 * kb, Kp);
        char* p = s_wf + ((size_t)(f * 3) * 64 + lane) * 16;
        *reinterpret_cast<bf16x8_t*>(p) = w.hi;
        *reinterpret_cast<bf16x8_t*>(p + 1024) = w.mid;
        *reinterpret_cast<bf16x8_t*>(p + 2048) = w.lo;
    }
    __syncthreads();
    const bool wt = a.N <= kWtMaxRows;
    for (; tile < ntiles; ++n, tile = unit(n, cs0, cs1)) {
        Bf16x3 as[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) as[s] = split8(af[s][0], af[s][1]);
        load_a(unit(n + 1, cs0n, cs1n), af);                   // the wave's next unit (if any) flies under this one
        const size_t trow = (size_t)__builtin_amdgcn_readfirstlane(tile) * 16;
        float* o1 = a.out1 + trow * a.ldo1;
        float* o2 = a.out2 + trow * a.ldo2;
#pragma unroll 1
        for (int cs = cs0; cs < cs1; ++cs) {
            Bf16x3 wreg[KS][4];
            const char* wf = s_wf + ((size_t)(cs * KS * 4 * 3) * 64 + lane) * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const char* p = wf + (size_t)((s * 4 + t) * 3) * 1024;
                    wreg[s][t].hi = *reinterpret_cast<const bf16x8_t*>(p);
                    wreg[s][t].mid = *reinterpret_cast<const bf16x8_t*>(p + 1024);
                    wreg[s][t].lo = *reinterpret_cast<const bf16x8_t*>(p + 2048);
                }
            const int m0 = cs * 64 + 4 * c;
            const float4 b = ld4(s_bias + m0);
            v4f_t acc[4], accb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[t] = (v4f_t){0.f, 0.f, 0.f, 0.f}; accb[t] = acc[t]; }
            // (the chains of k_ts_gemm_x3: small partial products of both k steps, the middle ones, hi x hi in a chain of its own)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma_x3_small(as[s], wreg[s][t], acc[t]);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma_x3_mid(as[s], wreg[s][t], acc[t]);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) accb[t] = mfma_x3_big(as[s], wreg[s][t], accb[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += accb[t];
            // C layout: tile column = lane & 15 (-> logical columns cs * 64 + 4 c + t), row = (lane >> 4) * 4 + i
            if (m0 < M) {
                const bool first = m0 < a.M1;
                const unsigned lane_off = first ? (unsigned)(kb * 4 * a.ldo1 + m0) * 4u : (unsigned)(kb * 4 * a.ldo2 + (m0 - a.M1)) * 4u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (tile * 16 + kb * 4 + i >= a.N) continue;
                    const float4 v = make_float4(acc[0][i] + b.x, acc[1][i] + b.y, acc[2][i] + b.z, acc[3][i] + b.w);
                    if (first) st4o_sel(wt, o1, lane_off, v, (unsigned)(i * a.ldo1) * 4u);
                    else st4o_sel(wt, o2, lane_off, v, (unsigned)(i * a.ldo2) * 4u);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G[I, J] = [P1 | P2 | 1]^T @ Q.  A wave owns a 64 x 64 output slab (16 accumulator tiles, 64 VGPRs) over its own
// row range: per 4-row step ONE float4 load of P feeds the A operands of the 4 row tiles and ONE float4 load of Q
// the B operands of the 4 column tiles (stride-4 permutation on both sides: tile t holds columns 4c + t), i.e.
// 16 MFMAs per two fully coalesced 256-byte-per-row loads.  Block = (64-column slab of P, row split), 8 waves on
// disjoint row ranges, summed through LDS in wave order.
#ifndef GLAM_WG_STEPS
#define GLAM_WG_STEPS 4
#endif
#ifndef GLAM_WG_PAIR_BLOCKS
#define GLAM_WG_PAIR_BLOCKS 256
#endif
constexpr int kWgBlock = 512;
constexpr int kWgWaves = kWgBlock / 64;

// lanes whose 4 columns are the virtual ones column / zero padding read their operand from here with a row stride of 0 (no selects in
// the loop, no registers for the constants)
__device__ float4 g_wg_const[2] = {{1.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // not const: global address space, plain global loads
__device__ float4 g_wg_pass = {1.f, 1.f, 1.f, 1.f};       // the mask of the lanes a PMASK product does not mask (stride 0)

#ifndef GLAM_WG_ROLL
#define GLAM_WG_ROLL 1
#endif

// PMASK: P1 enters as P1 * (pmask > 0) — the backward of a ReLU in front of the product (dy of LinearBlock + ReLU, src_1gp/layer.py:232-237)
// applied where dy is consumed instead of by an elementwise launch of its own
template <bool CELU, bool SEG, bool PMASK = false>
__global__ void __launch_bounds__(kWgBlock) k_wgrad(WgArgs2 two) {
    static_assert(!(SEG && PMASK), "the masked product has one operand set");
    __shared__ float s_red[kWgWaves * 32 * 64];        // half of the 64 accumulator registers at a time: 64 KB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, kq = lane >> 4;
    const bool second = (int)blockIdx.x >= two.first_b;
    const WgArgs a = second ? two.b : two.a;      // by value (scalar selects): a reference made the compiler re-read the argument block
    // XCD-aware decode (blockIdx % 8 = XCD): the slabs of one row split share an XCD, so Q is fetched into one L2
    const int bid = second ? blockIdx.x - two.first_b : blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int slab = loc % a.ntile, split = (loc / a.ntile) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int row0 = __builtin_amdgcn_readfirstlane(min((split * kWgWaves + wave) * a.rows_per_wave, a.N));     // wave-uniform: scalar loop control
    const int row1 = min(row0 + a.rows_per_wave, a.N);
    const int pcol = slab * 64 + 4 * c, qcol = 4 * c;
    const int I12 = a.I1 + a.I2;
    // resolve this lane's P and Q sources once (I1, I2, J are multiples of 4: no straddling): base pointer + row stride in floats
    const float* psrc = reinterpret_cast<const float*>(&g_wg_const[(a.ones && pcol == I12) ? 0 : 1]);
    int pld = 0;
    if (pcol < a.I1) { psrc = a.P1 + pcol; pld = a.ldp1; }
    else if (pcol < I12) { psrc = a.P2 + (pcol - a.I1); pld = a.ldp2; }
    const float* qsrc = reinterpret_cast<const float*>(&g_wg_const[(a.qones && qcol == a.J) ? 0 : 1]);
    int qld = 0;
    if (qcol < a.J) { qsrc = a.Q + qcol; qld = a.ldq; }
    const float* msrc = reinterpret_cast<const float*>(&g_wg_pass);
    int mld = 0;
    if (PMASK && pcol < a.I1) { msrc = a.pmask + pcol; mld = a.ldp1; }

    v4f acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = (v4f){0.f, 0.f, 0.f, 0.f};

    constexpr int kSteps = GLAM_WG_STEPS;   // rows/4 per batch
    auto mma = [&](float4 pv, float4 qv, float4 mv = f4zero()) {
        if (CELU && a.q_celu) qv = celu4(qv);        // wave-uniform; celu is the identity on the constant lanes' 1 and 0
        if (PMASK) { pv.x = mv.x > 0.f ? pv.x : 0.f; pv.y = mv.y > 0.f ? pv.y : 0.f; pv.z = mv.z > 0.f ? pv.z : 0.f; pv.w = mv.w > 0.f ? pv.w : 0.f; }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(pv, ti), f4get(qv, tj), acc[ti][tj], 0, 0, 0);
    };
    // the rows [r0, r1) of one operand set (ps / qs: this lane's column of P / Q, row 0)
    // (always_inline: the SETS instantiation calls it twice, and as a real function its by-reference captures — the argument struct, the
    //  64 accumulator registers — went through scratch memory: 408 bytes per lane, 87 scratch instructions)
    auto rows = [&](int r0, int r1, const float* ps, const float* qs) __attribute__((always_inline)) {
        // full batches (4 kSteps rows, no masks), then one masked batch for the rest
        const int nfull = (r1 - r0) / (4 * kSteps);
        // ONE running pointer per operand: requests go out in row order (step after step, batch after batch)
        const float* pn = ps + (size_t)(r0 + kq) * pld;
        const float* qn = qs + (size_t)(r0 + kq) * qld;
        const int pstep = 4 * pld, qstep = 4 * qld;        // floats between consecutive steps of a lane
        if constexpr (PMASK) {
            // (the masked product is the small one behind the first linear: the plain batch loop, the mask rows requested with the operands)
            const float* mn = msrc + (size_t)(r0 + kq) * mld;
            const int mstep = 4 * mld;
            for (int n0 = r0; n0 < r1; n0 += 4 * kSteps) {
                float4 pt[kSteps], qt[kSteps], mt[kSteps];
#pragma unroll
                for (int st = 0; st < kSteps; ++st) {
                    const int rr = min(n0 + 4 * st + kq, r1 - 1) - (r0 + kq);      // clamped: unconditional loads, zeroed below
                    pt[st] = ld4g(pn + (size_t)rr * pld);
                    qt[st] = ld4g(qn + (size_t)rr * qld);
                    mt[st] = ld4g(mn + (size_t)rr * mld);
                }
#pragma unroll
                for (int st = 0; st < kSteps; ++st) {
                    if (n0 + 4 * st < r1) {                 // wave-uniform
                        if (n0 + 4 * st + kq >= r1) pt[st] = f4zero();
                        mma(pt[st], qt[st], mt[st]);
                    }
                }
            }
            (void)pstep; (void)qstep; (void)mstep; (void)nfull;
            return;
        }
    #if GLAM_WG_ROLL
        // rolling prefetch: the operands of step st of the NEXT batch are requested as soon as this batch's step st has issued its MFMAs
        // (its registers are free from then on): every load flies under 16 (kSteps - 1) of the wave's own MFMAs besides the other waves'
        float4 pl[kSteps], ql[kSteps];
        if (nfull > 0) {
    #pragma unroll
            for (int st = 0; st < kSteps; ++st) { pl[st] = ld4g(pn); ql[st] = ld4g(qn); pn += pstep; qn += qstep; }
        }
        for (int b = 0; b + 1 < nfull; ++b) {
    #pragma unroll
            for (int st = 0; st < kSteps; ++st) {
                mma(pl[st], ql[st]);
                pl[st] = ld4g(pn);
                ql[st] = ld4g(qn);
                pn += pstep;
                qn += qstep;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nfull > 0) {
    #pragma unroll
            for (int st = 0; st < kSteps; ++st) mma(pl[st], ql[st]);
        }
    #else
        for (int b = 0; b < nfull; ++b) {
            float4 pl[kSteps], ql[kSteps];
    #pragma unroll
            for (int st = 0; st < kSteps; ++st) { pl[st] = ld4g(pn); ql[st] = ld4g(qn); pn += pstep; qn += qstep; }
    #pragma unroll
            for (int st = 0; st < kSteps; ++st) mma(pl[st], ql[st]);
        }
    #endif
        {
            const int n0 = r0 + nfull * 4 * kSteps;
            if (n0 < r1) {                                 // wave-uniform
                float4 pt[kSteps], qt[kSteps];
    #pragma unroll
                for (int st = 0; st < kSteps; ++st) {
                    const bool nok = n0 + 4 * st + kq < r1;
                    pt[st] = nok ? ld4g(pn + st * pstep) : f4zero();
                    qt[st] = nok ? ld4g(qn + st * qstep) : f4zero();
                }
    #pragma unroll
                for (int st = 0; st < kSteps; ++st)
                    if (n0 + 4 * st < r1) mma(pt[st], qt[st]);       // wave-uniform: an empty step costs no MFMAs
            }
        }
    };
    if (!SEG) {
        rows(row0, row1, psrc, qsrc);
    } else {
        // several operand sets of seg_rows rows each behind one another (the applications of a block that shares its weights): a wave's
        // range lies in one set or straddles ONE boundary (rows_per_wave <= seg_rows)
        const int s0 = __builtin_amdgcn_readfirstlane(row0 / a.seg_rows), bnd = (s0 + 1) * a.seg_rows;
        const int e0 = min(row1, bnd);
        // The nine base pointers as VALUES in scalar registers, each loaded from BOTH argument structs and pinned before anything selects
        // between them.  A select between two argument FIELDS (`sg == 0 ? a.Q : a.segQ[0]`, `a.segQ[sg - 1]`, `second ? two.b.x : two.a.x`)
        // is folded into a load from a selected ADDRESS, and a kernel argument addressed dynamically is first copied to scratch memory:
        // 168-408 bytes per lane and 87 scratch instructions in this kernel (k_gru_fwd_ws had the same fold as per-lane pointer loads).
#define GLAM_WG_PICK(name, field)                                                          \
        const float* name;                                                                 \
        { const float* x_ = two.a.field; const float* y_ = two.b.field; asm volatile("" : "+s"(x_), "+s"(y_)); name = second ? y_ : x_; }
        GLAM_WG_PICK(q0, Q) GLAM_WG_PICK(q1, segQ[0]) GLAM_WG_PICK(q2, segQ[1])
        GLAM_WG_PICK(p10, P1) GLAM_WG_PICK(p11, segP1[0]) GLAM_WG_PICK(p12, segP1[1])
        GLAM_WG_PICK(p20, P2) GLAM_WG_PICK(p21, segP2[0]) GLAM_WG_PICK(p22, segP2[1])
#undef GLAM_WG_PICK
        // (plain values, no capturing lambda: see above)
        const bool p_first = pcol < a.I1;
        const int poff = p_first ? pcol : pcol - a.I1;
        const float* pa0 = p_first ? p10 : p20; const float* pa1 = p_first ? p11 : p21; const float* pa2 = p_first ? p12 : p22;
        const float* pb_s0 = pld == 0 ? psrc : (s0 == 0 ? pa0 : s0 == 1 ? pa1 : pa2) + poff;
        const float* pb_s1 = pld == 0 ? psrc : (s0 == 0 ? pa1 : pa2) + poff;
        const float* qb_s0 = qld == 0 ? qsrc : (s0 == 0 ? q0 : s0 == 1 ? q1 : q2) + qcol;
        const float* qb_s1 = qld == 0 ? qsrc : (s0 == 0 ? q1 : q2) + qcol;
        rows(row0 - s0 * a.seg_rows, e0 - s0 * a.seg_rows, pb_s0, qb_s0);
        if (row1 > bnd) rows(0, row1 - bnd, pb_s1, qb_s1);
    }
    // ---- sum the 8 waves lane-for-lane (identical register layouts) in wave order.  Accumulator tile t = ti*4 + tj
    //      of a lane is one float4 (r = 0..3): 8 tiles per half go to LDS as b128 stores [wave][t][lane], thread
    //      (t, lane) adds the 8 waves' float4 and stores the block partial as out[(t*64 + lane)*4 + r] ----
    float* out = a.partial + ((size_t)slab * a.nsplit + split) * kWgSlabStride;
    float4* s_red4 = reinterpret_cast<float4*>(s_red);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int tt = half * 8 + t;
            const v4f v = acc[tt >> 2][tt & 3];
            s_red4[(wave * 8 + t) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        {
            const int t = tid >> 6;                    // 8 tiles x 64 lanes = 512 threads
            float4 sum = s_red4[t * 64 + lane];
#pragma unroll
            for (int w = 1; w < kWgWaves; ++w) {
                const float4 v = s_red4[(w * 8 + t) * 64 + lane];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            st4o_sel(a.N <= kWtMaxRows, out, (unsigned)(((half * 8 + t) * 64 + lane) * 16), sum);      // (see st4o_wt)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Final fixed-order reduction of up to 3 partial sets in ONE launch.
//   kind 0 (k_wgrad partials): element offset `o` in [0, nslab*4096) decodes to r = o%4, lane = (o/4)%64,
//          t = (o%4096)/256, ti = t/4, tj = t%4, kq = lane/16, c = lane%16  ->  i = 64*slab + 4*(4*kq + r) + ti,
//          j = 4*c + tj;  out[i*si + j*sj] = sum_s partial[(slab*nsplit + s)*kWgSlabStride + o%4096]
//   kind 1 (flat block partials [nsplit][n]): out[e] (e < split_at) or out2[e - split_at] = sum_s partial[s*n + e]
__global__ void __launch_bounds__(kBlock) k_final_reduce(ReduceArgs ra) {
    __shared__ float s_part[16][17];
    int jb = 0;
#pragma unroll
    for (int q = 1; q < 3; ++q)
        if (q < ra.njobs && (int)blockIdx.x >= ra.job[q].first_block) jb = q;
    const ReduceJob& J = ra.job[jb];
    const int c = threadIdx.x & 15, rl = threadIdx.x >> 4;       // 16 elements x 16 split lanes
    const int e = ((int)blockIdx.x - J.first_block) * 16 + c;
    float s0 = 0.f, s1 = 0.f;
    if (e < J.n) {
        const float* p;
        size_t stride;
        if (J.kind == 0) { p = J.partial + (size_t)(e >> 12) * J.nsplit * kWgSlabStride + (e & 4095); stride = kWgSlabStride; }
        else { p = J.partial + e; stride = (size_t)J.n; }
        int s = rl;
        for (; s + 16 < J.nsplit; s += 32) { s0 += p[(size_t)s * stride]; s1 += p[(size_t)(s + 16) * stride]; }
        for (; s < J.nsplit; s += 16) s0 += p[(size_t)s * stride];
    }
    s_part[rl][c] = s0 + s1;
    __syncthreads();
    if (rl == 0 && e < J.n) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += s_part[r][c];
        if (J.kind == 0) {
            const int o = e & 4095, r = o & 3, lane = (o >> 2) & 63, t = o >> 8;
            const int i = (e >> 12) * 64 + 4 * ((lane >> 4) * 4 + r) + (t >> 2), j = 4 * (lane & 15) + (t & 3);
            if (i < J.I && j < J.J) {
                if (J.out_b && j == J.J - 1) {
                    J.out_b[i] = J.add_b ? s + J.add_b[i] : s;
                } else if (J.jw == 0 || j < J.jw) {
                    const size_t o2 = (size_t)i * J.si + (size_t)j * J.sj;
                    J.out[o2] = J.addend ? s + J.addend[o2] : s;
                }
            }
        } else if (e < J.split_at) {
            if (J.out) J.out[e] = s;
        } else {
            J.out2[e - J.split_at] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (shared with layer.hip)
// ------------------------------------------------------------------------------------------------
// Kernel variants <MT, GMAX> (column tiles of 16, k-groups of 16) by shape; make_image and the GEMM agree through this:
//   0: M <= 64,  K <= 192   <4, 12>    48 KB image         2: K <= 96,  M <= 320   <20, 6>   120 KB  (hid_dim_alpha = 6:
//   1: K <= 64,  M <= 192   <12, 4>    48 KB                                                           92 -> 276 + 8)
// (a <8, 18> variant for K <= 288, M <= 128 was measured at 28.8 us against the library GEMM's 19.2 us for 276 -> 92 at
//  N = 20.7 k — a 144 KB image per block and 1.26 items per wave — and is not kept)
//   3: K <= 288, M <= 96   3 x bf16 only (tall_x3.hip: no LDS image, MT = 8 positions per row)   (hid_dim_alpha = 6: 276 -> 92)
static int ts_variant(int K, int M) {
    if (M <= 0 || K <= 0) return -1;
    if (M <= 64 && K <= 192) return 0;
    if (K <= 64 && M <= 192) return 1;
    if (K <= 96 && M <= 320) return 2;
    if (K <= 288 && M <= 96) return 3;
    if (K <= 320 && M <= 64) return 4;      // 3 x bf16 only (NNConv's relation product [N, De C] x [De C, C]: 300 -> 60)
    return -1;
}
static int ts_mt(int variant) { return variant == 0 || variant == 4 ? 4 : variant == 1 ? 12 : variant == 2 ? 20 : 8; }

size_t ts_image_floats(int K, int M) {
    const int Kp = (K + 15) & ~15, v = ts_variant(K, M);
    return (size_t)Kp * 16 * ts_mt(v < 0 ? 1 : v);
}

static int ts_shape_ok(const char* fn, int K, int M) {
    if (ts_variant(K, M) < 0)
        return fail(GLAM_E_UNSUPPORTED, "%s: K=%d with M=%d outside the kernel table (K<=192,M<=64 | K<=64,M<=192 | K<=96,M<=320 | K<=288,M<=96 | K<=320,M<=64)", fn, K, M);
    return GLAM_OK;
}

int launch_ts_make_image(const float* W, int ldw, int transW, int K, int M, float* img, hipStream_t s) {
    if (int rc = ts_shape_ok("ts_gemm image", K, M)) return rc;
    const int MT = ts_mt(ts_variant(K, M));
    hipLaunchKernelGGL(k_ts_make_image, dim3(grid_for((int64_t)ts_image_floats(K, M), kBlock)), dim3(kBlock), 0, s, W, ldw,
                       transW, K, M, MT, img);
    GLAM_LAUNCH_CHECK("ts_make_image");
    return GLAM_OK;
}

// one job of a k_ts_make_images-style launch: returns the number of blocks it owns, < 0 (an error code) for a shape outside the table
int image_job(ImageJob& j, const char* fn, const float* W, int ldw, int transW, int K, int M, float* img, int first) {
    if (int rc = ts_shape_ok(fn, K, M)) return rc;
    j = ImageJob{W, ldw, transW, K, M, ts_mt(ts_variant(K, M)), img, first, 0};
    return (int)((ts_image_floats(K, M) + kBlock - 1) / kBlock);
}

static int ts_plan(const TsArgs& a, int* variant, int* grid) {
    const int M = a.M1 + a.M2, K = a.K1 + a.K2;
    if (int rc = ts_shape_ok("ts_gemm", K, M)) return rc;
    if ((a.K1 & 3) || (a.K2 & 3) || (a.lda1 & 3) || (a.K2 && (a.lda2 & 3)) || (a.M1 & 3) || (a.M2 & 3) || (a.ldo1 & 3) ||
        (a.M2 && (a.ldo2 & 3)))
        return fail(GLAM_E_UNSUPPORTED, "ts_gemm: K=%d+%d M=%d+%d and leading dimensions must be multiples of 4", a.K1, a.K2, a.M1, a.M2);
    const int ntiles = (a.N + 15) / 16;
    *variant = ts_variant(K, M);
    // column splits per row tile (MT / TPI).  K <= 192 x 64 columns: splitting would re-read the long A rows
    const int nitems = ntiles * (*variant == 0 || *variant >= 3 ? 1 : *variant == 1 ? 3 : 5);
    int g = nitems < 2048 ? nitems : (nitems + 7) / 8;   // < one item per wave slot: one block per item first (see `spread`)
    if (g > 256) g = 256;                // one 8-wave block per CU, items dealt round-robin over every wave of the grid
    *grid = g;
    return GLAM_OK;
}
// the register-B form of the 192-column variant: one block per CU
static bool ts_rb_enabled() { return true; }       // (the register-B form of the wide products: an A/B switch until round 5)
static bool tall_x3_enabled() { const char* e = getenv("GLAM_TALL_X3"); return !e || atoi(e) != 0; }     // A/B switch of tall_x3.hip
// GLAM_X3=0: the dense products stay on the fp32 matrix instructions (A/B switch, read per call)
bool ts_x3_enabled() { const char* e = getenv("GLAM_X3"); return !e || atoi(e) != 0; }
static bool ts_sw_enabled() { const char* e = getenv("GLAM_TS_SW"); return !e || atoi(e) != 0; }      // A/B switch of k_ts_gemm_x3_sw
static bool ts_rb_big(int N) { return N >= 131072; }       // 12-wave blocks once the launch streams from HBM
static int ts_rb_grid(int N) {
    const int ntiles = (N + 15) / 16;
    if (ts_rb_big(N)) return 256;        // 12 waves per block: any block count is a multiple of the 3 column splits
    int g = (ntiles * 3 + 7) / 8;        // one item per wave
    g = (g + 2) / 3 * 3;                 // 8 g waves: a multiple of the 3 column splits
    return g > 255 ? 255 : g;
}

// b == nullptr: one product; otherwise two products of the SAME variant in one launch
int launch_ts_gemm2(const TsArgs& a, const TsArgs* b, hipStream_t s) {
    if (a.N <= 0) return GLAM_OK;
    int variant = -1, grid_a = 0, grid_b = 0;
    if (int rc = ts_plan(a, &variant, &grid_a)) return rc;
    TsArgs2 two{a, b ? *b : a, grid_a};
    if (b) {
        int vb = -1;
        if (int rc = ts_plan(*b, &vb, &grid_b)) return rc;
        if (vb != variant || b->N != a.N) return fail(GLAM_E_UNSUPPORTED, "ts_gemm pair: the two products must share N and the kernel variant");
    }
    // each job stages ITS OWN image ((K + 15) / 16 groups): the allocation must hold the larger of the two
    size_t lds = ts_image_floats(a.K1 + a.K2, a.M1 + a.M2) * sizeof(float);
    if (b) { const size_t lb = ts_image_floats(b->K1 + b->K2, b->M1 + b->M2) * sizeof(float); if (lb > lds) lds = lb; }
    const int grid = grid_a + grid_b;
    // the long-reduction shapes on the bf16 matrix cores (tall_x3.hip); GLAM_X3=0 keeps the fp32 matrix instructions where they exist
    if (b && (a.rng_state || b->rng_state || a.node_pre || b->node_pre))
        return fail(GLAM_E_UNSUPPORTED, "ts_gemm pair: the RReLU epilogue / the node product take a launch of their own");
    if (variant >= 3 || (variant == 0 && ts_x3_enabled() && tall_x3_enabled())) return launch_tall_x3(a, b, variant, s);
    if (a.out_relu || (b && b->out_relu) || a.rng_state || a.node_pre) return fail(GLAM_E_UNSUPPORTED, "ts_gemm: the ReLU / RReLU epilogues exist in k_tall_x3 only (glam_ts_gemm_relu_supported)");
    if (variant == 2 && ts_x3_enabled() && tall_x3_enabled() && !a.cgrad_src && !a.addend && !(b && (b->cgrad_src || b->addend)))
        return launch_tall_x3(a, b, variant, s);
    if (variant == 0) hipLaunchKernelGGL((k_ts_gemm<4, 12, 4>), dim3(grid), dim3(kTsBlock), lds, s, two);
    else if (variant == 1 && !b && ts_rb_enabled() && a.K2 == 0 && !a.a_celu && !a.cgrad_src && !a.addend) {
        // the register-B form is the plain product only (one A source, no folded CELU, no gradient epilogue, fp32 out): its registers
        // decide its occupancy
        two.first_b = ts_rb_grid(a.N);
        GLAM_PROF_LABEL("k_ts_gemm<12, 4, 4>");
        if (ts_x3_enabled()) {      // 3 x bf16 on the bf16 matrix cores (fp32 accuracy, bf16x3.h)
            if (ts_sw_enabled()) {      // W split once per block, whole row tiles per wave (k_ts_gemm_x3_sw), at every size
                static bool big[64] = {};      // > 64 KB of dynamic LDS is opted into once per device
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
                if (!big[dev] || dev == 63) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ts_gemm_x3_sw<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 73 * 1024) != hipSuccess)
                        return fail(GLAM_E_HIP, "ts_gemm_x3_sw: opting into 73 KB of dynamic LDS failed");
                    big[dev] = true;
                }
                const int ntiles = (a.N + 15) / 16, g = (ntiles + 1) / 2;
                hipLaunchKernelGGL((k_ts_gemm_x3_sw<2, 3>), dim3(g < 256 ? g : 256), dim3(512), 3 * 2 * 4 * 3 * 1024 + 1024, s, a);
            } else if (ts_rb_big(a.N)) hipLaunchKernelGGL((k_ts_gemm_x3<768>), dim3(two.first_b), dim3(768), 0, s, a, two.first_b);
            else hipLaunchKernelGGL((k_ts_gemm_x3<512>), dim3(two.first_b), dim3(512), 0, s, a, two.first_b);
        } else if (ts_rb_big(a.N)) hipLaunchKernelGGL((k_ts_gemm<12, 4, 4, 768>), dim3(two.first_b), dim3(768), 0, s, two);
        else hipLaunchKernelGGL((k_ts_gemm<12, 4, 4, 512>), dim3(two.first_b), dim3(512), 0, s, two);
    } else if (variant == 1) hipLaunchKernelGGL((k_ts_gemm<12, 4, 4>), dim3(grid), dim3(kTsBlock), lds, s, two);
    else if (variant == 2 && !b && ts_rb_enabled() && ts_x3_enabled() && a.K2 == 0 && !a.a_celu && !a.cgrad_src && !a.addend) {
        // hid_dim_alpha = 6 (92 -> 276 + 8): three 32-k steps x four column tiles of W in 144 registers, five 64-column splits
        const int ntiles = (a.N + 15) / 16;
        int g = ((ntiles * 5 + 7) / 8 + 4) / 5 * 5;      // one item per wave; 8 g waves: a multiple of the 5 column splits
        if (g > 255) g = 255;
        GLAM_PROF_LABEL("k_ts_gemm<20, 6, 4>");
        hipLaunchKernelGGL((k_ts_gemm_x3<512, 3, 5>), dim3(g), dim3(512), 0, s, a, g);
    } else {
        static bool big2 = false;      // > 64 KB of dynamic LDS is opted into once
        if (!big2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ts_gemm<20, 6, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            big2 = true;
        }
        hipLaunchKernelGGL((k_ts_gemm<20, 6, 4>), dim3(grid), dim3(kTsBlock), lds, s, two);
    }
    GLAM_LAUNCH_CHECK("ts_gemm");
    return GLAM_OK;
}
int launch_ts_gemm(const TsArgs& a, hipStream_t s) { return launch_ts_gemm2(a, nullptr, s); }

constexpr int kWgradBlocks = 256;      // about one 8-wave block per CU: the right grid while the operands are cache resident
constexpr int kWgradBlocksBig = 512;   // two per CU (four waves per SIMD) once they stream from HBM: B = 16 384: 169 vs 182 us, B = 1 024: 15.0 vs 14.4
static int wgrad_big_rows() { return 131072; }
static int wgrad_budget(int N) { return N >= wgrad_big_rows() ? kWgradBlocksBig : kWgradBlocks; }

size_t wgrad_workspace_floats() { return (size_t)(3 * 256 + 24) * kWgSlabStride; }     // per product: one 64 x 64 slab per (slab, split); k_wgrad_x3: <= 768

// fills the launch geometry of one product (at most `budget` blocks) and the matching reduce job
static int plan_wgrad(WgArgs& a, float* out, int si, int sj, int budget, ReduceJob* job, int* blocks) {
    const int I = a.I1 + a.I2 + (a.ones ? 1 : 0);
    const int Jt = a.J + (a.qones ? 1 : 0);
    if (I > 320 || Jt > 64 || I <= 0 || a.J <= 0 || (a.J & 3) || (a.ldq & 3) || (a.I1 & 3) || (a.I2 & 3) || (a.ldp1 & 3) ||
        (a.I2 && (a.ldp2 & 3)))
        return fail(GLAM_E_UNSUPPORTED, "wgrad: I=%d+%d J=%d outside the kernel table (I <= 320, J <= 64, multiples of 4)",
                    a.I1, a.I2, a.J);
    const int nslab = (I + 63) / 64;
    int nsplit = budget / nslab / 8 * 8;         // multiple of 8: one XCD per row split
    if (nsplit < 8) nsplit = 8;
    // keep at least 16 rows per wave when the problem is small
    const int max_split = (int)(((int64_t)a.N + 16 * kWgWaves - 1) / (16 * kWgWaves));
    if (nsplit > max_split) nsplit = max_split < 1 ? 1 : max_split;
    int rpw = (int)(((int64_t)a.N + nsplit * kWgWaves - 1) / (nsplit * kWgWaves));
    rpw = (rpw + 3) & ~3;
    if (rpw < 4) rpw = 4;
    a.rows_per_wave = rpw;
    a.nsplit = nsplit;
    a.ntile = nslab;
    *blocks = (nsplit + 7) / 8 * 8 * nslab;
    *job = ReduceJob{0, a.partial, nsplit, nslab * 4096, I, Jt, si, sj, out, nullptr, 0, 0};
    return GLAM_OK;
}

// k_wgrad_x3 (wgrad_x3.hip): one 8-wave block per CU whatever the size; a block = (group of up to three slabs, row split) with
// rows_per_split a multiple of the 32-row step; the splits of a multi-set product are dealt set by set (nsplit = splits per set x nseg).
// Every split leaves one partial slab per slab of the product, so the split count is also bounded by the workspace (kWxMaxSlabs).
constexpr int kWxSlabsPerBlock = 3, kWxMaxSlabs = 3 * 256;
static bool wgrad_x3_enabled() {      // (GLAM_WGRAD_X3: A/B switch; GLAM_X3=0 puts every product back on the fp32 matrix instructions)
    static const bool v = [] { const char* e = getenv("GLAM_WGRAD_X3"); return !e || atoi(e) != 0; }();
    return v && ts_x3_enabled();
}
static int plan_wgrad_x3(WgArgs& a, float* out, int si, int sj, int budget, ReduceJob* job, int* blocks) {
    const int I = a.I1 + a.I2 + (a.ones ? 1 : 0);
    const int Jt = a.J + (a.qones ? 1 : 0);
    if (I > 320 || Jt > 64 || I <= 0 || a.J <= 0 || (a.J & 3) || (a.ldq & 3) || (a.I1 & 3) || (a.I2 & 3) || (a.ldp1 & 3) ||
        (a.I2 && (a.ldp2 & 3)))
        return fail(GLAM_E_UNSUPPORTED, "wgrad: I=%d+%d J=%d outside the kernel table (I <= 320, J <= 64, multiples of 4)",
                    a.I1, a.I2, a.J);
    const int nslab = (I + 63) / 64, ngrp = (nslab + kWxSlabsPerBlock - 1) / kWxSlabsPerBlock;
    const int nseg = a.nseg > 1 ? a.nseg : 1;
    const int64_t rows = nseg > 1 ? a.seg_rows : a.N;          // rows of one operand set
    int nsplit = budget / ngrp;
    if (nsplit > kWxMaxSlabs / nslab) nsplit = kWxMaxSlabs / nslab;
    nsplit = nsplit / 8 * 8;                     // multiple of 8: one XCD per row split
    if (nsplit < 8) nsplit = 8;
    int per = nsplit / nseg;
    const int max_per = (int)((rows + 63) / 64);               // at least two steps per block when the problem is small
    if (per > max_per) per = max_per;
    if (per < 1) per = 1;
    int rps = (int)((rows + per - 1) / per);
    rps = (rps + 31) & ~31;
    if (rps < 32) rps = 32;
    per = (int)((rows + rps - 1) / rps);                       // no empty split
    if (per < 1) per = 1;
    a.rows_per_split = rps;
    a.rows_per_wave = rps;
    a.nsplit = per * nseg;
    a.ntile = nslab;
    *blocks = (a.nsplit + 7) / 8 * 8 * ngrp;
    *job = ReduceJob{0, a.partial, a.nsplit, nslab * 4096, I, Jt, si, sj, out, nullptr, 0, 0};
    return GLAM_OK;
}

// rows from which k_wgrad_x3 takes a launch (GLAM_WGRAD_X3_ROWS: A/B switch): below, its fixed cost — 8-wave blocks behind a barrier,
// the first 48 KB stage — and the larger reduction cancel what the bf16 matrix instructions save (N = 20 400: 13.55 vs 13.57 us)
static int wgrad_x3_rows() {       // (read at every launch: the tests switch it)
    const char* e = getenv("GLAM_WGRAD_X3_ROWS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 32768;
}

int launch_wgrad_partials(WgArgs a, float* out, int si, int sj, hipStream_t s, ReduceJob* job, bool many_splits) {
    int blocks = 0;
    if (many_splits && a.N >= wgrad_x3_rows() && wgrad_x3_enabled() && !a.pmask) {
        if (int rc = plan_wgrad_x3(a, out, si, sj, kWgradBlocks, job, &blocks)) return rc;
        return launch_wgrad_x3(WgArgs2{a, a, blocks}, blocks, s);
    }
    if (int rc = plan_wgrad(a, out, si, sj, wgrad_budget(a.N), job, &blocks)) return rc;
    if (a.nseg > 1 && a.rows_per_wave > a.seg_rows)
        return fail(GLAM_E_UNSUPPORTED, "wgrad: %d operand sets of %d rows are too short for a wave's %d rows (run them one by one)", a.nseg,
                    a.seg_rows, a.rows_per_wave);
    WgArgs2 two{a, a, blocks};
    GLAM_PROF_LABEL(a.pmask ? "k_wgrad<relu mask>" : a.nseg > 1 ? "k_wgrad<true, sets>" : a.q_celu ? "k_wgrad<true>" : "k_wgrad<false>");      // (the labels bench.py's kernel table is keyed by)
    if (a.pmask) {
        if (a.nseg > 1 || a.q_celu) return fail(GLAM_E_UNSUPPORTED, "wgrad: the masked product takes one operand set and no folded CELU");
        hipLaunchKernelGGL((k_wgrad<false, false, true>), dim3(blocks), dim3(kWgBlock), 0, s, two);
    } else if (a.nseg > 1) hipLaunchKernelGGL((k_wgrad<true, true>), dim3(blocks), dim3(kWgBlock), 0, s, two);
    else if (a.q_celu) hipLaunchKernelGGL((k_wgrad<true, false>), dim3(blocks), dim3(kWgBlock), 0, s, two);
    else hipLaunchKernelGGL((k_wgrad<false, false>), dim3(blocks), dim3(kWgBlock), 0, s, two);
    GLAM_LAUNCH_CHECK("wgrad");
    return GLAM_OK;
}

// two products in one launch; the block budget is shared in proportion to their slab counts
int launch_wgrad_partials2(WgArgs a, float* out_a, int si_a, int sj_a, ReduceJob* job_a, WgArgs b, float* out_b, int si_b,
                           int sj_b, ReduceJob* job_b, hipStream_t s, bool many_splits) {
    const int ta = (a.I1 + a.I2 + (a.ones ? 1 : 0) + 63) / 64, tb = (b.I1 + b.I2 + (b.ones ? 1 : 0) + 63) / 64;
    const bool x3 = many_splits && a.N >= wgrad_x3_rows() && b.N >= wgrad_x3_rows() && wgrad_x3_enabled();
    const int total = (!x3 && a.N >= wgrad_big_rows()) ? 2 * GLAM_WG_PAIR_BLOCKS : GLAM_WG_PAIR_BLOCKS;
    const int ba = ta + tb > 0 ? total * ta / (ta + tb) : total / 2;
    int na = 0, nb = 0;
    if (x3) {
        const int ga = (ta + kWxSlabsPerBlock - 1) / kWxSlabsPerBlock, gb = (tb + kWxSlabsPerBlock - 1) / kWxSlabsPerBlock;
        const int xa = total * ga / (ga + gb);                  // the budget follows the slab groups (the blocks per row split)
        if (int rc = plan_wgrad_x3(a, out_a, si_a, sj_a, xa, job_a, &na)) return rc;
        if (int rc = plan_wgrad_x3(b, out_b, si_b, sj_b, total - xa, job_b, &nb)) return rc;
        return launch_wgrad_x3(WgArgs2{a, b, na}, na + nb, s);
    }
    if (int rc = plan_wgrad(a, out_a, si_a, sj_a, ba, job_a, &na)) return rc;
    if (int rc = plan_wgrad(b, out_b, si_b, sj_b, total - ba, job_b, &nb)) return rc;
    if ((a.nseg > 1 && a.rows_per_wave > a.seg_rows) || (b.nseg > 1 && b.rows_per_wave > b.seg_rows))
        return fail(GLAM_E_UNSUPPORTED, "wgrad: %d operand sets of %d rows are too short for a wave's %d rows (run them one by one)", a.nseg,
                    a.seg_rows, a.rows_per_wave);
    WgArgs2 two{a, b, na};
    GLAM_PROF_LABEL(a.nseg > 1 ? "k_wgrad<true, sets>" : (a.q_celu || b.q_celu) ? "k_wgrad<true>" : "k_wgrad<false>");
    if (a.nseg > 1) hipLaunchKernelGGL((k_wgrad<true, true>), dim3(na + nb), dim3(kWgBlock), 0, s, two);
    else if (a.q_celu || b.q_celu) hipLaunchKernelGGL((k_wgrad<true, false>), dim3(na + nb), dim3(kWgBlock), 0, s, two);
    else hipLaunchKernelGGL((k_wgrad<false, false>), dim3(na + nb), dim3(kWgBlock), 0, s, two);
    GLAM_LAUNCH_CHECK("wgrad(pair)");
    return GLAM_OK;
}

// N = 0 (an empty batch / shard): the products are all-zero matrices
__global__ void __launch_bounds__(kBlock) k_zero_strided(float* out, int I, int J, int si, int sj) {
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < I * J; e += gridDim.x * kBlock) out[(size_t)(e / J) * si + (size_t)(e % J) * sj] = 0.f;
}
static int zero_product(float* out, int I, int J, int si, int sj, hipStream_t s) {
    if (I * J > 0) hipLaunchKernelGGL(k_zero_strided, dim3(grid_for((int64_t)I * J, kBlock)), dim3(kBlock), 0, s, out, I, J, si, sj);
    GLAM_LAUNCH_CHECK("wgrad(N = 0)");
    return GLAM_OK;
}

int launch_final_reduce(ReduceArgs ra, hipStream_t s) {
    int blocks = 0;
    for (int q = 0; q < ra.njobs; ++q) {
        ra.job[q].first_block = blocks;
        blocks += (ra.job[q].n + 15) / 16;
    }
    if (blocks == 0) return GLAM_OK;
    hipLaunchKernelGGL(k_final_reduce, dim3(blocks), dim3(kBlock), 0, s, ra);
    GLAM_LAUNCH_CHECK("final_reduce");
    return GLAM_OK;
}

}  // namespace glam

#ifdef GLAM_TS_PROF
extern "C" int glam_debug_ts_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_ts_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

using namespace glam;

extern "C" int glam_route_enabled(const char* route) {
    GLAM_REQUIRE(route, "glam_route_enabled: null name");
    if (!strcmp(route, "x3")) return ts_x3_enabled() ? 1 : 0;
    if (!strcmp(route, "wgrad_x3")) return wgrad_x3_enabled() ? 1 : 0;
    return fail(GLAM_E_INVALID, "glam_route_enabled: unknown route '%s'", route);
}

extern "C" size_t glam_ts_gemm_image_bytes(int K, int M) { return ts_image_floats(K, M) * sizeof(float); }

extern "C" int glam_ts_gemm_make_image(const float* W, int ldw, int transW, int K, int M, float* img, void* stream) {
    GLAM_REQUIRE(W && img && aligned16(img), "glam_ts_gemm_make_image: null / misaligned pointer");
    return launch_ts_make_image(W, ldw, transW, K, M, img, (hipStream_t)stream);
}

extern "C" int glam_ts_gemm(const float* A1, int K1, int lda1, const float* A2, int K2, int lda2, const float* Wimg,
                            const float* bias, float* out1, int M1, int ldo1, float* out2, int M2, int ldo2, int64_t N,
                            void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_ts_gemm: N out of range");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(A1 && Wimg && out1 && (K2 == 0 || A2) && (M2 == 0 || out2), "glam_ts_gemm: null pointer");
    GLAM_REQUIRE(aligned16(A1) && aligned16(A2) && aligned16(Wimg) && aligned16(out1) && aligned16(out2) && aligned16(bias),
                 "glam_ts_gemm: pointers must be 16-byte aligned");
    TsArgs a{A1, K1, lda1, A2, K2, lda2, Wimg, bias, out1, M1, ldo1, out2, M2, ldo2, (int)N};
    return launch_ts_gemm(a, (hipStream_t)stream);
}

// out = max(A @ W + bias, 0): the linear + ReLU of a LinearBlock whose shape runs on k_tall_x3 (the input embeddings 15 -> 60 ...)
extern "C" int glam_ts_gemm_relu_supported(int K, int M) {
    if (K <= 0 || M <= 0 || (K & 3) || (M & 3)) return 0;
    const int v = ts_variant(K, M);        // (< 0: outside the kernel table)
    return (v >= 3 || (v == 0 && ts_x3_enabled() && tall_x3_enabled())) ? 1 : 0;
}
extern "C" int glam_ts_gemm_relu(const float* A, int K, int lda, const float* Wimg, const float* bias, float* out, int M, int ldo, int64_t N,
                                 void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_ts_gemm_relu: N out of range");
    if (!glam_ts_gemm_relu_supported(K, M)) return fail(GLAM_E_UNSUPPORTED, "glam_ts_gemm_relu: K=%d M=%d does not run on k_tall_x3", K, M);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(A && Wimg && out, "glam_ts_gemm_relu: null pointer");
    GLAM_REQUIRE(aligned16(A) && aligned16(Wimg) && aligned16(out) && aligned16(bias), "glam_ts_gemm_relu: pointers must be 16-byte aligned");
    TsArgs a{A, K, lda, nullptr, 0, 0, Wimg, bias, out, M, ldo, nullptr, 0, 0, (int)N};
    a.out_relu = 1;
    return launch_ts_gemm(a, (hipStream_t)stream);
}

extern "C" int glam_ts_gemm_rrelu_supported(int K, int M) {
    if (K <= 0 || M <= 0 || (K & 3) || (M & 3) || K > 64) return 0;
    return (ts_variant(K, M) == 0 && ts_x3_enabled() && tall_x3_enabled()) ? 1 : 0;
}

// out = RReLU(lower, upper)(A @ W + bias) in TRAINING mode (src_1gp/model.py:31: the reference's default activation) and, out_drop
// non-null, out_drop = Dropout(drop_p)(out) — the twin the block behind starts with (layer.py:255-256) — from the device-side Philox
// stream (rng.h): the words glam_bias_res_act_rng_fwd would draw for the same elements at the same stream position, so the pair
// (glam_ts_gemm, glam_bias_res_act_rng_fwd) and this one launch write the same bits; rng_eff receives the (seed, offset) pair for
// glam_bias_res_act_rng_bwd.  out, out_drop: [N, M] contiguous.  Shapes: glam_ts_gemm_rrelu_supported (the input embeddings).
extern "C" int glam_ts_gemm_rrelu(const float* A, int K, int lda, const float* Wimg, const float* bias, int M, int64_t N, float rr_lower,
                                  float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* out, float* out_drop,
                                  void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_ts_gemm_rrelu: N out of range");
    if (!glam_ts_gemm_rrelu_supported(K, M)) return fail(GLAM_E_UNSUPPORTED, "glam_ts_gemm_rrelu: K=%d M=%d outside k_tall_x3's RReLU epilogue (K <= 64, M <= 64)", K, M);
    GLAM_REQUIRE(rr_lower > 0.f && rr_lower <= rr_upper && drop_p >= 0.f && drop_p < 1.f, "glam_ts_gemm_rrelu: needs 0 < lower <= upper and 0 <= p < 1 "
                 "(got %g, %g, %g)", rr_lower, rr_upper, drop_p);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(A && Wimg && out && rng_state && rng_eff, "glam_ts_gemm_rrelu: null pointer");
    GLAM_REQUIRE(aligned16(A) && aligned16(Wimg) && aligned16(out) && aligned16(bias) && aligned16(out_drop), "glam_ts_gemm_rrelu: pointers must be 16-byte aligned");
    TsArgs a{A, K, lda, nullptr, 0, 0, Wimg, bias, out, M, M, nullptr, 0, 0, (int)N};
    a.rng_state = reinterpret_cast<long long*>(rng_state);
    a.rng_eff = reinterpret_cast<long long*>(rng_eff);
    a.rr_lo = rr_lower; a.rr_hi = rr_upper; a.drop_p = drop_p; a.out_drop = out_drop;
    return launch_ts_gemm(a, (hipStream_t)stream);
}

// The input embedding in front of a TripletMessage (src_1gp/model.py:49, :53): out = act(A @ W + bias) — act 0: none, 1: ReLU,
// 4: training-mode RReLU (+ the dropped twin, as glam_ts_gemm_rrelu) — AND the TripletMessage's node product of those rows in the same
// launch: xw[N, node_cols] | a_ij[N, 8] = out (out_drop when given) @ [W_node | Wa] from node_pre, the layer's pre-split fragment image
// (staged + glam_triplet_staged_node_fragments) — the values glam_ts_gemm writes for the same rows, bit for bit (node_product.h);
// glam_triplet_layer_fwd_ell with x = NULL then starts at its aggregate launch.  M <= 64 (M = the layer's Cp), K <= 64.
extern "C" int glam_ts_gemm_act_node(const float* A, int K, int lda, const float* Wimg, const float* bias, int M, int64_t N, int act,
                                     float rr_lower, float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* out,
                                     float* out_drop, const void* node_pre, int node_cols, float* xw, float* a_ij, void* stream) {
    const char* fn = "glam_ts_gemm_act_node";
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "%s: N out of range", fn);
    if (!glam_ts_gemm_rrelu_supported(K, M)) return fail(GLAM_E_UNSUPPORTED, "%s: K=%d M=%d outside k_tall_x3's node-product form (K <= 64, M <= 64)", fn, K, M);
    if (!(act == 0 || act == 1 || act == 4)) return fail(GLAM_E_UNSUPPORTED, "%s: activation code %d (0 none, 1 ReLU, 4 training-mode RReLU)", fn, act);
    if (!(node_cols > 0 && (node_cols & 3) == 0 && node_cols + 8 > 64 && node_cols + 8 <= 192 && M >= 24))
        return fail(GLAM_E_UNSUPPORTED, "%s: the node product takes 56 < H*Cp <= 184 columns, a multiple of 4 (%d)", fn, node_cols);
    if (act == 4)
        GLAM_REQUIRE(rr_lower > 0.f && rr_lower <= rr_upper && drop_p >= 0.f && drop_p < 1.f, "%s: needs 0 < lower <= upper and 0 <= p < 1 (got %g, %g, %g)",
                     fn, rr_lower, rr_upper, drop_p);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(A && Wimg && out && node_pre && xw && a_ij && (act != 4 || (rng_state && rng_eff)) && (act == 4 || !out_drop), "%s: null pointer / a dropped "
                 "twin without the RReLU form", fn);
    GLAM_REQUIRE(aligned16(A) && aligned16(Wimg) && aligned16(out) && aligned16(bias) && aligned16(out_drop) && aligned16(node_pre) && aligned16(xw) &&
                     aligned16(a_ij), "%s: pointers must be 16-byte aligned", fn);
    TsArgs a{A, K, lda, nullptr, 0, 0, Wimg, bias, out, M, M, nullptr, 0, 0, (int)N};
    a.out_relu = act == 1;
    if (act == 4) {
        a.rng_state = reinterpret_cast<long long*>(rng_state);
        a.rng_eff = reinterpret_cast<long long*>(rng_eff);
        a.rr_lo = rr_lower; a.rr_hi = rr_upper; a.drop_p = drop_p; a.out_drop = out_drop;
    }
    a.node_pre = node_pre; a.node_xw = xw; a.node_a = a_ij; a.node_m1 = node_cols;
    return launch_ts_gemm(a, (hipStream_t)stream);
}

// room for two products: glam_wgrad_gemm_pair, and glam_wgrad_gemm when it splits 64 < J <= 128 into two column chunks
extern "C" size_t glam_wgrad_workspace_bytes(void) { return 2 * wgrad_workspace_floats() * sizeof(float) + 256; }

extern "C" int glam_ts_gemm_celu(const float* A, int K, int lda, int a_celu, const float* Wimg, const float* bias, float* out, int M,
                                 int ldo, const float* cgrad_src, int ld_cgrad, int64_t N, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_ts_gemm_celu: N out of range");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(A && Wimg && out, "glam_ts_gemm_celu: null pointer");
    GLAM_REQUIRE(aligned16(A) && aligned16(Wimg) && aligned16(out) && aligned16(bias) && aligned16(cgrad_src) && (ld_cgrad & 3) == 0,
                 "glam_ts_gemm_celu: pointers must be 16-byte aligned");
    TsArgs a{A, K, lda, nullptr, 0, 0, Wimg, bias, out, M, ldo, nullptr, 0, 0, (int)N, a_celu, cgrad_src, ld_cgrad};
    return launch_ts_gemm(a, (hipStream_t)stream);
}

// Two products of one kernel variant in ONE launch, every per-product option of the single entry points available to each:
//   out_x[N, M_x] = act_x(A_x)[N, K_x] @ W_x (+ bias_x) (* celu'(cgrad_x)) (+ addend_x),  act = CELU when a_celu_x
extern "C" int glam_ts_gemm_pair(const float* Aa, int Ka, int lda, int a_celu_a, const float* Wimg_a, const float* bias_a, float* out_a,
                                 int Ma, int ldo_a, const float* cgrad_a, int ld_cgrad_a, const float* addend_a, int ld_add_a,
                                 const float* Ab, int Kb, int ldb, int a_celu_b, const float* Wimg_b, const float* bias_b, float* out_b,
                                 int Mb, int ldo_b, const float* cgrad_b, int ld_cgrad_b, const float* addend_b, int ld_add_b,
                                 int64_t N, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_ts_gemm_pair: N out of range");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(Aa && Wimg_a && out_a && Ab && Wimg_b && out_b, "glam_ts_gemm_pair: null pointer");
    GLAM_REQUIRE(aligned16(Aa) && aligned16(Wimg_a) && aligned16(out_a) && aligned16(bias_a) && aligned16(cgrad_a) && aligned16(addend_a) &&
                     aligned16(Ab) && aligned16(Wimg_b) && aligned16(out_b) && aligned16(bias_b) && aligned16(cgrad_b) && aligned16(addend_b) &&
                     !(ld_cgrad_a & 3) && !(ld_cgrad_b & 3) && !(ld_add_a & 3) && !(ld_add_b & 3),
                 "glam_ts_gemm_pair: pointers must be 16-byte aligned, leading dimensions multiples of 4");
    TsArgs a{Aa, Ka, lda, nullptr, 0, 0, Wimg_a, bias_a, out_a, Ma, ldo_a, nullptr, 0, 0, (int)N, a_celu_a, cgrad_a, ld_cgrad_a, addend_a, ld_add_a};
    TsArgs b{Ab, Kb, ldb, nullptr, 0, 0, Wimg_b, bias_b, out_b, Mb, ldo_b, nullptr, 0, 0, (int)N, a_celu_b, cgrad_b, ld_cgrad_b, addend_b, ld_add_b};
    return launch_ts_gemm2(a, &b, (hipStream_t)stream);
}

// out[N, M] = A[N, K] @ W + bias + addend: the GRU backward's d_h = d_gh @ W_hh^T + (the direct z * g path), one launch
extern "C" int glam_ts_gemm_add(const float* A, int K, int lda, const float* Wimg, const float* bias, float* out, int M, int ldo,
                                const float* addend, int ld_add, int64_t N, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_ts_gemm_add: N out of range");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(A && Wimg && out && addend, "glam_ts_gemm_add: null pointer");
    GLAM_REQUIRE(aligned16(A) && aligned16(Wimg) && aligned16(out) && aligned16(bias) && aligned16(addend) && (ld_add & 3) == 0,
                 "glam_ts_gemm_add: pointers must be 16-byte aligned");
    TsArgs a{A, K, lda, nullptr, 0, 0, Wimg, bias, out, M, ldo, nullptr, 0, 0, (int)N, 0, nullptr, 0, addend, ld_add};
    return launch_ts_gemm(a, (hipStream_t)stream);
}

static int wgrad_pair_impl(const char* fn, const float* Pa, int Ia, int ldpa, int ones_a, const float* Qa, int Ja, int ldqa, int qones_a,
                           int qcelu_a, float* out_a, int si_a, int sj_a, const float* Pb, int Ib, int ldpb, int ones_b,
                           const float* Qb, int Jb, int ldqb, int qones_b, int qcelu_b, float* out_b, int si_b, int sj_b,
                           int64_t N, void* ws, size_t ws_bytes, const float* add_a, const float* add_b, hipStream_t s,
                           float* bias_a = nullptr, const float* addb_a = nullptr, float* bias_b = nullptr,
                           const float* addb_b = nullptr) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "%s: N out of range", fn);
    if (N == 0) {
        GLAM_REQUIRE(out_a && out_b, "%s: null pointer", fn);
        GLAM_REQUIRE(!add_a && !add_b, "%s: N = 0 with an addend is not supported (add on the host side)", fn);
        if (int rc = zero_product(out_a, Ia + (ones_a ? 1 : 0), Ja + (qones_a ? 1 : 0), si_a, sj_a, s)) return rc;
        return zero_product(out_b, Ib + (ones_b ? 1 : 0), Jb + (qones_b ? 1 : 0), si_b, sj_b, s);
    }
    GLAM_REQUIRE(Pa && Qa && out_a && Pb && Qb && out_b && ws, "%s: null pointer", fn);
    GLAM_REQUIRE(ws_bytes >= glam_wgrad_workspace_bytes(), "%s: workspace too small", fn);
    GLAM_REQUIRE(aligned16(Qa) && aligned16(Pa) && aligned16(Qb) && aligned16(Pb), "%s: P / Q must be 16-byte aligned", fn);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    WgArgs a{Pa, Ia, ldpa, nullptr, 0, 0, ones_a, Qa, Ja, ldqa, qones_a, (int)N, 0, partial, 0, 0, qcelu_a};
    WgArgs b{Pb, Ib, ldpb, nullptr, 0, 0, ones_b, Qb, Jb, ldqb, qones_b, (int)N, 0, partial + wgrad_workspace_floats(), 0, 0, qcelu_b};
    ReduceArgs ra{};
    ra.njobs = 2;
    if (int rc = launch_wgrad_partials2(a, out_a, si_a, sj_a, &ra.job[0], b, out_b, si_b, sj_b, &ra.job[1], s)) return rc;
    ra.job[0].addend = add_a;
    ra.job[1].addend = add_b;
    ra.job[0].out_b = bias_a; ra.job[0].add_b = addb_a;
    ra.job[1].out_b = bias_b; ra.job[1].add_b = addb_b;
    return launch_final_reduce(ra, s);
}

extern "C" int glam_wgrad_gemm_pair(const float* Pa, int Ia, int ldpa, int ones_a, const float* Qa, int Ja, int ldqa, int qones_a,
                                    int qcelu_a, float* out_a, int si_a, int sj_a, const float* Pb, int Ib, int ldpb, int ones_b,
                                    const float* Qb, int Jb, int ldqb, int qones_b, int qcelu_b, float* out_b, int si_b, int sj_b,
                                    int64_t N, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_pair_impl("glam_wgrad_gemm_pair", Pa, Ia, ldpa, ones_a, Qa, Ja, ldqa, qones_a, qcelu_a, out_a, si_a, sj_a, Pb, Ib, ldpb,
                           ones_b, Qb, Jb, ldqb, qones_b, qcelu_b, out_b, si_b, sj_b, N, ws, ws_bytes, nullptr, nullptr, (hipStream_t)stream);
}

// [d_W | d_b] of two linears y = [x | 1] W^T in one launch + one reduction, weights and biases into SEPARATE contiguous tensors:
// dw_x[I, J] = P_x^T Q_x, db_x[I] = column sums of P_x (the ones column of [Q | 1]); optional addends laid out like the outputs.
extern "C" int glam_wgrad_gemm_pair_split(const float* Pa, int Ia, int ldpa, const float* Qa, int Ja, int ldqa, int qcelu_a, float* dw_a,
                                          float* db_a, const float* Pb, int Ib, int ldpb, const float* Qb, int Jb, int ldqb, int qcelu_b,
                                          float* dw_b, float* db_b, int64_t N, void* ws, size_t ws_bytes, const float* add_w_a,
                                          const float* add_b_a, const float* add_w_b, const float* add_b_b, void* stream) {
    GLAM_REQUIRE(dw_a && db_a && dw_b && db_b, "glam_wgrad_gemm_pair_split: null output");
    if (N == 0) {
        GLAM_REQUIRE(!add_w_a && !add_b_a && !add_w_b && !add_b_b, "glam_wgrad_gemm_pair_split: N = 0 with an addend (add on the host side)");
        hipStream_t s = (hipStream_t)stream;
        (void)hipMemsetAsync(dw_a, 0, (size_t)Ia * Ja * sizeof(float), s);
        (void)hipMemsetAsync(db_a, 0, (size_t)Ia * sizeof(float), s);
        (void)hipMemsetAsync(dw_b, 0, (size_t)Ib * Jb * sizeof(float), s);
        (void)hipMemsetAsync(db_b, 0, (size_t)Ib * sizeof(float), s);
        return GLAM_OK;
    }
    return wgrad_pair_impl("glam_wgrad_gemm_pair_split", Pa, Ia, ldpa, 0, Qa, Ja, ldqa, 1, qcelu_a, dw_a, Ja, 1, Pb, Ib, ldpb, 0, Qb, Jb,
                           ldqb, 1, qcelu_b, dw_b, Jb, 1, N, ws, ws_bytes, add_w_a, add_w_b, (hipStream_t)stream, db_a, add_b_a, db_b,
                           add_b_b);
}

// The same pair of products summed over nseg <= 3 operand sets of N rows each (the applications of a block that shares its weights:
// message_steps GRU steps, src_1gp/model.py:53-54) in ONE launch + ONE reduction.
extern "C" int glam_wgrad_gemm_pair_split_seg(int nseg, const float* const* Pa, int Ia, int ldpa, const float* const* Qa, int Ja, int ldqa,
                                              int qcelu_a, float* dw_a, float* db_a, const float* const* Pb, int Ib, int ldpb,
                                              const float* const* Qb, int Jb, int ldqb, int qcelu_b, float* dw_b, float* db_b, int64_t N,
                                              void* ws, size_t ws_bytes, const float* add_w_a, const float* add_b_a, const float* add_w_b,
                                              const float* add_b_b, void* stream) {
    const char* fn = "glam_wgrad_gemm_pair_split_seg";
    GLAM_REQUIRE(nseg >= 1 && nseg <= 3, "%s: %d operand sets (1..3)", fn, nseg);
    GLAM_REQUIRE(Pa && Qa && Pb && Qb && dw_a && db_a && dw_b && db_b && ws, "%s: null pointer", fn);
    GLAM_REQUIRE(N >= 1 && N * nseg < INT32_MAX, "%s: N out of range", fn);
    GLAM_REQUIRE(ws_bytes >= glam_wgrad_workspace_bytes(), "%s: workspace too small", fn);
    for (int q = 0; q < nseg; ++q)
        GLAM_REQUIRE(Pa[q] && Qa[q] && Pb[q] && Qb[q] && aligned16(Pa[q]) && aligned16(Qa[q]) && aligned16(Pb[q]) && aligned16(Qb[q]),
                     "%s: operand set %d: null / misaligned pointer", fn, q);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    WgArgs a{Pa[0], Ia, ldpa, nullptr, 0, 0, 0, Qa[0], Ja, ldqa, 1, (int)(N * nseg), 0, partial, 0, 0, qcelu_a};
    WgArgs b{Pb[0], Ib, ldpb, nullptr, 0, 0, 0, Qb[0], Jb, ldqb, 1, (int)(N * nseg), 0, partial + wgrad_workspace_floats(), 0, 0, qcelu_b};
    a.nseg = b.nseg = nseg;
    a.seg_rows = b.seg_rows = (int)N;
    for (int q = 1; q < nseg; ++q) {
        a.segP1[q - 1] = Pa[q]; a.segQ[q - 1] = Qa[q];
        b.segP1[q - 1] = Pb[q]; b.segQ[q - 1] = Qb[q];
    }
    ReduceArgs ra{};
    ra.njobs = 2;
    if (int rc = launch_wgrad_partials2(a, dw_a, Ja, 1, &ra.job[0], b, dw_b, Jb, 1, &ra.job[1], (hipStream_t)stream)) return rc;
    ra.job[0].addend = add_w_a;
    ra.job[1].addend = add_w_b;
    ra.job[0].out_b = db_a; ra.job[0].add_b = add_b_a;
    ra.job[1].out_b = db_b; ra.job[1].add_b = add_b_b;
    return launch_final_reduce(ra, (hipStream_t)stream);
}

// Both weight gradients of a GRU step from the ONE gate-gradient matrix glam_gru_bwd_ws writes when d_gh is null: D[n] = [d_pr | d_pz |
// d_pn | d_pn r] (4C floats per row).  d_gi = D[:, 0:3C] and d_gh = [D[:, 0:2C] | D[:, 3C:4C]] are column blocks of it (the [P1 | P2]
// operand of k_wgrad / k_wgrad_x3), so
//   dw_ih[3C, C] = d_gi^T X, db_ih = column sums of d_gi,   dw_hh[3C, C] = d_gh^T H, db_hh = column sums of d_gh
// run as glam_wgrad_gemm_pair_split_seg does (same launch, same reduction, same order of additions: bit-identical to it on the expanded
// matrices), summed over nseg <= 3 operand sets (D[s], X[s], H[s]) of N rows each; X, H with row strides ldx, ldh.
extern "C" int glam_wgrad_gemm_gru_gates_seg(int nseg, const float* const* D, int C, const float* const* X, int ldx, int qcelu,
                                             const float* const* H, int ldh, float* dw_ih, float* db_ih, float* dw_hh, float* db_hh,
                                             int64_t N, void* ws, size_t ws_bytes, const float* add_w_ih, const float* add_b_ih,
                                             const float* add_w_hh, const float* add_b_hh, void* stream) {
    const char* fn = "glam_wgrad_gemm_gru_gates_seg";
    GLAM_REQUIRE(nseg >= 1 && nseg <= 3, "%s: %d operand sets (1..3)", fn, nseg);
    GLAM_REQUIRE(C > 0 && (C & 3) == 0 && C + 1 <= 64 && ldx >= C && ldh >= C && (ldx & 3) == 0 && (ldh & 3) == 0,
                 "%s: C=%d must be a multiple of 4 below 64 (row strides %d, %d: multiples of 4, >= C)", fn, C, ldx, ldh);
    GLAM_REQUIRE(D && X && H && dw_ih && db_ih && dw_hh && db_hh, "%s: null pointer", fn);
    GLAM_REQUIRE(N >= 0 && N * nseg < INT32_MAX, "%s: N out of range", fn);
    if (N == 0) {
        GLAM_REQUIRE(!add_w_ih && !add_b_ih && !add_w_hh && !add_b_hh, "%s: N = 0 with an addend (add on the host side)", fn);
        hipStream_t s = (hipStream_t)stream;
        (void)hipMemsetAsync(dw_ih, 0, (size_t)3 * C * C * sizeof(float), s);
        (void)hipMemsetAsync(db_ih, 0, (size_t)3 * C * sizeof(float), s);
        (void)hipMemsetAsync(dw_hh, 0, (size_t)3 * C * C * sizeof(float), s);
        (void)hipMemsetAsync(db_hh, 0, (size_t)3 * C * sizeof(float), s);
        return GLAM_OK;
    }
    GLAM_REQUIRE(ws && ws_bytes >= glam_wgrad_workspace_bytes(), "%s: workspace missing / too small", fn);
    for (int q = 0; q < nseg; ++q)
        GLAM_REQUIRE(D[q] && X[q] && H[q] && aligned16(D[q]) && aligned16(X[q]) && aligned16(H[q]), "%s: operand set %d: null / misaligned pointer",
                     fn, q);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    WgArgs a{D[0], 3 * C, 4 * C, nullptr, 0, 0, 0, X[0], C, ldx, 1, (int)(N * nseg), 0, partial, 0, 0, qcelu};
    WgArgs b{D[0], 2 * C, 4 * C, D[0] + 3 * C, C, 4 * C, 0, H[0], C, ldh, 1, (int)(N * nseg), 0, partial + wgrad_workspace_floats(), 0, 0, 0};
    a.nseg = b.nseg = nseg;
    a.seg_rows = b.seg_rows = (int)N;
    for (int q = 1; q < nseg; ++q) {
        a.segP1[q - 1] = D[q]; a.segQ[q - 1] = X[q];
        b.segP1[q - 1] = D[q]; b.segP2[q - 1] = D[q] + 3 * C; b.segQ[q - 1] = H[q];
    }
    ReduceArgs ra{};
    ra.njobs = 2;
    if (int rc = launch_wgrad_partials2(a, dw_ih, C, 1, &ra.job[0], b, dw_hh, C, 1, &ra.job[1], (hipStream_t)stream)) return rc;
    ra.job[0].addend = add_w_ih;
    ra.job[1].addend = add_w_hh;
    ra.job[0].out_b = db_ih; ra.job[0].add_b = add_b_ih;
    ra.job[1].out_b = db_hh; ra.job[1].add_b = add_b_hh;
    return launch_final_reduce(ra, (hipStream_t)stream);
}

static int wgrad_gemm_impl(const float* P1, int I1, int ldp1, const float* P2, int I2, int ldp2, int ones,
                           const float* Q, int J, int ldq, int qones, int64_t N, float* out, int stride_i,
                           int stride_j, const float* addend, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_wgrad_gemm: N out of range");
    if (N == 0) {
        GLAM_REQUIRE(out, "glam_wgrad_gemm: null pointer");
        GLAM_REQUIRE(!addend, "glam_wgrad_gemm_add: N = 0 with an addend (the sum is the addend: nothing to launch)");
        return zero_product(out, I1 + I2 + (ones ? 1 : 0), J + (qones ? 1 : 0), stride_i, stride_j, (hipStream_t)stream);
    }
    GLAM_REQUIRE(P1 && Q && out && ws && (I2 == 0 || P2), "glam_wgrad_gemm: null pointer");
    GLAM_REQUIRE(ws_bytes >= glam_wgrad_workspace_bytes(), "glam_wgrad_gemm: workspace too small");
    GLAM_REQUIRE(aligned16(Q) && aligned16(P1) && aligned16(P2), "glam_wgrad_gemm: P / Q must be 16-byte aligned");
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    if (J > 64 && J + (qones ? 1 : 0) <= 128 && (J & 3) == 0) {
        // two column chunks of Q (64 | J - 64 [| 1]) as the two products of one launch: G[:, :64] and G[:, 64:]
        WgArgs a{P1, I1, ldp1, P2, I2, ldp2, ones, Q, 64, ldq, 0, (int)N, 0, partial, 0, 0};
        WgArgs b{P1, I1, ldp1, P2, I2, ldp2, ones, Q + 64, J - 64, ldq, qones, (int)N, 0, partial + wgrad_workspace_floats(), 0, 0};
        ReduceArgs rb{};
        rb.njobs = 2;
        if (int rc = launch_wgrad_partials2(a, out, stride_i, stride_j, &rb.job[0], b, out + (size_t)64 * stride_j, stride_i, stride_j,
                                            &rb.job[1], (hipStream_t)stream))
            return rc;
        if (addend) { rb.job[0].addend = addend; rb.job[1].addend = addend + (size_t)64 * stride_j; }
        return launch_final_reduce(rb, (hipStream_t)stream);
    }
    WgArgs a{P1, I1, ldp1, P2, I2, ldp2, ones, Q, J, ldq, qones, (int)N, 0, partial, 0, 0};
    ReduceArgs ra{};
    ra.njobs = 1;
    if (int rc = launch_wgrad_partials(a, out, stride_i, stride_j, (hipStream_t)stream, &ra.job[0])) return rc;
    ra.job[0].addend = addend;
    return launch_final_reduce(ra, (hipStream_t)stream);
}

extern "C" int glam_wgrad_gemm(const float* P1, int I1, int ldp1, const float* P2, int I2, int ldp2, int ones,
                               const float* Q, int J, int ldq, int qones, int64_t N, float* out, int stride_i,
                               int stride_j, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_gemm_impl(P1, I1, ldp1, P2, I2, ldp2, ones, Q, J, ldq, qones, N, out, stride_i, stride_j, nullptr, ws, ws_bytes, stream);
}

// out = the product + addend (f32, laid out like out: same strides) — the gradient carry of a weight applied several times per
// forward, summed by the reduction instead of an add launch.  N > 0.
extern "C" int glam_wgrad_gemm_add(const float* P1, int I1, int ldp1, const float* P2, int I2, int ldp2, int ones,
                                   const float* Q, int J, int ldq, int qones, int64_t N, float* out, int stride_i,
                                   int stride_j, const float* addend, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(addend, "glam_wgrad_gemm_add: null addend");
    return wgrad_gemm_impl(P1, I1, ldp1, P2, I2, ldp2, ones, Q, J, ldq, qones, N, out, stride_i, stride_j, addend, ws, ws_bytes, stream);
}

// glam_wgrad_gemm (one P block) summed over nseg <= 3 operand sets of N rows each, + an optional addend: the weight gradient of a
// tall matmul applied message_steps times with shared weights (NNConv's relation product, src_1gp/layer.py:115-122) as ONE product.
extern "C" int glam_wgrad_gemm_sets(int nseg, const float* const* P, int I, int ldp, int ones, const float* const* Q, int J, int ldq,
                                    int64_t N, float* out, int stride_i, int stride_j, const float* addend, void* ws, size_t ws_bytes,
                                    void* stream) {
    const char* fn = "glam_wgrad_gemm_sets";
    GLAM_REQUIRE(nseg >= 1 && nseg <= 3, "%s: %d operand sets (1..3)", fn, nseg);
    GLAM_REQUIRE(P && Q && out && ws, "%s: null pointer", fn);
    GLAM_REQUIRE(N >= 1 && N * nseg < INT32_MAX, "%s: N out of range", fn);
    GLAM_REQUIRE(J <= 64, "%s: J = %d > 64 (chunked products run set by set)", fn, J);
    GLAM_REQUIRE(ws_bytes >= glam_wgrad_workspace_bytes(), "%s: workspace too small", fn);
    for (int q = 0; q < nseg; ++q)
        GLAM_REQUIRE(P[q] && Q[q] && aligned16(P[q]) && aligned16(Q[q]), "%s: operand set %d: null / misaligned pointer", fn, q);
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    WgArgs a{P[0], I, ldp, nullptr, 0, 0, ones, Q[0], J, ldq, 0, (int)(N * nseg), 0, partial, 0, 0};
    a.nseg = nseg;
    a.seg_rows = (int)N;
    for (int q = 1; q < nseg; ++q) { a.segP1[q - 1] = P[q]; a.segQ[q - 1] = Q[q]; }
    ReduceArgs ra{};
    ra.njobs = 1;
    if (int rc = launch_wgrad_partials(a, out, stride_i, stride_j, (hipStream_t)stream, &ra.job[0])) return rc;
    ra.job[0].addend = addend;
    return launch_final_reduce(ra, (hipStream_t)stream);
}

// glam_wgrad_gemm in full ([P1 | P2 | 1]^T Q, any output strides, J up to 128 as two column chunks) summed over nseg <= 3 operand sets
// of N rows each: the two N-deep weight gradients of the wide TripletMessage (hid_dim_alpha = 6) over all applications of the layer.
extern "C" int glam_wgrad_gemm_sets2(int nseg, const float* const* P1, int I1, int ldp1, const float* const* P2, int I2, int ldp2, int ones,
                                     const float* const* Q, int J, int ldq, int64_t N, float* out, int stride_i, int stride_j,
                                     const float* addend, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "glam_wgrad_gemm_sets2";
    GLAM_REQUIRE(nseg >= 1 && nseg <= 3 && P1 && Q && (I2 == 0 || P2), "%s: %d operand sets (1..3) / null table", fn, nseg);
    GLAM_REQUIRE(N >= 1 && N * nseg < INT32_MAX && out && ws, "%s: N out of range / null pointer", fn);
    GLAM_REQUIRE(ws_bytes >= glam_wgrad_workspace_bytes(), "%s: workspace too small", fn);
    GLAM_REQUIRE(J > 0 && (J & 3) == 0 && J <= 128, "%s: J = %d must be a multiple of 4 up to 128", fn, J);
    for (int q = 0; q < nseg; ++q)
        GLAM_REQUIRE(P1[q] && Q[q] && (I2 == 0 || P2[q]) && aligned16(P1[q]) && aligned16(Q[q]) && (I2 == 0 || aligned16(P2[q])),
                     "%s: operand set %d: null / misaligned pointer", fn, q);
    hipStream_t s = (hipStream_t)stream;
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    auto make = [&](int qoff, int Jc, float* part) {
        WgArgs w{P1[0], I1, ldp1, I2 ? P2[0] : nullptr, I2, ldp2, ones, Q[0] + qoff, Jc, ldq, 0, (int)(N * nseg), 0, part, 0, 0};
        w.nseg = nseg; w.seg_rows = (int)N;
        for (int q = 1; q < nseg; ++q) { w.segP1[q - 1] = P1[q]; w.segP2[q - 1] = I2 ? P2[q] : nullptr; w.segQ[q - 1] = Q[q] + qoff; }
        return w;
    };
    ReduceArgs ra{};
    if (J <= 64) {
        WgArgs a = make(0, J, partial);
        ra.njobs = 1;
        if (int rc = launch_wgrad_partials(a, out, stride_i, stride_j, s, &ra.job[0])) return rc;
        ra.job[0].addend = addend;
        return launch_final_reduce(ra, s);
    }
    WgArgs a = make(0, 64, partial), b = make(64, J - 64, partial + wgrad_workspace_floats());
    ra.njobs = 2;
    if (int rc = launch_wgrad_partials2(a, out, stride_i, stride_j, &ra.job[0], b, out + (size_t)64 * stride_j, stride_i, stride_j, &ra.job[1], s))
        return rc;
    if (addend) { ra.job[0].addend = addend; ra.job[1].addend = addend + (size_t)64 * stride_j; }
    return launch_final_reduce(ra, s);
}

// [d_W | d_b] of one linear y = [act(x) | 1] W^T with up to 127 inputs: dw[I, J] = P^T act(Q) and db[I] = column sums of P in SEPARATE
// contiguous outputs (optional addends laid out like them: the gradient carry), act = CELU when q_celu (the CELU in front of a
// MessageBlock's GRU folded into the gate product, src_1gp/layer.py:261).  J <= 64: one product; beyond: the two column chunks of Q
// (64 | J - 64 | 1) as the two products of one launch.
static int wgrad_linear_impl(const char* fn, int nseg, const float* const* Ps, const float* const* Qs, int I, int ldp, int J, int ldq, int q_celu,
                             float* dw, float* db, const float* add_w, const float* add_b, int64_t N, void* ws, size_t ws_bytes, void* stream);

extern "C" int glam_wgrad_gemm_linear(const float* P, int I, int ldp, const float* Q, int J, int ldq, int q_celu, float* dw, float* db,
                                      const float* add_w, const float* add_b, int64_t N, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_linear_impl("glam_wgrad_gemm_linear", 1, &P, &Q, I, ldp, J, ldq, q_celu, dw, db, add_w, add_b, N, ws, ws_bytes, stream);
}

// ... summed over nseg <= 3 operand sets of N rows each (the applications of a block that shares the linear's weights)
extern "C" int glam_wgrad_gemm_linear_sets(int nseg, const float* const* P, int I, int ldp, const float* const* Q, int J, int ldq, int q_celu,
                                           float* dw, float* db, const float* add_w, const float* add_b, int64_t N, void* ws, size_t ws_bytes,
                                           void* stream) {
    GLAM_REQUIRE(nseg >= 1 && nseg <= 3 && P && Q, "glam_wgrad_gemm_linear_sets: %d operand sets (1..3) / null table", nseg);
    return wgrad_linear_impl("glam_wgrad_gemm_linear_sets", nseg, P, Q, I, ldp, J, ldq, q_celu, dw, db, add_w, add_b, N, ws, ws_bytes, stream);
}

static int wgrad_linear_impl(const char* fn, int nseg, const float* const* Ps, const float* const* Qs, int I, int ldp, int J, int ldq, int q_celu,
                             float* dw, float* db, const float* add_w, const float* add_b, int64_t N, void* ws, size_t ws_bytes, void* stream) {
    const float* P = Ps[0];
    const float* Q = Qs[0];
    GLAM_REQUIRE(N >= 1 && N * nseg < INT32_MAX, "%s: N out of range (N = 0: zero the outputs on the host side)", fn);
    for (int q = 0; q < nseg; ++q) GLAM_REQUIRE(Ps[q] && Qs[q] && aligned16(Ps[q]) && aligned16(Qs[q]), "%s: operand set %d: null / misaligned", fn, q);
    GLAM_REQUIRE(P && Q && dw && db && ws, "%s: null pointer", fn);
    GLAM_REQUIRE(I > 0 && J > 0 && (J & 3) == 0 && J + 1 <= 128, "%s: J = %d must be a multiple of 4 with J + 1 <= 128", fn, J);
    GLAM_REQUIRE(ws_bytes >= glam_wgrad_workspace_bytes(), "%s: workspace too small", fn);
    GLAM_REQUIRE(aligned16(P) && aligned16(Q), "%s: P / Q must be 16-byte aligned", fn);
    hipStream_t s = (hipStream_t)stream;
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    ReduceArgs ra{};
    auto sets = [&](WgArgs& w, int qoff) {
        w.nseg = nseg; w.seg_rows = (int)N;
        for (int q = 1; q < nseg; ++q) { w.segP1[q - 1] = Ps[q]; w.segQ[q - 1] = Qs[q] + qoff; }
    };
    if (J + 1 <= 64) {
        WgArgs a{P, I, ldp, nullptr, 0, 0, 0, Q, J, ldq, 1, (int)(N * nseg), 0, partial, 0, 0, q_celu};
        sets(a, 0);
        ra.njobs = 1;
        if (int rc = launch_wgrad_partials(a, dw, J, 1, s, &ra.job[0])) return rc;
        ra.job[0].addend = add_w; ra.job[0].out_b = db; ra.job[0].add_b = add_b;
        return launch_final_reduce(ra, s);
    }
    WgArgs a{P, I, ldp, nullptr, 0, 0, 0, Q, 64, ldq, 0, (int)(N * nseg), 0, partial, 0, 0, q_celu};
    WgArgs b{P, I, ldp, nullptr, 0, 0, 0, Q + 64, J - 64, ldq, 1, (int)(N * nseg), 0, partial + wgrad_workspace_floats(), 0, 0, q_celu};
    sets(a, 0);
    sets(b, 64);
    ra.njobs = 2;
    if (int rc = launch_wgrad_partials2(a, dw, J, 1, &ra.job[0], b, dw + 64, J, 1, &ra.job[1], s)) return rc;
    ra.job[0].addend = add_w;
    ra.job[1].addend = add_w ? add_w + 64 : nullptr;
    ra.job[1].out_b = db; ra.job[1].add_b = add_b;
    return launch_final_reduce(ra, s);
}

// [d_W | d_b] of one linear y = [x | 1] W^T, weight and bias into SEPARATE contiguous tensors (autograd takes them as they are; a
// strided view of a combined buffer costs a copy launch each): dw[I, J] = P^T Q, db[I] = column sums of P.  J + 1 <= 64.
static int wgrad_gemm_split(const float* P, const float* pmask, int I, int ldp, const float* Q, int J, int ldq, float* dw, float* db, int64_t N,
                            void* ws, size_t ws_bytes, void* stream);
extern "C" int glam_wgrad_gemm_split(const float* P, int I, int ldp, const float* Q, int J, int ldq, float* dw, float* db, int64_t N,
                                     void* ws, size_t ws_bytes, void* stream) {
    return wgrad_gemm_split(P, nullptr, I, ldp, Q, J, ldq, dw, db, N, ws, ws_bytes, stream);
}
// ... of a linear whose output went through a ReLU (LinearBlock + ReLU, src_1gp/layer.py:232-237): P = dy * (Y > 0), Y f32[N, I] with
// P's row stride = the saved output — the ReLU's backward inside the product instead of an elementwise launch in front of it
extern "C" int glam_wgrad_gemm_split_relu(const float* P, const float* Y, int I, int ldp, const float* Q, int J, int ldq, float* dw, float* db,
                                          int64_t N, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE((Y || N == 0) && aligned16(Y), "glam_wgrad_gemm_split_relu: null / misaligned mask");
    return wgrad_gemm_split(P, Y, I, ldp, Q, J, ldq, dw, db, N, ws, ws_bytes, stream);
}
static int wgrad_gemm_split(const float* P, const float* pmask, int I, int ldp, const float* Q, int J, int ldq, float* dw, float* db, int64_t N,
                            void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_wgrad_gemm_split: N out of range");
    GLAM_REQUIRE(dw && db, "glam_wgrad_gemm_split: null output");
    // J need not be a multiple of 4 when the rows of Q are (ldq >= ceil4(J)): the product runs over the padded width and the reduction
    // writes the J real columns of dw[I, J] only — the gradient of a weight narrower than its zero-padded input (15 -> 16 columns)
    // arrives contiguous instead of as a strided view that autograd copies
    const int Jw = J;
    J = (J + 3) & ~3;
    GLAM_REQUIRE(I > 0 && Jw > 0 && J + 1 <= 64 && ldq >= J, "glam_wgrad_gemm_split: ceil4(J) + 1 must be <= 64 and ldq >= ceil4(J)");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        (void)hipMemsetAsync(dw, 0, (size_t)I * Jw * sizeof(float), s);
        (void)hipMemsetAsync(db, 0, (size_t)I * sizeof(float), s);
        return GLAM_OK;
    }
    GLAM_REQUIRE(P && Q && ws, "glam_wgrad_gemm_split: null pointer");
    GLAM_REQUIRE(ws_bytes >= glam_wgrad_workspace_bytes(), "glam_wgrad_gemm_split: workspace too small");
    GLAM_REQUIRE(aligned16(Q) && aligned16(P), "glam_wgrad_gemm_split: P / Q must be 16-byte aligned");
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    WgArgs a{P, I, ldp, nullptr, 0, 0, 0, Q, J, ldq, 1, (int)N, 0, partial, 0, 0};
    a.pmask = pmask;
    ReduceArgs ra{};
    ra.njobs = 1;
    if (int rc = launch_wgrad_partials(a, dw, Jw, 1, s, &ra.job[0])) return rc;
    ra.job[0].out_b = db;
    ra.job[0].jw = Jw;
    return launch_final_reduce(ra, s);
}

// The four weight images of one linear pair y_a = x W_a^T, y_b = h W_b^T (W_* f32[M, K] as torch stores them) in ONE launch:
// forward images (logical [K, M] = W^T) and input-gradient images (logical [M, K] = W).
static int make_image_set(const char* fn, const float* Wa, const float* Wb, int K, int M, float* img_a_fwd, float* img_b_fwd,
                          float* img_a_bwd, float* img_b_bwd, float* gate_a, float* gate_b, void* stream) {
    GLAM_REQUIRE(Wa && Wb && img_a_fwd && img_b_fwd && img_a_bwd && img_b_bwd, "%s: null pointer", fn);
    if (int rc = ts_shape_ok(fn, K, M)) return rc;
    if (int rc = ts_shape_ok(fn, M, K)) return rc;
    ImageJobs js{};
    const float* W[4] = {Wa, Wb, Wa, Wb};
    float* img[4] = {img_a_fwd, img_b_fwd, img_a_bwd, img_b_bwd};
    int blocks = 0;
    for (int q = 0; q < 4; ++q) {
        const bool fwd = q < 2;
        const int Kq = fwd ? K : M, Mq = fwd ? M : K;
        js.job[q] = ImageJob{W[q], K, fwd ? 1 : 0, Kq, Mq, ts_mt(ts_variant(Kq, Mq)), img[q], blocks, 0};
        blocks += (int)((ts_image_floats(Kq, Mq) + kBlock - 1) / kBlock);
    }
    js.njobs = 4;
    if (gate_a) {
        GLAM_REQUIRE(gate_b && M == 3 * K && K <= 64 && (K & 3) == 0 && aligned16(gate_a) && aligned16(gate_b),
                     "%s: gate-padded images need M = 3 K, K <= 64 and a multiple of 4", fn);
        float* gimg[2] = {gate_a, gate_b};
        for (int q = 0; q < 2; ++q) {
            js.job[4 + q] = ImageJob{W[q], K, 1, K, 192, 12, gimg[q], blocks, K};
            blocks += 64 * 192 / kBlock;
        }
        js.njobs = 6;
    }
    hipLaunchKernelGGL(k_ts_make_images, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, js);
    GLAM_LAUNCH_CHECK(fn);
    return GLAM_OK;
}

extern "C" int glam_ts_gemm_make_image_quad(const float* Wa, const float* Wb, int K, int M, float* img_a_fwd, float* img_b_fwd,
                                            float* img_a_bwd, float* img_b_bwd, void* stream) {
    return make_image_set("glam_ts_gemm_make_image_quad", Wa, Wb, K, M, img_a_fwd, img_b_fwd, img_a_bwd, img_b_bwd, nullptr, nullptr, stream);
}

// The quad plus the two gate-padded images of the fused GRU step (glam_gru_fused_make_images, block.hip) in the same launch: all six
// are re-layouts of the same two gate matrices W_ih, W_hh f32[3 C, C] and change together.
extern "C" int glam_gru_make_images(const float* w_ih, const float* w_hh, int C, float* img_a_fwd, float* img_b_fwd, float* img_a_bwd,
                                    float* img_b_bwd, float* fused_ih, float* fused_hh, void* stream) {
    GLAM_REQUIRE(fused_ih && fused_hh, "glam_gru_make_images: null pointer");
    return make_image_set("glam_gru_make_images", w_ih, w_hh, C, 3 * C, img_a_fwd, img_b_fwd, img_a_bwd, img_b_bwd, fused_ih, fused_hh, stream);
}
