// k_tall_x3: the tall-skinny product with a LONG reduction, C[N, M <= 96] = A[N, K <= 288] @ W[K, M] (+ the k_ts_gemm epilogues), on the
// bf16 matrix cores in 3 x bf16 form (bf16x3.h: fp32 accuracy), warp-specialised like the update epilogue of k_triplet_fwd_ws.
// These are the products whose A rows are the wide side: the GRU's two input-gradient products d_gi[N, 3C] @ W_ih (src_1gp/layer.py:262
// backward; with the celu' and addend epilogues), TripletMessage.update of the shapes outside the warp-specialised layer kernels
// (src_1gp/layer.py:57-61), its input gradient, and — K <= 288 — the same products at hid_dim_alpha = 6 (src_1gp/glam.py:60), which ran
// on the library GEMM.  The fp32-MFMA form (k_ts_gemm<4, 12, 4>) spends 34.6 cycles per 16 x 16 x 4 step on the fp32 vector datapath;
// here a 16-row tile is split into bf16 (hi, mid, lo) planes ONCE by producer waves (the split is the vector work: doing it in every
// consumer would make the kernel VALU bound) and every consumer wave keeps its 16-column slice of W, split in the prologue, in registers.
//   block = P producer + NC consumer waves; LDS ring of RING tiles [3 planes][16 rows][pitch]; producers check in per slot (s_ready),
//   consumers release it (s_taken): the block's only barrier is the one after the flag initialisation.
#include "dense.h"
#include "rng.h"
#include "triplet_pipe.h"
#include "node_product.h"
#include <type_traits>

namespace glam {

struct TallArgs2 { TsArgs a, b; int first_b; };

// row pitch of a plane in bytes: 64 KS of data + padding to 40 words mod 64 (the consumers' ds_read_b128 fragment reads — lane = row | 16-byte
// k block — are then conflict free in all four 16-lane groups; triplet_pipe.h: kX3RowBytes)
__host__ __device__ constexpr int tall_pitch(int KS) { return KS <= 2 ? 160 : KS <= 6 ? 416 : 672; }
// bytes per row of an epilogue plane: 16 NC floats + 4 (the consumers' four row groups then read disjoint banks)
__host__ __device__ constexpr int tall_epi_pitch(int NC) { return (16 * NC + 4) * 4; }
__host__ __device__ constexpr int tall_tile_bytes(int KS, int NC, bool epi) { return 3 * 16 * tall_pitch(KS) + (epi ? 2 * 16 * tall_epi_pitch(NC) : 0); }
constexpr int kTallHeader = 128 + 320 * 4; // s_ready[16] | s_taken[16] | bias[320] (zero where there is none)

// KS 32-k steps (K <= 32 KS); NC consumer waves of CT 16-column tiles each (M <= 16 NC CT; consumer w owns the logical columns
// 16 CT w .. 16 CT (w + 1) - 1: a contiguous 64 CT bytes of every output row); P producer waves; MP positions per image row
// EPI: the tile also carries the epilogue's operands (celu' source rows and addend rows of the 16 x 64 output block, fp32): the producers
// fetch them with the same look-ahead as A, so the consumers' loop has no global loads at all — a load there is issued behind the previous
// tile's stores and waiting for it means waiting for them (one in-order counter): a full memory round trip per tile on the MFMA waves
template <int KS, int NC, int CT, int P, int RING, int MP, bool EPI, bool RNG = false, bool NODE = false>
__global__ void __launch_bounds__((P + NC) * 64) k_tall_x3(TallArgs2 two) {
    static_assert(!EPI || (CT == 1 && P >= NC), "the epilogue planes are 16 x 16 NC, at most one float4 chunk per producer lane");
    static_assert(!NODE || (CT == 1 && NC == 4 && P == 4 && RING == 4 && !EPI), "the node product: four consumers of 16 channels, four producers");
    constexpr int PITCH = tall_pitch(KS), PLANE = 16 * PITCH, TILE = tall_tile_bytes(KS, NC, EPI);
    constexpr int EPITCH = tall_epi_pitch(NC), EPLANE = 16 * EPITCH;   // bytes
    constexpr int QN = KS * 8;                                  // float4 chunks of a tile row (data + zero fill up to 32 KS)
    constexpr int J = (16 * QN + 64 * P - 1) / (64 * P);        // chunks per producer lane and tile
    constexpr int D = 3;                                        // tiles a producer keeps in flight (registers): the launch streams A, and
                                                                // one tile per wave in flight is ~18 KB per CU — latency bound (measured 3.1 TB/s)
    extern __shared__ __attribute__((aligned(16))) char s_tall[];
    int* s_ready = reinterpret_cast<int*>(s_tall);
    int* s_taken = s_ready + 16;
    char* s_ring = s_tall + kTallHeader;
    // NODE: the rows this launch writes feed a TripletMessage — the consumers leave every finished tile in a second ring, the producers
    // multiply it by [W_node | Wa] (node_product.h)
    int* s_xready = s_ready + 10;
    int* s_xtaken = s_taken + 10;
    char* s_xn = s_ring + RING * tall_tile_bytes(KS, NC, EPI);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool second = (int)blockIdx.x >= two.first_b;
    const TsArgs a = second ? two.b : two.a;
    const int bid = second ? (int)blockIdx.x - two.first_b : (int)blockIdx.x;
    const int nblk = second ? (int)gridDim.x - two.first_b : two.first_b;
    const int K = a.K1 + a.K2, M = a.M1 + a.M2, ntiles = (a.N + 15) >> 4;
    float* s_bias = reinterpret_cast<float*>(s_tall + 128);
    if (tid < 32) s_ready[tid] = 0;
    if (tid < 16 * NC * CT) s_bias[tid] = (a.bias && tid < a.M1) ? a.bias[tid] : 0.f;
    __syncthreads();

    if (wave < P) {
        // ---- producers: rows of A -> three bf16 planes.  Chunk idx of the tile = (row, q): k = 4 q .. 4 q + 3.  Every load is
        //      unconditional (a chunk outside the matrix reads A1[0..3] and is zeroed when it is used): D tiles of loads are in flight
        //      and the wave waits for exactly the set it is about to split ----
        auto chunk_ok = [&](int tile, int j) {
            const int idx = lane + 64 * (wave + P * j), r = idx / QN, k = (idx - r * QN) * 4;
            return idx < 16 * QN && tile < ntiles && tile * 16 + r < a.N && k < K;
        };
        auto load = [&](int tile, float4 (&v)[J]) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int idx = lane + 64 * (wave + P * j), r = idx / QN, k = (idx - r * QN) * 4;
                const size_t row = (size_t)(tile * 16 + r);
                const float* src = k < a.K1 ? a.A1 + row * a.lda1 + k : a.A2 + row * a.lda2 + (k - a.K1);
                v[j] = ld4(chunk_ok(tile, j) ? src : a.A1);
            }
        };
        // epilogue planes: lane -> (row er, columns 4 eq .. 4 eq + 3) of the 16 x 64 block
        const int er = (lane + 64 * wave) / (4 * NC), eq = (lane + 64 * wave) % (4 * NC);
        auto epi_ok = [&](int tile) { return er < 16 && tile < ntiles && tile * 16 + er < a.N && 4 * eq < a.M1; };
        auto load_epi = [&](int tile, float4 (&e)[2]) {
            const size_t row = (size_t)(tile * 16 + er);
            e[0] = ld4((a.cgrad_src && epi_ok(tile)) ? a.cgrad_src + row * a.ld_cgrad + 4 * eq : a.A1);
            e[1] = ld4((a.addend && epi_ok(tile)) ? a.addend + row * a.ld_add + 4 * eq : a.A1);
        };
        float4 buf[D][J];
        float4 ebuf[D][2];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load(bid + d * nblk, buf[d]);
            if constexpr (EPI) load_epi(bid + d * nblk, ebuf[d]);
        }
        // one tile out of register set d into its ring slot (the slot is free), and that set's next loads
        auto publish = [&](auto dc, int it) {
            constexpr int d = decltype(dc)::value;
            const int tile = bid + it * nblk, slot = it % RING;
            {
                {
                    asm volatile("" ::: "memory");
                    char* tl = s_ring + slot * TILE;
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int idx = lane + 64 * (wave + P * j), r = idx / QN, q = idx - r * QN;
                        float4 v = chunk_ok(tile, j) ? buf[d][j] : f4zero();
                        if (a.a_celu) v = celu4(v);
                        if (idx < 16 * QN) {                    // (zero fill up to 32 KS: the consumers read every k step)
                            unsigned h0, m0, l0, h1, m1, l1;
                            split2(v.x, v.y, h0, m0, l0);
                            split2(v.z, v.w, h1, m1, l1);
                            char* p = tl + r * PITCH + q * 8;
                            *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                            *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
                            *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
                        }
                    }
                    if constexpr (EPI) {
                        const bool ok = epi_ok(tile);
                        char* ep = tl + 3 * PLANE + er * EPITCH + eq * 16;
                        if (er < 16) {
                            *reinterpret_cast<float4*>(ep) = (a.cgrad_src && ok) ? ebuf[d][0] : f4zero();
                            *reinterpret_cast<float4*>(ep + EPLANE) = (a.addend && ok) ? ebuf[d][1] : f4zero();
                        }
                        load_epi(tile + D * nblk, ebuf[d]);
                    }
                    load(tile + D * nblk, buf[d]);              // this register set's next tile, D tiles ahead
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) flag_bump(s_ready + slot);
                }
            }
        };
        static_assert(D == 3, "register sets");
        // NODE (the schedule of k_gru_fwd_ws<.., NODE>, block.hip): the first tile out in front of the loop, the product's weights asked
        // for behind it, then iteration `it` multiplies finished tile it - RING before it publishes tile `it`; the last tiles behind the loop
        Bf16x3 wn[2][3];
        const int my_tiles = bid < ntiles ? (ntiles - bid + nblk - 1) / nblk : 0;
        int jn = 0;
        const NodeOut nout{a.node_xw, a.node_a, a.node_m1, a.N};
        auto node_wait = [&](int j) { while (flag_load(s_xready + (j & 3)) < NC * ((j >> 2) + 1)) __builtin_amdgcn_s_sleep(1); };
        if constexpr (NODE) {
            if (my_tiles > 0) publish(std::integral_constant<int, 0>{}, 0);
            node_load_fragments(wn, a.node_pre, wave, lane);
        }
        for (int it0 = 0; bid + it0 * nblk < ntiles; it0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int it = it0 + d, tile = bid + it * nblk;
                if (tile < ntiles && !(NODE && it == 0)) {
                    const int slot = it % RING, round = it / RING;
                    while (flag_load(s_taken + slot) < NC * round) __builtin_amdgcn_s_sleep(1);
                    if constexpr (NODE) {
                        if (it >= RING) { node_wait(jn); node_product_tile(wn, s_xn, s_xtaken, jn & 3, bid + jn * nblk, wave, lane, nout); ++jn; }
                    }
                    if (d == 0) publish(std::integral_constant<int, 0>{}, it);
                    else if (d == 1) publish(std::integral_constant<int, 1>{}, it);
                    else publish(std::integral_constant<int, 2>{}, it);
                }
            }
        }
        if constexpr (NODE) {
            for (; jn < my_tiles; ++jn) { node_wait(jn); node_product_tile(wn, s_xn, s_xtaken, jn & 3, bid + jn * nblk, wave, lane, nout); }
        }
        return;
    }

    // ---- consumers: wave w owns the logical columns 16 CT w .. 16 CT (w + 1) - 1 ----
    const int w = wave - P, c = lane & 15, kb = lane >> 4;
    const int Kp = (K + 15) & ~15;                              // rows of the weight image (zero beyond K)
    Bf16x3 wreg[KS][CT];
    float4 wraw[KS][CT][2];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int col = 16 * (CT * w + j) + c;
        int pos = (col & ~63) + (col & 3) * 16 + ((col & 63) >> 2);            // position of logical column `col` in an image row
        asm volatile("" : "+v"(pos));      // (keeps the weight loads on the consumers' side of the role branch: hoisted, the producers wait for them too)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // unconditional loads (rows beyond the image re-read its last row group and are zeroed): all 2 KS CT requests of the prologue
            // are in flight together — a load under a condition is waited for on the spot
            const int k0 = 32 * s + 8 * kb, k1 = k0 + 4;
            const float4 lo4 = ld4(a.Wimg + ((size_t)(min(k0, Kp - 4) >> 2) * MP + pos) * 4);
            const float4 hi4 = ld4(a.Wimg + ((size_t)(min(k1, Kp - 4) >> 2) * MP + pos) * 4);
            wraw[s][j][0] = lo4; wraw[s][j][1] = hi4;
        }
    }
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 32 * s + 8 * kb;
            wreg[s][j] = split8w(k0 < Kp ? wraw[s][j][0] : f4zero(), k0 + 4 < Kp ? wraw[s][j][1] : f4zero());
        }
    // W is the matrix instruction's FIRST operand (rows of the result tile = output columns) and the data tile the second: a lane of
    // the result holds data row c and the four CONSECUTIVE output columns 16 tile_j + 4 kb .. + 3 — one float4 store per column tile
    // (with the operands the other way round it held one column of four rows: four scalar stores, 4 x the store instructions)
    const int col0 = 16 * CT * w + 4 * kb;                      // this lane's first column of tile j: col0 + 16 j
    // training-mode RReLU (+ the next Dropout's twin) in the epilogue: this launch's stream position, read by the waves that draw
    Philox ph{};
    if constexpr (RNG) {      // (an instantiation of its own: as a run-time branch it cost every shape ~10 registers, and the widest one spilled)
        const long long seed = __hip_atomic_load(a.rng_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long off = __hip_atomic_load(a.rng_state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.rng_eff && blockIdx.x == 0 && tid == P * 64) { a.rng_eff[0] = seed; a.rng_eff[1] = off; }
        const long long pair[2] = {seed, off};
        ph = philox_init(pair);
    }
    const bool use_cg = EPI && a.cgrad_src && col0 < a.M1, use_ad = EPI && a.addend && col0 < a.M1;
    auto do_tile = [&](int tile, int it) {
        const int slot = it % RING, want = P * (it / RING + 1);
        const int row = 16 * tile + c;
        while (flag_load(s_ready + slot) < want) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const char* tl = s_ring + slot * TILE + c * PITCH + kb * 16;       // row c, k = 32 s + 8 kb ..
        v4f_t acc[CT];
        float4 cg = f4zero(), ad = f4zero();
        if constexpr (CT == 1) {
            // long reductions: three accumulator chains (small / middle / large partial products), every k step (the producers zero-fill)
            v4f_t acc_s = {0.f, 0.f, 0.f, 0.f}, acc_m = acc_s, acc_b = acc_s;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Bf16x3 x;
                x.hi = *reinterpret_cast<const bf16x8_t*>(tl + 64 * s);
                x.mid = *reinterpret_cast<const bf16x8_t*>(tl + PLANE + 64 * s);
                x.lo = *reinterpret_cast<const bf16x8_t*>(tl + 2 * PLANE + 64 * s);
                acc_s = mfma_x3_small(wreg[s][0], x, acc_s);
                acc_m = mfma_x3_mid(wreg[s][0], x, acc_m);
                acc_b = mfma_x3_big(wreg[s][0], x, acc_b);
            }
            if constexpr (EPI) {
                const char* ep = s_ring + slot * TILE + 3 * PLANE + c * EPITCH + (16 * w + 4 * kb) * 4;
                cg = *reinterpret_cast<const float4*>(ep);
                ad = *reinterpret_cast<const float4*>(ep + EPLANE);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_taken + slot);          // every fragment is in registers: the slot may be refilled
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[0][i] = (acc_s[i] + acc_m[i]) + acc_b[i];
        } else {
            // short reductions, several column tiles: CT independent chains — small partial products of every k step first, then the
            // middle ones, then hi x hi; the fragments are re-read per phase (18 instead of 9 LDS reads: the registers of all KS fragments
            // on top of the CT KS weight slices would not fit)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[j] = (v4f_t){0.f, 0.f, 0.f, 0.f};
            v4f_t accb[CT];
#pragma unroll
            for (int j = 0; j < CT; ++j) accb[j] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Bf16x3 x;
                x.hi = *reinterpret_cast<const bf16x8_t*>(tl + 64 * s);
                x.mid = *reinterpret_cast<const bf16x8_t*>(tl + PLANE + 64 * s);
                x.lo = *reinterpret_cast<const bf16x8_t*>(tl + 2 * PLANE + 64 * s);
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[j] = mfma_x3_small(wreg[s][j], x, acc[j]);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Bf16x3 x;
                x.hi = *reinterpret_cast<const bf16x8_t*>(tl + 64 * s);
                x.mid = *reinterpret_cast<const bf16x8_t*>(tl + PLANE + 64 * s);
                x.lo = x.mid;
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[j] = mfma_x3_mid(wreg[s][j], x, acc[j]);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Bf16x3 x;
                x.hi = *reinterpret_cast<const bf16x8_t*>(tl + 64 * s);
                x.mid = x.hi; x.lo = x.hi;
                if (s == KS - 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) flag_bump(s_taken + slot);
                }
#pragma unroll
                for (int j = 0; j < CT; ++j) accb[j] = mfma_x3_big(wreg[s][j], x, accb[j]);
            }
            // (the large products in a chain of their own: see k_ts_gemm_x3)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[j] += accb[j];
        }
        float4 xnext = f4zero();                                // NODE: the lane's piece of the rows the TripletMessage reads
        if constexpr (NODE) {
            if (it >= kNodeXRing) while (flag_load(s_xtaken + (it & 3)) < P * (it >> 2)) __builtin_amdgcn_s_sleep(1);
        }
        if (row < a.N) {
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int col = col0 + 16 * j;
                if (col >= M) continue;
                const float4 bj = *reinterpret_cast<const float4*>(s_bias + col);
                float4 v = make_float4(acc[j][0] + bj.x, acc[j][1] + bj.y, acc[j][2] + bj.z, acc[j][3] + bj.w);
                if (use_cg) { v.x *= celu1_grad(cg.x); v.y *= celu1_grad(cg.y); v.z *= celu1_grad(cg.z); v.w *= celu1_grad(cg.w); }
                if (use_ad) { v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w; }
                if (a.out_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if constexpr (RNG) {      // (the quad of the flat [N, M1] output this lane holds: the words of the stand-alone launch)
                    const size_t e = (size_t)row * a.ldo1 + col;
                    const uint4 w4 = philox4(ph, e >> 2);
                    float4 od;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned wd = philox_word(w4, q);
                        const float vq = f4get(v, q), o = vq > 0.f ? vq : vq * rrelu_slope_w(wd, a.rr_lo, a.rr_hi);
                        (&v.x)[q] = o; (&od.x)[q] = o * drop_scale_w(wd, a.drop_p);
                    }
                    if (a.out_drop) st4(a.out_drop + e, od);
                    if constexpr (NODE) { if (a.out_drop) xnext = od; else xnext = v; }
                } else if constexpr (NODE) xnext = v;
                if (col < a.M1) st4(a.out1 + (size_t)row * a.ldo1 + col, v);
                else st4(a.out2 + (size_t)row * a.ldo2 + (col - a.M1), v);
            }
        }
        if constexpr (NODE) {
            *reinterpret_cast<float4*>(s_xn + (it & 3) * kNodeXTile + c * kNodeXPitch + col0 * 4) = xnext;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_xready + (it & 3));
        }
    };
    int it = 0;
    for (int tile = bid; tile < ntiles; tile += nblk, ++it) do_tile(tile, it);
    if constexpr (RNG) {
        // the block's ticket for the stream position (rng.h): the last consumer wave to finish takes it — the producers never read the pair
        int last = 0;
        if (lane == 0) last = __hip_atomic_fetch_add(s_ready + 15, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == NC - 1;
        if (last) rng_ticket(a.rng_state, ph);
    }
}

template <int KS, int NC, int CT, int P, int RING, int MP, bool EPI, bool RNG = false, bool NODE = false>
static int launch_tall(const TallArgs2& two, int grid, hipStream_t s) {
    static bool big[64] = {};
    constexpr size_t lds = kTallHeader + (size_t)RING * tall_tile_bytes(KS, NC, EPI) + (NODE ? kNodeXBytes : 0);
    static_assert(lds <= 160 * 1024, "ring exceeds the LDS of a CU");
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_tall_x3<KS, NC, CT, P, RING, MP, EPI, RNG, NODE>), big, "tall_x3")) return rc;
    hipLaunchKernelGGL((k_tall_x3<KS, NC, CT, P, RING, MP, EPI, RNG, NODE>), dim3(grid), dim3((P + NC) * 64), lds, s, two);
    return GLAM_OK;
}

// shape classes (the weight image is the k_ts_gemm image of the same (K, M): its row length MP follows ts_variant in gemm.hip):
//   0: K <= 192, M <= 64  (MP 64)     3: K <= 288, M <= 96 (MP 128)     2: K <= 96, M <= 320 (MP 320), plain epilogue only
//   4: K <= 320, M <= 64  (MP 64)
int launch_tall_x3(const TsArgs& a, const TsArgs* b, int variant, hipStream_t s) {
    const int ntiles = (a.N + 15) / 16;
    const int g = ntiles < 256 ? ntiles : 256;                 // one block per CU and product
    TallArgs2 two{a, b ? *b : a, g};
    const int grid = b ? 2 * g : g;
    const bool epi = a.cgrad_src || a.addend || (b && (b->cgrad_src || b->addend));
    int rc;
    // the narrow layers of the search space (hid_dim 15 / 30: K = 16 .. 104) do not pay for six 32-k steps per tile
    const int Kmax = (a.K1 + a.K2) > (b ? b->K1 + b->K2 : 0) ? (a.K1 + a.K2) : (b->K1 + b->K2);
    if (a.rng_state || a.node_pre) {
        // the training-mode RReLU epilogue / the node product of the TripletMessage behind: the input embeddings (K <= 64, M <= 64), one
        // product per launch
        if (!(variant == 0 && Kmax <= 64 && !epi && !b && a.M2 == 0 && a.ldo1 == a.M1))
            return fail(GLAM_E_UNSUPPORTED, "tall_x3: the RReLU epilogue / the node product take K <= 64, M <= 64, one contiguous output");
        if (a.rng_state && a.node_pre) { GLAM_PROF_LABEL("k_tall_x3<2, 4, 1>+rrelu+node"); rc = launch_tall<2, 4, 1, 4, 4, 64, false, true, true>(two, grid, s); }
        else if (a.node_pre) { GLAM_PROF_LABEL("k_tall_x3<2, 4, 1>+node"); rc = launch_tall<2, 4, 1, 4, 4, 64, false, false, true>(two, grid, s); }
        else { GLAM_PROF_LABEL("k_tall_x3<2, 4, 1>+rrelu"); rc = launch_tall<2, 4, 1, 4, 4, 64, false, true>(two, grid, s); }
    }
    else if (variant == 0 && Kmax <= 64 && epi) { GLAM_PROF_LABEL("k_tall_x3<2, 4, 1, epi>"); rc = launch_tall<2, 4, 1, 4, 4, 64, true>(two, grid, s); }
    else if (variant == 0 && Kmax <= 64) { GLAM_PROF_LABEL("k_tall_x3<2, 4, 1>"); rc = launch_tall<2, 4, 1, 4, 4, 64, false>(two, grid, s); }
    else if (variant == 0 && Kmax <= 128 && epi) { GLAM_PROF_LABEL("k_tall_x3<4, 4, 1, epi>"); rc = launch_tall<4, 4, 1, 4, 4, 64, true>(two, grid, s); }
    else if (variant == 0 && Kmax <= 128) { GLAM_PROF_LABEL("k_tall_x3<4, 4, 1>"); rc = launch_tall<4, 4, 1, 4, 4, 64, false>(two, grid, s); }
    else if (variant == 0 && epi) { GLAM_PROF_LABEL("k_tall_x3<6, 4, 1, epi>"); rc = launch_tall<6, 4, 1, 4, 4, 64, true>(two, grid, s); }
    else if (variant == 0) { GLAM_PROF_LABEL("k_tall_x3<6, 4, 1>"); rc = launch_tall<6, 4, 1, 4, 4, 64, false>(two, grid, s); }
    else if (variant == 3 && epi) { GLAM_PROF_LABEL("k_tall_x3<9, 6, 1, epi>"); rc = launch_tall<9, 6, 1, 6, 3, 128, true>(two, grid, s); }
    else if (variant == 3) { GLAM_PROF_LABEL("k_tall_x3<9, 6, 1>"); rc = launch_tall<9, 6, 1, 6, 4, 128, false>(two, grid, s); }
    else if (variant == 4 && epi) { GLAM_PROF_LABEL("k_tall_x3<10, 4, 1, epi>"); rc = launch_tall<10, 4, 1, 4, 3, 64, true>(two, grid, s); }
    else if (variant == 4) { GLAM_PROF_LABEL("k_tall_x3<10, 4, 1>"); rc = launch_tall<10, 4, 1, 4, 4, 64, false>(two, grid, s); }
    else if (variant == 2 && !epi) { GLAM_PROF_LABEL("k_tall_x3<3, 4, 5>"); rc = launch_tall<3, 4, 5, 4, 4, 320, false>(two, grid, s); }
    else return fail(GLAM_E_UNSUPPORTED, "tall_x3: variant %d with a celu' / addend epilogue is outside the kernel table", variant);
    if (rc) return rc;
    GLAM_LAUNCH_CHECK("tall_x3");
    return GLAM_OK;
}

}  // namespace glam
