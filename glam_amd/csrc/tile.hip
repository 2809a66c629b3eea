// Molecule-tile forward of the whole TripletMessage layer (reference: src_1gp/layer.py:36-61) in ONE launch.
//
// Molecular batches are block diagonal: every edge joins two atoms of the same molecule and the atoms of a
// molecule are contiguous, so a contiguous node range cut at molecule boundaries (a "tile", <= 112 nodes /
// 512 edges, planned once per batch by glam_tile_plan) is closed under the neighbour gather.  One 8-wave block
// owns a tile and keeps everything the tile needs in the CU's 160 KB LDS:
//
//   prologue  weight image [W_node | Wa_i | Wa_j] -> LDS (global_load_lds, no VGPR staging); the tile's CSR slice
//             (local row pointers, local source ids) and its edge features -> LDS; W_edge -> LDS
//   phase A   xw|a_ij tile = x_tile @ Wcat on the fp32 matrix cores (work item = 16 rows x 64 columns, dealt
//             round-robin to the 8 waves) -> LDS tile (and global: the backward pass needs xw, a_ij)
//   phase B   gather / separable attention logits / segment softmax / weighted sum, a 16-lane group per target
//             node exactly as k_triplet_fwd, but every operand now comes from LDS (~100 cycles instead of three
//             dependent HBM/L2 round trips); the update-GEMM image streams into LDS underneath
//   phase C   out_tile = aggr_tile @ W_scale + bias on the matrix cores (work item = 16 rows x 32 columns)
//
// The arithmetic (k order of every GEMM, edge order of every segment) is the one of the general path
// (k_ts_gemm -> k_triplet_fwd), so both paths return bit-identical results; graphs whose plan fails (an
// edge-closed range larger than a tile: proteins) simply stay on the general path.
#include "dense.h"
#include "triplet_kernels.h"

namespace glam {

constexpr int kTileBlock = 512;
constexpr int kTileWaves = kTileBlock / 64;
constexpr int kTileNodes = 112;      // 7 MFMA row tiles
constexpr int kTileEdges = 512;
constexpr int kImgFloats = 64 * 192;

struct TileFwdArgs {
    const float* x; const float* edge_attr; const float* img_node; const float* img_upd; const float* we_p;
    const float* M; const float* bias_p;
    const int* rowptr; const int* nbr; const int* eid; const int* tile_ptr; int T;
    int Cp; float slope;
    float* xw; float* a_ij; float* aggr; float* stats; float* out;
};

#ifdef GLAM_TILE_PROF   // developer aid (tools/tile_prof.py): per-phase shader-clock stamps of every tile
__device__ long long g_tile_prof[8192 * 8];
#define TILE_STAMP(k) do { if (tid == 0 && t < 8192) g_tile_prof[t * 8 + (k)] = clock64(); } while (0)
#else
#define TILE_STAMP(k) do { } while (0)
#endif

// barrier that orders LDS traffic only: global stores issued before it (xw, a_ij, out: never re-read by this block)
// keep draining instead of being waited for as __syncthreads() would
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


template <int H, int DE>
__global__ void __launch_bounds__(kTileBlock) k_tile_fwd(TileFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, kq = lane >> 4;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp, XS = HC + 8;
    const int G1 = (Cp + 15) >> 4;       // 16-k groups of the node GEMM   (<= 4)
    const int NCG = (XS + 63) >> 6;      // 64-column groups of its output (<= 3)
    const int G2 = (HC + 15) >> 4;       // 16-k groups of the update GEMM (<= 12)
    float* s_img = smem;
    float* s_xw = s_img + kImgFloats;                 // [kTileNodes][XS]: xw | a_i | a_j
    float* s_we = s_xw + kTileNodes * XS;             // [DE][H][Cp]
    float* s_ea = s_we + DE * HC;                     // [kTileEdges][DE]
    int* s_src = reinterpret_cast<int*>(s_ea + kTileEdges * DE);
    int* s_rp = s_src + kTileEdges;

    for (int i = tid; i < DE * HC / 4; i += kTileBlock) st4(s_we + 4 * i, ld4(a.we_p + 4 * i));
    float Mr[DE][H];
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) Mr[k][h] = a.M[k * 4 + h];

    const int lg = tid & 15, grp = tid >> 4;
    const bool okq = lg < Q;
    const int qq = okq ? lg : 0;

    for (int t = blockIdx.x; t < a.T; t += gridDim.x) {
        const int n0 = a.tile_ptr[t], nn = a.tile_ptr[t + 1] - n0;
        if (nn <= 0) continue;
        const int nrt = (nn + 15) >> 4;
        TILE_STAMP(0);
        // ---------------- prologue ----------------
        lds_copy_async<kTileBlock>(a.img_node, s_img, G1 * 768, tid);
        // A fragments of this wave's (<= 3) node-GEMM items: in flight together with everything else of the prologue
        float4 afA[3][4];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int it = wave + u * kTileWaves;
            const int row = (it / NCG) * 16 + c;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k0 = 16 * g + 4 * kq;
                afA[u][g] = (it < nrt * NCG && row < nn && k0 < Cp) ? ld4(a.x + (size_t)(n0 + row) * Cp + k0) : f4zero();
            }
        }
        const int e0 = a.rowptr[n0], ne = a.rowptr[n0 + nn] - e0;
        for (int i = tid; i <= nn; i += kTileBlock) s_rp[i] = a.rowptr[n0 + i] - e0;
        for (int e = tid; e < ne; e += kTileBlock) {
            s_src[e] = a.nbr[e0 + e] - n0;
            const float* p = a.edge_attr + (size_t)a.eid[e0 + e] * DE;
#pragma unroll
            for (int i = 0; i < DE / 4; ++i) st4(s_ea + e * DE + 4 * i, ld4(p + 4 * i));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        TILE_STAMP(1);

        // ---------------- phase A: node GEMM ----------------
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int it = wave + u * kTileWaves;
            if (it < nrt * NCG) {
                const int rt = it / NCG, cg = it - rt * NCG;
                v4f acc[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < G1) {
                        float4 bv[4];
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) bv[tt] = ld4(s_img + ((4 * g + kq) * 192 + cg * 64 + tt * 16 + c) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
                                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(afA[u][g], j), f4get(bv[tt], j), acc[tt], 0, 0, 0);
                    }
                }
                const int m0 = cg * 64 + 4 * c;
                if (m0 < XS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = rt * 16 + kq * 4 + i;
                        if (rr >= nn) continue;
                        const float4 v = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                        st4(s_xw + rr * XS + m0, v);
                        if (m0 < HC) st4(a.xw + (size_t)(n0 + rr) * HC + m0, v);
                        else st4(a.a_ij + (size_t)(n0 + rr) * 8 + (m0 - HC), v);
                    }
                }
            }
        }
        lds_barrier();
        TILE_STAMP(2);
        lds_copy_async<kTileBlock>(a.img_upd, s_img, G2 * 256, tid);     // lands during phase B

        // ---------------- phase B: gather / softmax / aggregate, all operands in LDS ----------------
        constexpr int CH = 4;
        for (int i = grp; i < nn; i += kTileBlock / 16) {
            const int beg = s_rp[i], end = s_rp[i + 1];
            const float4 aiv = ld4(s_xw + i * XS + HC);
            float ai[H], m[H], ssum[H];
            float4 acc[H];
#pragma unroll
            for (int h = 0; h < H; ++h) { ai[h] = f4get(aiv, h); m[h] = -INFINITY; ssum[h] = 0.f; acc[h] = f4zero(); }
            int sidx[CH], eloc[CH];
            bool val[CH];
            float ea[CH][DE], lk[CH][H];
            auto load_idx = [&](int eb) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    val[k] = eb + k < end;
                    eloc[k] = val[k] ? eb + k : end - 1;
                    sidx[k] = s_src[eloc[k]];
                }
            };
            auto load_logits = [&]() {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
#pragma unroll
                    for (int u = 0; u < DE / 4; ++u) {
                        const float4 v = ld4(s_ea + eloc[k] * DE + 4 * u);
                        ea[k][4 * u] = v.x; ea[k][4 * u + 1] = v.y; ea[k][4 * u + 2] = v.z; ea[k][4 * u + 3] = v.w;
                    }
                    const float4 aj = ld4(s_xw + sidx[k] * XS + HC + 4);
                    float pre[H];
                    edge_pre<H, DE>(ai, aj, ea[k], Mr, pre);
#pragma unroll
                    for (int h = 0; h < H; ++h) lk[k][h] = leaky(pre[h], a.slope);
                }
            };
            auto take_max = [&]() {
#pragma unroll
                for (int k = 0; k < CH; ++k)
#pragma unroll
                    for (int h = 0; h < H; ++h)
                        if (val[k]) m[h] = fmaxf(m[h], lk[k][h]);
            };
            auto accumulate = [&]() {
                // one head at a time: its neighbour rows (LDS, ~100 cycles) and W_edge chunk are live for that head only
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float4 wv[DE], rows[CH];
#pragma unroll
                    for (int kk = 0; kk < DE; ++kk) wv[kk] = ld4(s_we + (kk * H + h) * Cp + qq * 4);
#pragma unroll
                    for (int k = 0; k < CH; ++k) rows[k] = ld4(s_xw + sidx[k] * XS + h * Cp + qq * 4);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (!val[k]) continue;
                        const float p = softmax_exp(lk[k][h] - m[h]);
                        ssum[h] += p;
                        float4 e4 = f4zero();
#pragma unroll
                        for (int kk = 0; kk < DE; ++kk) fma4(e4, ea[k][kk], wv[kk]);
                        fma4(acc[h], p, e4 * rows[k]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (end - beg <= CH) {
                if (end > beg) { load_idx(beg); load_logits(); take_max(); accumulate(); }
            } else {
                for (int eb = beg; eb < end; eb += CH) { load_idx(eb); load_logits(); take_max(); }
                for (int eb = beg; eb < end; eb += CH) { load_idx(eb); load_logits(); accumulate(); }
            }
            float* orow = a.aggr + (size_t)(n0 + i) * HC;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float inv = 1.f / (ssum[h] + 1e-16f);
                if (okq) st4(orow + h * Cp + qq * 4, inv * acc[h]);
            }
            if (lg == 0) {
                float4 mv = f4zero(), sv = f4zero();
                float* mp = &mv.x; float* sp = &sv.x;
#pragma unroll
                for (int h = 0; h < H; ++h) { mp[h] = (end > beg) ? m[h] : 0.f; sp[h] = ssum[h]; }
                st4(a.stats + (size_t)(n0 + i) * 8, mv);
                st4(a.stats + (size_t)(n0 + i) * 8 + 4, sv);
            }
        }
        TILE_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // aggr rows of the tile are visible to the whole block; update image is in LDS

        TILE_STAMP(4);
        // ---------------- phase C: update GEMM ----------------
        // work item = 16 rows x 32 columns (column tiles 2*half, 2*half + 1 share the A fragment); a wave has at most
        // two items and loads both A fragments (aggr rows, L2 hits) before the first MFMA
        {
            float4 afC[2][12];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = wave + u * kTileWaves;
                const int row = (it >> 1) * 16 + c;
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    const int k0 = 16 * g + 4 * kq;
                    afC[u][g] = (it < nrt * 2 && row < nn && k0 < HC) ? ld4(a.aggr + (size_t)(n0 + row) * HC + k0) : f4zero();
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = wave + u * kTileWaves;
                if (it < nrt * 2) {
                    const int rt = it >> 1, half = it & 1;
                    v4f c0 = (v4f){0.f, 0.f, 0.f, 0.f}, c1 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < 12; ++g) {
                        if (g < G2) {
                            const float4 b0 = ld4(s_img + ((4 * g + kq) * 64 + (2 * half) * 16 + c) * 4);
                            const float4 b1 = ld4(s_img + ((4 * g + kq) * 64 + (2 * half + 1) * 16 + c) * 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(afC[u][g], j), f4get(b0, j), c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(afC[u][g], j), f4get(b1, j), c1, 0, 0, 0);
                            }
                        }
                    }
                    const int col = 4 * c + 2 * half;
                    if (col < Cp) {
                        const float bb0 = a.bias_p[col], bb1 = a.bias_p[col + 1];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rr = rt * 16 + kq * 4 + i;
                            if (rr < nn) *reinterpret_cast<float2*>(a.out + (size_t)(n0 + rr) * Cp + col) = make_float2(c0[i] + bb0, c1[i] + bb1);
                        }
                    }
                }
            }
        }
        lds_barrier();        // s_img / s_rp / s_src / s_ea are rewritten by the next tile's prologue
        TILE_STAMP(5);
    }
}

static size_t tile_fwd_lds_bytes(int H, int Cp, int Dp) {
    const int HC = H * Cp, XS = HC + 8;
    return ((size_t)kImgFloats + (size_t)kTileNodes * XS + (size_t)Dp * HC + (size_t)kTileEdges * Dp) * sizeof(float) +
           ((size_t)kTileEdges + kTileNodes + 4) * sizeof(int);
}

bool tile_fwd_supported(int H, int Cp, int Dp) {
    const int XS = H * Cp + 8;
    return H >= 1 && H <= 4 && (Cp & 3) == 0 && Cp >= 4 && Cp <= 64 && XS > 64 && XS <= 192 && (Dp == 4 || Dp == 8) &&
           tile_fwd_lds_bytes(H, Cp, Dp) <= 160 * 1024;
}

template <int H, int DE>
static int tile_fwd_launch_t(const TileFwdArgs& a, size_t lds, hipStream_t s) {
    static bool attr_set = false;     // > 64 KB of dynamic LDS has to be opted into once per kernel
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_fwd<H, DE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return fail(GLAM_E_HIP, "tile_fwd: cannot enable 160 KB of LDS");
        attr_set = true;
    }
    hipLaunchKernelGGL((k_tile_fwd<H, DE>), dim3(a.T), dim3(kTileBlock), lds, s, a);
    GLAM_LAUNCH_CHECK("tile_fwd");
    return GLAM_OK;
}

int tile_fwd_launch(const float* x, const float* edge_attr, const float* img_node, const float* img_upd, const float* we_p,
                    const float* M, const float* bias_p, const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                    const int32_t* tile_ptr, int T, int H, int Cp, int Dp, float slope, float* xw, float* a_ij, float* aggr,
                    float* stats, float* out, hipStream_t s) {
    if (!tile_fwd_supported(H, Cp, Dp)) return fail(GLAM_E_UNSUPPORTED, "tile_fwd: H=%d Cp=%d Dp=%d", H, Cp, Dp);
    if (T <= 0) return GLAM_OK;
    TileFwdArgs a{x, edge_attr, img_node, img_upd, we_p, M, bias_p, rowptr, src, eid, tile_ptr, T, Cp, slope, xw, a_ij, aggr, stats, out};
    const size_t lds = tile_fwd_lds_bytes(H, Cp, Dp);
#define GLAM_TILE_CASE(HH, DD) if (H == HH && Dp == DD) return tile_fwd_launch_t<HH, DD>(a, lds, s);
    GLAM_TILE_CASE(1, 4) GLAM_TILE_CASE(1, 8) GLAM_TILE_CASE(2, 4) GLAM_TILE_CASE(2, 8)
    GLAM_TILE_CASE(3, 4) GLAM_TILE_CASE(3, 8) GLAM_TILE_CASE(4, 4) GLAM_TILE_CASE(4, 8)
#undef GLAM_TILE_CASE
    return fail(GLAM_E_UNSUPPORTED, "tile_fwd: H=%d Dp=%d", H, Dp);
}

#ifdef GLAM_TILE_PROF
extern "C" int glam_debug_tile_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tile_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

void tile_limits(int* max_nodes, int* max_edges) { *max_nodes = kTileNodes; *max_edges = kTileEdges; }

}  // namespace glam
