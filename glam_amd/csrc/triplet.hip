// Host side of the fused gather / attention-softmax / scatter-add kernels (templates in triplet_kernels.h,
// instantiated per head count in triplet_h1..h4.hip): shape checks, launch sequencing, C ABI.
#include "triplet_kernels.h"

namespace glam {

template <typename Args>
static bool dispatch(int kind, int H, int De, int emul, Shape sh, const Args& a, int nodes, size_t lds, hipStream_t s,
                     int cap, int* grid_out) {
    switch (H) {
        case 1: return triplet_launch_h1(kind, De, emul, sh, &a, nodes, lds, s, cap, grid_out);
        case 2: return triplet_launch_h2(kind, De, emul, sh, &a, nodes, lds, s, cap, grid_out);
        case 3: return triplet_launch_h3(kind, De, emul, sh, &a, nodes, lds, s, cap, grid_out);
        case 4: return triplet_launch_h4(kind, De, emul, sh, &a, nodes, lds, s, cap, grid_out);
    }
    return false;
}

// out[i] = sum_b partial[b][i] in a fixed order (deterministic).  A block owns 16 columns; its 256
// threads are 16 columns x 16 row lanes, each lane sums rows rl, rl+16, ... with 4 independent chains,
// then the 16 lanes are combined through LDS in index order.
__global__ void __launch_bounds__(kBlock) k_reduce_partials(const float* partial, int nblk, int P, int WSZ,
                                                           float* d_w_edge, float* d_M) {
    __shared__ float s_part[16][17];
    const int c = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < P) {
        int b = rl;
        for (; b + 48 < nblk; b += 64) {
            s0 += partial[(size_t)(b + 0) * P + i]; s1 += partial[(size_t)(b + 16) * P + i];
            s2 += partial[(size_t)(b + 32) * P + i]; s3 += partial[(size_t)(b + 48) * P + i];
        }
        for (; b < nblk; b += 16) s0 += partial[(size_t)b * P + i];
    }
    s_part[rl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && i < P) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += s_part[r][c];
        if (i < WSZ) { if (d_w_edge) d_w_edge[i] = s; }
        else d_M[i - WSZ] = s;
    }
}

static int check_dims(const char* fn, int64_t N, int64_t E, int H, int Cp, int De, Shape* sh) {
    if (N < 0 || E < 0 || N > INT32_MAX || E > INT32_MAX) return fail(GLAM_E_INVALID, "%s: N/E out of range", fn);
    if (Cp <= 0 || (Cp & 3)) return fail(GLAM_E_INVALID, "%s: Cp=%d must be a positive multiple of 4", fn, Cp);
    if (De != 4 && De != 8) return fail(GLAM_E_UNSUPPORTED, "%s: De=%d (host must zero-pad edge features to 4 or 8)", fn, De);
    if (H < 1 || H > 4) return fail(GLAM_E_UNSUPPORTED, "%s: heads=%d not in 1..4", fn, H);
    // the kernels address every tensor with 32-bit byte offsets
    if ((uint64_t)N * H * Cp * 4 >= (1ull << 32) || (uint64_t)E * De * 4 >= (1ull << 32))
        return fail(GLAM_E_UNSUPPORTED, "%s: a tensor of N=%lld x H*Cp=%d (or E=%lld x De) floats exceeds 4 GiB", fn, (long long)N,
                    H * Cp, (long long)E);
    if (!pick_shape(Cp >> 2, sh)) return fail(GLAM_E_UNSUPPORTED, "%s: Cp=%d exceeds 256 channels per head", fn, Cp);
    return GLAM_OK;
}

constexpr int kFusedBlocks = 512;   // fused-GEMM variants: two blocks per CU, each keeps its weight image in LDS

}  // namespace glam

using namespace glam;

extern "C" int glam_triplet_fwd(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge,
                                const float* M, const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                int64_t N, int64_t E, int H, int Cp, int De, int emul, float slope, float* aggr,
                                float* stats, void* stream) {
    Shape sh;
    if (int rc = check_dims("glam_triplet_fwd", N, E, H, Cp, De, &sh)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(xw && a_ij && M && rowptr && aggr && stats && (E == 0 || (src && eid && edge_attr)) && (!emul || w_edge),
                 "glam_triplet_fwd: null pointer");
    GLAM_REQUIRE(aligned16(xw) && aligned16(a_ij) && aligned16(aggr) && aligned16(stats) && aligned16(edge_attr) &&
                     aligned16(w_edge), "glam_triplet_fwd: pointers must be 16-byte aligned");
    FwdArgs a{xw, a_ij, edge_attr, w_edge, M, rowptr, src, eid, (int)N, Cp, slope, aggr, stats, nullptr, nullptr, nullptr};
    const size_t lds = emul ? (size_t)De * H * Cp * sizeof(float) : 0;
    GLAM_PROF_LABEL("k_triplet_fwd");
    if (!dispatch(kTripletFwd, H, De, emul, sh, a, (int)N, lds, (hipStream_t)stream, kMaxBlocks, nullptr))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_fwd: no kernel for H=%d De=%d emul=%d", H, De, emul);
    GLAM_LAUNCH_CHECK("glam_triplet_fwd");
    return GLAM_OK;
}

extern "C" size_t glam_triplet_bwd_workspace_bytes(int64_t N, int64_t E, int H, int Cp, int De) {
    const size_t P = (size_t)De * H * Cp + (size_t)De * 4;
    return ((size_t)E * 8 + (size_t)kBwdBlocks * P) * sizeof(float) + 256;
}

namespace glam {
// Forward aggregate with the update GEMM fused in (layer.py:42-61 in one launch).  Returns
// GLAM_E_UNSUPPORTED when the shape has no fused variant (caller falls back to aggregate + k_ts_gemm).
bool triplet_fwd_can_fuse_update(int H, int Cp, int De) {
    Shape sh;
    return pick_shape(Cp >> 2, &sh) && sh.G == 16 && sh.ITER == 1 && H * Cp <= 192 && Cp <= 64 && (De == 4 || De == 8);
}
int triplet_fwd_fused_update(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge,
                             const float* M, const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t N,
                             int64_t E, int H, int Cp, int De, float slope, float* aggr, float* stats,
                             const float* img_upd, const float* bias_p, float* out, hipStream_t s) {
    Shape sh;
    if (int rc = check_dims("glam_triplet_fwd(fused)", N, E, H, Cp, De, &sh)) return rc;
    if (N == 0) return GLAM_OK;
    FwdArgs a{xw, a_ij, edge_attr, w_edge, M, rowptr, src, eid, (int)N, Cp, slope, aggr, stats, img_upd, bias_p, out};
    const size_t lds = ((size_t)De * H * Cp + 16 * (size_t)(H * Cp + 4) + 16 * 64 + (size_t)((H * Cp + 15) & ~15) * 64) * sizeof(float);
    GLAM_PROF_LABEL("k_triplet_fwd+update");
    if (!dispatch(kTripletFwd, H, De, 1, sh, a, (int)N, lds, s, kFusedBlocks, nullptr))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_fwd(fused): no kernel for H=%d De=%d", H, De);
    GLAM_LAUNCH_CHECK("glam_triplet_fwd(fused)");
    return GLAM_OK;
}

// B2 with the input-gradient GEMM fused in (same shapes as the forward's fused update)
bool triplet_bwd_can_fuse_dx(int H, int Cp, int De) { return triplet_fwd_can_fuse_update(H, Cp, De) && H * Cp + 8 <= 192; }
// B1 with d_aggr = d_out @ W_scale^T fused in as a per-tile prologue
bool triplet_bwd_can_fuse_dagg(int H, int Cp, int De) { return triplet_fwd_can_fuse_update(H, Cp, De); }

// Backward launches.  reduce_now = true: d_w_edge / d_M are final on return (3 launches);
// reduce_now = false: the B1 block partials [*nblk_out][P] are left at *partial_out for a merged reduction.
int triplet_bwd_impl(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M,
                     const float* aggr, const float* stats, const float* d_aggr, const int32_t* rowptr,
                     const int32_t* src, const int32_t* eid, const int32_t* colptr, const int32_t* dst,
                     const int32_t* eid_t, int64_t N, int64_t E, int H, int Cp, int De, int emul, float slope,
                     float* d_xw, float* d_a_ij, float* d_w_edge, float* d_M, float* d_edge_attr, void* ws,
                     size_t ws_bytes, hipStream_t s, bool reduce_now, const float** partial_out, int* nblk_out,
                     const float* img_dx, float* d_x, const float* img_dagg,
                     const float* d_out, const int32_t* ell_dst, const int32_t* ell_eid_t, int edge_onehot, const int32_t* ell_src,
                     const int32_t* ell_eid, const float* dx_addend, const float* dagg_pre) {
    Shape sh;
    if (int rc = check_dims("glam_triplet_bwd", N, E, H, Cp, De, &sh)) return rc;
    const int WSZ = emul ? De * H * Cp : 0;
    const int P = WSZ + De * 4;
    GLAM_REQUIRE(xw && a_ij && M && aggr && stats && d_aggr && rowptr && colptr && d_xw && d_a_ij && ws,
                 "glam_triplet_bwd: null pointer");
    GLAM_REQUIRE(ws_bytes >= glam_triplet_bwd_workspace_bytes(N, E, H, Cp, De), "glam_triplet_bwd: workspace too small");
    GLAM_REQUIRE(aligned16(xw) && aligned16(a_ij) && aligned16(aggr) && aligned16(stats) && aligned16(d_aggr) &&
                     aligned16(d_xw) && aligned16(d_a_ij) && aligned16(edge_attr) && aligned16(w_edge) &&
                     aligned16(d_edge_attr), "glam_triplet_bwd: pointers must be 16-byte aligned");
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255;
    float* alpha_e = reinterpret_cast<float*>(base);
    float* dpre_e = alpha_e + (size_t)E * 4;
    float* partial = dpre_e + (size_t)E * 4;

    // LDS reduction buffer of B1: one row per lane group when that fits beside W_edge (two blocks per CU), else one per wave
    const int gpb = kBlock / sh.G;
    const int red_groups = ((size_t)WSZ + (size_t)gpb * P) * sizeof(float) <= 60 * 1024 ? gpb : 4;
    // d_aggr = d_out @ W_scale^T computed tile by tile inside B1 (d_aggr is then written, not read)
    const bool fuse_dagg = img_dagg && d_out && triplet_bwd_can_fuse_dagg(H, Cp, De) && emul;
    if ((img_dagg || d_out) && !fuse_dagg) return fail(GLAM_E_UNSUPPORTED, "glam_triplet_bwd: no fused d_aggr variant for H=%d Cp=%d", H, Cp);
    int nblk = 0;
    // molecular graphs with one-hot bond features: the warp-specialised B1 (matrix waves produce the d_aggr tiles ahead of the vector waves)
    const bool b1_ws = ell_src && ell_eid && fuse_dagg && !d_edge_attr && triplet_bwd_dst_ws_supported(H, Cp, De, edge_onehot);
    const bool b2_ws = ell_dst && ell_eid_t && emul && Cp <= 64 && img_dx && d_x && triplet_bwd_src_ws_supported(H, Cp, De, edge_onehot);
    if (b1_ws) {
        if (int rc = triplet_bwd_dst_ws(xw, a_ij, edge_attr, w_edge, M, aggr, stats, d_out, img_dagg, ell_src, ell_eid, N, E, H, Cp, De,
                                        edge_onehot, slope, const_cast<float*>(d_aggr), alpha_e, dpre_e, d_a_ij, partial, &nblk, s, dagg_pre))
            return rc;
    } else {
    BwdDstArgs b1{xw, a_ij, edge_attr, w_edge, M, aggr, stats, d_aggr, rowptr, src, eid, (int)N, Cp, slope,
                  alpha_e, dpre_e, d_a_ij, partial, red_groups,
                  fuse_dagg ? img_dagg : nullptr, fuse_dagg ? d_out : nullptr, fuse_dagg ? const_cast<float*>(d_aggr) : nullptr};
    const size_t red_floats = (size_t)red_groups * P;
    const size_t img_floats = (size_t)((Cp + 15) & ~15) * (H * Cp <= 64 ? 64 : 192) + 16 * (size_t)(H * Cp + 4) + 16 * 64;   // image, d_aggr tile, A tile
    const size_t lds1 = ((size_t)WSZ + (fuse_dagg && img_floats > red_floats ? img_floats : red_floats)) * sizeof(float);
    GLAM_PROF_LABEL(fuse_dagg ? "d_aggr+k_triplet_bwd_dst" : "k_triplet_bwd_dst");
    if (!dispatch(kTripletBwdDst, H, De, emul, sh, b1, (int)N, lds1, s, kBwdBlocks, &nblk))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_bwd: no kernel for H=%d De=%d emul=%d", H, De, emul);
    GLAM_LAUNCH_CHECK("glam_triplet_bwd(B1)");
    }
    if (reduce_now) {
        hipLaunchKernelGGL(k_reduce_partials, dim3((P + 15) / 16), dim3(kBlock), 0, s, partial, nblk, P, WSZ, d_w_edge, d_M);
        GLAM_LAUNCH_CHECK("glam_triplet_bwd(reduce)");
    } else {
        *partial_out = partial;
        *nblk_out = nblk;
    }
    if (d_edge_attr) {
        BwdDeaArgs bd{xw, w_edge, M, d_aggr, alpha_e, dpre_e, rowptr, src, eid, (int)N, Cp, d_edge_attr};
        GLAM_PROF_LABEL("k_triplet_bwd_dea");
        if (!dispatch(kTripletBwdDea, H, De, emul, sh, bd, (int)N, (size_t)WSZ * sizeof(float), s, kMaxBlocks, nullptr))
            return fail(GLAM_E_UNSUPPORTED, "glam_triplet_bwd: no kernel for H=%d De=%d emul=%d", H, De, emul);
        GLAM_LAUNCH_CHECK("glam_triplet_bwd(d_edge_attr)");
    }
    // molecular graphs with one-hot bond features: B2 over ELL records by source with the d_x GEMM inside, warp-specialised
    if (b2_ws)
        return triplet_bwd_src_ws(d_aggr, alpha_e, dpre_e, edge_attr, w_edge, ell_dst, ell_eid_t, N, E, H, Cp, De, edge_onehot, d_xw,
                                  d_a_ij, img_dx, d_x, s, dx_addend);
    if (dx_addend) return fail(GLAM_E_UNSUPPORTED, "glam_triplet_bwd: a d_x addend needs the warp-specialised backward by source");
    const bool fuse_dx = img_dx && d_x && triplet_bwd_can_fuse_dx(H, Cp, De) && emul;
    if ((img_dx || d_x) && !fuse_dx) return fail(GLAM_E_UNSUPPORTED, "glam_triplet_bwd: no fused d_x variant for H=%d Cp=%d", H, Cp);
    BwdSrcArgs b2{edge_attr, w_edge, d_aggr, alpha_e, dpre_e, colptr, dst, eid_t, (int)N, Cp, d_xw, d_a_ij,
                  fuse_dx ? img_dx : nullptr, fuse_dx ? d_x : nullptr};
    const int KX = H * Cp + 8, LDT = KX + ((68 - (KX & 63)) & 63);
    const size_t lds2 = ((size_t)WSZ + (fuse_dx ? 16 * (size_t)LDT + 16 * 64 + (size_t)((KX + 15) & ~15) * 64 : 0)) * sizeof(float);
    GLAM_PROF_LABEL(fuse_dx ? "k_triplet_bwd_src+dx" : "k_triplet_bwd_src");
    if (!dispatch(kTripletBwdSrc, H, De, emul, sh, b2, (int)N, lds2, s, fuse_dx ? kFusedBlocks : kMaxBlocks, nullptr))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_bwd: no kernel for H=%d De=%d emul=%d", H, De, emul);
    GLAM_LAUNCH_CHECK("glam_triplet_bwd(B2)");
    return GLAM_OK;
}
}  // namespace glam

extern "C" int glam_triplet_bwd(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge,
                                const float* M, const float* aggr, const float* stats, const float* d_aggr,
                                const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int32_t* colptr,
                                const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E, int H, int Cp, int De,
                                int emul, float slope, float* d_xw, float* d_a_ij, float* d_w_edge, float* d_M,
                                float* d_edge_attr, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(d_M && (!emul || d_w_edge), "glam_triplet_bwd: null gradient output");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (emul) (void)hipMemsetAsync(d_w_edge, 0, (size_t)De * H * Cp * 4, s);
        (void)hipMemsetAsync(d_M, 0, (size_t)De * 16, s);
        return GLAM_OK;
    }
    return triplet_bwd_impl(xw, a_ij, edge_attr, w_edge, M, aggr, stats, d_aggr, rowptr, src, eid, colptr, dst, eid_t, N, E,
                            H, Cp, De, emul, slope, d_xw, d_a_ij, d_w_edge, d_M, d_edge_attr, ws, ws_bytes, s, true,
                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
}
