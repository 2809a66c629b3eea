// MessageBlock remainder (reference: src_1gp/layer.py:261-263): one step of torch.nn.GRU(C, C) with seq_len 1.
// The two gate GEMMs run on k_ts_gemm (gemm.hip); this file holds the fused gate math and its backward.
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h
// gi = celu(x) @ W_ih^T + b_ih, gh = h @ W_hh^T + b_hh are [N, 3C] (gate order r | z | n, as torch stores them).
#include <type_traits>
#include "rng.h"
#include "dense.h"
#include "triplet_pipe.h"
#include "node_product.h"

namespace glam {

// Gate non-linearities on the hardware transcendental unit: v_exp_f32 (softmax_exp: |x| 2^-24 relative from the rounded x log2(e)
// plus 1 ulp) and v_rcp_f32 (1 ulp) — absolute error < 3e-7 on a gate in [0, 1] / [-1, 1], far inside the 1e-5 parity bar the block
// goldens run at.  The libm sequences (expf + IEEE divide, tanhf: ~380 vector instructions per element with the three gates) made the
// GRU tail kernels VALU bound: the forward tail processed 6 cycles per element per CU, the same rate as the fused kernel's epilogue.
// Forward and backward share the functions, so the recomputed gates equal the forward pass's.  -DGLAM_EXACT_EXP restores libm.
#ifdef GLAM_EXACT_EXP
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float tanh_(float v) { return tanhf(v); }
#else
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.f + softmax_exp(-v)); }
__device__ __forceinline__ float tanh_(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + softmax_exp(2.f * v)); }
#endif

__global__ void __launch_bounds__(kBlock) k_gru_gates_fwd(const float* gi, const float* gh, const float* h, int N, int C,
                                                         float* h_new) {
    const size_t total = (size_t)N * C;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const size_t n = i / C, c = i % C, b = n * 3 * C + c;
        const float r = sigmoidf_(gi[b] + gh[b]);
        const float z = sigmoidf_(gi[b + C] + gh[b + C]);
        const float nn = tanh_(gi[b + 2 * C] + r * gh[b + 2 * C]);
        h_new[i] = (1.f - z) * nn + z * h[i];
    }
}

// gates are recomputed from gi / gh (cheaper than saving three [N,C] tensors)
__global__ void __launch_bounds__(kBlock) k_gru_gates_bwd(const float* gi, const float* gh, const float* h,
                                                         const float* d_hnew, int N, int C, float* d_gi, float* d_gh,
                                                         float* d_h) {
    const size_t total = (size_t)N * C;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const size_t n = i / C, c = i % C, b = n * 3 * C + c;
        const float ghn = gh[b + 2 * C];
        const float r = sigmoidf_(gi[b] + gh[b]);
        const float z = sigmoidf_(gi[b + C] + gh[b + C]);
        const float nn = tanh_(gi[b + 2 * C] + r * ghn);
        const float g = d_hnew[i];
        const float d_n = g * (1.f - z), d_z = g * (h[i] - nn);
        const float d_pn = d_n * (1.f - nn * nn);
        const float d_pr = d_pn * ghn * r * (1.f - r);
        const float d_pz = d_z * z * (1.f - z);
        d_gi[b] = d_pr; d_gi[b + C] = d_pz; d_gi[b + 2 * C] = d_pn;
        d_gh[b] = d_pr; d_gh[b + C] = d_pz; d_gh[b + 2 * C] = d_pn * r;
        d_h[i] = g * z;
    }
}

// ---- the rest of MessageBlock.forward after the GRU (layer.py:263-266) folded into the gate kernel:
//   h' = GRU gates;  y = h' + identity (res);  out = act(y),  act in {none, ReLU, LeakyReLU(slope), CELU(alpha = 1)}
//   training mode of the reference's default configuration (model.py:30-31): act = RReLU (slope ~ U(lo, hi) per element where
//   y <= 0, torch.nn.RReLU) and an optional SECOND output out_drop = Dropout(p)(out) — the input of the next message step's
//   conv (layer.py:256) — both drawn from the device-side Philox stream of rng.h; the backward kernels regenerate the numbers.
enum { kActNone = 0, kActRelu = 1, kActLeaky = 2, kActCelu = 3, kActRRelu = 4 };

struct TailRng { long long* state; long long* eff; float lo, hi, p; float* out_drop; int vec; };      // forward; vec: C % 4 == 0 and
                                                                                                        // every pointer 16-byte aligned
struct TailRngB { const long long* eff; float lo, hi, p; const float* d_out_drop; int vec; };          // backward

// One Philox word per element: the RReLU slope comes from its high 16 bits, the Dropout decision from its low 16 bits
// (independent halves of one uniform word).  Word of element i = philox4(i / 4)[i % 4], whatever thread computes it: the
// vectorised paths below draw one Philox block per FOUR consecutive elements, the scalar paths one per element.
// (rrelu_slope_w / drop_scale_w: rng.h)
__device__ __forceinline__ unsigned rng_word(const Philox& ph, size_t i) { return philox_word(philox4(ph, i >> 2), (int)(i & 3)); }
__device__ __forceinline__ float4 ld4c(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float act_fwd(float y, int act, float slope) {
    switch (act) {
        case kActRelu: return fmaxf(y, 0.f);
        case kActLeaky: return y > 0.f ? y : y * slope;
        case kActCelu: return fmaxf(y, 0.f) + fminf(0.f, expf(y) - 1.f);
        default: return y;
    }
}
// d act / d y from the OUTPUT (out > 0 <=> y > 0 for all three; CELU: exp(y) = out + 1 on the negative side)
// (selects, not a switch: as a switch on the wave-uniform `act` every element of an unrolled epilogue became four small basic blocks,
//  each with its own conservative wait for the loads in flight)
__device__ __forceinline__ float act_grad_from_out(float out, int act, float slope) {
    const float neg = act == kActRelu ? 0.f : act == kActLeaky ? slope : act == kActCelu ? out + 1.f : 1.f;
    return out > 0.f ? 1.f : neg;
}

template <bool RNG>
__global__ void __launch_bounds__(kBlock) k_gru_tail_fwd(const float* gi, const float* gh, const float* h, const float* identity,
                                                        int N, int C, int act, float slope, float* h_new, float* out, TailRng rg) {
    const size_t total = (size_t)N * C;
    Philox ph{};
    if constexpr (RNG) ph = rng_begin(rg.state, rg.eff);
    if constexpr (RNG) {
        if (rg.vec) {              // four consecutive channels of one row per thread: one Philox block, float4 traffic
            for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < total / 4; q += (size_t)gridDim.x * kBlock) {
                const size_t i = 4 * q, n = i / C, c = i % C, b = n * 3 * C + c;
                const float4 ir = ld4c(gi + b), iz = ld4c(gi + b + C), in = ld4c(gi + b + 2 * C);
                const float4 hr = ld4c(gh + b), hz = ld4c(gh + b + C), hnn = ld4c(gh + b + 2 * C);
                const float4 hv = ld4c(h + i), idv = identity ? ld4c(identity + i) : f4zero();
                const uint4 w4 = philox4(ph, q);
                float4 hn4, o4, od4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r = sigmoidf_(f4get(ir, j) + f4get(hr, j));
                    const float z = sigmoidf_(f4get(iz, j) + f4get(hz, j));
                    const float nn = tanh_(f4get(in, j) + r * f4get(hnn, j));
                    const float hn = (1.f - z) * nn + z * f4get(hv, j);
                    const float y = hn + f4get(idv, j);
                    const unsigned w = philox_word(w4, j);
                    const float o = act == kActRRelu ? (y > 0.f ? y : y * rrelu_slope_w(w, rg.lo, rg.hi)) : act_fwd(y, act, slope);
                    (&hn4.x)[j] = hn; (&o4.x)[j] = o; (&od4.x)[j] = o * drop_scale_w(w, rg.p);
                }
                st4(h_new + i, hn4);
                st4(out + i, o4);
                if (rg.out_drop) st4(rg.out_drop + i, od4);
            }
            rng_end(rg.state, ph);
            return;
        }
    }
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const size_t n = i / C, c = i % C, b = n * 3 * C + c;
        const float r = sigmoidf_(gi[b] + gh[b]);
        const float z = sigmoidf_(gi[b + C] + gh[b + C]);
        const float nn = tanh_(gi[b + 2 * C] + r * gh[b + 2 * C]);
        const float hn = (1.f - z) * nn + z * h[i];
        h_new[i] = hn;
        const float y = identity ? hn + identity[i] : hn;
        float o;
        unsigned w = 0u;
        if constexpr (RNG) w = rng_word(ph, i);
        if (RNG && act == kActRRelu) o = y > 0.f ? y : y * rrelu_slope_w(w, rg.lo, rg.hi);
        else o = act_fwd(y, act, slope);
        out[i] = o;
        if constexpr (RNG) { if (rg.out_drop) rg.out_drop[i] = o * drop_scale_w(w, rg.p); }
    }
    if constexpr (RNG) rng_end(rg.state, ph);
}

// d_out: gradient of the block output; d_hstate (may be null): gradient arriving at h' through the next step's GRU
template <bool RNG>
__global__ void __launch_bounds__(kBlock) k_gru_tail_bwd(const float* gi, const float* gh, const float* h, const float* out,
                                                        const float* d_out, const float* d_hstate, int N, int C, int act,
                                                        float slope, float* d_gi, float* d_gh, float* d_h, float* d_identity,
                                                        TailRngB rg) {
    const size_t total = (size_t)N * C;
    Philox ph{};
    if constexpr (RNG) ph = philox_init(rg.eff);
    if (rg.vec) {
        // four consecutive channels of one row per thread: every load first (10 x 16 bytes), then 8 float4 stores (the scalar loop
        // below issues 4 x the instructions and, on its second trip, waits for the first trip's stores before it may load)
        for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < total / 4; q += (size_t)gridDim.x * kBlock) {
            const size_t i = 4 * q, n = i / C, c = i % C, b = n * 3 * C + c;
            const float4 ir = ld4c(gi + b), iz = ld4c(gi + b + C), in = ld4c(gi + b + 2 * C);
            const float4 hr = ld4c(gh + b), hz = ld4c(gh + b + C), hnn = ld4c(gh + b + 2 * C);
            const float4 hv = ld4c(h + i), ov = ld4c(out + i);
            const float4 go = d_out ? ld4c(d_out + i) : f4zero();
            const float4 gs = d_hstate ? ld4c(d_hstate + i) : f4zero();
            float4 gd = f4zero();
            if constexpr (RNG) { if (rg.d_out_drop) gd = ld4c(rg.d_out_drop + i); }
            uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (RNG) w4 = philox4(ph, q);
            float4 dy4, pr4, pz4, pn4, pnr4, dh4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dy;
                if constexpr (RNG) {
                    const unsigned w = philox_word(w4, j);
                    float g = f4get(go, j);
                    if (rg.d_out_drop) g = fmaf(f4get(gd, j), drop_scale_w(w, rg.p), g);
                    dy = g * (act == kActRRelu ? (f4get(ov, j) > 0.f ? 1.f : rrelu_slope_w(w, rg.lo, rg.hi)) : act_grad_from_out(f4get(ov, j), act, slope));
                } else {
                    dy = f4get(go, j) * act_grad_from_out(f4get(ov, j), act, slope);
                }
                const float g = d_hstate ? dy + f4get(gs, j) : dy;
                const float ghn = f4get(hnn, j);
                const float r = sigmoidf_(f4get(ir, j) + f4get(hr, j));
                const float z = sigmoidf_(f4get(iz, j) + f4get(hz, j));
                const float nn = tanh_(f4get(in, j) + r * ghn);
                const float d_n = g * (1.f - z), d_z = g * (f4get(hv, j) - nn);
                const float d_pn = d_n * (1.f - nn * nn);
                (&dy4.x)[j] = dy;
                (&pr4.x)[j] = d_pn * ghn * r * (1.f - r);
                (&pz4.x)[j] = d_z * z * (1.f - z);
                (&pn4.x)[j] = d_pn;
                (&pnr4.x)[j] = d_pn * r;
                (&dh4.x)[j] = g * z;
            }
            if (d_identity) st4(d_identity + i, dy4);
            st4(d_gi + b, pr4); st4(d_gi + b + C, pz4); st4(d_gi + b + 2 * C, pn4);
            st4(d_gh + b, pr4); st4(d_gh + b + C, pz4); st4(d_gh + b + 2 * C, pnr4);
            st4(d_h + i, dh4);
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const size_t n = i / C, c = i % C, b = n * 3 * C + c;
        float dy;
        if constexpr (RNG) {
            const unsigned w = rng_word(ph, i);
            float g = d_out ? d_out[i] : 0.f;
            if (rg.d_out_drop) g = fmaf(rg.d_out_drop[i], drop_scale_w(w, rg.p), g);
            dy = g * (act == kActRRelu ? (out[i] > 0.f ? 1.f : rrelu_slope_w(w, rg.lo, rg.hi)) : act_grad_from_out(out[i], act, slope));
        } else {
            dy = d_out[i] * act_grad_from_out(out[i], act, slope);
        }
        if (d_identity) d_identity[i] = dy;
        const float g = d_hstate ? dy + d_hstate[i] : dy;
        const float ghn = gh[b + 2 * C];
        const float r = sigmoidf_(gi[b] + gh[b]);
        const float z = sigmoidf_(gi[b + C] + gh[b + C]);
        const float nn = tanh_(gi[b + 2 * C] + r * ghn);
        const float d_n = g * (1.f - z), d_z = g * (h[i] - nn);
        const float d_pn = d_n * (1.f - nn * nn);
        const float d_pr = d_pn * ghn * r * (1.f - r);
        const float d_pz = d_z * z * (1.f - z);
        d_gi[b] = d_pr; d_gi[b + C] = d_pz; d_gi[b + 2 * C] = d_pn;
        d_gh[b] = d_pr; d_gh[b + C] = d_pz; d_gh[b + 2 * C] = d_pn * r;
        d_h[i] = g * z;
    }
}


// ---- block tail without a GRU (GCN / GAT blocks, src_1gp/layer.py:248, 263-266): out = act(y + bias + identity) ----
template <bool RNG>
__global__ void __launch_bounds__(kBlock) k_bias_res_act_fwd(const float* y, const float* bias, const float* identity, int N, int C,
                                                            int act, float slope, float* out, TailRng rg) {
    const size_t total = (size_t)N * C;
    Philox ph{};
    if constexpr (RNG) ph = rng_begin(rg.state, rg.eff);
    if constexpr (RNG) {
        if (rg.vec) {
            for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < total / 4; q += (size_t)gridDim.x * kBlock) {
                const size_t i = 4 * q;
                float4 v = ld4c(y + i);
                if (bias) { const float4 bv = ld4c(bias + i % C); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                if (identity) { const float4 iv = ld4c(identity + i); v.x += iv.x; v.y += iv.y; v.z += iv.z; v.w += iv.w; }
                const uint4 w4 = philox4(ph, q);
                float4 o4, od4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w = philox_word(w4, j);
                    const float vj = f4get(v, j);
                    const float o = act == kActRRelu ? (vj > 0.f ? vj : vj * rrelu_slope_w(w, rg.lo, rg.hi)) : act_fwd(vj, act, slope);
                    (&o4.x)[j] = o; (&od4.x)[j] = o * drop_scale_w(w, rg.p);
                }
                if (out) st4(out + i, o4);
                if (rg.out_drop) st4(rg.out_drop + i, od4);
            }
            rng_end(rg.state, ph);
            return;
        }
    }
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        float v = y[i];
        if (bias) v += bias[i % C];
        if (identity) v += identity[i];
        float o;
        unsigned w = 0u;
        if constexpr (RNG) w = rng_word(ph, i);
        if (RNG && act == kActRRelu) o = v > 0.f ? v : v * rrelu_slope_w(w, rg.lo, rg.hi);
        else o = act_fwd(v, act, slope);
        if (out) out[i] = o;
        if constexpr (RNG) { if (rg.out_drop) rg.out_drop[i] = o * drop_scale_w(w, rg.p); }
    }
    if constexpr (RNG) rng_end(rg.state, ph);
}
// `out` may be null for act == none (a pure Dropout): the activation derivative is then 1
template <bool RNG>
__global__ void __launch_bounds__(kBlock) k_bias_res_act_bwd(const float* out, const float* d_out, int N, int C, int act, float slope,
                                                            float* d_y, TailRngB rg) {
    const size_t total = (size_t)N * C;
    Philox ph{};
    if constexpr (RNG) {
        ph = philox_init(rg.eff);
        if (rg.vec) {
            for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < total / 4; q += (size_t)gridDim.x * kBlock) {
                const size_t i = 4 * q;
                const float4 g0 = d_out ? ld4c(d_out + i) : f4zero(), gd = rg.d_out_drop ? ld4c(rg.d_out_drop + i) : f4zero();
                const float4 o4 = out ? ld4c(out + i) : make_float4(1.f, 1.f, 1.f, 1.f);
                const uint4 w4 = philox4(ph, q);
                float4 r4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w = philox_word(w4, j);
                    const float g = fmaf(f4get(gd, j), drop_scale_w(w, rg.p), f4get(g0, j)), o = f4get(o4, j);
                    (&r4.x)[j] = g * (act == kActRRelu ? (o > 0.f ? 1.f : rrelu_slope_w(w, rg.lo, rg.hi)) : act_grad_from_out(o, act, slope));
                }
                st4(d_y + i, r4);
            }
            return;
        }
    }
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        if constexpr (RNG) {
            const unsigned w = rng_word(ph, i);
            float g = d_out ? d_out[i] : 0.f;
            if (rg.d_out_drop) g = fmaf(rg.d_out_drop[i], drop_scale_w(w, rg.p), g);
            const float o = out ? out[i] : 1.f;
            d_y[i] = g * (act == kActRRelu ? (o > 0.f ? 1.f : rrelu_slope_w(w, rg.lo, rg.hi)) : act_grad_from_out(o, act, slope));
        } else {
            d_y[i] = d_out[i] * act_grad_from_out(out[i], act, slope);
        }
    }
}

// ---- Set2Set readout (reference: model.py:41, PyG Set2Set(C, processing_steps = 3)): gate math of one torch.nn.LSTM
//      cell step.  gates f32[B, 4C] = W_ih q* + b_ih + W_hh h + b_hh in torch's order i | f | g | o;
//      c' = sigmoid(f) c + sigmoid(i) tanh(g),  h' = sigmoid(o) tanh(c').  The backward recomputes the gates. ----
__global__ void __launch_bounds__(kBlock) k_lstm_cell_fwd(const float* gates, const float* c_prev, int B, int C, float* h_new,
                                                         float* c_new) {
    const size_t total = (size_t)B * C;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        const size_t n = idx / C, c = idx % C, b = n * 4 * C + c;
        const float i = sigmoidf_(gates[b]), f = sigmoidf_(gates[b + C]), g = tanh_(gates[b + 2 * C]), o = sigmoidf_(gates[b + 3 * C]);
        const float cn = f * c_prev[idx] + i * g;
        c_new[idx] = cn;
        h_new[idx] = o * tanh_(cn);
    }
}

__global__ void __launch_bounds__(kBlock) k_lstm_cell_bwd(const float* gates, const float* c_prev, const float* d_h,
                                                         const float* d_c, int B, int C, float* d_gates, float* d_c_prev) {
    const size_t total = (size_t)B * C;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        const size_t n = idx / C, c = idx % C, b = n * 4 * C + c;
        const float i = sigmoidf_(gates[b]), f = sigmoidf_(gates[b + C]), g = tanh_(gates[b + 2 * C]), o = sigmoidf_(gates[b + 3 * C]);
        const float cp = c_prev[idx];
        const float tc = tanh_(f * cp + i * g);
        const float dh = d_h ? d_h[idx] : 0.f;
        const float dc = (d_c ? d_c[idx] : 0.f) + dh * o * (1.f - tc * tc);
        d_gates[b] = dc * g * i * (1.f - i);
        d_gates[b + C] = dc * cp * f * (1.f - f);
        d_gates[b + 2 * C] = dc * i * (1.f - g * g);
        d_gates[b + 3 * C] = dh * tc * o * (1.f - o);
        d_c_prev[idx] = dc * f;
    }
}

// ------------------------------------------------------------------------------------------------
// The GRU step of a MessageBlock (src_1gp/layer.py:261-266) in ONE launch: both gate linears on the fp32 matrix cores and the gate
// math, residual, activation (+ RReLU / Dropout) in their epilogue.  As separate launches the pair of gate GEMMs wrote gi and gh
// (2 x [N, 3C]) and the tail kernel read them back: ~16 + 12 us per application at B = 1024.
//   weight images: K = C (<= 64) x 192 columns with every gate padded to 64 (position p = gate*64 + t*16 + c holds channel 4c + t of
//   that gate), so lane (c, kq) of a wave ends up with the r, z, n pre-activations of the SAME four channels 4c..4c+3 for its four rows
//   kq*4 + i: the gate equations run in registers.  Work item = one 16-row tile, all three gates of both products (384 MFMAs);
//   block = 8 waves with both images (96 KB) in LDS, one block per CU.  gi and gh are still written (the backward recomputes the
//   gates from them), in the unpadded [N, 3C] layout.  Same arithmetic, same order as glam_ts_gemm + glam_gru_tail_*: bit-identical.
constexpr int kGruBlock = 512;
constexpr int kGruImgFloats = 64 * 192;

struct GruFusedArgs {
    const float* x; const float* h; const float* identity; const float* img_ih; const float* img_hh; const float* b_ih; const float* b_hh;
    float* gi; float* gh; float* h_new; float* out;      // k_gru_fwd_ws, gh null: gi is [N, 4C] = [r | z | n | gh_n], the gates themselves
                                                         // (what k_gru_bwd_ws<.., true> reads: 4C instead of 6C floats per row written
                                                         // here and read there, and no sigmoid / tanh in the backward)
    int N, C, celu_in, act; float slope;
    float* x_celu;      // k_gru_fwd_ws, may be null: celu(x) [N, C] as the producers compute it — what the backward (celu_in = 2) and the weight
                        // gradient (Q without its CELU) then read INSTEAD of x: no exponential in either
    const void* pre;    // k_gru_fwd_ws, may be null: both gate matrices as the consumers' 3 x bf16 fragments (k_gru_ws_pre) — img_ih / img_hh unused
    // k_gru_fwd_ws, xw non-null: the block is applied again (src_1gp/model.py:53-54) and this step's output rows ARE the next application's
    // conv input: the producers multiply every finished tile by [W_node | Wa] (node_pre: the matrix as their pre-split operand fragments,
    // layer.hip: kNodePreFloats; K = C, node_m1 + 8 <= 192 columns) before it leaves LDS and write xw[N, node_m1] | a_ij[N, 8] — the next
    // TripletMessage starts at its aggregate launch
    const void* node_pre; float* xw; float* a_ij; int node_m1;
};

__global__ void __launch_bounds__(kBlock) k_gru_fused_images(const float* w_ih, const float* w_hh, int C, float* img_ih, float* img_hh) {
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < 2 * kGruImgFloats; idx += gridDim.x * kBlock) {
        const int which = idx >= kGruImgFloats, e = idx - which * kGruImgFloats;
        const int j = e & 3, p = (e >> 2) % 192, k = (e >> 2) / 192 * 4 + j;
        const int m = ts_col_of_pos(p), gate = m >> 6, ch = m & 63;
        const float* w = which ? w_hh : w_ih;
        (which ? img_hh : img_ih)[e] = (k < C && ch < C) ? w[(size_t)(gate * C + ch) * C + k] : 0.f;
    }
}

#ifdef GLAM_GRU_PROF   // developer aid (tools/gru_prof.py): cycle stamps of wave 0 of every block
__device__ long long g_gru_prof[256 * 8];
#define GRU_STAMP(k) do { if (tid == 0 && blockIdx.x < 256) g_gru_prof[blockIdx.x * 8 + (k)] = clock64(); } while (0)
#else
#define GRU_STAMP(k) do { } while (0)
#endif

template <bool RNG>
__global__ void __launch_bounds__(kGruBlock) k_gru_fused_fwd(GruFusedArgs a, TailRng rg) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float s_img[];          // [2][64 * 192]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, kq = lane >> 4;
    const int C = a.C, GK = (C + 15) >> 4, ntiles = (a.N + 15) >> 4;
    constexpr int WPB = kGruBlock / 64;
    Philox ph{};
    if constexpr (RNG) ph = rng_begin(rg.state, rg.eff);
    // fewer tiles than wave slots: one tile per block first (a SIMD then runs one MFMA stream, not two)
    const bool spread = ntiles < (int)gridDim.x * WPB;
    const int stride = gridDim.x * WPB;
    int tile = spread ? (int)blockIdx.x + wave * (int)gridDim.x : (int)blockIdx.x * WPB + wave;
    auto load_a = [&](int t, float4 (&ax)[4], float4 (&ah)[4]) {
        const int row = t * 16 + c;
        const bool rok = t < ntiles && row < a.N;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k0 = 16 * g + 4 * kq;
            ax[g] = f4zero(); ah[g] = f4zero();
            if (rok && k0 < C) { ax[g] = ld4(a.x + (size_t)row * C + k0); ah[g] = ld4(a.h + (size_t)row * C + k0); }
        }
        if (a.celu_in) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ax[g] = celu4(ax[g]);
        }
    };
    float4 ax[4], ah[4];
    GRU_STAMP(0);
    load_a(tile, ax, ah);                      // flies while the images are staged
    {
        constexpr int kStage = 2 * kGruImgFloats / 4 / kGruBlock;       // 12 float4 per thread
        float4 buf[kStage];
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int idx = tid + i * kGruBlock;                        // float4 index over [ih | hh]
            buf[i] = idx < kGruImgFloats / 4 ? ld4(a.img_ih + 4 * idx) : ld4(a.img_hh + 4 * (idx - kGruImgFloats / 4));
        }
#pragma unroll
        for (int i = 0; i < kStage; ++i) st4(s_img + 4 * (tid + i * kGruBlock), buf[i]);
    }
    __syncthreads();
    GRU_STAMP(1);
    // The two waves of a SIMD would issue their MFMAs alternately, finish them together and then meet again in the store-bound
    // epilogue (every CU of the chip at once: the phase ran at the chip's write bandwidth).  Priority staggers them: the first
    // four waves take the matrix pipe first and store while the other four multiply.
    if (wave < 4) __builtin_amdgcn_s_setprio(3);
    int pass_no = 0;
    const float* wl_a = s_img + (kq * 192 + c) * 4;
    const float* wl_b = wl_a + kGruImgFloats;
    for (; tile < ntiles; tile += stride) {
        v4f acc_a[3][4], acc_b[3][4];
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3)
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc_a[g3][t] = (v4f){0.f, 0.f, 0.f, 0.f}; acc_b[g3][t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < GK) {
#pragma unroll
                for (int g3 = 0; g3 < 3; ++g3) {
                    float4 ba[4], bb[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        ba[t] = ld4(wl_a + g * 16 * 192 + (g3 * 64 + t * 16) * 4);
                        bb[t] = ld4(wl_b + g * 16 * 192 + (g3 * 64 + t * 16) * 4);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xa = f4get(ax[g], j), xh = f4get(ah[g], j);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            acc_a[g3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, f4get(ba[t], j), acc_a[g3][t], 0, 0, 0);
                            acc_b[g3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xh, f4get(bb[t], j), acc_b[g3][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
#ifdef GLAM_GRU_PROF
        asm volatile("s_nop 0" :: "v"(acc_a[0][0]), "v"(acc_b[2][3]));
#endif
        if (pass_no == 0) GRU_STAMP(2);
        const int cur = tile;
        // ---- epilogue: lane (c, kq) owns channels 4c..4c+3 of rows kq*4 + i.  Every load first, then nothing but arithmetic and
        //      stores: a load issued between stores makes the compiler wait for vmcnt(0), i.e. for the previous row's stores to
        //      complete (4 x ~4k cycles: the first version's epilogue took 19-23k cycles) ----
        const int ch = 4 * c;
        if (ch < C) {
            float4 bi[3], bh[3], hv[4], idv[4];
#pragma unroll
            for (int g3 = 0; g3 < 3; ++g3) { bi[g3] = ld4(a.b_ih + g3 * C + ch); bh[g3] = ld4(a.b_hh + g3 * C + ch); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = min(cur * 16 + kq * 4 + i, a.N - 1);
                const size_t e = (size_t)row * C + ch;
                hv[i] = ld4(a.h + e);
                idv[i] = a.identity ? ld4(a.identity + e) : f4zero();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = cur * 16 + kq * 4 + i;
                if (row >= a.N) continue;
                float4 gi4[3], gh4[3];
#pragma unroll
                for (int g3 = 0; g3 < 3; ++g3) {
                    gi4[g3] = make_float4(acc_a[g3][0][i] + bi[g3].x, acc_a[g3][1][i] + bi[g3].y, acc_a[g3][2][i] + bi[g3].z, acc_a[g3][3][i] + bi[g3].w);
                    gh4[g3] = make_float4(acc_b[g3][0][i] + bh[g3].x, acc_b[g3][1][i] + bh[g3].y, acc_b[g3][2][i] + bh[g3].z, acc_b[g3][3][i] + bh[g3].w);
                    st4(a.gi + (size_t)row * 3 * C + g3 * C + ch, gi4[g3]);
                    st4(a.gh + (size_t)row * 3 * C + g3 * C + ch, gh4[g3]);
                }
                const size_t e = (size_t)row * C + ch;
                uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (RNG) w4 = philox4(ph, e >> 2);
                float4 hn4, o4, od4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r = sigmoidf_(f4get(gi4[0], j) + f4get(gh4[0], j));
                    const float z = sigmoidf_(f4get(gi4[1], j) + f4get(gh4[1], j));
                    const float nn = tanh_(f4get(gi4[2], j) + r * f4get(gh4[2], j));
                    const float hn = (1.f - z) * nn + z * f4get(hv[i], j);
                    const float y = a.identity ? hn + f4get(idv[i], j) : hn;
                    const unsigned w = philox_word(w4, j);
                    const float o = (RNG && a.act == kActRRelu) ? (y > 0.f ? y : y * rrelu_slope_w(w, rg.lo, rg.hi)) : act_fwd(y, a.act, a.slope);
                    (&hn4.x)[j] = hn; (&o4.x)[j] = o;
                    if constexpr (RNG) (&od4.x)[j] = o * drop_scale_w(w, rg.p);
                }
                st4(a.h_new + e, hn4);
                st4(a.out + e, o4);
                if constexpr (RNG) { if (rg.out_drop) st4(rg.out_drop + e, od4); }
            }
        }
        load_a(tile + stride, ax, ah);         // the next tile's operands (after the stores: see above)
        if (pass_no == 0) GRU_STAMP(3);
        ++pass_no;
    }
    GRU_STAMP(4);
#ifdef GLAM_GRU_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GRU_STAMP(5);
    if constexpr (RNG) rng_end(rg.state, ph);
}


// ------------------------------------------------------------------------------------------------
// The same GRU step, warp-specialised and on the bf16 matrix cores in 3 x bf16 form (bf16x3.h: fp32 accuracy) — round 4.
// k_gru_fused_fwd above is a latency chain at B = 1024: 96 KB of weight images staged per block, then 384 fp32 matrix instructions per
// 16-row tile on the SIMD's one fp32 datapath (5.5 us per tile and wave, two waves per SIMD), then the epilogue: 22.6 us per application.
// Here (the structure of k_tall_x3, tall_x3.hip) 4 producer waves load x and h rows three tiles ahead, split [celu(x) | h] into bf16
// (hi, mid, lo) planes of an LDS tile ring — plus the fp32 h and identity rows the gate equations need — and 4 consumer waves, each bound to
// 16 channels, keep the matching 6 x (16 x 64) slices of W_ih^T / W_hh^T (r, z, n of both products) split in 144 registers: 72
// v_mfma_f32_16x16x32_bf16 per tile and wave, W as the first operand so that a lane ends up with r, z, n pre-activations of FOUR
// CONSECUTIVE channels of one row — gates, residual, activation (+ RReLU / Dropout on the same Philox words as every other path) in
// registers, float4 stores.  The weights come from the k_ts_gemm images of the gate matrices (K = C, M = 3C, 192 positions per row):
// the ones the pair launch and the backward already use — no gate-padded images.
// ------------------------------------------------------------------------------------------------
#ifdef GLAM_WS_TL      // timeline stamps of the two warp-specialised GRU kernels (tools/gru_timeline.py; see triplet_ws.hip)
__device__ long long g_gru_tl[2 * 256 * 12 * 6];      // [forward | backward][block][wave][stamp], shader clock of the CU
__device__ long long g_gru_rt[2 * 256 * 12 * 6];
__device__ long long g_gru_finef[256 * 12 * 16];     // forward, consumer tiles it == 0 (slots 0..4) and it == 2 (slots 8..12)      // the same stamps on the device-wide 100 MHz counter
// (the clock reads are volatile assembly: the compiler moves a plain clock64() across waits and loads)
#define GRU_FINEF(k) do { unsigned long long c_, r_; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_), "=s"(r_) :: "memory"); \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) g_gru_finef[(blockIdx.x * 12 + (threadIdx.x >> 6)) * 16 + (k)] = (long long)c_; } while (0)
#define GRU_TL(kid, k) do { unsigned long long c_, r_; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_), "=s"(r_) :: "memory"); \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) { const int i_ = (((kid) * 256 + blockIdx.x) * 12 + (threadIdx.x >> 6)) * 6 + (k); \
    g_gru_tl[i_] = (long long)c_; g_gru_rt[i_] = (long long)r_; } } while (0)
#else
#define GRU_TL(kid, k) do { } while (0)
#define GRU_FINEF(k) do { } while (0)
#endif
#ifndef GLAM_GRU_NODE_PRIO
#define GLAM_GRU_NODE_PRIO 0
#endif
constexpr int kGwP = 4, kGwC = 4, kGwRing = 4, kGwD = 3;
constexpr int kGwPitch = 416, kGwPlane = 16 * kGwPitch;                 // bf16 planes: 128 k of [celu(x) | h] per row (triplet_pipe.h: kX3RowBytes)
constexpr int kGwEPitch = 272, kGwEPlane = 16 * kGwEPitch;              // fp32 planes (h, identity): 64 floats + 4 per row
constexpr int kGwTile = 3 * kGwPlane + 2 * kGwEPlane;                   // 28 672 bytes
constexpr int kGwHeader = 128 + 6 * 64 * 4;                             // flags | biases [ih, hh][gate][64]
constexpr size_t kGwLds = kGwHeader + (size_t)kGwRing * kGwTile;
// ... + the finished rows on their way to the node product (node_product.h)
constexpr int kGwXPitch = kNodeXPitch, kGwXTile = kNodeXTile, kGwXRing = kNodeXRing;
constexpr size_t kGwLdsNode = kGwLds + kNodeXBytes;

template <bool RNG, bool NODE>
__global__ void __launch_bounds__((kGwP + kGwC) * 64) k_gru_fwd_ws(GruFusedArgs a, TailRng rg) {
    constexpr int P = kGwP, NC = kGwC, RING = kGwRing, D = kGwD, PITCH = kGwPitch, PLANE = kGwPlane, EPITCH = kGwEPitch, EPLANE = kGwEPlane,
                  TILE = kGwTile, MP = 192;
    extern __shared__ __attribute__((aligned(16))) char s_gw[];
    int* s_ready = reinterpret_cast<int*>(s_gw);
    int* s_taken = s_ready + 16;
    char* s_ring = s_gw + kGwHeader;
    // node product (a.xw): consumers -> producers, a second ring.  s_xready[slot]: consumer waves that wrote their 16 channels of the
    // tile, s_xtaken[slot]: producer waves that hold its fragments in registers (both count up over the block's tiles)
    constexpr int XPITCH = kGwXPitch, XTILE = kGwXTile;
    int* s_xready = s_ready + 12;
    int* s_xtaken = s_taken + 12;
    char* s_xn = s_ring + RING * TILE;
    constexpr bool node = NODE;      // (an instantiation of its own: decided at run time, every wait for a prefetched row in front of the
                                     //  branch had to assume the shorter history — the one without the node weights' 18 loads)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int C = a.C, ntiles = (a.N + 15) >> 4, bid = blockIdx.x, nblk = gridDim.x;
    Philox ph{};
    GRU_TL(0, 0);
    if constexpr (RNG) ph = rng_begin(rg.state, rg.eff);
    // One barrier, at once (it only publishes the zeroed ring flags).  What the waves load first is ORDERED: the CU has one queue into
    // memory, and the consumers' weight slices — 96 loads, 98 KB per block, ~5 k cycles to arrive (tools/ubench/slice_loads.hip) — ahead of
    // the producers' first tiles kept the ring empty until 13 k cycles into a 33 k-cycle block (tools/gru_timeline.py).  The producers go
    // first; the consumers wait for their check-in (s_ready[8]) before they issue theirs.  The biases ride in the consumers' registers.
    if (tid < 32) s_ready[tid] = 0;
    __syncthreads();
    if (wave < P) GRU_TL(0, 4);

    if (wave < P) {
        // ---- producers.  A chunk of the tile = (row r, q): q < 16 -> x[row, 4 q ..], q >= 16 -> h[row, 4 (q - 16) ..]; every load is
        //      unconditional (chunks outside the matrices read x[0..3] and are zeroed when used): D tiles of loads in flight ----
        auto chunk_ok = [&](int tile, int j) {
            const int idx = lane + 64 * (wave + P * j), r = idx >> 5, q = idx & 31;
            return tile < ntiles && tile * 16 + r < a.N && 4 * (q & 15) < C;
        };
        const int er = (lane + 64 * wave) >> 4, eq = (lane + 64 * wave) & 15;      // identity rows: one chunk per lane
        auto id_ok = [&](int tile) { return a.identity && tile < ntiles && tile * 16 + er < a.N && 4 * eq < C; };
        // the three sources as VALUES in scalar registers: with `q < 16 ? a.x : a.h` written on the argument struct the compiler selected
        // between the ADDRESSES of the two fields and loaded the pointer per lane — a dependent memory round trip and a wait for
        // everything in flight in front of every row load (590 cycles per load in the prologue, and no prefetch in the loop)
        const float* px = a.x; const float* ph_ = a.h; const float* pid = a.identity;
        asm volatile("" : "+s"(px), "+s"(ph_), "+s"(pid));
        auto load = [&](int tile, float4 (&v)[3]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = lane + 64 * (wave + P * j), r = idx >> 5, q = idx & 31;
                const size_t e = (size_t)(tile * 16 + r) * C + 4 * (q & 15);
                v[j] = ld4g(chunk_ok(tile, j) ? (q < 16 ? px : ph_) + e : px);      // (ld4g: the pinned pointers are generic to the compiler)
            }
            v[2] = ld4g(id_ok(tile) ? pid + (size_t)(tile * 16 + er) * C + 4 * eq : px);
        };
        float4 buf[D][3];
#pragma unroll
        for (int d = 0; d < D; ++d) load(bid + d * nblk, buf[d]);
        if (lane == 0) flag_bump(s_ready + 8);      // (this wave's first loads are in the queue)
        GRU_TL(0, 1);
        // ---- the node product of a finished tile: x_next[16, C] @ [W_node | Wa] — producer wave p owns output columns 48 p .. 48 p + 47
        //      (three 16-column fragments, W as the FIRST matrix operand: a lane ends up with four consecutive columns of one row).
        //      The partial products of k_ts_gemm_x3_sw in its order (there x is the first operand): the same bits as the stand-alone
        //      launch.  The weights come split already, in lane order (the staging launch wrote them: 18 coalesced 1 KB loads per wave
        //      behind the first tiles' rows; from the fp32 image — 12 scattered loads per lane in front of everything the block's other
        //      waves ask for, then the splits — every first tile went out 3.4 k cycles later) ----
        Bf16x3 wn[2][3];
        const int my_tiles = bid < ntiles ? (ntiles - bid + nblk - 1) / nblk : 0;
        int jn = 0;                                  // the next finished tile to multiply
        auto node_ready = [&](int j) { return j < my_tiles && flag_load(s_xready + (j & 3)) >= NC * ((j >> 2) + 1); };
        const NodeOut nout{a.xw, a.a_ij, a.node_m1, a.N};
        auto node_product = [&](int j) { node_product_tile(wn, s_xn, s_xtaken, j & 3, bid + j * nblk, wave, lane, nout); };
        // one tile out of register set d into ring slot it % RING (the slot is free), and that set's next loads
        auto publish = [&](auto dc, int it) {
            constexpr int d = decltype(dc)::value;
            const int tile = bid + it * nblk, slot = it % RING;
            asm volatile("" ::: "memory");
            char* tl = s_ring + slot * TILE;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = lane + 64 * (wave + P * j), r = idx >> 5, q = idx & 31;
                float4 v = chunk_ok(tile, j) ? buf[d][j] : f4zero();
                if (q >= 16) *reinterpret_cast<float4*>(tl + 3 * PLANE + r * EPITCH + (q - 16) * 16) = v;      // h, exact
                else if (a.celu_in) {
                    v = celu4(v);
                    if (a.x_celu && chunk_ok(tile, j)) st4(a.x_celu + (size_t)(tile * 16 + r) * C + 4 * q, v);
                }
                unsigned h0, m0, l0, h1, m1, l1;
                split2(v.x, v.y, h0, m0, l0);
                split2(v.z, v.w, h1, m1, l1);
                char* p = tl + r * PITCH + q * 8;
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
            }
            *reinterpret_cast<float4*>(tl + 3 * PLANE + EPLANE + er * EPITCH + eq * 16) = id_ok(tile) ? buf[d][2] : f4zero();
            load(tile + D * nblk, buf[d]);              // this register set's next tile, D tiles ahead
            // beyond the LLC three tiles of loads in flight per producer make the launch SLOWER (N = 326 400: 248 -> 280 us; the
            // pointer loads this kernel used to do by accident — see `px` above — had kept it to one): there the producer waits
            // for its prefetch
            if (a.N > 131072) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_ready + slot);
        };
        // With the node product a STATIC schedule: iteration `it` first multiplies tile it - RING (finished a little after its ring slot
        // came free: the consumers bump s_taken in front of their epilogue and s_xready behind it), then publishes tile `it`; the last RING
        // tiles follow behind the loop.  The product's three stores share the memory counter with the row prefetch: at a fixed place in
        // the iteration the compiler counts them (the wait for a register set stays "all but the newer loads and stores"); taken whenever
        // a finished tile showed up — a loop over events — every wait had to assume the shortest history, the prefetch was one tile deep
        // and a producer pass took 5.5 k cycles instead of 2.7 k
        if constexpr (NODE) {
            // the block's first tile goes out in front of the loop and the node weights are asked for BEHIND it (straight-line code: the
            // counts the waits are built from stay exact): 72 KB per block that nobody needs before the consumers are through their first
            // tile — issued with the first rows, they stood in the CU's queue in front of the consumers' own 144 KB
            if (my_tiles > 0) publish(std::integral_constant<int, 0>{}, 0);
            node_load_fragments(wn, a.node_pre, wave, lane);
        }
        for (int it0 = 0; bid + it0 * nblk < ntiles; it0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int it = it0 + d, tile = bid + it * nblk;
                if (it == 1) GRU_TL(0, 2);
                if (tile < ntiles && !(NODE && it == 0)) {
                    const int slot = it % RING, round = it / RING;
                    while (flag_load(s_taken + slot) < NC * round) __builtin_amdgcn_s_sleep(1);
                    if (node && it >= RING) {
                        while (!node_ready(jn)) __builtin_amdgcn_s_sleep(1);
                        node_product(jn);
                        ++jn;
                    }
                    static_assert(D == 3, "register sets");
                    if (d == 0) publish(std::integral_constant<int, 0>{}, it);
                    else if (d == 1) publish(std::integral_constant<int, 1>{}, it);
                    else publish(std::integral_constant<int, 2>{}, it);
                }
            }
        }
        if (node) {
            for (; jn < my_tiles; ++jn) {
                while (!node_ready(jn)) __builtin_amdgcn_s_sleep(1);
                node_product(jn);
            }
        }
        GRU_TL(0, 3);
    } else {
        // ---- consumers: wave w owns channels 16 w .. 16 w + 15; as an operand lane (c, kb) holds W column (channel) 16 w + c, k block kb;
        //      as a result lane it holds data row c, channels 16 w + 4 kb .. + 3 ----
        const int w = wave - P, c = lane & 15, kb = lane >> 4;
        const int Kp = (C + 15) & ~15;
        Bf16x3 wih[2][3], whh[2][3];
        {
            WRaw8 ri[2][3], rh[2][3];
            int chw = 16 * w + c;
            asm volatile("" : "+v"(chw));          // (keeps these loads on the consumers' side of the role branch)
            const bool okw = chw < C;
            GRU_FINEF(5);
            // (the 36 coalesced loads of the pre-split image do not hold the producers' first tile up: no wait — 0.2 us at N <= 5 120)
            if (!a.pre) while (flag_load(s_ready + 8) < P) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            GRU_FINEF(6);
            if (a.pre) {
                // the fragments as k_gru_ws_pre wrote them: split already, in lane order — 36 coalesced 1 KB loads and no vector work
                // (split here, 256 blocks each spent ~4 k cycles of their ~31 k on the same 96 values per lane: tools/gru_timeline.py)
                const char* base = reinterpret_cast<const char*>(a.pre) + (size_t)w * (36 * 1024) + lane * 16;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const char* f = base + (g * 2 + s) * (6 * 1024);
                        wih[s][g].hi = ldfrag(f); wih[s][g].mid = ldfrag(f + 1024); wih[s][g].lo = ldfrag(f + 2048);
                        whh[s][g].hi = ldfrag(f + 3072); whh[s][g].mid = ldfrag(f + 4096); whh[s][g].lo = ldfrag(f + 5120);
                    }
                GRU_TL(0, 1);
            } else {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int col = g * C + min(chw, C - 1);
                const int pos = (col & ~63) + (col & 3) * 16 + ((col >> 2) & 15);      // ts_pos_of_col
#pragma unroll
                for (int s = 0; s < 2; ++s) { ri[s][g] = w_load8(a.img_ih, MP, pos, 32 * s + 8 * kb, Kp); rh[s][g] = w_load8(a.img_hh, MP, pos, 32 * s + 8 * kb, Kp); }
            }
            GRU_TL(0, 1);
#ifdef GLAM_WS_TL
            GRU_FINEF(7);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GRU_FINEF(13);
#endif
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    wih[s][g] = w_split8(ri[s][g], 32 * s + 8 * kb, Kp, okw);
                    whh[s][g] = w_split8(rh[s][g], 32 * s + 8 * kb, Kp, okw);
                }
            }
        }
        const int ch = 16 * w + 4 * kb;
        const bool gates = a.gh == nullptr;
#if GLAM_GRU_NODE_PRIO
        if (node) __builtin_amdgcn_s_setprio(GLAM_GRU_NODE_PRIO);      // the consumers set the pace: their matrix instructions before the node product's
#endif
        float4 bias_i[3], bias_h[3];               // the lane's four channels of every gate (zero beyond C)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int cb = g * C + min(ch, C - 4);      // (unconditional loads: one under a condition is waited for on the spot)
            bias_i[g] = ld4(a.b_ih + cb);
            bias_h[g] = ld4(a.b_hh + cb);
        }
        if (ch >= C) {
#pragma unroll
            for (int g = 0; g < 3; ++g) { bias_i[g] = f4zero(); bias_h[g] = f4zero(); }
        }
        GRU_TL(0, 4);
        int it = 0;
        for (int tile = bid; tile < ntiles; tile += nblk, ++it) {
            if (it == 1) GRU_TL(0, 2);
            const int slot = it % RING, want = P * (it / RING + 1);
            const int row = 16 * tile + c;
            if (it == 0) GRU_FINEF(0); else if (it == 2) GRU_FINEF(8);
            // (the slot of the node ring this tile's rows go to: its check rides along with the wait for the tile)
            int xt = 0;
            if (node && it >= kGwXRing) xt = __hip_atomic_load(s_xtaken + (it & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (flag_load(s_ready + slot) < want) __builtin_amdgcn_s_sleep(1);
            if (node && it >= kGwXRing) {
                xt = __builtin_amdgcn_readfirstlane(xt);
                while (xt < P * (it >> 2)) { __builtin_amdgcn_s_sleep(1); xt = flag_load(s_xtaken + (it & 3)); }
            }
            asm volatile("" ::: "memory");
            if (it == 0) GRU_FINEF(1); else if (it == 2) GRU_FINEF(9);
            const char* tl = s_ring + slot * TILE + c * PITCH + kb * 16;       // row c, k = 32 s + 8 kb ..  (s = 0, 1: celu(x); 2, 3: h)
            v4f_t ai[3], ah[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) { ai[g] = (v4f_t){0.f, 0.f, 0.f, 0.f}; ah[g] = ai[g]; }
            // three passes over the tile's fragments (re-read from LDS: 12 + 8 + 4 reads): small partial products of every k step first,
            // then the middle ones, then hi x hi — six independent accumulator chains.  The reads run ONE k step ahead of the matrix
            // instructions (a consumer is alone on its SIMD's matrix pipe: an LDS round trip per step would be exposed each time)
            auto rd = [&](int s, int terms) {
                Bf16x3 x;
                x.hi = *reinterpret_cast<const bf16x8_t*>(tl + 64 * s);
                x.mid = terms > 1 ? *reinterpret_cast<const bf16x8_t*>(tl + PLANE + 64 * s) : x.hi;
                x.lo = terms > 2 ? *reinterpret_cast<const bf16x8_t*>(tl + 2 * PLANE + 64 * s) : x.mid;
                return x;
            };
            const char* ep = s_ring + slot * TILE + 3 * PLANE + c * EPITCH + ch * 4;
            Bf16x3 xa = rd(0, 3);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const Bf16x3 xb = s < 3 ? rd(s + 1, 3) : rd(0, 2);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (s < 2) ai[g] = mfma_x3_small(wih[s][g], xa, ai[g]);
                    else ah[g] = mfma_x3_small(whh[s - 2][g], xa, ah[g]);
                }
                xa = xb;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const Bf16x3 xb = s < 3 ? rd(s + 1, 2) : rd(0, 1);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (s < 2) ai[g] = mfma_x3_mid(wih[s][g], xa, ai[g]);
                    else ah[g] = mfma_x3_mid(whh[s - 2][g], xa, ah[g]);
                }
                xa = xb;
            }
            float4 hv = f4zero(), idv = f4zero();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Bf16x3 xb = xa;
                if (s < 3) xb = rd(s + 1, 1);
                else {
                    hv = *reinterpret_cast<const float4*>(ep); idv = *reinterpret_cast<const float4*>(ep + EPLANE);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) flag_bump(s_taken + slot);      // every fragment is in registers: the slot may be refilled
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (s < 2) ai[g] = mfma_x3_big(wih[s][g], xa, ai[g]);
                    else ah[g] = mfma_x3_big(whh[s - 2][g], xa, ah[g]);
                }
                xa = xb;
            }
            if (it == 0) GRU_FINEF(2); else if (it == 2) GRU_FINEF(10);
            float* const xp = reinterpret_cast<float*>(s_xn + (it & 3) * XTILE + c * XPITCH) + ch;      // the lane's piece of the next conv input
            if (row < a.N && ch < C) {
                float4 gi4[3], gh4[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float4 bi = bias_i[g], bh = bias_h[g];
                    gi4[g] = make_float4(ai[g][0] + bi.x, ai[g][1] + bi.y, ai[g][2] + bi.z, ai[g][3] + bi.w);
                    gh4[g] = make_float4(ah[g][0] + bh.x, ah[g][1] + bh.y, ah[g][2] + bh.z, ah[g][3] + bh.w);
                    if (!gates) {
                        st4(a.gi + (size_t)row * 3 * C + g * C + ch, gi4[g]);
                        st4(a.gh + (size_t)row * 3 * C + g * C + ch, gh4[g]);
                    }
                }
                if (it == 0) GRU_FINEF(3); else if (it == 2) GRU_FINEF(11);
                const size_t e = (size_t)row * C + ch;
                uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (RNG) w4 = philox4(ph, e >> 2);
                float4 hn4, o4, od4, r4, z4, n4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r = sigmoidf_(f4get(gi4[0], j) + f4get(gh4[0], j));
                    const float z = sigmoidf_(f4get(gi4[1], j) + f4get(gh4[1], j));
                    const float nn = tanh_(f4get(gi4[2], j) + r * f4get(gh4[2], j));
                    const float hn = (1.f - z) * nn + z * f4get(hv, j);
                    const float y = a.identity ? hn + f4get(idv, j) : hn;
                    const unsigned wd = philox_word(w4, j);
                    const float o = (RNG && a.act == kActRRelu) ? (y > 0.f ? y : y * rrelu_slope_w(wd, rg.lo, rg.hi)) : act_fwd(y, a.act, a.slope);
                    (&hn4.x)[j] = hn; (&o4.x)[j] = o;
                    (&r4.x)[j] = r; (&z4.x)[j] = z; (&n4.x)[j] = nn;
                    if constexpr (RNG) (&od4.x)[j] = o * drop_scale_w(wd, rg.p);
                }
                if (node) {       // (in front of the stores: the LDS write is through by the time they are issued)
                    float4 xnext = o4;
                    if constexpr (RNG) { if (rg.out_drop) xnext = od4; }      // (the next conv reads the dropped twin)
                    *reinterpret_cast<float4*>(xp) = xnext;
                }
                if (gates) {      // [r | z | n | gh_n]: all the backward takes from the two pre-activation matrices (see GruFusedArgs)
                    float* gp = a.gi + (size_t)row * 4 * C + ch;
                    st4(gp, r4); st4(gp + C, z4); st4(gp + 2 * C, n4); st4(gp + 3 * C, gh4[2]);
                }
                st4(a.h_new + e, hn4);
                st4(a.out + e, o4);
                if constexpr (RNG) { if (rg.out_drop) st4(rg.out_drop + e, od4); }
            } else if (node) {
                *reinterpret_cast<float4*>(xp) = f4zero();      // (zero beyond the matrix: the product runs over 64 channels)
            }
            if (node) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) flag_bump(s_xready + (it & 3));
            }
            if (it == 0) GRU_FINEF(4); else if (it == 2) GRU_FINEF(12);
        }
        GRU_TL(0, 3);
    }
#ifdef GLAM_WS_TL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GRU_TL(0, 5);
#endif
    if constexpr (RNG) rng_end(rg.state, ph);
}


// ------------------------------------------------------------------------------------------------
// Backward of the GRU step in ONE launch: the gate gradients (k_gru_tail_bwd) and both input-gradient products
//   d_x = [d_pr | d_pz | d_pn] @ W_ih  (* celu'(x) with the folded CELU),   d_h = [d_pr | d_pz | d_pn r] @ W_hh + g z
// As two launches the gate kernel wrote d_gi and d_gh (2 x [N, 3C]) and the product pair read them back: 16.5 + 18.1 us per application
// at B = 1024.  Here 4 producer waves recompute the gates of a 16-row tile from gi / gh (two tiles of loads in flight), write d_gi, d_gh
// (the weight-gradient launch needs them) and d_identity, and publish [d_pr | d_pz | d_pn | d_pn r] — the first two gate blocks are
// shared by both products — as bf16 (hi, mid, lo) planes plus the fp32 rows celu'(x) and g z; 8 consumer waves (four 16-channel tiles
// per product, their 6 x (16 x 32) weight slices split in 72 registers) take 36 v_mfma_f32_16x16x32_bf16 per tile each.
// Same gate arithmetic and Philox words as k_gru_tail_bwd<RNG>; the products in 3 x bf16 form (bf16x3.h) like glam_ts_gemm_pair's.
// ------------------------------------------------------------------------------------------------
constexpr int kGbP = 4, kGbC = 8, kGbRing = 3, kGbD = 2;
constexpr int kGbPitch = 672, kGbPlane = 16 * kGbPitch;                 // 256 k per row: four gate blocks padded to 64 channels
constexpr int kGbTile = 3 * kGbPlane + 2 * kGwEPlane;                   // 40 960 bytes
constexpr size_t kGbLds = 128 + (size_t)kGbRing * kGbTile;

struct GruBwdArgs {
    const float* gi; const float* gh; const float* h; const float* out; const float* d_out; const float* d_hstate; const float* x;
    const float* img_ih_t; const float* img_hh_t;      // k_ts_gemm images of W_ih / W_hh as [3C, C] (the input-gradient products)
    float* d_gi; float* d_gh; float* d_identity; float* d_x; float* d_h;
    int N, C, act, celu_in; float slope;
    int merge_identity;      // 1: the skip connection and the GRU state are the SAME tensor (first application of a block, layer.py:254):
                             //    d_h += d_identity here, d_identity is not written
    const void* pre;         // may be null: both matrices as the consumers' 3 x bf16 fragments (k_gru_ws_pre) — img_ih_t / img_hh_t unused
};

template <bool RNG, bool GATES>
__global__ void __launch_bounds__((kGbP + kGbC) * 64) k_gru_bwd_ws(GruBwdArgs a, TailRngB rg) {
    constexpr int P = kGbP, NC = kGbC, RING = kGbRing, D = kGbD, PITCH = kGbPitch, PLANE = kGbPlane, EPITCH = kGwEPitch, EPLANE = kGwEPlane,
                  TILE = kGbTile;
    extern __shared__ __attribute__((aligned(16))) char s_gb[];
    int* s_ready = reinterpret_cast<int*>(s_gb);
    int* s_taken = s_ready + 16;
    char* s_ring = s_gb + 128;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int C = a.C, ntiles = (a.N + 15) >> 4, bid = blockIdx.x, nblk = gridDim.x;
    GRU_TL(1, 0);
    if (tid < 32) s_ready[tid] = 0;
    __syncthreads();

    if (wave < P) {
        // ---- producers: lane -> (row er, channels 4 eq .. 4 eq + 3) of the tile; unconditional loads (items outside the matrices read
        //      gi[0..3] and are zeroed), D tiles in flight ----
        Philox ph{};
        if constexpr (RNG) ph = philox_init(rg.eff);
        const int er = (lane + 64 * wave) >> 4, eq = (lane + 64 * wave) & 15;
        auto item_ok = [&](int tile) { return tile < ntiles && tile * 16 + er < a.N && 4 * eq < C; };
        // GATES: gi is [N, 4C] = [r | z | n | gh_n] as k_gru_fwd_ws wrote it (gh null) and d_gi is written as [N, 4C] =
        // [d_pr | d_pz | d_pn | d_pn r] (d_gh null): 8C instead of 12C floats per row through this launch's memory side
        constexpr int NG = GATES ? 4 : 6, GW = GATES ? 4 : 3, NL = NG + (RNG ? 6 : 5);
        auto load = [&](int tile, float4 (&v)[NL]) {
            const bool ok = item_ok(tile);
            const size_t i = ((size_t)(tile * 16 + er)) * C + 4 * eq, b = ((size_t)(tile * 16 + er)) * GW * C + 4 * eq;
            const float* z = a.gi;
            if constexpr (GATES) {
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = ld4(ok ? a.gi + b + g * C : z);
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g) { v[g] = ld4(ok ? a.gi + b + g * C : z); v[3 + g] = ld4(ok ? a.gh + b + g * C : z); }
            }
            v[NG] = ld4(ok ? a.h + i : z);
            v[NG + 1] = ld4(ok ? a.out + i : z);
            v[NG + 2] = ld4((ok && a.d_out) ? a.d_out + i : z);
            v[NG + 3] = ld4((ok && a.d_hstate) ? a.d_hstate + i : z);
            v[NG + 4] = ld4((ok && a.celu_in) ? a.x + i : z);
            if constexpr (RNG) v[NG + 5] = ld4((ok && rg.d_out_drop) ? rg.d_out_drop + i : z);
        };
        float4 buf[D][NL];
#pragma unroll
        for (int d = 0; d < D; ++d) load(bid + d * nblk, buf[d]);
        if (lane == 0) flag_bump(s_ready + 8);      // (this wave's first loads are in the CU's queue: the consumers may issue theirs)
        GRU_TL(1, 1);
        for (int it0 = 0; bid + it0 * nblk < ntiles; it0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int it = it0 + d, tile = bid + it * nblk;
                if (it == 1) GRU_TL(1, 2);
                if (tile < ntiles) {
                    const bool ok = item_ok(tile);
                    const float4 (&v)[NL] = buf[d];
                    const size_t i = ((size_t)(tile * 16 + er)) * C + 4 * eq, b = ((size_t)(tile * 16 + er)) * GW * C + 4 * eq;
                    uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
                    if constexpr (RNG) w4 = philox4(ph, i >> 2);
                    float4 dy4 = f4zero(), pr4 = f4zero(), pz4 = f4zero(), pn4 = f4zero(), pnr4 = f4zero(), gz4 = f4zero(), cf4 = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (ok) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float ov = f4get(v[NG + 1], j);
                            float dy;
                            if constexpr (RNG) {
                                const unsigned wd = philox_word(w4, j);
                                float g0 = a.d_out ? f4get(v[NG + 2], j) : 0.f;
                                if (rg.d_out_drop) g0 = fmaf(f4get(v[NG + 5], j), drop_scale_w(wd, rg.p), g0);
                                dy = g0 * (a.act == kActRRelu ? (ov > 0.f ? 1.f : rrelu_slope_w(wd, rg.lo, rg.hi)) : act_grad_from_out(ov, a.act, a.slope));
                            } else {
                                dy = f4get(v[NG + 2], j) * act_grad_from_out(ov, a.act, a.slope);
                            }
                            const float g = a.d_hstate ? dy + f4get(v[NG + 3], j) : dy;
                            float ghn, r, z, nn;
                            if constexpr (GATES) { r = f4get(v[0], j); z = f4get(v[1], j); nn = f4get(v[2], j); ghn = f4get(v[3], j); }
                            else {
                                ghn = f4get(v[5], j);
                                r = sigmoidf_(f4get(v[0], j) + f4get(v[3], j));
                                z = sigmoidf_(f4get(v[1], j) + f4get(v[4], j));
                                nn = tanh_(f4get(v[2], j) + r * ghn);
                            }
                            const float d_n = g * (1.f - z), d_z = g * (f4get(v[NG], j) - nn);
                            const float d_pn = d_n * (1.f - nn * nn);
                            (&dy4.x)[j] = dy;
                            (&pr4.x)[j] = d_pn * ghn * r * (1.f - r);
                            (&pz4.x)[j] = d_z * z * (1.f - z);
                            (&pn4.x)[j] = d_pn;
                            (&pnr4.x)[j] = d_pn * r;
                            (&gz4.x)[j] = a.merge_identity ? g * z + dy : g * z;
                            if (a.celu_in == 2) { const float xc = f4get(v[NG + 4], j); (&cf4.x)[j] = xc > 0.f ? 1.f : xc + 1.f; }      // x holds celu(x)
                            else if (a.celu_in) (&cf4.x)[j] = celu1_grad(f4get(v[NG + 4], j));
                        }
                        if (a.d_identity && !a.merge_identity) st4(a.d_identity + i, dy4);
                        st4(a.d_gi + b, pr4); st4(a.d_gi + b + C, pz4); st4(a.d_gi + b + 2 * C, pn4);
                        if constexpr (GATES) st4(a.d_gi + b + 3 * C, pnr4);
                        else { st4(a.d_gh + b, pr4); st4(a.d_gh + b + C, pz4); st4(a.d_gh + b + 2 * C, pnr4); }
                    }
                    load(tile + D * nblk, buf[d]);              // (before the LDS wait: this register set's next tile, D tiles ahead)
                    const int slot = it % RING, round = it / RING;
                    while (flag_load(s_taken + slot) < NC * round) __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                    char* tl = s_ring + slot * TILE;
                    const float4 parts[4] = {pr4, pz4, pn4, pnr4};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned h0, m0, l0, h1, m1, l1;
                        split2(parts[q].x, parts[q].y, h0, m0, l0);
                        split2(parts[q].z, parts[q].w, h1, m1, l1);
                        char* p = tl + er * PITCH + q * 128 + eq * 8;
                        *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
                        *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
                    }
                    *reinterpret_cast<float4*>(tl + 3 * PLANE + er * EPITCH + eq * 16) = cf4;
                    *reinterpret_cast<float4*>(tl + 3 * PLANE + EPLANE + er * EPITCH + eq * 16) = gz4;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) flag_bump(s_ready + slot);
                }
            }
        }
        GRU_TL(1, 3);
#ifdef GLAM_WS_TL
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GRU_TL(1, 5);
#endif
        return;
    }

    // ---- consumers: wave w -> product w >> 2 (0: d_x over W_ih, 1: d_h over W_hh), output channels 16 (w & 3) .. + 15 ----
    const int w = wave - P, prod = w >> 2, ct = w & 3, c = lane & 15, kb = lane >> 4;
    const float* img = prod ? a.img_hh_t : a.img_ih_t;
    const int Kp = (3 * C + 15) & ~15;
    Bf16x3 wreg[6];
    {
        WRaw8 raw[6];
        int col = 16 * ct + c;
        asm volatile("" : "+v"(col));              // (keeps these loads on the consumers' side of the role branch)
        const bool okc = col < C;
        const int cc = min(col, C - 1), pos = (cc & 3) * 16 + (cc >> 2);       // ts_pos_of_col within the 64 positions of a row
        // the producers' first tiles go into the CU's memory queue AHEAD of the 96 weight loads (see k_gru_fwd_ws): the producers are this
        // launch's long pole (4.7 k cycles per tile), and behind the weights their first tile was published 15 k cycles into the block
        while (flag_load(s_ready + 8) < P) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (a.pre) {      // split already, in lane order (k_gru_ws_pre): see k_gru_fwd_ws
            const char* base = reinterpret_cast<const char*>(a.pre) + (size_t)w * (18 * 1024) + lane * 16;
#pragma unroll
            for (int s = 0; s < 6; ++s) { wreg[s].hi = ldfrag(base + s * 3072); wreg[s].mid = ldfrag(base + s * 3072 + 1024); wreg[s].lo = ldfrag(base + s * 3072 + 2048); }
        } else {
#pragma unroll
        for (int s = 0; s < 6; ++s) raw[s] = w_load8(img, 64, pos, (s >> 1) * C + 32 * (s & 1) + 8 * kb, Kp);     // rows gate * C + channel
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int ch0 = 32 * (s & 1) + 8 * kb;                                // (a gate's rows end at channel C: the next gate's follow)
            wreg[s] = split8((okc && ch0 < C) ? raw[s].a : f4zero(), (okc && ch0 + 4 < C) ? raw[s].b : f4zero());
        }
        }
    }
    const int ch = 16 * ct + 4 * kb;
    float* const outp = prod ? a.d_h : a.d_x;
    int it = 0;
    GRU_TL(1, 1);
    for (int tile = bid; tile < ntiles; tile += nblk, ++it) {
        if (it == 1) GRU_TL(1, 2);
        const int slot = it % RING, want = P * (it / RING + 1);
        const int row = 16 * tile + c;
        while (flag_load(s_ready + slot) < want) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const char* tl = s_ring + slot * TILE + c * PITCH + kb * 16;
        v4f_t acc_s = {0.f, 0.f, 0.f, 0.f}, acc_m = acc_s, acc_b = acc_s;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int ls = ((prod && s >= 4) ? 6 : 2 * (s >> 1)) + (s & 1);      // d_h reads d_pn r (block 3) where d_x reads d_pn (block 2)
            Bf16x3 x;
            x.hi = *reinterpret_cast<const bf16x8_t*>(tl + 64 * ls);
            x.mid = *reinterpret_cast<const bf16x8_t*>(tl + PLANE + 64 * ls);
            x.lo = *reinterpret_cast<const bf16x8_t*>(tl + 2 * PLANE + 64 * ls);
            acc_s = mfma_x3_small(wreg[s], x, acc_s);
            acc_m = mfma_x3_mid(wreg[s], x, acc_m);
            acc_b = mfma_x3_big(wreg[s], x, acc_b);
        }
        const float4 e4 = *reinterpret_cast<const float4*>(s_ring + slot * TILE + 3 * PLANE + prod * EPLANE + c * EPITCH + ch * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flag_bump(s_taken + slot);
        if (row < a.N && ch < C) {
            float4 v = make_float4((acc_s[0] + acc_m[0]) + acc_b[0], (acc_s[1] + acc_m[1]) + acc_b[1], (acc_s[2] + acc_m[2]) + acc_b[2],
                                   (acc_s[3] + acc_m[3]) + acc_b[3]);
            if (prod) { v.x += e4.x; v.y += e4.y; v.z += e4.z; v.w += e4.w; }
            else { v.x *= e4.x; v.y *= e4.y; v.z *= e4.z; v.w *= e4.w; }      // celu'(x), or 1
            st4(outp + (size_t)row * C + ch, v);
        }
    }
    GRU_TL(1, 3);
#ifdef GLAM_WS_TL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GRU_TL(1, 5);
#endif
}

}  // namespace glam

using namespace glam;

static int rng_args_ok(const char* fn, int act, float lo, float hi, float p);
#ifdef GLAM_WS_TL
extern "C" int glam_debug_gru_finef(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_gru_finef), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
extern "C" int glam_debug_gru_tl(long long* host_out, int n, int device_wide) {
    return (device_wide ? hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_gru_rt), (size_t)n * sizeof(long long))
                        : hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_gru_tl), (size_t)n * sizeof(long long))) == hipSuccess ? 0 : 1;
}
#endif
#ifdef GLAM_GRU_PROF
extern "C" int glam_debug_gru_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_gru_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

static int gru_fused_launch(const GruFusedArgs& a, const TailRng* rg, hipStream_t s) {
    static bool big = false;       // 96 KB of dynamic LDS is opted into once
    if (!big) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_fused_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGruImgFloats * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_fused_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGruImgFloats * 4);
        big = true;
    }
    const int ntiles = (a.N + 15) / 16;
    int grid = ntiles < 2048 ? ntiles : (ntiles + 7) / 8;
    if (grid > 256) grid = 256;
    const size_t lds = 2 * kGruImgFloats * sizeof(float);
    if (rg) hipLaunchKernelGGL(k_gru_fused_fwd<true>, dim3(grid), dim3(kGruBlock), lds, s, a, *rg);
    else hipLaunchKernelGGL(k_gru_fused_fwd<false>, dim3(grid), dim3(kGruBlock), lds, s, a, TailRng{});
    GLAM_LAUNCH_CHECK("glam_gru_fused_fwd");
    return GLAM_OK;
}

// (layouts and values: gru_pre_fragment, dense.h — glam_prestage builds the same images inside its own launch)
__global__ void __launch_bounds__(64) k_gru_ws_pre(const float* w_ih, const float* w_hh, int C, char* pre_fwd, char* pre_bwd) {
    const bool bwd = blockIdx.x >= kGruPreFrags;
    const int f = blockIdx.x - (bwd ? kGruPreFrags : 0), m = f / 24;
    char* dst = bwd ? pre_bwd : pre_fwd;
    if (dst) gru_pre_fragment(m ? w_hh : w_ih, C, bwd, m, f % 24, threadIdx.x, dst);
}

static int gru_bwd_ws_launch(const GruBwdArgs& a, const TailRngB* rg, hipStream_t s) {
    static bool big0[64] = {}, big1[64] = {}, big2[64] = {}, big3[64] = {};
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_bwd_ws<false, false>), big0, "gru_bwd_ws")) return rc;
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_bwd_ws<true, false>), big1, "gru_bwd_ws")) return rc;
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_bwd_ws<false, true>), big2, "gru_bwd_ws")) return rc;
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_bwd_ws<true, true>), big3, "gru_bwd_ws")) return rc;
    const int ntiles = (a.N + 15) / 16, cap = ws_grid_cap(1024);
    const int grid = ntiles < cap ? ntiles : cap;
    const dim3 g(grid), b((kGbP + kGbC) * 64);
    if (!a.gh) {      // the gates themselves in, the four gradient blocks out (see the kernel)
        GLAM_PROF_LABEL(rg ? "k_gru_bwd_ws<true, gates>" : "k_gru_bwd_ws<false, gates>");
        if (rg) hipLaunchKernelGGL((k_gru_bwd_ws<true, true>), g, b, kGbLds, s, a, *rg);
        else hipLaunchKernelGGL((k_gru_bwd_ws<false, true>), g, b, kGbLds, s, a, TailRngB{});
    } else if (rg) hipLaunchKernelGGL((k_gru_bwd_ws<true, false>), g, b, kGbLds, s, a, *rg);
    else hipLaunchKernelGGL((k_gru_bwd_ws<false, false>), g, b, kGbLds, s, a, TailRngB{});
    GLAM_LAUNCH_CHECK("gru_bwd_ws");
    return GLAM_OK;
}

static int gru_bwd_ws_args_ok(const char* fn, const GruBwdArgs& a, int64_t N, const float* d_out_drop) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "%s: N out of range", fn);
    if (!(a.C >= 24 && a.C <= 64 && (a.C & 3) == 0)) return fail(GLAM_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4 in 24..64", fn, a.C);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(a.gi && a.h && a.out && (a.d_out || d_out_drop) && (a.pre || (a.img_ih_t && a.img_hh_t)) && a.d_gi && a.d_x && a.d_h &&
                     (!a.celu_in || a.x), "%s: null pointer", fn);
    GLAM_REQUIRE((a.gh == nullptr) == (a.d_gh == nullptr), "%s: gh and d_gh are both given (gi, gh, d_gi, d_gh as [N, 3C]) or both null (gi = the "
                 "gates [N, 4C] of a forward without gh, d_gi = [N, 4C])", fn);
    GLAM_REQUIRE(aligned16(a.gi) && aligned16(a.gh) && aligned16(a.h) && aligned16(a.out) && aligned16(a.d_out) && aligned16(a.d_hstate) && aligned16(a.x) &&
                     aligned16(a.img_ih_t) && aligned16(a.img_hh_t) && aligned16(a.d_gi) && aligned16(a.d_gh) && aligned16(a.d_identity) &&
                     aligned16(a.d_x) && aligned16(a.d_h) && aligned16(d_out_drop) && aligned16(a.pre), "%s: pointers must be 16-byte aligned", fn);
    return GLAM_OK;
}

static int gru_bwd_ws_impl(const char* fn, const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                           const float* d_hstate, const float* x, const float* img_ih_t, const float* img_hh_t, const void* pre, int64_t N, int C,
                           int celu_in, int act, float slope, int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x,
                           float* d_h, hipStream_t s) {
    if (act < kActNone || act > kActCelu) return fail(GLAM_E_UNSUPPORTED, "%s: activation code %d", fn, act);
    const GruBwdArgs a{gi, gh, h, out, d_out, d_hstate, x, img_ih_t, img_hh_t, d_gi, d_gh, d_identity, d_x, d_h, (int)N, C, act, celu_in, slope,
                       merge_identity ? 1 : 0, pre};
    if (int rc = gru_bwd_ws_args_ok(fn, a, N, nullptr)) return rc;
    if (N == 0) return GLAM_OK;
    return gru_bwd_ws_launch(a, nullptr, s);
}
extern "C" int glam_gru_bwd_ws(const float* gi, const float* gh, const float* h, const float* out, const float* d_out, const float* d_hstate,
                               const float* x, const float* img_ih_t, const float* img_hh_t, int64_t N, int C, int celu_in, int act, float slope,
                               int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream) {
    return gru_bwd_ws_impl("glam_gru_bwd_ws", gi, gh, h, out, d_out, d_hstate, x, img_ih_t, img_hh_t, nullptr, N, C, celu_in, act, slope,
                           merge_identity, d_gi, d_gh, d_identity, d_x, d_h, (hipStream_t)stream);
}
// ... with the matrices from the backward image of glam_gru_ws_make_pre instead of the two k_ts_gemm images (same values, bit for bit)
extern "C" int glam_gru_bwd_ws_pre(const float* gi, const float* gh, const float* h, const float* out, const float* d_out, const float* d_hstate,
                                   const float* x, const void* pre_bwd, int64_t N, int C, int celu_in, int act, float slope,
                                   int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream) {
    GLAM_REQUIRE(pre_bwd || N == 0, "glam_gru_bwd_ws_pre: null image");
    return gru_bwd_ws_impl("glam_gru_bwd_ws_pre", gi, gh, h, out, d_out, d_hstate, x, nullptr, nullptr, pre_bwd, N, C, celu_in, act, slope,
                           merge_identity, d_gi, d_gh, d_identity, d_x, d_h, (hipStream_t)stream);
}

static int rng_args_ok(const char* fn, int act, float lo, float hi, float p);
static int gru_bwd_ws_rng_impl(const char* fn, const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                               const float* d_out_drop, const float* d_hstate, const float* x, const float* img_ih_t, const float* img_hh_t,
                               const void* pre, int64_t N, int C, int celu_in, int act, float slope, float rr_lower, float rr_upper, float drop_p,
                               const int64_t* rng_eff, int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h,
                               hipStream_t s) {
    if (int rc = rng_args_ok(fn, act, rr_lower, rr_upper, drop_p)) return rc;
    const GruBwdArgs a{gi, gh, h, out, d_out, d_hstate, x, img_ih_t, img_hh_t, d_gi, d_gh, d_identity, d_x, d_h, (int)N, C, act, celu_in, slope,
                       merge_identity ? 1 : 0, pre};
    if (int rc = gru_bwd_ws_args_ok(fn, a, N, d_out_drop)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(rng_eff, "%s: null rng_eff", fn);
    const TailRngB rg{reinterpret_cast<const long long*>(rng_eff), rr_lower, rr_upper, drop_p, d_out_drop, 1};
    return gru_bwd_ws_launch(a, &rg, s);
}
extern "C" int glam_gru_bwd_ws_rng(const float* gi, const float* gh, const float* h, const float* out, const float* d_out, const float* d_out_drop,
                                   const float* d_hstate, const float* x, const float* img_ih_t, const float* img_hh_t, int64_t N, int C,
                                   int celu_in, int act, float slope, float rr_lower, float rr_upper, float drop_p, const int64_t* rng_eff,
                                   int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream) {
    return gru_bwd_ws_rng_impl("glam_gru_bwd_ws_rng", gi, gh, h, out, d_out, d_out_drop, d_hstate, x, img_ih_t, img_hh_t, nullptr, N, C, celu_in,
                               act, slope, rr_lower, rr_upper, drop_p, rng_eff, merge_identity, d_gi, d_gh, d_identity, d_x, d_h,
                               (hipStream_t)stream);
}
extern "C" int glam_gru_bwd_ws_rng_pre(const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                                       const float* d_out_drop, const float* d_hstate, const float* x, const void* pre_bwd, int64_t N, int C,
                                       int celu_in, int act, float slope, float rr_lower, float rr_upper, float drop_p, const int64_t* rng_eff,
                                       int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream) {
    GLAM_REQUIRE(pre_bwd || N == 0, "glam_gru_bwd_ws_rng_pre: null image");
    return gru_bwd_ws_rng_impl("glam_gru_bwd_ws_rng_pre", gi, gh, h, out, d_out, d_out_drop, d_hstate, x, nullptr, nullptr, pre_bwd, N, C,
                               celu_in, act, slope, rr_lower, rr_upper, drop_p, rng_eff, merge_identity, d_gi, d_gh, d_identity, d_x, d_h,
                               (hipStream_t)stream);
}

extern "C" size_t glam_gru_ws_pre_bytes(void) { return (size_t)kGruPreBytes; }
// both images (either may be null) from weight_ih / weight_hh [3C, C] (contiguous) in one launch
extern "C" int glam_gru_ws_make_pre(const float* w_ih, const float* w_hh, int C, void* pre_fwd, void* pre_bwd, void* stream) {
    if (!(C >= 24 && C <= 64 && (C & 3) == 0)) return fail(GLAM_E_UNSUPPORTED, "glam_gru_ws_make_pre: C=%d must be a multiple of 4 in 24..64", C);
    GLAM_REQUIRE(w_ih && w_hh && aligned16(pre_fwd) && aligned16(pre_bwd), "glam_gru_ws_make_pre: null / misaligned pointer");
    if (!pre_fwd && !pre_bwd) return GLAM_OK;
    hipLaunchKernelGGL(k_gru_ws_pre, dim3(2 * kGruPreFrags), dim3(64), 0, (hipStream_t)stream, w_ih, w_hh, C, (char*)pre_fwd, (char*)pre_bwd);
    GLAM_LAUNCH_CHECK("glam_gru_ws_make_pre");
    return GLAM_OK;
}

// the warp-specialised 3 x bf16 form: weights from the k_ts_gemm images of the gate matrices (K = C, M = 3 C with 64 < 3 C <= 192)
extern "C" int glam_gru_ws_supported(int C) { return C >= 24 && C <= 64 && (C & 3) == 0; }

static int gru_ws_launch(const GruFusedArgs& a, const TailRng* rg, hipStream_t s) {
    static bool big0[64] = {}, big1[64] = {}, big2[64] = {}, big3[64] = {};
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_fwd_ws<false, false>), big0, "gru_ws_fwd")) return rc;
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_fwd_ws<true, false>), big1, "gru_ws_fwd")) return rc;
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_fwd_ws<false, true>), big2, "gru_ws_fwd")) return rc;
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_gru_fwd_ws<true, true>), big3, "gru_ws_fwd")) return rc;
    const int ntiles = (a.N + 15) / 16, cap = ws_grid_cap(1024);
    const int grid = ntiles < cap ? ntiles : cap;
    const size_t lds = a.xw ? kGwLdsNode : kGwLds;
    if (a.xw) GLAM_PROF_LABEL(rg ? "k_gru_fwd_ws<true>+node" : "k_gru_fwd_ws<false>+node");
    else if (!a.gh) GLAM_PROF_LABEL(rg ? "k_gru_fwd_ws<true, gates>" : "k_gru_fwd_ws<false, gates>");
    const dim3 g(grid), b((kGwP + kGwC) * 64);
    if (a.xw) {
        if (rg) hipLaunchKernelGGL((k_gru_fwd_ws<true, true>), g, b, lds, s, a, *rg);
        else hipLaunchKernelGGL((k_gru_fwd_ws<false, true>), g, b, lds, s, a, TailRng{});
    } else {
        if (a.gh) GLAM_PROF_LABEL(rg ? "k_gru_fwd_ws<true>" : "k_gru_fwd_ws<false>");      // (the names the stringified launch used to have)
        if (rg) hipLaunchKernelGGL((k_gru_fwd_ws<true, false>), g, b, lds, s, a, *rg);
        else hipLaunchKernelGGL((k_gru_fwd_ws<false, false>), g, b, lds, s, a, TailRng{});
    }
    GLAM_LAUNCH_CHECK("gru_ws_fwd");
    return GLAM_OK;
}

static int gru_ws_args_ok(const char* fn, const GruFusedArgs& a, int64_t N) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "%s: N out of range", fn);
    if (!glam_gru_ws_supported(a.C)) return fail(GLAM_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4 in 24..64", fn, a.C);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(a.x && a.h && (a.pre || (a.img_ih && a.img_hh)) && a.b_ih && a.b_hh && a.gi && a.h_new && a.out, "%s: null pointer", fn);
    GLAM_REQUIRE(aligned16(a.x) && aligned16(a.h) && aligned16(a.identity) && aligned16(a.img_ih) && aligned16(a.img_hh) && aligned16(a.pre) && aligned16(a.gi) &&
                     aligned16(a.gh) && aligned16(a.h_new) && aligned16(a.out), "%s: pointers must be 16-byte aligned", fn);
    return GLAM_OK;
}

// the node product of the NEXT application of the block inside this launch (see GruFusedArgs): all four given or none
struct GruNode { const void* img; int cols; float* xw; float* a_ij; };
static int gru_node_ok(const char* fn, const GruNode& n, int C) {
    if (!n.img && !n.xw && !n.a_ij) return GLAM_OK;
    GLAM_REQUIRE(n.img && n.xw && n.a_ij && aligned16(n.img) && aligned16(n.xw) && aligned16(n.a_ij), "%s: node image, xw and a_ij are given together, 16-byte aligned", fn);
    if (!(n.cols > 0 && (n.cols & 3) == 0 && n.cols + 8 > 64 && n.cols + 8 <= 192 && C >= 24))
        return fail(GLAM_E_UNSUPPORTED, "%s: the node product takes 56 < H*Cp <= 184 columns, a multiple of 4 (%d)", fn, n.cols);
    return GLAM_OK;
}

static int gru_ws_fwd_impl(const char* fn, const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                           const void* pre, const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh,
                           float* h_new, float* out, float* x_celu, hipStream_t s, GruNode nd = GruNode{nullptr, 0, nullptr, nullptr}) {
    if (act < kActNone || act > kActCelu) return fail(GLAM_E_UNSUPPORTED, "%s: activation code %d", fn, act);
    GLAM_REQUIRE(!x_celu || (celu_in && aligned16(x_celu)), "%s: x_celu needs celu_in and 16-byte alignment", fn);
    const GruFusedArgs a{x, h, identity, img_ih, img_hh, b_ih, b_hh, gi, gh, h_new, out, (int)N, C, celu_in, act, slope, x_celu, pre,
                         nd.img, nd.xw, nd.a_ij, nd.cols};
    if (int rc = gru_ws_args_ok(fn, a, N)) return rc;
    if (int rc = gru_node_ok(fn, nd, C)) return rc;
    if (N == 0) return GLAM_OK;
    return gru_ws_launch(a, nullptr, s);
}
extern "C" int glam_gru_ws_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                               const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi,
                               float* gh, float* h_new, float* out, void* stream) {
    return gru_ws_fwd_impl("glam_gru_ws_fwd", x, h, identity, img_ih, img_hh, nullptr, b_ih, b_hh, N, C, celu_in, act, slope, gi, gh, h_new, out, nullptr,
                           (hipStream_t)stream);
}
// ... also writing x_celu[N, C] = celu(x) (celu_in must be set): the tensor to keep for the backward INSTEAD of x — glam_gru_bwd_ws with
// celu_in = 2 takes celu'(x) from it (x > 0 ? 1 : celu(x) + 1) and the weight gradient reads it as Q without a CELU of its own
extern "C" int glam_gru_ws_fwd_xc(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                                  const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi,
                                  float* gh, float* h_new, float* out, float* x_celu, void* stream) {
    return gru_ws_fwd_impl("glam_gru_ws_fwd_xc", x, h, identity, img_ih, img_hh, nullptr, b_ih, b_hh, N, C, celu_in, act, slope, gi, gh, h_new, out, x_celu,
                           (hipStream_t)stream);
}

static int rng_args_ok(const char* fn, int act, float lo, float hi, float p);
static int gru_ws_rng_fwd_impl(const char* fn, const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                               const void* pre, const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower,
                               float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh, float* h_new,
                               float* out, float* out_drop, float* x_celu, hipStream_t s, GruNode nd = GruNode{nullptr, 0, nullptr, nullptr}) {
    if (int rc = rng_args_ok(fn, act, rr_lower, rr_upper, drop_p)) return rc;
    GLAM_REQUIRE(!x_celu || (celu_in && aligned16(x_celu)), "%s: x_celu needs celu_in and 16-byte alignment", fn);
    const GruFusedArgs a{x, h, identity, img_ih, img_hh, b_ih, b_hh, gi, gh, h_new, out, (int)N, C, celu_in, act, slope, x_celu, pre,
                         nd.img, nd.xw, nd.a_ij, nd.cols};
    if (int rc = gru_ws_args_ok(fn, a, N)) return rc;
    if (int rc = gru_node_ok(fn, nd, C)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(rng_state && rng_eff && aligned16(out_drop), "%s: null RNG state / misaligned out_drop", fn);
    const TailRng rg{reinterpret_cast<long long*>(rng_state), reinterpret_cast<long long*>(rng_eff), rr_lower, rr_upper, drop_p, out_drop, 1};
    return gru_ws_launch(a, &rg, s);
}
extern "C" int glam_gru_ws_rng_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                                   const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope,
                                   float rr_lower, float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi,
                                   float* gh, float* h_new, float* out, float* out_drop, void* stream) {
    return gru_ws_rng_fwd_impl("glam_gru_ws_rng_fwd", x, h, identity, img_ih, img_hh, nullptr, b_ih, b_hh, N, C, celu_in, act, slope, rr_lower, rr_upper,
                               drop_p, rng_state, rng_eff, gi, gh, h_new, out, out_drop, nullptr, (hipStream_t)stream);
}
extern "C" int glam_gru_ws_rng_fwd_xc(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                                      const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope,
                                      float rr_lower, float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi,
                                      float* gh, float* h_new, float* out, float* out_drop, float* x_celu, void* stream) {
    return gru_ws_rng_fwd_impl("glam_gru_ws_rng_fwd_xc", x, h, identity, img_ih, img_hh, nullptr, b_ih, b_hh, N, C, celu_in, act, slope, rr_lower, rr_upper,
                               drop_p, rng_state, rng_eff, gi, gh, h_new, out, out_drop, x_celu, (hipStream_t)stream);
}

// ... with the gate matrices from the forward image of glam_gru_ws_make_pre instead of the two k_ts_gemm images (same values, bit for
// bit; x_celu may be null)
extern "C" int glam_gru_ws_fwd_pre(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih,
                                   const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh, float* h_new,
                                   float* out, float* x_celu, void* stream) {
    GLAM_REQUIRE(pre_fwd || N == 0, "glam_gru_ws_fwd_pre: null image");
    return gru_ws_fwd_impl("glam_gru_ws_fwd_pre", x, h, identity, nullptr, nullptr, pre_fwd, b_ih, b_hh, N, C, celu_in, act, slope, gi, gh, h_new,
                           out, x_celu, (hipStream_t)stream);
}
extern "C" int glam_gru_ws_rng_fwd_pre(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih,
                                       const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower, float rr_upper,
                                       float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh, float* h_new, float* out,
                                       float* out_drop, float* x_celu, void* stream) {
    GLAM_REQUIRE(pre_fwd || N == 0, "glam_gru_ws_rng_fwd_pre: null image");
    return gru_ws_rng_fwd_impl("glam_gru_ws_rng_fwd_pre", x, h, identity, nullptr, nullptr, pre_fwd, b_ih, b_hh, N, C, celu_in, act, slope,
                               rr_lower, rr_upper, drop_p, rng_state, rng_eff, gi, gh, h_new, out, out_drop, x_celu, (hipStream_t)stream);
}

// ... and the node product of the block's NEXT application (src_1gp/model.py:53-54: the same block is applied message_steps times, so
// this step's output rows — the dropped twin in the rng form when out_drop is given — are the next TripletMessage's input): the launch
// also writes xw[N, node_cols] | a_ij[N, 8] = out @ [W_node | Wa] from node_pre, that matrix as the producers' pre-split operand
// fragments (K = C rows, node_cols + 8 columns: glam_triplet_staged_node_fragments) — what glam_ts_gemm writes for the same rows, bit for bit;
// glam_triplet_layer_fwd_ell with x = NULL then starts at its aggregate launch.
extern "C" int glam_gru_ws_fwd_pre_node(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih,
                                        const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh,
                                        float* h_new, float* out, float* x_celu, const void* node_pre, int node_cols, float* xw,
                                        float* a_ij, void* stream) {
    GLAM_REQUIRE(pre_fwd || N == 0, "glam_gru_ws_fwd_pre_node: null image");
    GLAM_REQUIRE(node_pre && xw && a_ij, "glam_gru_ws_fwd_pre_node: null node image / output");
    return gru_ws_fwd_impl("glam_gru_ws_fwd_pre_node", x, h, identity, nullptr, nullptr, pre_fwd, b_ih, b_hh, N, C, celu_in, act, slope, gi, gh,
                           h_new, out, x_celu, (hipStream_t)stream, GruNode{node_pre, node_cols, xw, a_ij});
}
extern "C" int glam_gru_ws_rng_fwd_pre_node(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih,
                                            const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower,
                                            float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh,
                                            float* h_new, float* out, float* out_drop, float* x_celu, const void* node_pre,
                                            int node_cols, float* xw, float* a_ij, void* stream) {
    GLAM_REQUIRE(pre_fwd || N == 0, "glam_gru_ws_rng_fwd_pre_node: null image");
    GLAM_REQUIRE(node_pre && xw && a_ij, "glam_gru_ws_rng_fwd_pre_node: null node image / output");
    return gru_ws_rng_fwd_impl("glam_gru_ws_rng_fwd_pre_node", x, h, identity, nullptr, nullptr, pre_fwd, b_ih, b_hh, N, C, celu_in, act, slope,
                               rr_lower, rr_upper, drop_p, rng_state, rng_eff, gi, gh, h_new, out, out_drop, x_celu, (hipStream_t)stream,
                               GruNode{node_pre, node_cols, xw, a_ij});
}

extern "C" int glam_gru_fused_supported(int C) { return C >= 4 && C <= 64 && (C & 3) == 0; }
extern "C" size_t glam_gru_fused_image_bytes(void) { return (size_t)kGruImgFloats * sizeof(float); }

extern "C" int glam_gru_fused_make_images(const float* w_ih, const float* w_hh, int C, float* img_ih, float* img_hh, void* stream) {
    GLAM_REQUIRE(glam_gru_fused_supported(C), "glam_gru_fused_make_images: C=%d must be a multiple of 4, at most 64", C);
    GLAM_REQUIRE(w_ih && w_hh && img_ih && img_hh && aligned16(img_ih) && aligned16(img_hh), "glam_gru_fused_make_images: null / misaligned pointer");
    hipLaunchKernelGGL(k_gru_fused_images, dim3((2 * kGruImgFloats + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, w_ih, w_hh, C,
                       img_ih, img_hh);
    GLAM_LAUNCH_CHECK("glam_gru_fused_make_images");
    return GLAM_OK;
}

static int gru_fused_args_ok(const char* fn, const GruFusedArgs& a, int64_t N) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "%s: N out of range", fn);
    if (!glam_gru_fused_supported(a.C)) return fail(GLAM_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4, at most 64", fn, a.C);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(a.x && a.h && a.img_ih && a.img_hh && a.b_ih && a.b_hh && a.gi && a.gh && a.h_new && a.out, "%s: null pointer", fn);
    GLAM_REQUIRE(aligned16(a.x) && aligned16(a.h) && aligned16(a.identity) && aligned16(a.img_ih) && aligned16(a.img_hh) && aligned16(a.b_ih) &&
                     aligned16(a.b_hh) && aligned16(a.gi) && aligned16(a.gh) && aligned16(a.h_new) && aligned16(a.out),
                 "%s: pointers must be 16-byte aligned", fn);
    return GLAM_OK;
}

extern "C" int glam_gru_fused_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                                  const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi,
                                  float* gh, float* h_new, float* out, void* stream) {
    if (act < kActNone || act > kActCelu) return fail(GLAM_E_UNSUPPORTED, "glam_gru_fused_fwd: activation code %d", act);
    const GruFusedArgs a{x, h, identity, img_ih, img_hh, b_ih, b_hh, gi, gh, h_new, out, (int)N, C, celu_in, act, slope};
    if (int rc = gru_fused_args_ok("glam_gru_fused_fwd", a, N)) return rc;
    if (N == 0) return GLAM_OK;
    return gru_fused_launch(a, nullptr, (hipStream_t)stream);
}

extern "C" int glam_gru_fused_rng_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                                      const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope,
                                      float rr_lower, float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi,
                                      float* gh, float* h_new, float* out, float* out_drop, void* stream) {
    if (int rc = rng_args_ok("glam_gru_fused_rng_fwd", act, rr_lower, rr_upper, drop_p)) return rc;
    const GruFusedArgs a{x, h, identity, img_ih, img_hh, b_ih, b_hh, gi, gh, h_new, out, (int)N, C, celu_in, act, slope};
    if (int rc = gru_fused_args_ok("glam_gru_fused_rng_fwd", a, N)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(rng_state && rng_eff && aligned16(out_drop), "glam_gru_fused_rng_fwd: null RNG state / misaligned out_drop");
    const TailRng rg{reinterpret_cast<long long*>(rng_state), reinterpret_cast<long long*>(rng_eff), rr_lower, rr_upper, drop_p, out_drop, 1};
    return gru_fused_launch(a, &rg, (hipStream_t)stream);
}


extern "C" int glam_gru_tail_fwd(const float* gi, const float* gh, const float* h, const float* identity, int64_t N, int C,
                                 int act, float slope, float* h_new, float* out, void* stream) {
    GLAM_REQUIRE(N >= 0 && C > 0 && N * (int64_t)C < (int64_t)INT32_MAX * 64, "glam_gru_tail_fwd: bad sizes");
    if (act < kActNone || act > kActCelu) return fail(GLAM_E_UNSUPPORTED, "glam_gru_tail_fwd: activation code %d", act);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(gi && gh && h && h_new && out, "glam_gru_tail_fwd: null pointer");
    hipLaunchKernelGGL(k_gru_tail_fwd<false>, dim3(grid_for(N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gi, gh, h, identity,
                       (int)N, C, act, slope, h_new, out, TailRng{});
    GLAM_LAUNCH_CHECK("glam_gru_tail_fwd");
    return GLAM_OK;
}

extern "C" int glam_gru_tail_bwd(const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                                 const float* d_hstate, int64_t N, int C, int act, float slope, float* d_gi, float* d_gh,
                                 float* d_h, float* d_identity, void* stream) {
    GLAM_REQUIRE(N >= 0 && C > 0, "glam_gru_tail_bwd: bad sizes");
    if (act < kActNone || act > kActCelu) return fail(GLAM_E_UNSUPPORTED, "glam_gru_tail_bwd: activation code %d", act);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(gi && gh && h && out && d_out && d_gi && d_gh && d_h, "glam_gru_tail_bwd: null pointer");
    TailRngB rg{};
    rg.vec = (C & 3) == 0 && aligned16(gi) && aligned16(gh) && aligned16(h) && aligned16(out) && aligned16(d_out) && aligned16(d_hstate) &&
             aligned16(d_gi) && aligned16(d_gh) && aligned16(d_h) && aligned16(d_identity);
    hipLaunchKernelGGL(k_gru_tail_bwd<false>, dim3(grid_for(rg.vec ? N * C / 4 : N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gi, gh, h,
                       out, d_out, d_hstate, (int)N, C, act, slope, d_gi, d_gh, d_h, d_identity, rg);
    GLAM_LAUNCH_CHECK("glam_gru_tail_bwd");
    return GLAM_OK;
}

extern "C" int glam_gru_gates_fwd(const float* gi, const float* gh, const float* h, int64_t N, int C, float* h_new,
                                  void* stream) {
    GLAM_REQUIRE(N >= 0 && C > 0 && N * (int64_t)C < (int64_t)INT32_MAX * 64, "glam_gru_gates_fwd: bad sizes");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(gi && gh && h && h_new, "glam_gru_gates_fwd: null pointer");
    hipLaunchKernelGGL(k_gru_gates_fwd, dim3(grid_for(N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gi, gh, h, (int)N, C, h_new);
    GLAM_LAUNCH_CHECK("glam_gru_gates_fwd");
    return GLAM_OK;
}

extern "C" int glam_gru_gates_bwd(const float* gi, const float* gh, const float* h, const float* d_hnew, int64_t N, int C,
                                  float* d_gi, float* d_gh, float* d_h, void* stream) {
    GLAM_REQUIRE(N >= 0 && C > 0, "glam_gru_gates_bwd: bad sizes");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(gi && gh && h && d_hnew && d_gi && d_gh && d_h, "glam_gru_gates_bwd: null pointer");
    hipLaunchKernelGGL(k_gru_gates_bwd, dim3(grid_for(N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gi, gh, h, d_hnew,
                       (int)N, C, d_gi, d_gh, d_h);
    GLAM_LAUNCH_CHECK("glam_gru_gates_bwd");
    return GLAM_OK;
}

extern "C" int glam_lstm_cell_fwd(const float* gates, const float* c_prev, int64_t B, int C, float* h_new, float* c_new,
                                  void* stream) {
    GLAM_REQUIRE(B >= 0 && B < INT32_MAX && C > 0, "glam_lstm_cell_fwd: bad dims");
    if (B == 0) return GLAM_OK;
    GLAM_REQUIRE(gates && c_prev && h_new && c_new, "glam_lstm_cell_fwd: null pointer");
    hipLaunchKernelGGL(k_lstm_cell_fwd, dim3(grid_for(B * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gates, c_prev, (int)B, C,
                       h_new, c_new);
    GLAM_LAUNCH_CHECK("glam_lstm_cell_fwd");
    return GLAM_OK;
}

extern "C" int glam_lstm_cell_bwd(const float* gates, const float* c_prev, const float* d_h, const float* d_c, int64_t B, int C,
                                  float* d_gates, float* d_c_prev, void* stream) {
    GLAM_REQUIRE(B >= 0 && B < INT32_MAX && C > 0, "glam_lstm_cell_bwd: bad dims");
    if (B == 0) return GLAM_OK;
    GLAM_REQUIRE(gates && c_prev && d_gates && d_c_prev, "glam_lstm_cell_bwd: null pointer");
    hipLaunchKernelGGL(k_lstm_cell_bwd, dim3(grid_for(B * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gates, c_prev, d_h, d_c,
                       (int)B, C, d_gates, d_c_prev);
    GLAM_LAUNCH_CHECK("glam_lstm_cell_bwd");
    return GLAM_OK;
}

extern "C" int glam_bias_res_act_fwd(const float* y, const float* bias, const float* identity, int64_t N, int C, int act, float slope,
                                     float* out, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX && C > 0 && act >= 0 && act <= 3, "glam_bias_res_act_fwd: bad arguments");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(y && out, "glam_bias_res_act_fwd: null pointer");
    hipLaunchKernelGGL(k_bias_res_act_fwd<false>, dim3(grid_for(N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, y, bias, identity, (int)N, C,
                       act, slope, out, TailRng{});
    GLAM_LAUNCH_CHECK("glam_bias_res_act_fwd");
    return GLAM_OK;
}

extern "C" int glam_bias_res_act_bwd(const float* out, const float* d_out, int64_t N, int C, int act, float slope, float* d_y,
                                     void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX && C > 0 && act >= 0 && act <= 3, "glam_bias_res_act_bwd: bad arguments");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(out && d_out && d_y, "glam_bias_res_act_bwd: null pointer");
    hipLaunchKernelGGL(k_bias_res_act_bwd<false>, dim3(grid_for(N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, d_out, (int)N, C, act,
                       slope, d_y, TailRngB{});
    GLAM_LAUNCH_CHECK("glam_bias_res_act_bwd");
    return GLAM_OK;
}

// ---- training-mode variants: RReLU / Dropout from the device-side Philox stream (rng.h) ----
// RNG launches take one device-scope atomic per block for the stream-position ticket (~10 ns each when they hit one address: a single
// counter made a 2048-block launch 20 us, a 512-block one 5 us longer than its arithmetic; the two-level ticket of rng.h keeps every
// same-address chain at grid / 16 + 16 operations, so the grid no longer needs a special cap).
constexpr int kRngBlocks = 2048;

static int rng_args_ok(const char* fn, int act, float lo, float hi, float p) {
    if (act < kActNone || act > kActRRelu) return fail(GLAM_E_UNSUPPORTED, "%s: activation code %d", fn, act);
    if (act == kActRRelu && !(lo > 0.f && lo <= hi)) return fail(GLAM_E_INVALID, "%s: RReLU needs 0 < lower <= upper (got %g, %g)", fn, lo, hi);
    if (!(p >= 0.f && p < 1.f)) return fail(GLAM_E_INVALID, "%s: dropout p = %g outside [0, 1)", fn, p);
    return GLAM_OK;
}

extern "C" int glam_gru_tail_rng_fwd(const float* gi, const float* gh, const float* h, const float* identity, int64_t N, int C,
                                     int act, float slope, float rr_lower, float rr_upper, float drop_p, int64_t* rng_state,
                                     int64_t* rng_eff, float* h_new, float* out, float* out_drop, void* stream) {
    GLAM_REQUIRE(N >= 0 && C > 0 && N * (int64_t)C < (int64_t)INT32_MAX * 64, "glam_gru_tail_rng_fwd: bad sizes");
    if (int rc = rng_args_ok("glam_gru_tail_rng_fwd", act, rr_lower, rr_upper, drop_p)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(gi && gh && h && h_new && out && rng_state && rng_eff, "glam_gru_tail_rng_fwd: null pointer");
    const int vec = (C & 3) == 0 && aligned16(gi) && aligned16(gh) && aligned16(h) && aligned16(identity) && aligned16(h_new) &&
                    aligned16(out) && aligned16(out_drop);
    TailRng rg{reinterpret_cast<long long*>(rng_state), reinterpret_cast<long long*>(rng_eff), rr_lower, rr_upper, drop_p, out_drop, vec};
    hipLaunchKernelGGL(k_gru_tail_fwd<true>, dim3(grid_for(N * C, kBlock, kRngBlocks)), dim3(kBlock), 0, (hipStream_t)stream, gi, gh, h, identity,
                       (int)N, C, act, slope, h_new, out, rg);
    GLAM_LAUNCH_CHECK("glam_gru_tail_rng_fwd");
    return GLAM_OK;
}

extern "C" int glam_gru_tail_rng_bwd(const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                                     const float* d_out_drop, const float* d_hstate, int64_t N, int C, int act, float slope,
                                     float rr_lower, float rr_upper, float drop_p, const int64_t* rng_eff, float* d_gi, float* d_gh,
                                     float* d_h, float* d_identity, void* stream) {
    GLAM_REQUIRE(N >= 0 && C > 0, "glam_gru_tail_rng_bwd: bad sizes");
    if (int rc = rng_args_ok("glam_gru_tail_rng_bwd", act, rr_lower, rr_upper, drop_p)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(gi && gh && h && out && (d_out || d_out_drop) && d_gi && d_gh && d_h && rng_eff, "glam_gru_tail_rng_bwd: null pointer");
    const int vec = (C & 3) == 0 && aligned16(gi) && aligned16(gh) && aligned16(h) && aligned16(out) && aligned16(d_out) && aligned16(d_out_drop) &&
                    aligned16(d_hstate) && aligned16(d_gi) && aligned16(d_gh) && aligned16(d_h) && aligned16(d_identity);
    TailRngB rg{reinterpret_cast<const long long*>(rng_eff), rr_lower, rr_upper, drop_p, d_out_drop, vec};
    hipLaunchKernelGGL(k_gru_tail_bwd<true>, dim3(grid_for(vec ? N * C / 4 : N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gi, gh, h, out,
                       d_out, d_hstate, (int)N, C, act, slope, d_gi, d_gh, d_h, d_identity, rg);
    GLAM_LAUNCH_CHECK("glam_gru_tail_rng_bwd");
    return GLAM_OK;
}

extern "C" int glam_bias_res_act_rng_fwd(const float* y, const float* bias, const float* identity, int64_t N, int C, int act,
                                         float slope, float rr_lower, float rr_upper, float drop_p, int64_t* rng_state,
                                         int64_t* rng_eff, float* out, float* out_drop, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX && C > 0, "glam_bias_res_act_rng_fwd: bad sizes");
    if (int rc = rng_args_ok("glam_bias_res_act_rng_fwd", act, rr_lower, rr_upper, drop_p)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(y && (out || (out_drop && act == kActNone)) && rng_state && rng_eff, "glam_bias_res_act_rng_fwd: null pointer");
    const int vec = (C & 3) == 0 && aligned16(y) && aligned16(bias) && aligned16(identity) && aligned16(out) && aligned16(out_drop);
    TailRng rg{reinterpret_cast<long long*>(rng_state), reinterpret_cast<long long*>(rng_eff), rr_lower, rr_upper, drop_p, out_drop, vec};
    hipLaunchKernelGGL(k_bias_res_act_fwd<true>, dim3(grid_for(N * C, kBlock, kRngBlocks)), dim3(kBlock), 0, (hipStream_t)stream, y, bias, identity,
                       (int)N, C, act, slope, out, rg);
    GLAM_LAUNCH_CHECK("glam_bias_res_act_rng_fwd");
    return GLAM_OK;
}

extern "C" int glam_bias_res_act_rng_bwd(const float* out, const float* d_out, const float* d_out_drop, int64_t N, int C, int act,
                                         float slope, float rr_lower, float rr_upper, float drop_p, const int64_t* rng_eff,
                                         float* d_y, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX && C > 0, "glam_bias_res_act_rng_bwd: bad sizes");
    if (int rc = rng_args_ok("glam_bias_res_act_rng_bwd", act, rr_lower, rr_upper, drop_p)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE((out || act == kActNone) && (d_out || d_out_drop) && d_y && rng_eff, "glam_bias_res_act_rng_bwd: null pointer");
    const int vec = (C & 3) == 0 && aligned16(out) && aligned16(d_out) && aligned16(d_out_drop) && aligned16(d_y);
    TailRngB rg{reinterpret_cast<const long long*>(rng_eff), rr_lower, rr_upper, drop_p, d_out_drop, vec};
    hipLaunchKernelGGL(k_bias_res_act_bwd<true>, dim3(grid_for(N * C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, d_out, (int)N,
                       C, act, slope, d_y, rg);
    GLAM_LAUNCH_CHECK("glam_bias_res_act_rng_bwd");
    return GLAM_OK;
}
