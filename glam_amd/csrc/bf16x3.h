// fp32 products on the bf16 matrix cores without giving up fp32 accuracy ("3 x bf16"): every fp32 operand is split EXACTLY into three
// bf16 terms, x = hi + mid + lo (round-to-nearest at each step: |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|), and a product a * b is taken as
// the six partial products a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid — each EXACT in the fp32
// accumulator (8 x 8 significant bits) — dropping a_mid b_lo + a_lo b_mid + a_lo b_lo <= 2^-26 |a b|, a quarter of the rounding error a
// single fp32 multiply makes (2^-24).  Why (round 4, tools/ubench/mfma_valu_overlap.hip): on gfx950 v_mfma_f32_16x16x4_f32 runs at the
// fp32 VECTOR rate and does NOT overlap with the vector instructions of the other waves of its SIMD (together = 1.75 x the longer of the
// two alone: one fp32 datapath), whereas v_mfma_f32_16x16x32_bf16 does (1.20 x) and moves 8 x the k per instruction in 21 cycles instead
// of 34.6: six of them replace eight fp32 MFMAs (125 vs 277 cycles) AND leave the vector pipe to the gather waves.
#pragma once
#include "common.h"

namespace glam {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

struct Bf16x3 { bf16x8_t hi, mid, lo; };      // eight consecutive k of one row / column: 4 registers per term

// two floats -> their (hi, mid, lo) bf16 pairs, packed (element 0 in the low half): v_cvt_pk_bf16_f32 rounds to nearest even
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const f32x2_t v = {x0, x1};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r1 = {x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xffff0000u)};        // exact
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
    const f32x2_t r2 = {r1[0] - __builtin_bit_cast(float, m << 16), r1[1] - __builtin_bit_cast(float, m & 0xffff0000u)};  // exact
    hi = h; mid = m; lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}
// The same split for two values that do NOT sit in adjacent registers (k_wgrad_x3: the same column of two rows), spelled on scalars: with
// the <2 x float> form the compiler builds v_pk_add_f32 operands out of two v_mov each (8.4 vector instructions per value instead of 5.5,
// and packed fp32 instructions are the expensive ones beside matrix instructions)
__device__ __forceinline__ void split2s(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){x0, x1}, bf16x2_t));
    float a0 = x0 - __builtin_bit_cast(float, h << 16), a1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);            // exact
    asm volatile("" : "+v"(a0), "+v"(a1));      // (keeps the two subtractions scalar: no SLP packing)
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a0, a1}, bf16x2_t));
    float b0 = a0 - __builtin_bit_cast(float, m << 16), b1 = a1 - __builtin_bit_cast(float, m & 0xffff0000u);            // exact
    asm volatile("" : "+v"(b0), "+v"(b1));
    hi = h; mid = m; lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){b0, b1}, bf16x2_t));
}
// eight consecutive k (two float4) -> one operand of v_mfma_f32_16x16x32_bf16 per term
__device__ __forceinline__ Bf16x3 split8(float4 a, float4 b) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    unsigned h[4], m[4], l[4];
    split2(a.x, a.y, h[0], m[0], l[0]); split2(a.z, a.w, h[1], m[1], l[1]);
    split2(b.x, b.y, h[2], m[2], l[2]); split2(b.z, b.w, h[3], m[3], l[3]);
    Bf16x3 r;
    r.hi = __builtin_bit_cast(bf16x8_t, (u4){h[0], h[1], h[2], h[3]});
    r.mid = __builtin_bit_cast(bf16x8_t, (u4){m[0], m[1], m[2], m[3]});
    r.lo = __builtin_bit_cast(bf16x8_t, (u4){l[0], l[1], l[2], l[3]});
    return r;
}
// one stored operand fragment (16 bytes of a lane) from global memory
__device__ __forceinline__ bf16x8_t ldfrag(const char* p) { return __builtin_bit_cast(bf16x8_t, *(const glam_gv4*)(p)); }
// Rows k0 .. k0 + 7 of column position `pos` of a k_ts_gemm weight image (Kp rows in groups of four, MP positions per group; zero beyond
// Kp).  The two loads are UNCONDITIONAL — a row group beyond the image re-reads the last one and is zeroed by w_split8 — so a prologue can
// issue all of a wave's weight loads back to back and wait once: a load under a condition is waited for on the spot (one L2 round trip
// per k step, in series).
struct WRaw8 { float4 a, b; };
__device__ __forceinline__ WRaw8 w_load8(const float* img, int MP, int pos, int k0, int Kp) {
    const int g0 = min(k0, Kp - 4) >> 2, g1 = min(k0 + 4, Kp - 4) >> 2;
    return WRaw8{ld4(img + ((size_t)g0 * MP + pos) * 4), ld4(img + ((size_t)g1 * MP + pos) * 4)};
}

#ifdef GLAM_X3_FAKE_WSPLIT      // timing experiment only (wrong numbers): what the prologue's weight splits cost
__device__ __forceinline__ Bf16x3 split8w(float4 a, float4 b) {
    Bf16x3 r;
    r.hi = __builtin_bit_cast(bf16x8_t, a); r.mid = __builtin_bit_cast(bf16x8_t, b); r.lo = r.hi;
    return r;
}
#else
__device__ __forceinline__ Bf16x3 split8w(float4 a, float4 b) { return split8(a, b); }
#endif
__device__ __forceinline__ Bf16x3 w_split8(const WRaw8& r, int k0, int Kp, bool ok = true) {
    return split8w((ok && k0 < Kp) ? r.a : f4zero(), (ok && k0 + 4 < Kp) ? r.b : f4zero());
}

// acc += a * b over the 32 k of one step, small partial products first (they would lose their low bits against the large ones)
__device__ __forceinline__ v4f_t mfma_x3_small(const Bf16x3& a, const Bf16x3& b, v4f_t acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ v4f_t mfma_x3_mid(const Bf16x3& a, const Bf16x3& b, v4f_t acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.hi, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ v4f_t mfma_x3_big(const Bf16x3& a, const Bf16x3& b, v4f_t acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

}  // namespace glam
