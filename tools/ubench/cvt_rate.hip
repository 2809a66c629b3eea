// Issue rate of the instructions of the 3 x bf16 split (bf16x3.h: split2) on one wave: cycles per instruction for a chain-free stream of
// v_cvt_pk_bf16_f32, v_sub_f32, v_and_b32, v_lshlrev_b32, and for split2 itself (per pair of floats).
// build: hipcc --offload-arch=gfx950 -O3 -o cvt_rate cvt_rate.hip ; run: ./cvt_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, float seed) {
    float a[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed * (threadIdx.x + i + 1); u[i] = __builtin_bit_cast(unsigned, a[i]); }
    long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            if (OP == 0) { const f32x2_t v = {a[i], a[i + 1]}; u[i] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); a[i] = __builtin_bit_cast(float, u[i]); }
            if (OP == 1) { a[i] = a[i] - a[i + 1]; }
            if (OP == 2) { u[i] = u[i] & u[i + 1]; u[i + 1] ^= 0x10001u; }
            if (OP == 3) { u[i] = (u[i] << 16) | 1u; }
            if (OP == 4) {      // split2
                const f32x2_t v = {a[i], a[i + 1]};
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                const f32x2_t r1 = {a[i] - __builtin_bit_cast(float, h << 16), a[i + 1] - __builtin_bit_cast(float, h & 0xffff0000u)};
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
                const f32x2_t r2 = {r1[0] - __builtin_bit_cast(float, m << 16), r1[1] - __builtin_bit_cast(float, m & 0xffff0000u)};
                const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
                a[i] = __builtin_bit_cast(float, h ^ m); a[i + 1] = __builtin_bit_cast(float, l ^ u[i]);
            }
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + __builtin_bit_cast(float, u[i]);
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> static void run(const char* what, float* out, long long* cyc, int per_iter) {
    long long h = 0;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, cyc, 1.37f);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8lld cycles for 256 x %d  -> %.2f cycles each (one wave alone on its SIMD)\n", what, h, per_iter, (double)h / (256.0 * per_iter));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    run<0>("v_cvt_pk_bf16_f32 (+ xor)", out, cyc, 8);
    run<1>("v_sub_f32", out, cyc, 8);
    run<2>("v_and_b32 (+ xor)", out, cyc, 8);
    run<3>("v_lshl_or_b32", out, cyc, 8);
    run<4>("split2 (a pair of floats)", out, cyc, 8);
    return 0;
}
