// libglam_hip.so — version and per-thread error reporting of the C ABI (include/glam_hip.h).
#include "dense.h"

#include <stdlib.h>
#include <string.h>

namespace glam {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- per-launch kernel timing ----
bool g_prof_on = false;
const char* g_prof_label = nullptr;
namespace {
struct ProfRec { char name[96]; unsigned grid; hipEvent_t e0, e1; };
ProfRec* g_prof = nullptr;
int g_prof_cap = 0, g_prof_n = 0, g_prof_events = 0;
}  // namespace

bool prof_slot(const char* kernel, unsigned grid, hipEvent_t* start, hipEvent_t* stop) {
    if (g_prof_n >= g_prof_cap) return false;          // full: the launch goes out untimed
    ProfRec& r = g_prof[g_prof_n];
    if (g_prof_n >= g_prof_events) {                   // event pairs are created once and reused by later sessions
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return false;
        g_prof_events = g_prof_n + 1;
    }
    // "(k_name<A, B>)" -> "k_name<A, B>"
    const char* b = g_prof_label ? g_prof_label : kernel;
    g_prof_label = nullptr;
    while (*b == '(' || *b == ' ') ++b;
    size_t n = strlen(b);
    while (n > 0 && (b[n - 1] == ')' || b[n - 1] == ' ')) --n;
    if (n >= sizeof(r.name)) n = sizeof(r.name) - 1;
    memcpy(r.name, b, n);
    r.name[n] = 0;
    r.grid = grid;
    *start = r.e0;
    *stop = r.e1;
    ++g_prof_n;
    return true;
}

}  // namespace glam

extern "C" int glam_prof_begin(int capacity) {
    using namespace glam;
    if (capacity <= 0) return fail(GLAM_E_INVALID, "glam_prof_begin: capacity must be positive");
    if (capacity > g_prof_cap) {
        ProfRec* p = static_cast<ProfRec*>(realloc(g_prof, sizeof(ProfRec) * (size_t)capacity));
        if (!p) return fail(GLAM_E_INVALID, "glam_prof_begin: out of host memory");
        g_prof = p;
        g_prof_cap = capacity;
    }
    g_prof_n = 0;
    g_prof_on = true;
    return GLAM_OK;
}

extern "C" int glam_prof_end(void) {
    glam::g_prof_on = false;
    return glam::g_prof_n;
}

extern "C" int glam_prof_read(int i, char* name_host, int name_cap, int32_t* grid_host, float* usec_host) {
    using namespace glam;
    if (i < 0 || i >= g_prof_n || !name_host || name_cap <= 0 || !grid_host || !usec_host)
        return fail(GLAM_E_INVALID, "glam_prof_read: bad record index / null pointer");
    ProfRec& r = g_prof[i];
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.e0, r.e1);
    if (e != hipSuccess) return fail(GLAM_E_HIP, "glam_prof_read: %s", hipGetErrorString(e));
    snprintf(name_host, (size_t)name_cap, "%s", r.name);
    *grid_host = (int32_t)r.grid;
    *usec_host = ms * 1e3f;
    return GLAM_OK;
}

extern "C" int glam_abi_version(void) { return GLAM_ABI_VERSION; }
extern "C" const char* glam_last_error(void) { return glam::g_err; }

// ------------------------------------------------------------------------------------------------
// Zero-padded copies of a module's parameters in ONE launch (and their gradients back in one).
// The kernels work on rows of Cp = ceil4(C) floats; four of the five hidden widths of the reference's search space
// (src_1gp/glam.py:60: 15 / 30 / 45 / 60 / 90) are not multiples of four, so a GRU's gate matrices [3C, C] -> [3Cp, Cp] and
// a Linear's [M, K] -> [Mp, Kp] have to be re-laid once per model pass.  As library pads that was a fill + a copy per
// tensor forward and a slice copy per tensor backward: 18 launches per training step at ~4.7 us each.
// Tensor t is [d0, d1, d2] (row-major), its padded form [d0, p1, p2] with p1 >= d1, p2 >= d2.
// ------------------------------------------------------------------------------------------------
namespace glam {
constexpr int kPadGroupMax = 8;
struct PadGroupArgs {
    const float* src[kPadGroupMax];
    float* dst[kPadGroupMax];
    int d1[kPadGroupMax], d2[kPadGroupMax], p1[kPadGroupMax], p2[kPadGroupMax];
    int end[kPadGroupMax];          // running element count of the side the kernel iterates over (padded forward, plain backward)
    int n;
};

// forward: dst[t] (padded) <- src[t] (plain), zeros in the pad; backward: dst[t] (plain gradient) <- src[t] (padded gradient, or
// zeros when that gradient does not exist)
template <bool BWD>
__global__ void __launch_bounds__(kBlock) k_pad_group(PadGroupArgs a) {
    const int total = a.end[a.n - 1];
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < total; idx += gridDim.x * kBlock) {
        int t = 0;
        while (idx >= a.end[t]) ++t;
        const int e = idx - (t ? a.end[t - 1] : 0);
        const int d1 = a.d1[t], d2 = a.d2[t], p1 = a.p1[t], p2 = a.p2[t];
        if (BWD) {
            const int c = e % d2, r = (e / d2) % d1, o = e / (d2 * d1);
            a.dst[t][e] = a.src[t] ? a.src[t][((size_t)o * p1 + r) * p2 + c] : 0.f;
        } else {
            const int c = e % p2, r = (e / p2) % p1, o = e / (p2 * p1);
            a.dst[t][e] = (r < d1 && c < d2) ? a.src[t][((size_t)o * d1 + r) * d2 + c] : 0.f;
        }
    }
}
}  // namespace glam

extern "C" int glam_pad_group(int n, const float* const* src, float* const* dst, const int32_t* dims, int backward, void* stream) {
    using namespace glam;
    if (n == 0) return GLAM_OK;
    GLAM_REQUIRE(n > 0 && n <= kPadGroupMax, "glam_pad_group: n=%d not in 1..%d", n, kPadGroupMax);
    GLAM_REQUIRE(src && dst && dims, "glam_pad_group: null pointer");
    PadGroupArgs a{};
    int64_t run = 0;
    for (int t = 0; t < n; ++t) {
        const int32_t* d = dims + 5 * t;            // d0 d1 d2 p1 p2
        GLAM_REQUIRE(d[0] > 0 && d[1] > 0 && d[2] > 0 && d[3] >= d[1] && d[4] >= d[2], "glam_pad_group: tensor %d has dims %d %d %d -> %d %d",
                     t, d[0], d[1], d[2], d[3], d[4]);
        GLAM_REQUIRE(dst[t] && (backward || src[t]), "glam_pad_group: tensor %d: null pointer", t);
        a.src[t] = src[t]; a.dst[t] = dst[t];
        a.d1[t] = d[1]; a.d2[t] = d[2]; a.p1[t] = d[3]; a.p2[t] = d[4];
        run += backward ? (int64_t)d[0] * d[1] * d[2] : (int64_t)d[0] * d[3] * d[4];
        if (run >= INT32_MAX) return fail(GLAM_E_UNSUPPORTED, "glam_pad_group: more than 2^31 elements");
        a.end[t] = (int)run;
    }
    a.n = n;
    const int grid = grid_for(run, kBlock);
    if (backward) hipLaunchKernelGGL(k_pad_group<true>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_pad_group<false>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    GLAM_LAUNCH_CHECK("glam_pad_group");
    return GLAM_OK;
}

// ------------------------------------------------------------------------------------------------
// Content fingerprint of a batch's index tensors (glam_batch_fingerprint in include/glam_hip.h): a 64-bit hash of up to four device
// buffers, position-sensitive (every 32-bit word is mixed with its index and its buffer's number before the commutative combine), one
// launch.  The graphed-callable route of glam_amd.graphs recognises a batch it has staged and captured before by it — the reference's
// trainer hands the model a freshly collated device copy every iteration (src_1gp/trainer.py:294), so object identity says nothing.
// ------------------------------------------------------------------------------------------------
namespace glam {
struct FingerprintArgs { const uint32_t* p[4]; unsigned long long words[4]; int n; unsigned long long* out; };
__device__ __forceinline__ unsigned long long fp_mix(unsigned long long z) {        // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(kBlock) k_fingerprint(FingerprintArgs a) {
    __shared__ unsigned long long s_part[kBlock / 64];
    unsigned long long h = 0;
    for (int t = 0; t < a.n; ++t) {
        const unsigned long long salt = 0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1);
        for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < a.words[t]; i += (unsigned long long)gridDim.x * kBlock)
            h += fp_mix(((unsigned long long)a.p[t][i] << 32 | (i & 0xffffffffull)) ^ salt ^ (i >> 32));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += s_part[w];
        atomicAdd(a.out, s);             // integer addition: commutative, the result does not depend on the block order
    }
}
}  // namespace glam

extern "C" int glam_batch_fingerprint(int n, const void* const* bufs, const int64_t* nbytes, uint64_t* out_dev, void* stream) {
    using namespace glam;
    GLAM_REQUIRE(n >= 1 && n <= 4 && bufs && nbytes && out_dev, "glam_batch_fingerprint: n=%d not in 1..4 / null pointer", n);
    FingerprintArgs a{};
    unsigned long long total = 0;
    for (int t = 0; t < n; ++t) {
        GLAM_REQUIRE(nbytes[t] >= 0 && (nbytes[t] & 3) == 0 && (nbytes[t] == 0 || bufs[t]) && (reinterpret_cast<uintptr_t>(bufs[t]) & 3u) == 0,
                     "glam_batch_fingerprint: buffer %d must be 4-byte aligned with a size that is a multiple of 4", t);
        a.p[t] = static_cast<const uint32_t*>(bufs[t]);
        a.words[t] = (unsigned long long)nbytes[t] / 4;
        total += a.words[t];
    }
    a.n = n;
    a.out = reinterpret_cast<unsigned long long*>(out_dev);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out_dev, 0, sizeof(uint64_t), s) != hipSuccess) return fail(GLAM_E_HIP, "glam_batch_fingerprint: memset failed");
    if (total == 0) return GLAM_OK;
    hipLaunchKernelGGL(k_fingerprint, dim3(grid_for((int64_t)total, 4 * kBlock, 256)), dim3(kBlock), 0, s, a);
    GLAM_LAUNCH_CHECK("glam_batch_fingerprint");
    return GLAM_OK;
}
