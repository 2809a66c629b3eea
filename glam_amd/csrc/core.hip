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
