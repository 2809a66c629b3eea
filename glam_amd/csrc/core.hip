// libglam_hip.so — version and per-thread error reporting of the C ABI (include/glam_hip.h).
#include "common.h"

namespace glam {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace glam

extern "C" int glam_abi_version(void) { return GLAM_ABI_VERSION; }
extern "C" const char* glam_last_error(void) { return glam::g_err; }
