// Kernel instantiations of triplet_kernels.h for heads = 3 (one translation unit per head count so that the
// ~100 template variants compile in parallel).
#include "triplet_kernels.h"

namespace glam {
bool triplet_launch_h3(int kind, int De, int emul, Shape sh, const void* args, int nodes, size_t lds, hipStream_t s,
                        int cap, int* grid_out) {
    return triplet_launch_impl<3>(kind, De, emul, sh, args, nodes, lds, s, cap, grid_out);
}
}  // namespace glam

#ifdef GLAM_B1_PROF
extern "C" int glam_debug_b1_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_b1_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

#ifdef GLAM_FWD_PROF
extern "C" int glam_debug_fwd_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_fwd_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
