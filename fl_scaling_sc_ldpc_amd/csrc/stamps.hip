// Diagnostic build only (make -C csrc stamps): where the kernels drop their in-kernel phase stamps.
// Empty in the product build.
#ifdef SCLDPC_STAMPS
#include <hip/hip_runtime.h>
__device__ long long *g_scldpc_stamps = nullptr;
extern "C" int scldpc_debug_set_stamps(long long *d_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_scldpc_stamps), &d_buf, sizeof d_buf) == hipSuccess ? 0 : -3;
}
#endif
