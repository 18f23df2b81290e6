// Error reporting, parameter checks and small queries of libscldpc_hip.so.
#include "common.h"
#include <cstring>

namespace scldpc {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int check_params(const scldpc_code_params *p)
{
    if (!p) return set_error(SCLDPC_ERR_BAD_ARG, "null scldpc_code_params");
    if (p->dv < 1 || p->dc < 1 || p->L < 1 || p->cns_pos < 1 || p->vns_pos < 1)
        return set_error(SCLDPC_ERR_BAD_ARG, "non-positive ensemble parameter (dv=%d dc=%d L=%d cns_pos=%d vns_pos=%d)",
                         p->dv, p->dc, p->L, p->cns_pos, p->vns_pos);
    // generate_code wires socket dv*VNt+i of a position's cns_pos*dc sockets (BPF:1667,1712)
    if ((int64_t)p->dv * p->vns_pos != (int64_t)p->dc * p->cns_pos)
        return set_error(SCLDPC_ERR_BAD_ARG, "dv*vns_pos (%d*%d) must equal dc*cns_pos (%d*%d)",
                         p->dv, p->vns_pos, p->dc, p->cns_pos);
    if ((int64_t)p->vns_pos * p->L > (1ll << 30) || (int64_t)(p->L + p->dv - 1) * p->cns_pos > (1ll << 30))
        return set_error(SCLDPC_ERR_TOO_LARGE, "ensemble too large for 32-bit node ids");
    return SCLDPC_OK;
}

static void *g_ws[16][2] = {{nullptr}};
static size_t g_ws_bytes[16][2] = {{0}};

int workspace(size_t bytes, void **out, int slot)
{
    int dev = 0;
    SCLDPC_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || slot < 0 || slot > 1)
        return set_error(SCLDPC_ERR_BAD_ARG, "device ordinal %d / workspace slot %d out of range", dev, slot);
    if (bytes > g_ws_bytes[dev][slot]) {
        if (g_ws[dev][slot]) {
            SCLDPC_HIP_CHECK(hipDeviceSynchronize());
            SCLDPC_HIP_CHECK(hipFree(g_ws[dev][slot]));
            g_ws[dev][slot] = nullptr; g_ws_bytes[dev][slot] = 0;
        }
        const size_t want = bytes + bytes / 4;
        SCLDPC_HIP_CHECK(hipMalloc(&g_ws[dev][slot], want));
        g_ws_bytes[dev][slot] = want;
    }
    *out = g_ws[dev][slot];
    return SCLDPC_OK;
}

}  // namespace scldpc

extern "C" int scldpc_abi_version(void) { return SCLDPC_ABI_VERSION; }
extern "C" const char *scldpc_last_error(void) { return scldpc::g_err; }

extern "C" int scldpc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
