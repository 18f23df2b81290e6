// Error reporting, parameter checks and small queries of libscldpc_hip.so.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

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

int allow_max_lds(const void *kernel)
{
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;      // (kernel, device): the attribute is kept per device
    int dev = 0;
    SCLDPC_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &kd : done)
        if (kd.first == kernel && kd.second == dev) return SCLDPC_OK;
    SCLDPC_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes));
    done.emplace_back(kernel, dev);
    return SCLDPC_OK;
}

size_t debug_lds_pad(const char *which)
{
    char name[64];
    snprintf(name, sizeof name, "SCLDPC_DEBUG_LDS_PAD_%s", which);
    const char *v = getenv(name);
    if (!v) return 0;
    const long x = strtol(v, nullptr, 10);
    return x > 0 && x < kMaxLdsBytes ? ((size_t)x + 15) & ~(size_t)15 : 0;
}

int debug_grid(const char *which, int ntrials)
{
    char name[64];
    snprintf(name, sizeof name, "SCLDPC_DEBUG_GRID_%s", which);
    const char *v = getenv(name);
    if (!v) return ntrials;
    const long x = strtol(v, nullptr, 10);
    return x > 0 && x < ntrials ? (int)x : ntrials;
}

int take_scratch(const char *who, const Scratch &s, size_t need, void **out)
{
    *out = nullptr;
    if (need == 0) return SCLDPC_OK;
    if (!s.ptr || s.bytes < need)
        return set_error(SCLDPC_ERR_BAD_ARG, "%s: this ensemble needs %zu bytes of device workspace (scldpc_workspace_bytes), "
                         "the caller passed %llu", who, need, (unsigned long long)(s.ptr ? s.bytes : 0));
    if (reinterpret_cast<uintptr_t>(s.ptr) & 255u)
        return set_error(SCLDPC_ERR_BAD_ARG, "%s: the workspace must be 256-byte aligned", who);
    *out = s.ptr;
    return SCLDPC_OK;
}

}  // namespace scldpc

extern "C" int64_t scldpc_workspace_bytes(int32_t op, const scldpc_code_params *p, int32_t ntrials, int32_t arg0, int32_t arg1)
{
    switch (op) {
    case SCLDPC_WS_SAMPLE:     return scldpc_sample_workspace_query(p, ntrials);
    case SCLDPC_WS_FULL_BP:    return scldpc_full_bp_workspace_query(p, ntrials, arg0);
    case SCLDPC_WS_SW_BP:      return scldpc_sw_bp_workspace_query(p, ntrials, arg0);
    case SCLDPC_WS_PEEL_SWEEP: return scldpc_peel_sweep_workspace_query(p, ntrials, arg0);
    case SCLDPC_WS_PEEL_PICK:  return scldpc_peel_pick_workspace_query(p, ntrials, arg0, arg1);
    }
    return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_workspace_bytes: unknown operation %d", op);
}

extern "C" int scldpc_abi_version(void) { return SCLDPC_ABI_VERSION; }
extern "C" const char *scldpc_last_error(void) { return scldpc::g_err; }

extern "C" int scldpc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
