// Shared host-side helpers of libscldpc_hip.so (error reporting, parameter checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/scldpc.h"

namespace scldpc {

int set_error(int code, const char *fmt, ...);

inline int n_of(const scldpc_code_params *p)  { return p->vns_pos * p->L; }
inline int nk_of(const scldpc_code_params *p) { return (p->L + p->dv - 1) * p->cns_pos; }
inline int nw_of(const scldpc_code_params *p) { return (n_of(p) + 31) / 32; }

// 0 when the geometry is one the reference's generate_code can produce (BPF:1656-1716).
int check_params(const scldpc_code_params *p);

// Library-owned device scratch (CN words of ensembles beyond the LDS budget): one allocation per process and
// device and slot, grown on demand (never inside a stream capture: a growth synchronises the device before freeing).
// slot 0: CN words of the decoders; slot 1: the big-ensemble sampler's scratch (the two may run on different streams).
int workspace(size_t bytes, void **out, int slot = 0);

// x / d == umulhi(x, magic) for every x < limit?  (monotone step function: checking the steps suffices)
inline bool magic_of(int d, int64_t limit, uint32_t *magic)
{
    *magic = (uint32_t)((1ull << 32) / (uint32_t)d) + 1u;
    for (int64_t q = 0; q * d < limit + d; q++) {
        const uint64_t x0 = (uint64_t)q * d, x1 = x0 ? x0 - 1 : 0;
        if (((x0 * *magic) >> 32) != (uint64_t)q || ((x1 * *magic) >> 32) != x1 / (uint64_t)d) return false;
    }
    return true;
}

constexpr int kMaxLdsBytes = 160 * 1024;   // gfx950: 160 KiB per CU, one workgroup may take all of it

#define SCLDPC_HIP_CHECK(expr)                                                                     \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::scldpc::set_error(SCLDPC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace scldpc
