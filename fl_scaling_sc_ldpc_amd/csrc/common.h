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

constexpr int kMaxLdsBytes = 160 * 1024;   // gfx950: 160 KiB per CU, one workgroup may take all of it

#define SCLDPC_HIP_CHECK(expr)                                                                     \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::scldpc::set_error(SCLDPC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace scldpc
