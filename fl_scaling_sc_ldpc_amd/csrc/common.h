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

// Caller-owned device scratch (CN words of ensembles beyond the LDS budget, the big-ensemble sampler's tables): the entry
// points that can need it take (d_workspace, workspace_bytes) and scldpc_workspace_bytes() says how much — the library
// allocates nothing and keeps no state between calls.  Launch helpers take a Scratch: in query mode they only report.
struct Scratch {
    void *ptr; uint64_t bytes;      // what the caller passed
    uint64_t *query;                // non-null: store the requirement there and return before launching
};
// out = the caller's buffer if it holds `need` bytes; error otherwise.  need == 0: out = nullptr.
int take_scratch(const char *who, const Scratch &s, size_t need, void **out);

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

// Lifts a kernel's dynamic-LDS limit to the CU's 160 KiB, ONCE per kernel and process, under a lock.  The attribute lives on
// the process-global function object: set per call to the launch's own size, two host threads launching the same kernel for
// different ensembles could lower it between the other thread's set and its launch.  0 or SCLDPC_ERR_HIP.
int allow_max_lds(const void *kernel);
// cn_build.hip: the CN words [T][nk] of the dv = 4 chain with 2-byte rows built through an LDS ring; false = not applicable
bool cn_build_launch(const scldpc_code_params *p, int ntrials, const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                     uint32_t *d_words, bool deg, void *stream);

// Diagnostics only (tools/ab_occupancy.py): extra bytes of dynamic LDS per workgroup from the environment variable
// SCLDPC_DEBUG_LDS_PAD_<which>, to measure a kernel at fewer workgroups per CU than its own footprint allows.  0 if unset.
size_t debug_lds_pad(const char *which);
// Diagnostics only: SCLDPC_DEBUG_GRID_<which> = number of workgroups of a persistent launch (each loops over its trials);
// unset or out of range: one workgroup per trial.
int debug_grid(const char *which, int ntrials);

#define SCLDPC_HIP_CHECK(expr)                                                                     \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::scldpc::set_error(SCLDPC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace scldpc

// per-file requirement queries behind scldpc_workspace_bytes (each runs its launch function's own decision logic)
int64_t scldpc_full_bp_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t want_rows);
int64_t scldpc_sw_bp_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t W);
int64_t scldpc_peel_sweep_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t adj16);
int64_t scldpc_peel_pick_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t total_size, int32_t rng_mt);
int64_t scldpc_sample_workspace_query(const scldpc_code_params *p, int32_t ntrials);
