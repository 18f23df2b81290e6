// The CN words of a trial — [cnt:4 | deg:4 | sum of the erased neighbours' ids:24] per check node, in a global-memory workspace —
// built through LDS.
//
// The decoders that keep their CN words in the workspace (ensembles whose words exceed the LDS: full_bp.hip's WideG policy,
// e.g. bp_traj's shipped Def_M = 2500; peel_pick.hip at the notebook's N = 10000) used to count every erased VN into its dv
// CNs with one global atomic per edge on a random word of the trial's words — with thousands of trials doing so at once,
// every atomic is a read-modify-write of a cold 128-byte line in DRAM: 680 of the 1960 ms of a launch of BASELINE config 3,
// half of full_bp's time at N = 5000.  The chain's structure keeps it local instead: VN position q only reaches CN positions
// q .. q + dv - 1 (BPF:1712).  One 1024-thread workgroup per trial sweeps the VN positions with a ring of dv CN positions in
// LDS — LDS atomics — and a CN position leaves for the workspace as whole lines once VN position q has been counted into it:
// the rows are read and the words written once each.
#include "common.h"
#include "kernel_util.h"

namespace {

using namespace scldpc_dev;

struct BArgs {
    int V, C, L, n, nk, nw, deg;                // deg: every neighbour also counts into the degree field (trajectory mode)
    const unsigned long long *rows;             // [T][n] four 16-bit position-local CN ids per VN
    const uint32_t *chan;                       // [T][nw]
    uint32_t *ws;                               // [T][nk]
};

__global__ __launch_bounds__(1024) void cn_build_kernel(const BArgs a)
{
    extern __shared__ uint32_t ring[];                                    // [4][C]: the words of CN positions q .. q + 3
    const int tid = threadIdx.x, trial = blockIdx.x, V = a.V, C = a.C;
    const unsigned long long *rows = a.rows + (size_t)trial * a.n;
    const uint32_t *chan = a.chan + (size_t)trial * a.nw;
    uint32_t *cn = a.ws + (size_t)trial * a.nk;
    for (int i = tid; i < 4 * C; i += 1024) ring[i] = 0;
    __syncthreads();
    for (int q = 0; q < a.L + 3; q++) {
        if (q < a.L) {
            for (int t = tid; t < V; t += 1024) {
                const int j = q * V + t;
                const bool er = (chan[j >> 5] >> (j & 31)) & 1u;
                if (er || a.deg) {
                    const unsigned long long r = rows[j];
                    const uint32_t add = (a.deg ? kDegOne : 0u) + (er ? kCntOne + (uint32_t)j : 0u);
#pragma unroll
                    for (int i = 0; i < 4; i++) atomicAdd(&ring[((q + i) & 3) * C + (int)((r >> (16 * i)) & 0xFFFFull)], add);
                }
            }
            __syncthreads();
        }
        // CN position q has all its neighbours (VN positions q - 3 .. q): out, and its slot cleared for CN position q + 4
        uint32_t *slot = ring + (q & 3) * C;
        for (int c = tid; c < C; c += 1024) {
            if ((size_t)q * C + c < (size_t)a.nk) cn[(size_t)q * C + c] = slot[c];
            slot[c] = 0;
        }
        __syncthreads();
    }
}

}  // namespace

namespace scldpc {

// Launches the build on `stream` and returns true; false (nothing launched) when the ensemble is not the dv = 4 chain with
// 2-byte rows' shape this kernel takes or its ring of four CN positions exceeds the LDS.
bool cn_build_launch(const scldpc_code_params *p, int ntrials, const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                     uint32_t *d_words, bool deg, void *stream)
{
    const size_t ring = 4u * 4u * (size_t)p->cns_pos;
    if (p->dv != 4 || p->cns_pos > 65536 || ring > (size_t)kMaxLdsBytes || ntrials <= 0) return false;
    BArgs a{};
    a.V = p->vns_pos; a.C = p->cns_pos; a.L = p->L; a.n = n_of(p); a.nk = nk_of(p); a.nw = nw_of(p); a.deg = deg ? 1 : 0;
    a.rows = reinterpret_cast<const unsigned long long *>(d_vn_adj16); a.chan = d_chan_bits; a.ws = d_words;
    if (allow_max_lds(reinterpret_cast<const void *>(cn_build_kernel)) != 0) return false;
    hipLaunchKernelGGL(cn_build_kernel, dim3(ntrials), dim3(1024), ring, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess;
}

}  // namespace scldpc
