// Per-step integer moments of a batch of degree-1 trajectories — the device half of the variance workflow
// (main_simulate_variance PD:1264-1294 → fl_scaling/est_scaling_params.py calc_nu_chunk :90-94, calc_var_chunk
// :131-138): for every peeling step s, over the trials of the batch,
//     cnt[s] = #{r1[t][s] != 0},  s1[s] = Σ r1[t][s],  s2[s] = Σ r1[t][s]²     (int64, accumulated in place)
// from which the host forms ssquares = Σ_{r1≠0} (r1/M − θ/M)² = (s2 − 2θ·s1 + cnt·θ²)/M² and counts = cnt
// for any theory curve θ.  One thread per step: consecutive threads read consecutive columns (coalesced).
#include "common.h"

namespace {
__global__ void r1_moments_kernel(int ntrials, int ncols, const int32_t *r1, long long *mom)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ncols) return;
    long long c = 0, s1 = 0, s2 = 0;
    for (int t = 0; t < ntrials; t++) {
        const long long v = r1[(size_t)t * ncols + s];
        c += v != 0; s1 += v; s2 += v * v;
    }
    mom[s] += c; mom[(size_t)ncols + s] += s1; mom[2 * (size_t)ncols + s] += s2;
}
}  // namespace

extern "C" int scldpc_r1_moments_device(int32_t ntrials, int32_t ncols, const int32_t *d_r1, int64_t *d_moments,
                                        void *stream)
{
    if (ntrials < 0 || ncols < 0 || ((ntrials > 0 && ncols > 0) && (!d_r1 || !d_moments)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_r1_moments_device: null buffer or negative size");
    if (ntrials == 0 || ncols == 0) return SCLDPC_OK;
    hipLaunchKernelGGL(r1_moments_kernel, dim3((ncols + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       ntrials, ncols, d_r1, reinterpret_cast<long long *>(d_moments));
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}
