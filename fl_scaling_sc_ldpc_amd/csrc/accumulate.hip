// plr_computation + willIstop over a batch of per-trial counters, in trial order
// (BPF:1503-1520, 440-451, 2140-2144).  One workgroup; the batch is a few MB at most.
#include "common.h"

namespace {

constexpr int kBlock = 1024;

__global__ __launch_bounds__(kBlock) void accumulate_run_kernel(int ntrials, const int32_t *counters,
                                                                long long stop_frame_err, long long *run)
{
    __shared__ int seg_fe[kBlock];
    __shared__ long long sums[SCLDPC_NRUN];
    const int tid = threadIdx.x;
    const int seg = (ntrials + kBlock - 1) / kBlock;
    const int t0 = min(tid * seg, ntrials), t1 = min(t0 + seg, ntrials);
    if (tid < SCLDPC_NRUN) sums[tid] = 0;

    int fe = 0;
    for (int t = t0; t < t1; t++) fe += counters[(size_t)t * SCLDPC_NCOUNTERS + SCLDPC_C_NUM_ERASURES] > 0;
    seg_fe[tid] = fe;
    __syncthreads();
    // exclusive prefix over the 1024 segment counts (Hillis–Steele in place)
    for (int o = 1; o < kBlock; o <<= 1) {
        const int v = tid >= o ? seg_fe[tid - o] : 0;
        __syncthreads();
        seg_fe[tid] += v;
        __syncthreads();
    }
    const long long before = (long long)seg_fe[tid] - fe;           // frame errors in earlier segments
    const long long already = run[SCLDPC_R_FRAME_ERR];
    const long long need = stop_frame_err > 0 ? stop_frame_err - already : (1ll << 62);

    long long loc[SCLDPC_NRUN] = {0};
    if (need > 0 && before < need) {
        long long seen = before;
        for (int t = t0; t < t1; t++) {
            const int32_t *c = counters + (size_t)t * SCLDPC_NCOUNTERS;
            const int ne = c[SCLDPC_C_NUM_ERASURES], ee = c[SCLDPC_C_NUM_ERASURES_EXP];
            if (ne > 0) {                                            // BPF:1506-1511
                loc[SCLDPC_R_USERS_ERR] += ne;
                loc[SCLDPC_R_FRAME_ERR] += 1;
                loc[SCLDPC_R_BLOCK_ERR] += c[SCLDPC_C_NUM_BLOCKS_ERR];
            }
            if (ee > 0) {                                            // BPF:1512-1517
                loc[SCLDPC_R_USERS_ERR_EXP] += ee;
                loc[SCLDPC_R_FRAME_ERR_EXP] += 1;
                loc[SCLDPC_R_BLOCK_ERR_EXP] += c[SCLDPC_C_NUM_BLOCKS_ERR_EXP];
            }
            if (c[SCLDPC_C_NUM_ERASURES_P1] > 0) loc[SCLDPC_R_FRAME_ERR_P1] += 1;   // BPF:1518-1519
            loc[SCLDPC_R_FRAMES] += 1;
            loc[SCLDPC_R_ITERATIONS] += c[SCLDPC_C_ITERATIONS];
            if (ne > 0 && ++seen >= need) break;                     // willIstop: this frame is the last (BPF:2143-2144)
        }
    }
    for (int k = 0; k < SCLDPC_NRUN; k++)
        if (loc[k]) atomicAdd(reinterpret_cast<unsigned long long *>(&sums[k]), (unsigned long long)loc[k]);
    __syncthreads();
    if (tid < SCLDPC_NRUN) run[tid] += sums[tid];
}

}  // namespace

extern "C" int scldpc_accumulate_run_device(int32_t ntrials, const int32_t *d_counters, int64_t stop_frame_err,
                                            int64_t *d_run, void *stream)
{
    if (ntrials < 0 || !d_run || (ntrials > 0 && !d_counters))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_accumulate_run_device: null buffer or negative ntrials");
    if (ntrials == 0) return SCLDPC_OK;
    hipLaunchKernelGGL(accumulate_run_kernel, dim3(1), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       ntrials, d_counters, (long long)stop_frame_err, reinterpret_cast<long long *>(d_run));
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}
