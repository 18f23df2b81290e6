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

// simulate_sc_ldpc's per-trial bookkeeping and stop rule, in trial order (PD:668-699): rows of scldpc_peel_sweep_device.
// Same shape as accumulate_run_kernel: a prefix count of the failed trials per segment finds where max_fuckups trips.
__global__ __launch_bounds__(kBlock) void accumulate_peel_kernel(int ntrials, const int32_t *out, long long max_fuckups,
                                                                 long long *run)
{
    __shared__ int seg_f[kBlock];
    __shared__ long long sums[SCLDPC_NPEELRUN];
    const int tid = threadIdx.x;
    const int seg = (ntrials + kBlock - 1) / kBlock;
    const int t0 = min(tid * seg, ntrials), t1 = min(t0 + seg, ntrials);
    if (tid < SCLDPC_NPEELRUN) sums[tid] = 0;
    int f = 0;
    for (int t = t0; t < t1; t++) f += out[(size_t)t * 8] >= 1;
    seg_f[tid] = f;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {
        const int v = tid >= o ? seg_f[tid - o] : 0;
        __syncthreads();
        seg_f[tid] += v;
        __syncthreads();
    }
    const long long before = (long long)seg_f[tid] - f;
    const long long need = max_fuckups > 0 ? max_fuckups - run[SCLDPC_PR_FUCKUPS] : (1ll << 62);
    long long loc[SCLDPC_NPEELRUN] = {0};
    if (need > 0 && before < need) {
        long long seen = before;
        for (int t = t0; t < t1; t++) {
            const int32_t *c = out + (size_t)t * 8;
            const int lost = c[0], lost_exp = c[1];
            loc[SCLDPC_PR_TRIALS] += 1;
            loc[SCLDPC_PR_FUCKUPS] += lost >= 1;                       // PD:668-670
            loc[SCLDPC_PR_LOST] += lost;
            loc[SCLDPC_PR_FUCKUPS_EXP] += lost_exp > 0;                // PD:691-693
            loc[SCLDPC_PR_LOST_EXP] += lost_exp;
            loc[SCLDPC_PR_BLOCKS_EXP] += c[2];
            if (lost >= 1 && ++seen >= need) break;                    // PD:698: this trial is the last
        }
    }
    for (int k = 0; k < SCLDPC_NPEELRUN; k++)
        if (loc[k]) atomicAdd(reinterpret_cast<unsigned long long *>(&sums[k]), (unsigned long long)loc[k]);
    __syncthreads();
    if (tid < SCLDPC_NPEELRUN) run[tid] += sums[tid];
}

// soft doping (gen_users_sc_ldpc_doping, PD:176-183): VNs lo .. hi-1 of every trial are never erased
__global__ void clear_channel_range_kernel(int ntrials, int nw, int lo, int hi, uint32_t *chan)
{
    const int w0 = lo >> 5, w1 = (hi - 1) >> 5, nword = w1 - w0 + 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)ntrials * nword) return;
    const int t = (int)(i / nword), w = w0 + (int)(i % nword);
    uint32_t keep = 0;
    if (w == w0) keep |= (1u << (lo & 31)) - 1u;
    if (w == w1 && (hi & 31)) keep |= ~((1u << (hi & 31)) - 1u);
    chan[(size_t)t * nw + w] &= keep;
}

}  // namespace

extern "C" int scldpc_accumulate_peel_device(int32_t ntrials, const int32_t *d_out, int64_t max_fuckups, int64_t *d_run,
                                             void *stream)
{
    if (ntrials < 0 || !d_run || (ntrials > 0 && !d_out))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_accumulate_peel_device: null buffer or negative ntrials");
    if (ntrials == 0) return SCLDPC_OK;
    hipLaunchKernelGGL(accumulate_peel_kernel, dim3(1), dim3(kBlock), 0, static_cast<hipStream_t>(stream), ntrials, d_out,
                       (long long)max_fuckups, reinterpret_cast<long long *>(d_run));
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_clear_channel_range_device(const scldpc_code_params *p, int32_t ntrials, int32_t vn_lo, int32_t vn_hi,
                                                 uint32_t *d_chan_bits, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    const int n = scldpc::n_of(p);
    if (ntrials < 0 || vn_lo < 0 || vn_hi > n || vn_lo > vn_hi || (ntrials > 0 && !d_chan_bits))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_clear_channel_range_device: need 0 <= vn_lo <= vn_hi <= n = %d "
                                 "(got %d, %d) and a buffer", n, vn_lo, vn_hi);
    if (ntrials == 0 || vn_lo == vn_hi) return SCLDPC_OK;
    const long long items = (long long)ntrials * (((vn_hi - 1) >> 5) - (vn_lo >> 5) + 1);
    hipLaunchKernelGGL(clear_channel_range_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ntrials, scldpc::nw_of(p), vn_lo, vn_hi, d_chan_bits);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

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
