// Throughput-mode ensemble + channel sampling on the device (gfx950).
//
// Same ensemble law as the reference's generate_code / channel_doped (BPF:1656-1761, 1547-1574):
// for each of the D = L+dv-1 CN positions one uniformly random permutation of the S = cns_pos*dc
// sockets, socket dv*t+i of position pos+i wired to edge i of VN (pos,t); i.i.d. Bernoulli(eps)
// erasures at the reference's 2^-31 resolution (random()/RAND_MAX, BPF:370); doped positions known.
// The reference's sequential glibc stream cannot be drawn in parallel, so this path is keyed by a
// counter-based generator instead (Philox4x32-10, Salmon et al. SC'11):
//     counter = (index, stream, trial_lo, trial_hi), key = (seed_lo, seed_hi)
//     stream p < D  : word (s&3) of index s>>2 is the sort key of socket s of CN position p
//     stream 2^31   : word (j&3) of index j>>2 is the channel draw of VN j
// A permutation is the rank of each socket's 32-bit key (ties by socket index), computed in LDS by
// one counting pass over the key's top bits plus a tiny in-bucket comparison (buckets hold ~1 key).
// One workgroup samples one trial: it walks the D positions, keeps the last dv permutations in an
// LDS ring, and emits VN position p-dv+1 as whole 16-byte rows, so HBM sees only full-line stores.
// The exact-replay sampler (glibc stream, identical seeds) is glibc_sampler.cpp.
#include "common.h"

namespace {

constexpr int kThreads = 512;
constexpr int kMaxDoped = 32;

struct SArgs {
    int dv, dc, L, cns_pos, vns_pos, n, S, D, nb, shift, lgchunk, dc_shift, nw;
    int ndoped;
    int doped[kMaxDoped];
    uint32_t seed_lo, seed_hi;
    unsigned long long trial0;
    uint32_t thresh;            // erased iff (draw >> 1) < thresh
    int off_gkey, off_gidx, off_win, off_wsum;   // LDS offsets in 32-bit words
    int32_t *vn_adj;
    uint32_t *chan;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// KMAX = Philox calls per thread per permutation (4 sockets each)
template <int KMAX>
__global__ __launch_bounds__(kThreads) void sample_philox_kernel(const SArgs a)
{
    extern __shared__ uint32_t lds[];
    uint32_t *hist = lds;                                           // nb counters → per-slice exclusive prefix
    uint32_t *gkey = lds + a.off_gkey;                              // S keys grouped by bucket
    uint16_t *gidx = reinterpret_cast<uint16_t *>(lds + a.off_gidx);   // S socket ids, same order
    uint16_t *win = reinterpret_cast<uint16_t *>(lds + a.off_win);     // ring of dv × S CN-local ids
    uint32_t *wsum = lds + a.off_wsum;                              // per-wave totals for the scan

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long trial = a.trial0 + blockIdx.x;
    const uint32_t t_lo = (uint32_t)trial, t_hi = (uint32_t)(trial >> 32);
    const int S = a.S, nb = a.nb, dv = a.dv;
    const int ncalls = (S + 3) >> 2;
    int32_t *adj = a.vn_adj + (size_t)blockIdx.x * a.n * dv;

    for (int p = 0; p < a.D; p++) {
        for (int b = tid; b < nb; b += kThreads) hist[b] = 0;
        __syncthreads();

        // ---- keys + bucket histogram; the atomic's return value is the arrival slot in the bucket
        uint32_t key[KMAX * 4], slot[KMAX * 4];
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const int q = tid + k * kThreads;
            if (q < ncalls) {
                uint32_t r[4];
                philox4x32_10((uint32_t)q, (uint32_t)p, t_lo, t_hi, a.seed_lo, a.seed_hi, r);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    key[k * 4 + u] = r[u];
                    if (q * 4 + u < S) slot[k * 4 + u] = atomicAdd(&hist[r[u] >> a.shift], 1u);
                }
            }
        }
        __syncthreads();

        // ---- exclusive scan of the bucket counts.  Each wave scans its own nb/8 contiguous slice
        //      (64 consecutive counters per step: conflict-free); the slice bases go to wpre[] and are
        //      added by the readers:  base(b) = hist[b] + wpre[b >> lgchunk].
        {
            const int chunk = nb >> 3;
            uint32_t carry = 0;
            for (int i = 0; i < chunk; i += 64) {
                const int idx = wave * chunk + i + lane;
                const uint32_t v = hist[idx];
                uint32_t inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t y = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += y;
                }
                hist[idx] = carry + inc - v;
                carry += __shfl(inc, 63, 64);
            }
            if (lane == 0) wsum[wave] = carry;
        }
        __syncthreads();
        uint32_t wpre[8];
        {
            uint32_t acc = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) { wpre[w] = acc; acc += wsum[w]; }
        }
        auto bucket_base = [&](uint32_t b) -> uint32_t {
            if (b >= (uint32_t)nb) return (uint32_t)S;
            const uint32_t h = hist[b], w = b >> a.lgchunk;
            uint32_t add = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) add = (w == (uint32_t)k) ? wpre[k] : add;
            return h + add;
        };

        // ---- group (key, socket) by bucket
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const int q = tid + k * kThreads;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = q * 4 + u;
                if (q < ncalls && s < S) {
                    const uint32_t g = bucket_base(key[k * 4 + u] >> a.shift) + slot[k * 4 + u];
                    gkey[g] = key[k * 4 + u];
                    gidx[g] = (uint16_t)s;
                }
            }
        }
        __syncthreads();

        // ---- rank = bucket base + #(smaller (key, socket) pairs in the bucket); CN-local id = rank / dc
        uint16_t *wp = win + (size_t)(p % dv) * S;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const int q = tid + k * kThreads;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = q * 4 + u;
                if (q < ncalls && s < S) {
                    const uint32_t kk = key[k * 4 + u], b = kk >> a.shift;
                    const uint32_t g0 = bucket_base(b), g1 = bucket_base(b + 1);
                    uint32_t rank = g0;
                    for (uint32_t g = g0; g < g1; g++) {
                        const uint32_t k2 = gkey[g];
                        rank += (k2 < kk) || (k2 == kk && gidx[g] < (uint16_t)s);
                    }
                    wp[s] = (uint16_t)(a.dc_shift >= 0 ? rank >> a.dc_shift : rank / (uint32_t)a.dc);
                }
            }
        }
        __syncthreads();

        // ---- VN position q = p-dv+1 now has all its dv permutations in the ring (BPF:1703-1716)
        const int qpos = p - (dv - 1);
        if (qpos >= 0) {
            for (int t = tid; t < a.vns_pos; t += kThreads) {
                const int j = qpos * a.vns_pos + t;
                if (dv == 4) {
                    int4 v;
                    v.x = (qpos + 0) * a.cns_pos + win[(size_t)((qpos + 0) % 4) * S + 4 * t + 0];
                    v.y = (qpos + 1) * a.cns_pos + win[(size_t)((qpos + 1) % 4) * S + 4 * t + 1];
                    v.z = (qpos + 2) * a.cns_pos + win[(size_t)((qpos + 2) % 4) * S + 4 * t + 2];
                    v.w = (qpos + 3) * a.cns_pos + win[(size_t)((qpos + 3) % 4) * S + 4 * t + 3];
                    reinterpret_cast<int4 *>(adj)[j] = v;
                } else {
                    for (int i = 0; i < dv; i++)
                        adj[(size_t)j * dv + i] = (qpos + i) * a.cns_pos + win[(size_t)((qpos + i) % dv) * S + dv * t + i];
                }
            }
        }
        // (the ring slot the next position overwrites is rewritten only after three more barriers)
    }

    // ---- channel: 32 VNs per output word, 8 Philox calls
    uint32_t *chan = a.chan + (size_t)blockIdx.x * a.nw;
    for (int w = tid; w < a.nw; w += kThreads) {
        uint32_t word = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint32_t r[4];
            philox4x32_10((uint32_t)(w * 8 + c), 0x80000000u, t_lo, t_hi, a.seed_lo, a.seed_hi, r);
#pragma unroll
            for (int u = 0; u < 4; u++) word |= (uint32_t)((r[u] >> 1) < a.thresh) << (c * 4 + u);
        }
        const int j0 = w * 32;
        if (j0 + 32 > a.n) word &= (1u << (a.n - j0)) - 1u;
        for (int d = 0; d < a.ndoped; d++) {                        // doped positions are never erased (BPF:1566-1573)
            const int lo = max(a.doped[d] * a.vns_pos, j0) - j0, hi = min((a.doped[d] + 1) * a.vns_pos, j0 + 32) - j0;
            if (lo < hi) word &= ~(((hi - lo) == 32 ? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u)) << lo);
        }
        chan[w] = word;
    }
}

}  // namespace

extern "C" int scldpc_sample_philox_device(const scldpc_code_params *p, uint64_t seed, uint64_t trial0,
                                           int32_t ntrials, double eps, int32_t ndoped, const int32_t *doped_positions,
                                           int32_t *d_vn_adj, uint32_t *d_chan_bits, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (ntrials < 0 || (ntrials > 0 && (!d_vn_adj || !d_chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_sample_philox_device: null buffer or negative ntrials");
    if (ndoped < 0 || ndoped > kMaxDoped || (ndoped > 0 && !doped_positions))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_sample_philox_device: 0 <= ndoped <= %d", kMaxDoped);
    if (!(eps >= 0.0 && eps <= 1.0))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_sample_philox_device: eps=%g outside [0,1]", eps);
    if (ntrials == 0) return SCLDPC_OK;

    SArgs a{};
    a.dv = p->dv; a.dc = p->dc; a.L = p->L; a.cns_pos = p->cns_pos; a.vns_pos = p->vns_pos;
    a.n = scldpc::n_of(p); a.S = p->cns_pos * p->dc; a.D = p->L + p->dv - 1; a.nw = scldpc::nw_of(p);
    if (a.S > 8192 || p->dv > 8)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE,
                                 "scldpc_sample_philox_device: %d sockets per position > 8192 (LDS-resident ranking)", a.S);
    int lg = 9;                                     // nb = power of two >= max(S, kThreads)
    while ((1 << lg) < a.S) lg++;
    a.nb = 1 << lg; a.shift = 32 - lg; a.lgchunk = lg - 3;
    a.dc_shift = -1;
    for (int k = 0; k < 8; k++) if ((1 << k) == p->dc) a.dc_shift = k;
    a.ndoped = ndoped;
    for (int d = 0; d < ndoped; d++) {
        if (doped_positions[d] < 0 || doped_positions[d] >= p->L)
            return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "doped position %d outside [0,%d)", doped_positions[d], p->L);
        a.doped[d] = doped_positions[d];
    }
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.trial0 = trial0;
    // erased iff r/RAND_MAX < eps with r = 31-bit draw  ⇔  r < ceil(eps * RAND_MAX)   (BPF:370,1554-1562)
    {
        const double x = eps * 2147483647.0;
        double c = (double)(uint64_t)x;
        if (c < x) c += 1.0;
        a.thresh = (uint32_t)c;
    }
    int off = ((a.nb + 1) + 3) & ~3;
    a.off_gkey = off; off += (a.S + 3) & ~3;
    a.off_gidx = off; off += ((a.S + 1) / 2 + 3) & ~3;
    a.off_win = off;  off += (((size_t)p->dv * a.S + 1) / 2 + 3) & ~3;
    a.off_wsum = off; off += 16;
    const size_t lds_bytes = 4u * (size_t)off;
    if (lds_bytes > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_sample_philox_device: needs %zu B of LDS", lds_bytes);
    a.vn_adj = d_vn_adj; a.chan = d_chan_bits;

    const int kmax = ((a.S + 3) / 4 + kThreads - 1) / kThreads;     // 1..4
    void (*kern)(const SArgs) = kmax <= 1 ? sample_philox_kernel<1> : kmax == 2 ? sample_philox_kernel<2>
                                                                                 : sample_philox_kernel<4>;
    SCLDPC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kThreads), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}
