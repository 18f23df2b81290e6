// The fixpoint of unlimited flooding BP (decodeBP, BPF:900-1140) from the VN -> CN table alone — gfx950.
//
// full_bp_small.hip keeps four bits per CN and, when a count drops to one, gathers the CN's row of the CN -> VN table to
// find the neighbour that is still erased: two dependent gathers per release (CN row, then the VN's row), 25 k line requests
// per trial of BASELINE's C2 — the fabric's random-request ceiling.  Here a CN keeps
//
//        field = (number of erased neighbours) + 8 * (sum of the erased neighbours' sockets)
//
// where the socket of edge i of VN t of a position is s = dv * t + i < 2048.  A release subtracts 8 s + 1 with ONE returning
// LDS atomic — count and sum change together, so whoever takes a count to one reads the last neighbour's socket out of the
// same returned word, turns it into a VN index (CN position - (s & 3), VN s >> 2) and queues THAT.  No CN -> VN table is
// read (the sampler need not write it), a release is ONE gather (the VN's 8-byte row), half the dependent round trips
// per level and 13 k line requests per trial.
//
// Two CNs share a 32-bit LDS word: the even one in bits 0-16 (count <= 8 and sum <= 8 * 2047 never carry out of 17 bits,
// and a subtraction takes away what an addition put there, so the field never borrows), the odd one in bits 17-31, where
// carries and borrows fall off the top of the word: its field is kept modulo 2^15, enough to hold 8 s + 1 when the count
// is one.  A count of eight reads as zero with the sum one too high — the field is one integer, only "count == 1 (mod 8)"
// and "count == 2 (mod 8)" are ever asked, and nine or ten erased neighbours do not exist.  26.5 KiB of CN words + 3 KiB of
// VN bits + queues: five trials per CU (four with the wider queues; scldpc_full_bp_vn16_supported).
//
// Safe without a barrier per level: a VN is released by whoever clears its bit in U first (atomic test-and-clear); a
// queue entry whose VN is already claimed is dropped.  The field a releasing thread gets back is exact for the moment of
// its atomic: every other release of that CN is either wholly in it or wholly not.
//
// Outputs: the counters of scldpc_full_bp_fixpoint_device (everything decodeBP reports except the iteration count), bit for
// bit (tests/test_gpu_v2.py).  Size-2 stopping sets (BPF:1067-1133): a CN with two erased neighbours names the other one as
// sum - own socket.
#include "common.h"
#include "kernel_util.h"
#include <algorithm>

namespace {

using namespace scldpc_dev;

enum { SC_NE = 0, SC_REM, SC_N0, SC_N1, SC_OVF, SC_Q, SC_N = 8 };

struct SumArgs {
    int L, V, C, n, nk, cn_lim, nw, nsw;            // nsw = words of two CN fields
    uint32_t magic_v, magic_c;
    int ntrials;
    int kswitch;                                    // frontier width below which the waves go private
    int off_U, off_q0, off_q1, off_pos, off_scal, total, qcap;      // LDS offsets in 32-bit words; qcap in entries (u16)
    const uint16_t *vn_adj16;                       // [T][n][4]   CN index local to its position
    const uint32_t *chan;
    int32_t *counters;
    uint32_t *erased_out;
};

constexpr uint32_t kLowMask = 0x1FFFFu;             // the even CN's field
constexpr int kHighShift = 17;

__device__ __forceinline__ uint32_t field_of(uint32_t word, int c) { return (c & 1) ? word >> kHighShift : word & kLowMask; }
__device__ __forceinline__ uint32_t amount_of(int c, uint32_t s) { return (8u * s + 1u) << ((c & 1) * kHighShift); }

template <int BLOCK, int PER_CU>
__global__ __launch_bounds__(BLOCK, PER_CU) __attribute__((amdgpu_num_sgpr(96))) void full_bp_sum_kernel(const SumArgs a)
{
    constexpr int kWaves = BLOCK / 64;
    extern __shared__ uint32_t lds[];
    uint32_t *cw = lds;                                                  // nsw words
    uint32_t *U = lds + a.off_U;
    uint16_t *q[2] = {reinterpret_cast<uint16_t *>(lds + a.off_q0), reinterpret_cast<uint16_t *>(lds + a.off_q1)};
    int *pos_cnt = reinterpret_cast<int *>(lds + a.off_pos);
    int *pos_ss = pos_cnt + a.L;
    int *scal = reinterpret_cast<int *>(lds + a.off_scal);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int trial = blockIdx.x;
    const int n = a.n, cn_lim = a.cn_lim, nw = a.nw, V = a.V, C = a.C, L = a.L, qcap = a.qcap;
    const uint2 *vrow = reinterpret_cast<const uint2 *>(a.vn_adj16) + (size_t)trial * n;
    const uint32_t *ch = a.chan + (size_t)trial * nw;

    // ---- channel bits, clear the fields ---------------------------------------------------------------------------
    for (int c = tid; c < a.nsw; c += BLOCK) cw[c] = 0;
    int ne_local = 0;
    for (int w = tid; w < nw; w += BLOCK) {
        uint32_t x = ch[w];
        if (w == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        U[w] = x;
        ne_local += __popc(x);
    }
    if (tid < SC_N) scal[tid] = 0;
    for (int i = tid; i < 2 * L; i += BLOCK) pos_cnt[i] = 0;
    __syncthreads();
    ne_local = wave_sum(ne_local);
    if (lane == 0 && ne_local) atomicAdd(&scal[SC_NE], ne_local);

    // the four CNs of VN j and the amounts its edges contribute to their fields
    auto edges = [&](int j, const uint2 r, int &pos, int (&cc)[4], uint32_t (&am)[4]) {
        pos = (int)__umulhi((uint32_t)j, a.magic_v);
        const uint32_t s0 = 4u * (uint32_t)(j - pos * V);
        const int base = pos * C;
        cc[0] = base + (int)(r.x & 0xFFFFu); cc[1] = base + C + (int)(r.x >> 16);
        cc[2] = base + 2 * C + (int)(r.y & 0xFFFFu); cc[3] = base + 3 * C + (int)(r.y >> 16);
#pragma unroll
        for (int i = 0; i < 4; i++) am[i] = amount_of(cc[i], s0 + (uint32_t)i);
    };

    // ---- build: every erased VN adds itself to its 4 CNs; rows are loaded unconditionally (coalesced 8-B loads) ------
    for (int j0 = tid; j0 < n; j0 += 4 * BLOCK) {
        uint2 r[4];
        bool er[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * BLOCK;
            er[u] = false;
            if (j < n) { r[u] = vrow[j]; er[u] = (U[j >> 5] >> (j & 31)) & 1u; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (er[u]) {
                int pos, cc[4];
                uint32_t am[4];
                edges(j0 + u * BLOCK, r[u], pos, cc, am);
#pragma unroll
                for (int i = 0; i < 4; i++) atomicAdd(&cw[cc[i] >> 1], am[i]);
            }
        }
    }
    __syncthreads();
    STAMP_DECL
    STAMP(0);                                                            // channel + build
    const int nch = scal[SC_NE];

    // ---- one release: VN j is the last erased neighbour of some CN ----------------------------------------------------
    // out[i] = 1 + the VN that the CN on edge i is left with if this release took its count to one, else 0
    int removed = 0;
    auto release = [&](int j, uint32_t (&out)[4]) {
        out[0] = out[1] = out[2] = out[3] = 0;
        const uint2 r = vrow[j];                                         // issued before the claim: overlaps its round trip
        const uint32_t bit = 1u << (j & 31);
        if (!(atomicAnd(&U[j >> 5], ~bit) & bit)) return;                // released through another CN already
        removed++;
        int pos, cc[4];
        uint32_t am[4], o[4];
        edges(j, r, pos, cc, am);
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = atomicSub(&cw[cc[i] >> 1], am[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t f = field_of(o[i] - am[i], cc[i]);
            if ((f & 7u) == 1u && cc[i] < cn_lim) {
                const uint32_t s = (f >> 3) & 0xFFFu;
                out[i] = 1u + (uint32_t)((pos + i - (int)(s & 3u)) * V) + (s >> 2);
            }
        }
    };
    // the VNs named by the CNs of word w whose count is one (CNs >= cn_lim dropped): 1 + VN, or 0
    auto ones_of = [&](int w, uint32_t (&vn)[2]) {
        const uint32_t x = cw[w];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = 2 * w + h;
            const uint32_t f = field_of(x, c);
            vn[h] = 0;
            if ((f & 7u) == 1u && c < cn_lim) {
                const uint32_t s = (f >> 3) & 0xFFFu;
                const int pc = (int)__umulhi((uint32_t)c, a.magic_c);
                vn[h] = 1u + (uint32_t)((pc - (int)(s & 3u)) * V) + (s >> 2);
            }
        }
    };
    // a wave appends its lanes' out[] entries to queue qn behind *push (one prefix scan + one LDS atomic per wave)
    auto append = [&](const uint32_t (&out)[4], int *push, uint16_t *qn, bool &overflow) {
        const int mine = (out[0] != 0u) + (out[1] != 0u) + (out[2] != 0u) + (out[3] != 0u);
        const int incl = (int)wave_inclusive_scan((uint32_t)mine);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (tot) {
            int base = 0;
            if (lane == 0) base = atomicAdd(push, tot);
            int idx = __builtin_amdgcn_readfirstlane(base) + incl - mine;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (out[i]) { if (idx < qcap) qn[idx] = (uint16_t)(out[i] - 1u); else overflow = true; idx++; }
        }
    };

    // ---- peel: barrier rounds over a shared queue while the frontier is wide (a scan opens the run and repairs an
    //      overflow), then every wave runs the VNs its own releases name from a private queue, level after level ----------
    int rounds = 0;
    {
        const int wcap = (qcap / kWaves) & ~1, half_cap = wcap / 2;
        const int kSwitch = min(a.kswitch, kWaves * min(half_cap, 64));   // a wave's share of the frontier fits its private queue and its lanes
        int ncur = 0;
        bool scan = true;
        for (;;) {
            uint16_t *qc = q[rounds & 1], *qn = q[(rounds + 1) & 1];
            if (scan) {
                // every CN < cn_lim whose count is one right now: its VN into qc
                for (int w0 = wave * 64; w0 < a.nsw; w0 += BLOCK) {
                    const int w = w0 + lane;
                    uint32_t vn[2] = {0, 0};
                    if (w < a.nsw) ones_of(w, vn);
                    const int mine = (vn[0] != 0u) + (vn[1] != 0u);
                    const int incl = (int)wave_inclusive_scan((uint32_t)mine);
                    const int tot = __builtin_amdgcn_readlane(incl, 63);
                    if (tot == 0) continue;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&scal[SC_Q], tot);
                    int idx = __builtin_amdgcn_readfirstlane(base) + incl - mine;
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        if (vn[h]) { if (idx < qcap) qc[idx] = (uint16_t)(vn[h] - 1u); idx++; }
                }
                __syncthreads();
                ncur = scal[SC_Q];
                if (ncur > qcap) { ncur = qcap; if (tid == 0) scal[SC_OVF] = 1; }      // the rest: next scan
                __syncthreads();
            }
            if (tid == 0) { scal[SC_Q] = 0; scal[SC_N0 + ((rounds + 1) & 1)] = 0; }
            int *push = &scal[SC_N0 + (rounds & 1)];
            bool overflow = false;
            if (ncur > kSwitch || half_cap < 32) {
                for (int k0 = wave * 64; k0 < ncur; k0 += BLOCK) {
                    uint32_t out[4] = {0, 0, 0, 0};
                    if (k0 + lane < ncur) release((int)qc[k0 + lane], out);
                    append(out, push, qn, overflow);
                }
            } else {
                // private phase: wave w takes entries w, w + kWaves, … into its own part of qn and runs to exhaustion
                uint16_t *mine = qn + wave * wcap;
                int cntw = (ncur - wave + kWaves - 1) / kWaves, cur = 0;
                if (cntw < 0) cntw = 0;
                if (lane < cntw) mine[lane] = qc[wave + lane * kWaves];
                while (cntw > 0) {
                    uint16_t *src = mine + cur * half_cap, *dst = mine + (cur ^ 1) * half_cap;
                    int ncnt = 0;
                    for (int b0 = 0; b0 < cntw; b0 += 64) {
                        uint32_t out[4] = {0, 0, 0, 0};
                        if (b0 + lane < cntw) release((int)src[b0 + lane], out);
                        const int mine_n = (out[0] != 0u) + (out[1] != 0u) + (out[2] != 0u) + (out[3] != 0u);
                        const int incl = (int)wave_inclusive_scan((uint32_t)mine_n);
                        int idx = ncnt + incl - mine_n;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (out[i]) { if (idx < half_cap) dst[idx] = (uint16_t)(out[i] - 1u); else overflow = true; idx++; }
                        ncnt += __builtin_amdgcn_readlane(incl, 63);
                    }
                    cntw = min(ncnt, half_cap);
                    cur ^= 1;
                }
            }
            if (overflow) scal[SC_OVF] = 1;
            __syncthreads();
            rounds++;
            const int pushed = *push;
            scan = scal[SC_OVF] != 0;            // a full queue dropped entries: find their CNs by a scan
            __syncthreads();
            if (tid == 0) scal[SC_OVF] = 0;
            ncur = scan ? 0 : min(pushed, qcap);
            if (!scan && ncur == 0) break;
        }
    }
    STAMP(1);                                                            // peeling
    removed = wave_sum(removed);
    if (lane == 0 && removed) atomicAdd(&scal[SC_REM], removed);
    __syncthreads();
    const int ne = nch - scal[SC_REM];

    // ---- erased VNs per position (word w of U may straddle two positions) --------------------------------------------
    int be = 0, ee = 0, bee = 0;
    if (ne > 0) {
        for (int w = tid; w < nw; w += BLOCK) {
            uint32_t x = U[w];
            int p0 = (int)__umulhi((uint32_t)(w * 32), a.magic_v);
            int room = (p0 + 1) * V - w * 32;                            // bits of this word left in position p0
            while (x) {
                const uint32_t lo = room >= 32 ? x : (x & ((1u << room) - 1u));
                if (lo) atomicAdd(&pos_cnt[p0], __popc(lo));
                x = room >= 32 ? 0u : (x >> room);
                p0++;
                room = V;
            }
        }
        __syncthreads();
        // ---- size-2 stopping sets (BPF:1067-1133) of the first failing position(s) only ----------------------------
        int q0 = 0;
        for (;;) {
            while (q0 < L && pos_cnt[q0] == 0) q0++;
            if (q0 >= L) break;
            for (int t = tid; t < V; t += BLOCK) {
                const int j = q0 * V + t;
                if (!((U[j >> 5] >> (j & 31)) & 1u)) continue;
                int pos, cc[4];
                uint32_t am[4];
                edges(j, vrow[j], pos, cc, am);
                bool pair = true;
                int partner = -1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t f = field_of(cw[cc[i] >> 1], cc[i]);
                    if ((f & 7u) != 2u) { pair = false; continue; }
                    const uint32_t s = ((f >> 3) - (4u * (uint32_t)t + (uint32_t)i)) & 0xFFFu;     // the CN's other erased neighbour
                    const int other = (q0 + i - (int)(s & 3u)) * V + (int)(s >> 2);
                    if (i > 0 && other != partner) pair = false;
                    partner = other;
                }
                if (pair && (int)__umulhi((uint32_t)partner, a.magic_v) == q0) atomicAdd(&pos_ss[q0], 1);
            }
            __syncthreads();
            const int e = pos_cnt[q0] - pos_ss[q0];
            if (e > 0) { ee = e; bee = 1; break; }                       // only the FIRST such position (BPF:1126-1132)
            q0++;
        }
        for (int pos = 0; pos < L; pos++) be += pos_cnt[pos] > 0;
    }
    STAMP(2);                                                            // per-position counts + expurgation
    STAMP_FLUSH();
    if (a.erased_out)
        for (int w = tid; w < nw; w += BLOCK) a.erased_out[(size_t)trial * nw + w] = U[w];
    if (tid == 0) {
        int32_t *o = a.counters + (size_t)trial * SCLDPC_NCOUNTERS;
        o[SCLDPC_C_NUM_ERASURES] = ne;
        o[SCLDPC_C_NUM_BLOCKS_ERR] = be;
        o[SCLDPC_C_NUM_ERASURES_EXP] = ee;
        o[SCLDPC_C_NUM_BLOCKS_ERR_EXP] = bee;
        o[SCLDPC_C_NUM_ERASURES_P1] = 0;
        o[SCLDPC_C_ITERATIONS] = rounds;                                 // barrier rounds, not flooding iterations
        o[SCLDPC_C_STATUS] = 0;
        o[SCLDPC_C_CHANNEL_ERASURES] = nch;
    }
}

int make_args(const scldpc_code_params *p, int32_t is_term, SumArgs *a, int per_cu)
{
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    a->L = p->L; a->V = p->vns_pos; a->C = p->cns_pos; a->n = n; a->nk = nk;
    a->cn_lim = is_term ? nk : p->L * p->cns_pos;                        // BPT:944-948
    a->nw = (n + 31) / 32; a->nsw = (nk + 1) / 2;
    int off = 0;
    auto take = [&](int words) { int o = off; off += (words + 3) & ~3; return o; };
    take(a->nsw);
    a->off_U = take(a->nw);
    a->off_pos = take(2 * p->L);
    a->off_scal = take(SC_N);
    const int budget = scldpc::kMaxLdsBytes / per_cu / 4 - 192;           // words per workgroup (five of them: 25 granules of 1280 B each)
    int qwords = ((budget - off) / 2) & ~3;                              // per queue; two uint16 entries per word
    if (qwords > 2048) qwords = 2048;
    if (qwords < 64) return -1;                                          // four private wave queues of 2 x 32 entries at least
    a->qcap = 2 * qwords;
    a->off_q0 = take(qwords);
    a->off_q1 = take(qwords);
    a->total = off;
    return 0;
}

constexpr int kBlockSum = 256;          // threads per trial
constexpr int kSwitchWidth = 128;       // frontier entries below which the waves go private

int per_cu_of(const scldpc_code_params *p)
{
    SumArgs a{};
    int per_cu = 5;
    if (const char *v = getenv("SCLDPC_DEBUG_SUM_PER_CU")) per_cu = std::max(1, std::min(5, atoi(v)));
    while (per_cu > 1 && make_args(p, 1, &a, per_cu) != 0) per_cu--;
    return make_args(p, 1, &a, per_cu) == 0 ? per_cu : 0;
}

}  // namespace

// 1 when scldpc_full_bp_fixpoint_device_vn16 takes this ensemble: the (4,8) chain with at most 2048 sockets per position
// (N <= 512), VN indices and CN indices of 16 bits, state within the LDS
extern "C" int scldpc_full_bp_vn16_supported(const scldpc_code_params *p)
{
    if (scldpc::check_params(p)) return 0;
    uint32_t m;
    return p->dv == 4 && p->dc == 8 && p->vns_pos * p->dv <= 2048 && scldpc::n_of(p) <= 65536 && scldpc::nk_of(p) <= 65536 &&
           p->cns_pos <= 65536 && per_cu_of(p) > 0 && scldpc::magic_of(p->vns_pos, scldpc::n_of(p) + 32, &m) &&
           scldpc::magic_of(p->cns_pos, scldpc::nk_of(p) + 2, &m);
}

extern "C" int scldpc_full_bp_fixpoint_device_vn16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                                   const uint32_t *d_chan_bits, int32_t is_term, int32_t *d_counters,
                                                   uint32_t *d_erased_bits, void *stream)
{
    const char *who = "scldpc_full_bp_fixpoint_device_vn16";
    if (int rc = scldpc::check_params(p)) return rc;
    if (!scldpc_full_bp_vn16_supported(p))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: takes dv = 4, dc = 8, N <= 512 and at most 65536 VNs per trial", who);
    if (ntrials < 0 || (ntrials > 0 && (!d_counters || !d_vn_adj16 || !d_chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: null buffer or negative ntrials", who);
    if (ntrials == 0) return SCLDPC_OK;
    SumArgs a{};
    const int per_cu = per_cu_of(p);
    make_args(p, is_term, &a, per_cu);
    scldpc::magic_of(p->vns_pos, a.n + 32, &a.magic_v);
    scldpc::magic_of(p->cns_pos, a.nk + 2, &a.magic_c);
    a.vn_adj16 = d_vn_adj16; a.chan = d_chan_bits;
    a.counters = d_counters; a.erased_out = d_erased_bits;
    a.kswitch = kSwitchWidth;
    a.ntrials = ntrials;
    void (*kern)(const SumArgs) = per_cu >= 5 ? full_bp_sum_kernel<kBlockSum, 5> : full_bp_sum_kernel<kBlockSum, 4>;
    size_t lds_bytes = 4u * (size_t)a.total;
    lds_bytes = std::min(lds_bytes + scldpc::debug_lds_pad("DECODER"), (size_t)scldpc::kMaxLdsBytes);
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kBlockSum), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}
