// Sweep peeling + stopping-set statistics of the reference's Python error-rate simulator — gfx950 kernel.
//
// Replaces, per trial, the body of the `for o in the_range` loop of simulate_sc_ldpc
// (simulators_sc_ldpc/peeling_decoding/peeling_decoding.py = PD, PD:632-691):
//   * `for t in range(start, total_size): sic_round(schedule, t)`  (PD:656-657, sic_round PD:270-287,
//     subtract_interference PD:294-313).  Its residual does not depend on the sweep order: it is the closure of
//     "a CN holding exactly one VN releases it", where a CN index < total_size may fire on any TRANSITION to one VN
//     (PD:305-308), a CN >= sweep_start also when it holds one VN from the outset (PD:273-277), and CNs >= total_size
//     never (the truncated tail of a non-terminated chain).  → the same frontier machine as full_bp.hip, with the
//     first round restricted to CNs >= sweep_start.
//   * `lost` = VNs still attached to a CN of [lost_lo, lost_hi) whose CNs are all < total_size (PD:659-666);
//   * extract_stopping_sets (PD:1077-1095) = connected components of `lost` through all shared CNs; "expurgated"
//     statistics ignore components of <= 2 VNs (PD:677-691): #lost_exp, and the number of distinct chain positions
//     int(birthday / cns_per_pos) among the VNs of larger components.
// Only "is this VN's component larger than 2" is ever asked, and that is a local question: once peeling is done the CN
// words are rebuilt over the LOST VNs alone; VN a sits in a component of more than two iff one of its CNs holds >= 3 lost
// VNs, or its CNs name two different partners, or its single partner b has a CN with >= 3 lost VNs or a partner other
// than a.  No union–find, so the 16-bit packed CN words serve here too (two workgroups per CU) — for position-structured
// graphs only, i.e. the 2-byte adjacency; arbitrary graphs (tail-biting, uncoupled) keep the 32-bit words.
#include "common.h"
#include "kernel_util.h"
#include "cn_words.h"
#include <cstdlib>

namespace {

using namespace scldpc_dev;

constexpr int kBlock = 1024;
enum { S_PUSH = 0, S_OVF = 3, S_REM = 6, S_NE = 9, S_LOST = 10, S_LOST_EXP = 11, S_NSCAL = 16 };

struct Layout { int cn_state, U, fbits, frozen, q0, q1, pos_flag, scal, total, qcap, nw; };

struct Args : Geo {             // Geo: vns_pos, magic_v, magic_c
    int dv, L, cns_pos, n, ncn, total_size, sweep_start, lost_lo, lost_hi;
    Layout lay;
    const void *vn_adj;
    const uint32_t *chan;
    int32_t *out;               // [T][8]: lost, lost_exp, blocks_failed_exp, 0, 0, rounds, 0, #erased
    uint32_t *lost_out;         // optional [T][nw]
    uint32_t *ws;               // [T][ncn] CN words in global memory (ensembles beyond the LDS)
    int prebuilt;               // … already built by cn_build.hip (through LDS, not by one global atomic per edge)
};

template <int DV, bool A16, class ST>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(72))) void peel_sweep_kernel(const Args a)   // see full_bp.hip
{
    extern __shared__ uint32_t lds[];
    uint32_t *cn_state = ST::kGlobal ? a.ws + (size_t)blockIdx.x * a.ncn : lds + a.lay.cn_state;
    uint32_t *U = lds + a.lay.U;
    uint32_t *fbits = lds + a.lay.fbits;
    uint32_t *frozen = lds + a.lay.frozen;
    uint32_t *q[2] = {lds + a.lay.q0, lds + a.lay.q1};
    int *pos_flag = reinterpret_cast<int *>(lds + a.lay.pos_flag);
    int *scal = reinterpret_cast<int *>(lds + a.lay.scal);

    const int tid = threadIdx.x, lane = tid & 63;
    const int trial = blockIdx.x;
    const int n = a.n, ncn = a.ncn, dv = (DV ? DV : a.dv), cn_lim = a.total_size, nw = a.lay.nw, qcap = a.lay.qcap;
    const char *adj = static_cast<const char *>(a.vn_adj) + (size_t)trial * n * dv * (A16 ? 2 : 4);
    const uint32_t *ch = a.chan + (size_t)trial * nw;
    auto pos_of = [&](int j) { return (int)__umulhi((uint32_t)j, a.magic_v); };

    const bool prebuilt = ST::kGlobal && a.prebuilt;
    if (!prebuilt) for (int c = tid; c < ST::words(ncn); c += kBlock) cn_state[c] = 0;
    auto make_vn = [&](int j) { Vn v; v.j = j; v.pos = pos_of(j); v.t = j - v.pos * a.vns_pos; return v; };
    int ne_local = 0;
    for (int w = tid; w < nw; w += kBlock) {
        uint32_t x = ch[w];
        if (w == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        U[w] = x;
        ne_local += __popc(x);
    }
    if (tid < S_NSCAL) scal[tid] = 0;
    for (int i = tid; i < a.L; i += kBlock) pos_flag[i] = 0;
    __syncthreads();
    {
        const uint32_t tot = wave_inclusive_scan((uint32_t)ne_local);
        if (lane == 63 && tot) atomicAdd(&scal[S_NE], (int)tot);
    }
    for (int j0 = prebuilt ? n : tid; j0 < n; j0 += 4 * kBlock) {
        int32_t c[4][8];
        bool er[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * kBlock;
            er[u] = false;
            if (j < n) {
                load_adj<DV, A16>(adj, dv, j, pos_of(j), a.cns_pos, c[u]);
                er[u] = (U[j >> 5] >> (j & 31)) & 1u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * kBlock;
            if (j < n && er[u]) {
                const Vn v = make_vn(j);
                for (int i = 0; i < dv; i++) ST::add(cn_state, c[u][i], v, i, a.vns_pos, true, false);
            }
        }
    }
    __syncthreads();

    // ---- peeling to the fixpoint (rounds as in full_bp.hip; the order is irrelevant for the residual) ----
    int iter = 0, ncur = 0;
    bool scan = true;
    for (;;) {
        const int g = iter % 3, gn = (iter + 1) % 3;
        uint32_t *qc = q[iter & 1], *qn = q[(iter + 1) & 1];
        if (tid == 0) { scal[S_PUSH + gn] = 0; scal[S_OVF + gn] = 0; scal[S_REM + gn] = 0; }
        int removed = 0;
        auto release = [&](int c) {
            const int j = ST::lone_vn(cn_state, c, a);
            if (j < 0) return;
            const Vn v = make_vn(j);
            int32_t cc[8];
            load_adj<DV, A16>(adj, dv, j, v.pos, a.cns_pos, cc);
            const uint32_t bit = 1u << (j & 31);
            if (!(atomicAnd(&U[j >> 5], ~bit) & bit)) return;
            removed++;
            uint32_t o[8];
            for (int i = 0; i < dv; i++) o[i] = ST::remove_cnt(cn_state, cc[i], v, i, a.vns_pos);
            for (int i = 0; i < dv; i++) ST::remove_fold(cn_state, cc[i], v, i, a.vns_pos);
            for (int i = 0; i < dv; i++) {
                if (o[i] == 2u && cc[i] < cn_lim) {                 // transition to one VN: may fire (PD:305-308)
                    const int idx = atomicAdd(&scal[S_PUSH + g], 1);
                    if (idx < qcap) qn[idx] = (uint32_t)cc[i]; else scal[S_OVF + g] = 1;
                }
            }
        };
        if (scan) {
            // Round 0: only swept CNs may fire from the outset (PD:656, 273-277); CNs below the sweep start that
            // hold one VN at the outset are remembered in `frozen` and never fire.  A later re-scan (queue
            // overflow) takes every other CN < total_size holding one VN: those got there by a transition.
            for (int base = 0; base < cn_lim; base += kBlock) {
                const int c = base + tid;
                const bool one = c < cn_lim && ST::cnt(cn_state, c) == 1u;
                bool v, fz;
                if (iter == 0) { v = one && c >= a.sweep_start; fz = one && c < a.sweep_start; }
                else           { fz = c < cn_lim && ((frozen[c >> 5] >> (c & 31)) & 1u); v = one && !fz; }
                const unsigned long long m = __ballot(v), mf = __ballot(fz);
                if (c - lane < cn_lim) {
                    if (lane == 0) { fbits[c >> 5] = (uint32_t)m; frozen[c >> 5] = (uint32_t)mf; }
                    if (lane == 32) { fbits[c >> 5] = (uint32_t)(m >> 32); frozen[c >> 5] = (uint32_t)(mf >> 32); }
                }
            }
            __syncthreads();
            for (int base = 0; base < cn_lim; base += kBlock) {
                const int c = base + tid;
                if (c < cn_lim && ((fbits[c >> 5] >> (c & 31)) & 1u)) release(c);
            }
        } else {
            for (int k = tid; k < ncur; k += kBlock) release((int)qc[k]);
        }
        {
            const uint32_t tot = wave_inclusive_scan((uint32_t)removed);
            if (lane == 63 && tot) atomicAdd(&scal[S_REM + g], (int)tot);
        }
        __syncthreads();
        const int rem = scal[S_REM + g];
        const bool ovf = scal[S_OVF + g] != 0;
        ncur = ovf ? 0 : scal[S_PUSH + g];
        iter++;
        if (ovf) { scan = true; continue; }
        scan = false;
        if (ncur == 0 && rem >= 0) break;                           // nothing queued: fixpoint
    }
    __syncthreads();

    // ---- lost set, components, expurgated statistics -----------------------------------------
    auto is_lost = [&](int j, int32_t (&cc)[8]) {
        load_adj<DV, A16>(adj, dv, j, pos_of(j), a.cns_pos, cc);
        bool in_range = false, all_inside = true;
        for (int i = 0; i < dv; i++) {
            in_range |= cc[i] >= a.lost_lo && cc[i] < a.lost_hi;    // PD:661-664
            all_inside &= cc[i] < cn_lim;                           // PD:665
        }
        return in_range && all_inside;
    };
    for (int c = tid; c < ST::words(ncn); c += kBlock) cn_state[c] = 0;      // CN words over the lost VNs only, from here on
    __syncthreads();
    for (int w = tid; w < nw; w += kBlock) {
        uint32_t x = U[w], keep = 0;
        while (x) {
            const int b = __ffs((int)x) - 1;
            x &= x - 1;
            int32_t cc[8];
            if (is_lost(w * 32 + b, cc)) {
                keep |= 1u << b;
                const Vn v = make_vn(w * 32 + b);
                for (int i = 0; i < dv; i++) ST::add(cn_state, cc[i], v, i, a.vns_pos, true, false);
            }
        }
        U[w] = keep;                                                // U := lost
    }
    __syncthreads();
    // what VN v's CNs say: -2 = some CN holds >= 3 lost VNs or two different partners show up; -1 = no partner; else the partner
    auto neighbourhood = [&](const Vn &v, const int32_t (&cc)[8]) {
        int partner = -1;
        for (int i = 0; i < dv; i++) {
            const uint32_t k = ST::cnt(cn_state, cc[i]);
            if (k >= 3u) return -2;
            if (k == 2u) {
                const int b = ST::partner(cn_state, cc[i], v, i, a);
                if (partner >= 0 && b != partner) return -2;
                partner = b;
            }
        }
        return partner;
    };
    int lost = 0, lost_exp = 0;
    for (int w = tid; w < nw; w += kBlock) {
        uint32_t x = U[w];
        while (x) {
            const int b = __ffs((int)x) - 1;
            x &= x - 1;
            const Vn va = make_vn(w * 32 + b);
            int32_t cc[8], cb[8];
            load_adj<DV, A16>(adj, dv, va.j, va.pos, a.cns_pos, cc);
            lost++;
            int pa = neighbourhood(va, cc);
            if (pa >= 0) {                                          // one partner: is {a, partner} closed?
                const Vn vb = make_vn(pa);
                load_adj<DV, A16>(adj, dv, vb.j, vb.pos, a.cns_pos, cb);
                const int pb = neighbourhood(vb, cb);
                if (pb != va.j) pa = -2;
            }
            if (pa == -2) {                                         // component of > 2 VNs
                lost_exp++;
                pos_flag[cc[0] / a.cns_pos] = 1;                    // int(u.birthday / cns_per_pos), PD:160,687
            }
        }
    }
    {
        const uint32_t t1 = wave_inclusive_scan((uint32_t)lost), t2 = wave_inclusive_scan((uint32_t)lost_exp);
        if (lane == 63 && t1) atomicAdd(&scal[S_LOST], (int)t1);
        if (lane == 63 && t2) atomicAdd(&scal[S_LOST_EXP], (int)t2);
    }
    __syncthreads();
    if (a.lost_out)
        for (int w = tid; w < nw; w += kBlock) a.lost_out[(size_t)trial * nw + w] = U[w];
    if (tid == 0) {
        int blocks = 0;
        for (int pos = 0; pos < a.L; pos++) blocks += pos_flag[pos];
        int32_t *o = a.out + (size_t)trial * 8;
        o[0] = scal[S_LOST]; o[1] = scal[S_LOST_EXP]; o[2] = blocks; o[3] = 0; o[4] = 0; o[5] = iter; o[6] = 0;
        o[7] = scal[S_NE];
    }
}

}  // namespace

static int launch_peel_sweep(const scldpc_code_params *p, int32_t ntrials, const void *d_vn_adj, bool adj16,
                             const uint32_t *d_chan_bits, int32_t total_size, int32_t sweep_start,
                             int32_t lost_lo, int32_t lost_hi, int32_t *d_out, uint32_t *d_lost_bits, const scldpc::Scratch &scratch, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (scratch.query) *scratch.query = 0;
    if (!scratch.query && (ntrials < 0 || (ntrials > 0 && (!d_out || !d_vn_adj || !d_chan_bits))))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_peel_sweep_device: null buffer or negative ntrials");
    const int n = scldpc::n_of(p), ncn = scldpc::nk_of(p);
    if (total_size < 0 || total_size > ncn || sweep_start < 0 || lost_lo < 0 || lost_hi > ncn)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_peel_sweep_device: CN ranges outside [0,%d]", ncn);
    if (ntrials <= 0) return SCLDPC_OK;
    if (p->dc > 15 || p->dv > 8 || (int64_t)p->dc * n >= (1ll << kDegShift))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_peel_sweep_device: needs dc <= 15, dv <= 8, dc*n < 2^24");
    Args a{};
    int off = 0;
    auto take = [&](int words) { int o = off; off += (words + 3) & ~3; return o; };
    a.lay.nw = (n + 31) / 32;
    // CN words: packed 16 bit (position-structured graphs only = the 2-byte adjacency), else 32 bit in LDS, else 32 bit in
    // the workspace.  Two workgroups per CU (half the LDS each) when queues of >= 1024 entries still fit.
    const bool can_pack = adj16 && (int64_t)p->dv * p->vns_pos <= 4096;
    int mode = -1, qcap = 0;                                        // 0 packed, 1 wide, 2 wide in the workspace
    for (int m = can_pack ? 0 : 1; m < 3 && mode < 0; m++) {
        off = 0;
        a.lay.cn_state = take(m == 0 ? Packed::lds_words(ncn) : m == 1 ? ncn : 0);
        a.lay.U = take(a.lay.nw);
        a.lay.fbits = take(((ncn + 63) / 64) * 2);
        a.lay.frozen = take(((ncn + 63) / 64) * 2);
        a.lay.pos_flag = take(p->L + p->dv);
        a.lay.scal = take(S_NSCAL);
        int left = scldpc::kMaxLdsBytes / 4 - off;
        if (scldpc::kMaxLdsBytes / 8 - 256 - off >= 2 * 1024) left = scldpc::kMaxLdsBytes / 8 - 256 - off;
        qcap = (left / 2) & ~3;
        if (qcap > 8192) qcap = 8192;
        if (qcap >= 256) mode = m;
    }
    if (mode < 0)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE,
                                 "scldpc_peel_sweep_device: %d VN bits + %d scan bits do not fit 160 KiB of LDS", n, ncn);
    const bool global_ws = mode == 2;
    a.lay.qcap = qcap; a.lay.q0 = take(qcap); a.lay.q1 = take(qcap); a.lay.total = off;
    if (global_ws) {
        const size_t need = (size_t)ntrials * ncn * sizeof(uint32_t);
        if (scratch.query) { *scratch.query = need; return SCLDPC_OK; }
        void *ws = nullptr;
        if (int rc = scldpc::take_scratch("scldpc_peel_sweep_device", scratch, need, &ws)) return rc;
        a.ws = static_cast<uint32_t *>(ws);
    }
    if (scratch.query) return SCLDPC_OK;
    a.dv = p->dv; a.L = p->L; a.vns_pos = p->vns_pos; a.cns_pos = p->cns_pos; a.n = n; a.ncn = ncn;
    a.total_size = total_size; a.sweep_start = sweep_start; a.lost_lo = lost_lo; a.lost_hi = lost_hi;
    if (!scldpc::magic_of(p->vns_pos, n > 4096 ? n : 4096, &a.magic_v) || !scldpc::magic_of(p->cns_pos, ncn, &a.magic_c))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_peel_sweep_device: reciprocal division inexact");
    a.vn_adj = d_vn_adj; a.chan = d_chan_bits; a.out = d_out; a.lost_out = d_lost_bits;
    void (*kern)(const Args) = nullptr;
#define PICK(ST) (p->dv == 4 ? (adj16 ? peel_sweep_kernel<4, true, ST> : peel_sweep_kernel<4, false, ST>) \
                             : (adj16 ? peel_sweep_kernel<0, true, ST> : peel_sweep_kernel<0, false, ST>))
    kern = mode == 0 ? PICK(Packed) : mode == 1 ? PICK(Wide) : PICK(WideG);
#undef PICK
    if (global_ws && adj16) {
        // the first build (every erased VN into its dv CNs) through an LDS ring of dv CN positions — see cn_build.hip; the
        // second one, over the lost VNs alone, stays in the kernel (a few VNs)
        bool pre = true;
        if (const char *v = getenv("SCLDPC_DEBUG_SWEEP_PREBUILD")) pre = atoi(v) != 0;                      // A/B, tests
        a.prebuilt = pre && scldpc::cn_build_launch(p, ntrials, static_cast<const uint16_t *>(d_vn_adj), d_chan_bits, a.ws, false, stream) ? 1 : 0;
    }
    const size_t lds_bytes = 4u * (size_t)a.lay.total;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kBlock), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_peel_sweep_device(const scldpc_code_params *p, int32_t ntrials,
                                        const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                                        int32_t total_size, int32_t sweep_start, int32_t lost_lo, int32_t lost_hi,
                                        int32_t *d_out, uint32_t *d_lost_bits, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_peel_sweep(p, ntrials, d_vn_adj, false, d_chan_bits, total_size, sweep_start, lost_lo, lost_hi,
                             d_out, d_lost_bits, scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

extern "C" int scldpc_peel_sweep_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                              const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                              int32_t total_size, int32_t sweep_start, int32_t lost_lo, int32_t lost_hi,
                                              int32_t *d_out, uint32_t *d_lost_bits, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_peel_sweep(p, ntrials, d_vn_adj16, true, d_chan_bits, total_size, sweep_start, lost_lo, lost_hi,
                             d_out, d_lost_bits, scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

// workspace of scldpc_peel_sweep_device(_adj16) for ntrials trials
int64_t scldpc_peel_sweep_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t adj16)
{
    uint64_t need = 0;
    const int rc = launch_peel_sweep(p, ntrials, nullptr, adj16 != 0, nullptr, 0, 0, 0, 0, nullptr, nullptr,
                                     scldpc::Scratch{nullptr, 0, &need}, nullptr);
    return rc ? (int64_t)rc : (int64_t)need;
}
