// Square sliding-window BP over the BEC for a batch of sampled SC-LDPC codes — gfx950 kernel.
//
// Replaces decodeBP_SW of the reference (BPW:628-912): for posW = 0..L-1 the CNs and the VNs of
// positions [posW, posW+W) are updated by flooding until the window's erasure count reaches 0,
// stops changing, or the iteration cap (init_it for posW == 0, max_it afterwards; BPW:699-702,839)
// is hit; messages persist from window to window; position posW is decided when its window ends;
// afterwards the size-2 stopping-set expurgation over all positions (BPW:850-908).
//
// Same node-level restatement as full_bp.hip: S = set of VNs the CNs still see as
// erased (1 bit per VN), one [cnt | idsum] word per CN kept exact for EVERY CN while S shrinks.
// Inside window posW only CNs of Cw = [posW, posW+W) can fire — CNs further right were never updated
// and send erasures, CNs further left touch no window VN — and only VNs at positions >= posW may be
// released (older positions are frozen).  One flooding iteration == one frontier round:
//     every CN of Cw with cnt == 1 at the start of the round releases its VN (if not frozen).
// The frontier is found by a scan of Cw when a window opens and kept in an LDS queue afterwards.
#include "common.h"
#include "kernel_util.h"
#include "cn_words.h"
#include <cstdlib>

namespace {

using namespace scldpc_dev;

constexpr int kBlock = 1024;

enum { S_CNT = 0, S_OVF = 3, S_REM = 6, S_NCH = 9, S_NSCAL = 16 };

struct Layout {             // offsets in 32-bit words into dynamic LDS
    int cn_state, S, fbits, q0, q1, pos_cnt, pos_ss, scal, total;
    int qcap, nw;
};

struct Args : Geo {             // Geo: vns_pos, magic_v, magic_c
    int dv, L, cns_pos, n, nk, W, max_it, init_it;
    int classical;                  // 0: square window (BPW:628-912); 1: classical window (BPF:627-897)
    int prebuilt;                   // WideG: the CN words were built by cn_build.hip (through LDS, not by global atomics)
    Layout lay;
    const void *vn_adj;             // int32 [T][n][dv] or uint16 [T][n][dv] (position-local ids)
    uint32_t *ws;                   // [T][nk] CN words in global memory (G only)
    const uint32_t *chan;
    int32_t *counters;
    uint32_t *erased_out;
};

// ST: CN-word policy (cn_words.h) — Packed (16 bit, two workgroups per CU), Wide (LDS) or WideG (global workspace, for
// ensembles beyond the LDS budget, e.g. L=100, N=2000)
template <int DV, bool A16, class ST>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(72))) void sw_bp_kernel(const Args a)   // see full_bp.hip
{
    extern __shared__ uint32_t lds[];
    uint32_t *cn_state;
    if constexpr (ST::kGlobal) cn_state = a.ws + (size_t)blockIdx.x * a.nk;
    else                       cn_state = lds + a.lay.cn_state;
    uint32_t *S = lds + a.lay.S;
    uint32_t *fbits = lds + a.lay.fbits;
    uint32_t *q[2] = {lds + a.lay.q0, lds + a.lay.q1};
    int *pos_cnt = reinterpret_cast<int *>(lds + a.lay.pos_cnt);
    int *pos_ss = reinterpret_cast<int *>(lds + a.lay.pos_ss);
    int *scal = reinterpret_cast<int *>(lds + a.lay.scal);

    const int tid = threadIdx.x, lane = tid & 63;
    const int trial = blockIdx.x;
    const int n = a.n, nk = a.nk, dv = (DV ? DV : a.dv), nw = a.lay.nw, qcap = a.lay.qcap;
    const int V = a.vns_pos, C = a.cns_pos, L = a.L, W = a.W;
    const char *adj = static_cast<const char *>(a.vn_adj) + (size_t)trial * n * dv * (A16 ? 2 : 4);
    const uint32_t *ch = a.chan + (size_t)trial * nw;

    const bool prebuilt = ST::kGlobal && a.prebuilt;
    if (!prebuilt) for (int c = tid; c < ST::words(nk); c += kBlock) cn_state[c] = 0;
    auto make_vn = [&](int j) { Vn v; v.j = j; v.pos = (int)__umulhi((uint32_t)j, a.magic_v); v.t = j - v.pos * V; return v; };
    for (int w = tid; w < nw; w += kBlock) {
        uint32_t x = ch[w];
        if (w == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        S[w] = x;
    }
    if (tid < S_NSCAL) scal[tid] = 0;
    for (int i = tid; i < L; i += kBlock) { pos_cnt[i] = 0; pos_ss[i] = 0; }
    __syncthreads();

    // ---- build the CN words and the per-position erasure counts from the channel ------------
    int nch = 0;
    if (prebuilt) {
        // the words are there: only the erasure counts per position are left, 32 VNs per step (V >= 32: a word of channel
        // bits meets at most two positions)
        for (int w = tid; w < nw; w += kBlock) {
            const uint32_t x = S[w];
            if (!x) continue;
            nch += __popc(x);
            const int p0 = (int)__umulhi((uint32_t)(w * 32), a.magic_v), split = (p0 + 1) * V - w * 32;
            const int lo = split >= 32 ? __popc(x) : __popc(x & ((1u << split) - 1u));
            if (lo) atomicAdd(&pos_cnt[p0], lo);
            if (__popc(x) - lo) atomicAdd(&pos_cnt[p0 + 1], __popc(x) - lo);
        }
    }
    for (int j0 = prebuilt ? n : tid; j0 < n; j0 += 4 * kBlock) {
        int32_t c[4][8];
        bool er[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * kBlock;
            er[u] = false;
            if (j < n) {
                load_adj<DV, A16>(adj, dv, j, j / V, C, c[u]);
                er[u] = (S[j >> 5] >> (j & 31)) & 1u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * kBlock;
            if (j < n && er[u]) {
                nch++;
                atomicAdd(&pos_cnt[j / V], 1);
                const Vn v = make_vn(j);
                for (int i = 0; i < dv; i++) ST::add(cn_state, c[u][i], v, i, V, true, false);
            }
        }
    }
    nch = wave_sum(nch);
    if (lane == 0 && nch) atomicAdd(&scal[S_NCH], nch);
    __syncthreads();

    int iters_total = 0;
    int gen = 0;                        // flooding-iteration counter: list gen lives in q[gen&1], its size in scal[S_CNT+gen%3]
    // Classical window (BPF:668-684): L+ms windows, the VN window reaches ms positions to the left of the CN window,
    // position posW-ms is decided when window posW closes, one iteration cap for all windows.
    const int ms_k = a.dv - 1, last = a.classical ? L + ms_k : L;
    for (int posW = 0; posW < last; posW++) {
        const int c0 = posW * C, c1 = min(c0 + W * C, nk);                  // BPW:674-676
        const int qlo = a.classical ? max(posW - ms_k, 0) : posW;           // BPW:691 / BPF:673-684
        const int jlo = qlo * V;                                            // VNs left of it are frozen
        const int qhi = min(posW + W, L);                                   // BPW:692-693
        const int cap = a.classical ? a.max_it : ((posW == 0) ? a.init_it : a.max_it);   // BPW:699-702 / BPF:824
        int iter = 0, prec = n, ncur = 0;
        bool scan = true;               // a window opens with a scan of its CNs (the carried list is dropped)
        // erasures inside the window (BPW:791-809).  Read here, where no release is in flight (the scan's
        // snapshot barrier comes first); afterwards kept current from the per-iteration release counter so
        // that every thread takes the same stop decision.
        int term = 0;
        for (int qq = qlo; qq < qhi; qq++) term += pos_cnt[qq];
        for (;;) {
            uint32_t *qc = q[gen & 1], *qn = q[(gen + 1) & 1];
            int *push_cnt = &scal[S_CNT + (gen + 1) % 3], *push_ovf = &scal[S_OVF + (gen + 1) % 3];
            int *rem_cnt = &scal[S_REM + gen % 3];
            int removed = 0;
            auto release = [&](uint32_t c) {
                const int j = ST::lone_vn(cn_state, (int)c, a);
                if (j < 0) return;                                          // its VN went via another CN this round
                if (j < jlo) return;                                        // frozen VN: stays erased for good
                const uint32_t bit = 1u << (j & 31);
                const uint32_t old = atomicAnd(&S[j >> 5], ~bit);
                if (!(old & bit)) return;
                const Vn v = make_vn(j);
                atomicSub(&pos_cnt[v.pos], 1);
                removed++;
                int32_t cc[8];
                load_adj<DV, A16>(adj, dv, j, v.pos, C, cc);
                uint32_t o[8];
                for (int i = 0; i < dv; i++) o[i] = ST::remove_cnt(cn_state, cc[i], v, i, V);
                for (int i = 0; i < dv; i++) ST::remove_fold(cn_state, cc[i], v, i, V);
                for (int i = 0; i < dv; i++) {
                    const uint32_t c2 = (uint32_t)cc[i];
                    // 2 → 1 inside the window: fires in the next iteration.  CNs right of the window
                    // are found by the scan of the first window that contains them; CNs left of it
                    // (classical window only: VNs of positions posW-ms..posW-1) never send again.
                    if (o[i] == 2u && (int)c2 < c1 && (int)c2 >= c0) {
                        const int idx = atomicAdd(push_cnt, 1);
                        if (idx < qcap) qn[idx] = c2; else *push_ovf = 1;
                    }
                }
            };
            if (tid == 0) { scal[S_CNT + (gen + 2) % 3] = 0; scal[S_OVF + (gen + 2) % 3] = 0; scal[S_REM + (gen + 1) % 3] = 0; }
            if (scan) {
                // snapshot {c in Cw : cnt == 1} into fbits first: releases of this round must not
                // promote CNs into the round's own frontier
                for (int base = c0 & ~63; base < c1; base += kBlock) {
                    const int c = base + tid;
                    const bool v = c >= c0 && c < c1 && ST::cnt(cn_state, c) == 1u;
                    const unsigned long long m = __ballot(v);
                    if (c - lane < c1) {
                        if (lane == 0) fbits[c >> 5] = (uint32_t)m;
                        if (lane == 32) fbits[c >> 5] = (uint32_t)(m >> 32);
                    }
                }
                __syncthreads();
                for (int base = c0 & ~63; base < c1; base += kBlock) {
                    const int c = base + tid;
                    if (c < c1 && ((fbits[c >> 5] >> (c & 31)) & 1u)) release((uint32_t)c);
                }
            } else {
                for (int k = tid; k < ncur; k += kBlock) release(qc[k]);
            }
            removed = wave_sum(removed);
            if (lane == 0 && removed) atomicAdd(rem_cnt, removed);
            __syncthreads();                                                // end of the flooding iteration
            iters_total++;
            term -= *rem_cnt;           // every released VN lies inside the window
            scan = *push_ovf != 0;      // queue overflow: rebuild the frontier by a scan
            ncur = min(*push_cnt, qcap);
            gen++;
            if (term == 0 || term == prec) break;                           // BPW:815-816
            prec = term;
            iter++;
            if (!(iter < cap)) break;                                       // BPW:839
        }
    }
    __syncthreads();

    // ---- totals, expurgation over every position (BPW:841-847, 850-908) ---------------------
    for (int w = tid; w < nw; w += kBlock) {
        uint32_t x = S[w];
        while (x) {
            const int b = __ffs((int)x) - 1;
            x &= x - 1;
            const int va = w * 32 + b, pos = va / V;
            int32_t cc[8];
            load_adj<DV, A16>(adj, dv, va, pos, C, cc);
            bool pair = true;
            int partner = -1;
            const Vn v = make_vn(va);
            for (int i = 0; i < dv; i++) {
                const int b2 = ST::partner(cn_state, cc[i], v, i, a);
                if (b2 < 0 || (i > 0 && b2 != partner)) { pair = false; break; }
                partner = b2;
            }
            if (pair && partner / V == pos) atomicAdd(&pos_ss[pos], 1);
        }
    }
    __syncthreads();
    if (a.erased_out)
        for (int w = tid; w < nw; w += kBlock) a.erased_out[(size_t)trial * nw + w] = S[w];
    if (tid == 0) {
        int ne = 0, be = 0, ee = 0, bee = 0, p1 = 0;
        const int ms = a.dv - 1;
        for (int pos = 0; pos < L; pos++) {
            const int cnt = pos_cnt[pos];
            ne += cnt;
            if (cnt > 0) be++;
            // NumErasuresP1: windows ms <= posW <= W-2 (BPW:846-847); the classical window decides position posW-ms
            if (a.classical ? (pos <= W - 2 - ms) : (pos >= ms && pos <= W - 2)) p1 += cnt;
            const int e = cnt - pos_ss[pos];
            if (e > 0) { ee += e; bee++; }                                  // every position (BPW:903-907)
        }
        int32_t *o = a.counters + (size_t)trial * SCLDPC_NCOUNTERS;
        o[SCLDPC_C_NUM_ERASURES] = ne;
        o[SCLDPC_C_NUM_BLOCKS_ERR] = be;
        o[SCLDPC_C_NUM_ERASURES_EXP] = ee;
        o[SCLDPC_C_NUM_BLOCKS_ERR_EXP] = bee;
        o[SCLDPC_C_NUM_ERASURES_P1] = p1;
        o[SCLDPC_C_ITERATIONS] = iters_total;
        o[SCLDPC_C_STATUS] = 0;
        o[SCLDPC_C_CHANNEL_ERASURES] = scal[S_NCH];
    }
}

template <class ST>
int make_layout(const scldpc_code_params *p, int W, Layout *lay)
{
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    int off = 0;
    auto take = [&](int words) { int o = off; off += (words + 3) & ~3; return o; };
    lay->nw = (n + 31) / 32;
    lay->cn_state = take(ST::lds_words(nk));
    lay->S = take(lay->nw);
    lay->fbits = take(((nk + 63) / 64) * 2);
    lay->pos_cnt = take(p->L);
    lay->pos_ss = take(p->L);
    lay->scal = take(S_NSCAL);
    // Two workgroups per CU (half the LDS each) whenever queues of at least 1024 entries still fit: the kernel waits on
    // LDS / L2 round trips most of the time and a second trial on the CU hides them.  Otherwise the whole LDS.
    int left = scldpc::kMaxLdsBytes / 4 - off;
    {
        const int half = scldpc::kMaxLdsBytes / 8 - 256 - off;
        if (half >= 2 * 1024) left = half;
    }
    int qcap = (left / 2) & ~3;
    if (qcap > 8192) qcap = 8192;
    if (qcap < 64) return -1;
    (void)W;
    lay->qcap = qcap;
    lay->q0 = take(qcap);
    lay->q1 = take(qcap);
    lay->total = off;
    return 0;
}

}  // namespace

static int launch_sw_bp(const scldpc_code_params *p, int32_t ntrials, const void *d_vn_adj, bool adj16, bool classical,
                        const uint32_t *d_chan_bits, int32_t W, int32_t max_it, int32_t init_it,
                        int32_t *d_counters, uint32_t *d_erased_bits, const scldpc::Scratch &scratch, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (scratch.query) *scratch.query = 0;
    if (!scratch.query && (ntrials < 0 || (ntrials > 0 && (!d_counters || !d_vn_adj || !d_chan_bits))))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_sw_bp_device: null buffer or negative ntrials");
    if (W < 1 || max_it < 0 || init_it < 0)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_sw_bp_device: need W >= 1, max_it >= 0, init_it >= 0");
    if (ntrials <= 0) return SCLDPC_OK;
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    if (p->dc > 15 || p->dv > 8 || (int64_t)p->dc * n >= (1ll << kDegShift))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_sw_bp_device: needs dc <= 15, dv <= 8, dc*n < 2^24");
    Args a{};
    // packed 16-bit CN words when the ensemble allows them (two workgroups per CU), else 32-bit words in LDS, else in the workspace
    const bool packed = (int64_t)p->dv * p->vns_pos <= 4096 && make_layout<Packed>(p, W, &a.lay) == 0;
    bool gws = false;
    if (!packed && make_layout<Wide>(p, W, &a.lay)) {
        if (make_layout<WideG>(p, W, &a.lay))
            return scldpc::set_error(SCLDPC_ERR_TOO_LARGE,
                                     "scldpc_sw_bp_device: n=%d VN bits + nk=%d scan bits do not fit 160 KiB of LDS", n, nk);
        gws = true;
        const size_t need = (size_t)ntrials * nk * sizeof(uint32_t);
        if (scratch.query) { *scratch.query = need; return SCLDPC_OK; }
        void *ws = nullptr;
        if (int rc = scldpc::take_scratch("scldpc_sw_bp_device", scratch, need, &ws)) return rc;
        a.ws = static_cast<uint32_t *>(ws);
    }
    if (scratch.query) return SCLDPC_OK;
    if (!scldpc::magic_of(p->vns_pos, n > 4096 ? n : 4096, &a.magic_v) || !scldpc::magic_of(p->cns_pos, nk, &a.magic_c))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_sw_bp_device: reciprocal division inexact for this size");
    a.dv = p->dv; a.L = p->L; a.vns_pos = p->vns_pos; a.cns_pos = p->cns_pos; a.n = n; a.nk = nk;
    a.W = W; a.max_it = max_it; a.init_it = init_it ? init_it : max_it;     // BPW:2101-2102
    a.classical = classical ? 1 : 0;
    a.vn_adj = d_vn_adj; a.chan = d_chan_bits; a.counters = d_counters; a.erased_out = d_erased_bits;

    void (*kern)(const Args);
#define PICK(ST) (p->dv == 4 ? (adj16 ? sw_bp_kernel<4, true, ST> : sw_bp_kernel<4, false, ST>) \
                             : (adj16 ? sw_bp_kernel<0, true, ST> : sw_bp_kernel<0, false, ST>))
    kern = packed ? PICK(Packed) : gws ? PICK(WideG) : PICK(Wide);
#undef PICK
    if (gws && adj16 && p->vns_pos >= 32) {
        bool pre = true;
        if (const char *v = getenv("SCLDPC_DEBUG_SW_PREBUILD")) pre = atoi(v) != 0;                         // A/B, tests
        a.prebuilt = pre && scldpc::cn_build_launch(p, ntrials, static_cast<const uint16_t *>(d_vn_adj), d_chan_bits, a.ws, false, stream) ? 1 : 0;
    }
    const size_t lds_bytes = 4u * (size_t)a.lay.total;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kBlock), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_sw_bp_device(const scldpc_code_params *p, int32_t ntrials,
                                   const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                                   int32_t W, int32_t max_it, int32_t init_it,
                                   int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_sw_bp(p, ntrials, d_vn_adj, false, false, d_chan_bits, W, max_it, init_it, d_counters, d_erased_bits,
                        scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

extern "C" int scldpc_sw_bp_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                         const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                         int32_t W, int32_t max_it, int32_t init_it,
                                         int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_sw_bp(p, ntrials, d_vn_adj16, true, false, d_chan_bits, W, max_it, init_it, d_counters, d_erased_bits,
                        scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

extern "C" int scldpc_swc_bp_device(const scldpc_code_params *p, int32_t ntrials,
                                    const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                                    int32_t W, int32_t max_it,
                                    int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_sw_bp(p, ntrials, d_vn_adj, false, true, d_chan_bits, W, max_it, max_it, d_counters, d_erased_bits,
                        scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

extern "C" int scldpc_swc_bp_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                          const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                          int32_t W, int32_t max_it,
                                          int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_sw_bp(p, ntrials, d_vn_adj16, true, true, d_chan_bits, W, max_it, max_it, d_counters, d_erased_bits,
                        scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

// workspace of scldpc_sw_bp_device / scldpc_swc_bp_device (and their _adj16 forms) for ntrials trials and window W
int64_t scldpc_sw_bp_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t W)
{
    uint64_t need = 0;
    const int rc = launch_sw_bp(p, ntrials, nullptr, true, false, nullptr, W, 1, 1, nullptr, nullptr,
                                scldpc::Scratch{nullptr, 0, &need}, nullptr);
    return rc ? (int64_t)rc : (int64_t)need;
}
