// Full flooding BP over the BEC for a batch of sampled SC-LDPC codes — gfx950 (MI355X) kernel.
//
// Replaces decodeBP of the reference (BPF:900-1140; BPT adds per-iteration rows and truncation,
// BPT:912-1051) bit for bit, but does O(E) work per trial instead of O(E·iterations):
// on the BEC, the a-posteriori erasure set after flooding iteration t equals the residual of
// level-synchronous parallel peeling after t rounds, so the kernel keeps
//   * one bit per VN  (U  = currently erased VNs), and
//   * one 32-bit word per CN: [cnt:4 | deg:4 | idsum:24] where cnt = #erased neighbours and
//     idsum = Σ of their VN ids — when cnt == 1 the idsum IS the id of the one erased neighbour,
//     so only the VN→CN table is ever read from HBM (no CN→VN lists, no per-edge messages),
// all of it in LDS (one workgroup = one trial), and walks a frontier of CNs with cnt == 1.
//
// Iteration t of the reference == one pass of the loop below:
//   phase B  count the valid frontier  → deg_1_iter        (BPF:969-978)
//   phase A  every frontier CN releases its VN; the releasing thread fetches that VN's dv CN ids
//            (one 16-B load), decrements those CNs and queues the ones that drop to cnt == 1
//   then     NumErasures / stop tests / optional trajectory row (BPF:1044-1065, BPT:988,1051)
// and afterwards the size-2 stopping-set expurgation (BPF:1067-1133) from the same CN words:
// an erased VN a belongs to such a pair iff every one of its CNs has cnt == 2 and the same partner
// idsum - a, located in a's position.
#include "common.h"
#include "kernel_util.h"

namespace {

using namespace scldpc_dev;

constexpr int kBlock = 1024;                 // 16 waves: one workgroup owns a CU's LDS
// scalar slots in LDS (double-buffered by iteration parity where noted)
enum { S_VALID0 = 0, S_VALID1, S_REMOVED0, S_REMOVED1, S_PUSH0, S_PUSH1, S_OVF0, S_OVF1,
       S_NE, S_EXTRA0, S_FIRST, S_NSCAL = 16 };

struct Layout {             // offsets in 32-bit words into dynamic LDS
    int cn_state, U, fbits, q0, q1, pos_cnt, pos_ss, scal, total;
    int qcap, nw, fwords;
};

struct Args {
    int dv, L, vns_pos, n, nk, cn_lim, max_it, rows_cap;
    Layout lay;
    const int32_t *vn_adj;
    const uint32_t *chan;
    int32_t *counters;
    int32_t *rows;
    uint32_t *erased_out;
};

template <bool TRAJ, int DV>
__global__ __launch_bounds__(kBlock) void full_bp_kernel(const Args a)
{
    extern __shared__ uint32_t lds[];
    uint32_t *cn_state = lds + a.lay.cn_state;
    uint32_t *U = lds + a.lay.U;
    uint32_t *fbits = lds + a.lay.fbits;
    uint32_t *q[2] = {lds + a.lay.q0, lds + a.lay.q1};
    int *pos_cnt = reinterpret_cast<int *>(lds + a.lay.pos_cnt);
    int *pos_ss = reinterpret_cast<int *>(lds + a.lay.pos_ss);
    int *scal = reinterpret_cast<int *>(lds + a.lay.scal);

    const int tid = threadIdx.x, lane = tid & 63;
    const int trial = blockIdx.x;
    const int n = a.n, nk = a.nk, dv = (DV ? DV : a.dv), cn_lim = a.cn_lim, nw = a.lay.nw, qcap = a.lay.qcap;
    const int32_t *adj = a.vn_adj + (size_t)trial * n * dv;
    const uint32_t *ch = a.chan + (size_t)trial * nw;

    // ---- load channel bits, clear CN words -------------------------------------------------
    for (int c = tid; c < nk; c += kBlock) cn_state[c] = 0;
    int ne_local = 0;
    for (int w = tid; w < nw; w += kBlock) {
        uint32_t x = ch[w];
        if (w == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        U[w] = x;
        ne_local += __popc(x);
    }
    if (tid < S_NSCAL) scal[tid] = 0;
    for (int i = tid; i < a.L; i += kBlock) { pos_cnt[i] = 0; pos_ss[i] = 0; }
    __syncthreads();
    ne_local = wave_sum(ne_local);
    if (lane == 0 && ne_local) atomicAdd(&scal[S_NE], ne_local);

    // ---- build: every erased VN adds (1, id) to its dv CNs (TRAJ: every VN also adds to deg) ----
    // Loads are unconditional so that each wave instruction reads 1 KiB contiguous (dv = 4).
    for (int j0 = tid; j0 < n; j0 += 4 * kBlock) {
        int32_t c[4][8];
        bool er[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * kBlock;
            er[u] = false;
            if (j < n) {
                load_adj<DV>(adj, dv, j, c[u]);
                er[u] = (U[j >> 5] >> (j & 31)) & 1u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * kBlock;
            if (j < n && (TRAJ || er[u])) {
                const uint32_t add = (TRAJ ? kDegOne : 0u) + (er[u] ? kCntOne + (uint32_t)j : 0u);
                for (int i = 0; i < dv; i++) atomicAdd(&cn_state[c[u][i]], add);
            }
        }
    }
    __syncthreads();

    int ne = scal[S_NE];
    const int nch = ne;
    int prec = n, iter = 0, ncur = 0, status = 0, rows_done = 0;
    bool scan = true;                   // iteration 0 has no queue yet: frontier = all CNs with cnt == 1
    int first_word = 0;

    for (;;) {
        const int par = iter & 1;
        uint32_t *qc = q[par], *qn = q[par ^ 1];

        // ---- phase B: validate + count the frontier (= deg_1_iter) -------------------------
        int valid = 0, extra = 0;
        if (scan) {
            for (int base = 0; base < cn_lim; base += kBlock) {
                const int c = base + tid;
                bool v = false;
                if (c < cn_lim) {
                    const uint32_t w = cn_state[c];
                    v = (w >> kCntShift) == 1u;
                    if (TRAJ && iter == 0)          // degree-1 CN whose only VN is known (BPF:973)
                        extra += ((w >> kCntShift) == 0u && ((w >> kDegShift) & kDegMask) == 1u);
                }
                const unsigned long long m = __ballot(v);
                if (c - lane < cn_lim) {            // this wave's 64-CN slice starts inside the range
                    if (lane == 0) fbits[c >> 5] = (uint32_t)m;
                    if (lane == 32) fbits[c >> 5] = (uint32_t)(m >> 32);
                }
                valid += v;
            }
        } else {
            for (int k = tid; k < ncur; k += kBlock) {
                const uint32_t c = qc[k];
                const bool v = (cn_state[c] >> kCntShift) == 1u;   // dropped to 0 within the round it was queued?
                if (!v) qc[k] = kInvalid;
                valid += v;
            }
        }
        valid = wave_sum(valid);
        if (lane == 0 && valid) atomicAdd(&scal[S_VALID0 + par], valid);
        if (TRAJ && iter == 0) {
            extra = wave_sum(extra);
            if (lane == 0 && extra) atomicAdd(&scal[S_EXTRA0], extra);
        }
        __syncthreads();                                            // (1)
        if (tid == 0) {         // counters of the NEXT iteration; last read before barrier (1)
            scal[S_VALID0 + (par ^ 1)] = 0; scal[S_REMOVED0 + (par ^ 1)] = 0;
            scal[S_PUSH0 + (par ^ 1)] = 0; scal[S_OVF0 + (par ^ 1)] = 0;
        }
        const int deg1 = scal[S_VALID0 + par] + ((TRAJ && iter == 0) ? scal[S_EXTRA0] : 0);

        // ---- phase A: release the VN of every frontier CN ----------------------------------
        int removed = 0;
        auto release = [&](uint32_t c) {
            const uint32_t w = cn_state[c];
            if ((w >> kCntShift) != 1u) return;                     // its VN was just released via another CN
            const uint32_t j = w & kSumMask, bit = 1u << (j & 31);
            const uint32_t old = atomicAnd(&U[j >> 5], ~bit);
            if (!(old & bit)) return;                               // lost the race for VN j
            removed++;
            int32_t cc[8];
            load_adj<DV>(adj, dv, (int)j, cc);
            for (int i = 0; i < dv; i++) {
                const uint32_t c2 = (uint32_t)cc[i];
                const uint32_t o = atomicSub(&cn_state[c2], kCntOne + j);
                if ((o >> kCntShift) == 2u && (int)c2 < cn_lim) {   // 2 → 1: candidate for the next round
                    const int idx = atomicAdd(&scal[S_PUSH0 + par], 1);
                    if (idx < qcap) qn[idx] = c2; else scal[S_OVF0 + par] = 1;
                }
            }
        };
        if (scan) {
            for (int base = 0; base < cn_lim; base += kBlock) {
                const int c = base + tid;
                if (c < cn_lim && ((fbits[c >> 5] >> (c & 31)) & 1u)) release((uint32_t)c);
            }
        } else {
            for (int k = tid; k < ncur; k += kBlock) {
                const uint32_t c = qc[k];
                if (c != kInvalid) release(c);
            }
        }
        removed = wave_sum(removed);
        if (lane == 0 && removed) atomicAdd(&scal[S_REMOVED0 + par], removed);
        __syncthreads();                                            // (2)

        // ---- bookkeeping, identical in every thread ----------------------------------------
        ne -= scal[S_REMOVED0 + par];
        const int recovered = prec - ne;
        if (TRAJ && (tid >> 6) == 0 && a.rows && rows_done < a.rows_cap) {
            // first erased VN (BPT:1037-1038): U only loses bits, so resume from the last hit
            int fw = first_word, first = n;
            while (fw < nw) {
                const uint32_t w = (fw + lane < nw) ? U[fw + lane] : 0u;
                const unsigned long long m = __ballot(w != 0u);
                if (m) {
                    const int l0 = __ffsll((long long)m) - 1;
                    const uint32_t w0 = __shfl(w, l0, 64);
                    fw += l0;
                    first = fw * 32 + (__ffs((int)w0) - 1);
                    break;
                }
                fw += 64;
            }
            first_word = fw;
            if (lane == 0) {
                int32_t *r = a.rows + ((size_t)trial * a.rows_cap + rows_done) * 3;
                r[0] = deg1; r[1] = recovered; r[2] = first / a.vns_pos;
            }
        }
        rows_done++;
        if (deg1 < recovered && iter > 0) { status = -1; break; }   // BPF:1035-1039
        if (ne == 0 || ne == prec) break;                           // BPF:1044-1045
        prec = ne;
        iter++;
        scan = scal[S_OVF0 + par] != 0;
        ncur = scan ? 0 : scal[S_PUSH0 + par];
        if (a.max_it > 0 && iter >= a.max_it) break;                // BPF:1065
    }
    __syncthreads();

    // ---- per-position erasure counts + size-2 stopping sets (BPF:1067-1133) -----------------
    if (ne > 0) {
        for (int w = tid; w < nw; w += kBlock) {
            uint32_t x = U[w];
            while (x) {
                const int b = __ffs((int)x) - 1;
                x &= x - 1;
                const int va = w * 32 + b, pos = va / a.vns_pos;
                atomicAdd(&pos_cnt[pos], 1);
                int32_t cc[8];
                load_adj<DV>(adj, dv, va, cc);
                bool pair = true;
                uint32_t partner = 0;
                for (int i = 0; i < dv; i++) {
                    const uint32_t s = cn_state[cc[i]];
                    const uint32_t b2 = (s & kSumMask) - (uint32_t)va;
                    if ((s >> kCntShift) != 2u || (i > 0 && b2 != partner)) { pair = false; break; }
                    partner = b2;
                }
                if (pair && (int)partner / a.vns_pos == pos) atomicAdd(&pos_ss[pos], 1);
            }
        }
    }
    __syncthreads();
    if (a.erased_out)
        for (int w = tid; w < nw; w += kBlock) a.erased_out[(size_t)trial * nw + w] = U[w];
    if (tid == 0) {
        int be = 0, ee = 0, bee = 0;
        for (int pos = 0; pos < a.L; pos++) {
            if (pos_cnt[pos] > 0) be++;
            const int e = pos_cnt[pos] - pos_ss[pos];
            if (e > 0 && bee == 0) { ee = e; bee = 1; }             // only the FIRST such position (BPF:1126-1132)
        }
        int32_t *o = a.counters + (size_t)trial * SCLDPC_NCOUNTERS;
        o[SCLDPC_C_NUM_ERASURES] = ne;
        o[SCLDPC_C_NUM_BLOCKS_ERR] = be;
        o[SCLDPC_C_NUM_ERASURES_EXP] = ee;
        o[SCLDPC_C_NUM_BLOCKS_ERR_EXP] = bee;
        o[SCLDPC_C_NUM_ERASURES_P1] = 0;
        o[SCLDPC_C_ITERATIONS] = rows_done;
        o[SCLDPC_C_STATUS] = status;
        o[SCLDPC_C_CHANNEL_ERASURES] = nch;
    }
}

int make_layout(const scldpc_code_params *p, Layout *lay)
{
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    int off = 0;
    auto take = [&](int words) { int o = off; off += (words + 3) & ~3; return o; };   // keep 16-B alignment
    lay->nw = (n + 31) / 32;
    lay->fwords = ((nk + 63) / 64) * 2;
    lay->cn_state = take(nk);
    lay->U = take(lay->nw + 64);              // +64: the first-erased scan may peek one wave past the end
    lay->fbits = take(lay->fwords);
    lay->pos_cnt = take(p->L);
    lay->pos_ss = take(p->L);
    lay->scal = take(S_NSCAL);
    const int left = scldpc::kMaxLdsBytes / 4 - off;
    int qcap = left / 2;
    qcap &= ~3;
    if (qcap > 8192) qcap = 8192;
    if (qcap < 64) return -1;
    lay->qcap = qcap;
    lay->q0 = take(qcap);
    lay->q1 = take(qcap);
    lay->total = off;
    return 0;
}

}  // namespace

extern "C" int64_t scldpc_full_bp_lds_bytes(const scldpc_code_params *p)
{
    if (int rc = scldpc::check_params(p)) return rc;
    Layout lay;
    if (make_layout(p, &lay)) return 4ll * (scldpc::nk_of(p) + scldpc::nw_of(p)) + (64 << 10);
    return 4ll * lay.total;
}

extern "C" int scldpc_full_bp_device(const scldpc_code_params *p, int32_t ntrials,
                                     const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                                     int32_t max_it, int32_t is_term,
                                     int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                                     uint32_t *d_erased_bits, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (ntrials < 0 || !d_counters || (ntrials > 0 && (!d_vn_adj || !d_chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_full_bp_device: null buffer or negative ntrials");
    if (d_rows && rows_cap <= 0)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_full_bp_device: d_rows given but rows_cap <= 0");
    if (ntrials == 0) return SCLDPC_OK;
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    if (p->dc > 15 || p->dv > 8 || (int64_t)p->dc * n >= (1ll << kDegShift))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE,
                                 "scldpc_full_bp_device: needs dc <= 15, dv <= 8, dc*n < 2^24 (got dc=%d dv=%d n=%d)",
                                 p->dc, p->dv, n);
    Args a{};
    if (make_layout(p, &a.lay))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE,
                                 "scldpc_full_bp_device: nk=%d CN words + n=%d VN bits do not fit 160 KiB of LDS", nk, n);
    a.dv = p->dv; a.L = p->L; a.vns_pos = p->vns_pos; a.n = n; a.nk = nk;
    a.cn_lim = is_term ? nk : p->L * p->cns_pos;                    // BPT:944-948
    a.max_it = max_it; a.rows_cap = d_rows ? rows_cap : 0;
    a.vn_adj = d_vn_adj; a.chan = d_chan_bits; a.counters = d_counters; a.rows = d_rows;
    a.erased_out = d_erased_bits;

    const bool traj = d_rows != nullptr;
    void (*kern)(const Args) = nullptr;
    if (p->dv == 4) kern = traj ? full_bp_kernel<true, 4> : full_bp_kernel<false, 4>;
    else            kern = traj ? full_bp_kernel<true, 0> : full_bp_kernel<false, 0>;
    const size_t lds_bytes = 4u * (size_t)a.lay.total;
    SCLDPC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kBlock), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}
