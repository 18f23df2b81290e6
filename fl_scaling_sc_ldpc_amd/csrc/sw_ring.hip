// Square sliding-window BP (decodeBP_SW, BPW:628-912) with the window's state — and nothing else — in LDS: gfx950.
//
// sw_bp.hip keeps one word per CN of the WHOLE chain (in LDS up to N = 1024, in a device workspace beyond: L2 atomics,
// two trials per CU).  But window posW only ever fires CNs of positions [posW, posW+W) and only releases VNs of positions
// [posW, posW+W), whose edges reach CN positions up to posW+W+2: everything to the left is frozen for good (BPW:672-693,
// SURVEY.md §7.4 B), everything further right has not been looked at yet.  So the kernel keeps a RING over positions:
//   * CN counts, 4 bits per CN, for positions [posW-4, posW+W+2]  (W+7 slots; the four slots behind the window serve the
//     size-2 stopping-set expurgation, see below),
//   * S bits (what the CNs still see as erased), for VN positions [posW-3, posW+W]  (W+4 slots),
// 21 KB per trial at (L=100, N=2000, W=10) instead of 412 KB of CN words in a workspace: seven 256-thread workgroups per
// CU with all atomics in LDS.  A VN position ENTERS the ring when the window first reaches it (its channel bits are read,
// its erased VNs count themselves into their dv CN positions — the right-most of which is a fresh slot), and leaves it
// dv positions after it froze.  CN counts see every erased neighbour that has entered; a window CN (position <= posW+W-1)
// has all its neighbours (positions >= its own - dv + 1) in the ring's past, so its count is exact, as in sw_bp.hip.
//
// Like full_bp_small.hip the CN state is the count only: a CN whose count is one finds its lone erased neighbour in the
// CN -> socket table (uint16 [nk][dc], socket s = dv*t + i = edge i of VN t of position CNpos - i; built from the VN -> CN
// table by scldpc_cn_sockets_device below) as the socket whose S bit is still set.
//
// One flooding iteration == one barrier round over the CNs whose count was one when it began, so iteration caps (init_it /
// max_it, BPW:699-702, 839) and the stop rule (window erasures zero or unchanged, BPW:815-816) are the reference's.  The
// frontier is a queue: the CNs the previous round took from two to one, and — when the window moves on — those plus the
// count-one CNs of the single CN position that entered the window (a scan of the whole window only opens the first window
// and repairs a queue overflow).
// Size-2 stopping sets (BPW:850-908: every failing position contributes): VN position q is examined when the window has
// moved dv positions past it — then its dv CN positions are final and still in the ring; a qualifying partner lies in the
// same position, so only S bits of position q are consulted.
#include "common.h"
#include "kernel_util.h"

namespace {

using namespace scldpc_dev;

constexpr int kBlock = 256;
enum { R_NCH = 0, R_PUSH = 1, R_OVF = 4, R_REM = 7, R_TMP = 10, R_N = 12 };      // PUSH / OVF / REM rotate three ways: one barrier per iteration

struct RArgs {
    int dv, dc, L, V, C, n, nk, W, max_it, init_it, nw;
    int R, RV, Cw, wpp;             // ring slots for CN / VN positions; words of count nibbles / S bits per position
    int off_S, off_fb, off_q0, off_q1, off_pos, off_scal, total, qcap;
    const uint16_t *vn_adj16;       // [T][n][dv]
    const uint16_t *cn_sock16;      // [T][nk][dc]
    const uint32_t *chan;
    int32_t *counters;
    uint32_t *erased_out;
};

// bits [b0, b0 + 32) of a packed bit array (b0 need not be word-aligned); words beyond nw read as zero
__device__ __forceinline__ uint32_t bits_at(const uint32_t *w, int nw, long long b0)
{
    const int i = (int)(b0 >> 5), sh = (int)(b0 & 31);
    const uint32_t lo = i < nw ? w[i] : 0u, hi = (sh && i + 1 < nw) ? w[i + 1] : 0u;
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
}

template <int DV, int DC>
__global__ __launch_bounds__(kBlock, 8) __attribute__((amdgpu_num_sgpr(80))) void sw_ring_kernel(const RArgs a)
{
    extern __shared__ uint32_t lds[];
    uint32_t *cnt = lds;                                                  // [R][Cw] words of 8 count nibbles
    uint32_t *S = lds + a.off_S;                                          // [RV][wpp]
    uint8_t *fbits = reinterpret_cast<uint8_t *>(lds + a.off_fb);         // snapshot of the window's count-one CNs: [W][Cw] bytes, bit k = CN 8w + k
    uint32_t *q[2] = {lds + a.off_q0, lds + a.off_q1};
    int *pos_cnt = reinterpret_cast<int *>(lds + a.off_pos);
    int *pos_ss = pos_cnt + a.L;
    int *scal = reinterpret_cast<int *>(lds + a.off_scal);

    const int tid = threadIdx.x, lane = tid & 63;
    const int trial = blockIdx.x;
    const int L = a.L, V = a.V, C = a.C, W = a.W, R = a.R, RV = a.RV, Cw = a.Cw, wpp = a.wpp, qcap = a.qcap;
    const int D = L + DV - 1;
    const uint2 *vrow = reinterpret_cast<const uint2 *>(a.vn_adj16) + (size_t)trial * a.n;
    const uint16_t *crow = a.cn_sock16 + (size_t)trial * a.nk * DC;
    const uint32_t *ch = a.chan + (size_t)trial * a.nw;
    uint32_t *eout = a.erased_out ? a.erased_out + (size_t)trial * a.nw : nullptr;

    // Ring slots without a division (a release would pay a dozen of them): every position touched while the window
    // stands at pw lies within [pw - (dv-1), pw + W + dv - 1], less than a ring apart, so one wrap of pw's own slot does.
    int pw = 0, cb = 0, vb = 0;                                           // window position, pw % R, pw % RV
    auto cslot = [&](int p) {                                             // word base of CN position p
        int sl = cb + (p - pw);
        sl -= sl >= R ? R : 0;
        sl += sl < 0 ? R : 0;
        return sl * Cw;
    };
    auto sslot = [&](int qq) {                                            // word base of VN position qq
        int sl = vb + (qq - pw);
        sl -= sl >= RV ? RV : 0;
        sl += sl < 0 ? RV : 0;
        return sl * wpp;
    };
    auto nib = [&](int p, int l) { return (cnt[cslot(p) + (l >> 3)] >> ((l & 7) * 4)) & 15u; };

    for (int i = tid; i < R * Cw; i += kBlock) cnt[i] = 0;
    for (int i = tid; i < RV * wpp; i += kBlock) S[i] = 0;
    for (int i = tid; i < 2 * L; i += kBlock) pos_cnt[i] = 0;
    if (tid < R_N) scal[tid] = 0;
    if (eout) for (int w = tid; w < a.nw; w += kBlock) eout[w] = 0;
    __syncthreads();

    // ---- a VN position enters the ring: channel bits, per-position count, its erased VNs into their CN positions ----
    auto enter = [&](int qq) {
        if (qq + DV - 1 < D) for (int i = tid; i < Cw; i += kBlock) cnt[cslot(qq + DV - 1) + i] = 0;     // a fresh CN slot
        int mine = 0;
        for (int w = tid; w < wpp; w += kBlock) {
            uint32_t x = bits_at(ch, a.nw, (long long)qq * V + 32ll * w);
            if (w * 32 + 32 > V) x &= (1u << (V - w * 32)) - 1u;
            S[sslot(qq) + w] = x;
            mine += __popc(x);
        }
        mine = wave_sum(mine);
        if (lane == 0 && mine) { atomicAdd(&pos_cnt[qq], mine); atomicAdd(&scal[R_NCH], mine); }
        __syncthreads();
        // rows are loaded unconditionally, four per thread in flight (coalesced 8-byte loads), then the erased ones count
        for (int t0 = tid; t0 < V; t0 += 4 * kBlock) {
            uint2 r[4];
            bool er[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + u * kBlock;
                er[u] = false;
                if (t < V) { r[u] = vrow[(size_t)qq * V + t]; er[u] = (S[sslot(qq) + (t >> 5)] >> (t & 31)) & 1u; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!er[u]) continue;
                const uint32_t l[4] = {r[u].x & 0xFFFFu, r[u].x >> 16, r[u].y & 0xFFFFu, r[u].y >> 16};
#pragma unroll
                for (int i = 0; i < DV; i++) atomicAdd(&cnt[cslot(qq + i) + (l[i] >> 3)], 1u << ((l[i] & 7) * 4));
            }
        }
        __syncthreads();
    };

    // ---- size-2 stopping sets of VN position qe, whose CN positions qe .. qe+dv-1 are final (BPW:857-908) ----------
    auto expurgate = [&](int qe) {
        if (pos_cnt[qe] == 0) return;                                     // uniform: pos_cnt is read after a barrier
        for (int t = tid; t < V; t += kBlock) {
            if (!((S[sslot(qe) + (t >> 5)] >> (t & 31)) & 1u)) continue;
            const uint2 r = vrow[(size_t)qe * V + t];
            const uint32_t l[4] = {r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16};
            bool pair = true;
#pragma unroll
            for (int i = 0; i < DV; i++) pair = pair && nib(qe + i, (int)l[i]) == 2u;
            if (!pair) continue;
            int partner = -1;
            for (int i = 0; i < DV && pair; i++) {                        // the other erased neighbour, if it is in position qe
                const uint16_t *row = crow + ((size_t)(qe + i) * C + l[i]) * DC;
                int other = -1;
                for (int k = 0; k < DC; k++) {
                    const uint32_t s = row[k];
                    if (s == 0xFFFFu) continue;
                    const int i2 = (int)(s % DV), t2 = (int)(s / DV);
                    if (i2 == i && t2 != t && ((S[sslot(qe) + (t2 >> 5)] >> (t2 & 31)) & 1u)) other = t2;
                }
                if (other < 0 || (i > 0 && other != partner)) pair = false;
                partner = other;
            }
            if (pair) atomicAdd(&pos_ss[qe], 1);
        }
    };
    // VNerased of a frozen position into the packed output bits
    auto emit_erased = [&](int qq) {
        if (!eout) return;
        for (int w = tid; w < wpp; w += kBlock) {
            const uint32_t x = S[sslot(qq) + w];
            if (!x) continue;
            const long long b0 = (long long)qq * V + 32ll * w;
            const int i = (int)(b0 >> 5), sh = (int)(b0 & 31);
            atomicOr(&eout[i], x << sh);
            if (sh && (x >> (32 - sh))) atomicOr(&eout[i + 1], x >> (32 - sh));
        }
    };

    for (int qq = 0; qq < min(W, L); qq++) enter(qq);

    int iters_total = 0, gen = 0;
    int carry_n = 0, phi_prev = 0;
    bool carry_ok = false;                                                // the last round's queue is complete (no overflow)
    for (int posW = 0; posW < L; posW++) {
        const int phi = min(posW + W, D);                                 // CN positions [posW, phi)   (BPW:674-676)
        const int qhi = min(posW + W, L);                                 // VN positions [posW, qhi)   (BPW:691-693)
        const int cap = posW == 0 ? a.init_it : a.max_it;                 // BPW:699-702
        int iter = 0, prec = a.n, ncur = 0;
        bool scan = true;                                                 // the first window opens with a scan of its CNs
        if (carry_ok) {
            // Later windows: the count-one CNs of [posW, phi) are (a) those the previous window's last round queued — a CN
            // that held one erased neighbour earlier has fired since (count zero) or waits for a frozen VN for good — and
            // (b) those of the one CN position that has just entered the window (never queued: pushes stop at phi), whose
            // counts the VN position entered above has completed.  No snapshot needed: nothing is in flight here.
            uint32_t *qc = q[gen & 1];
            if (tid == 0) scal[R_TMP] = carry_n;
            __syncthreads();
            if (phi > phi_prev) {
                const int pn = phi - 1;
                for (int w0 = (tid >> 6) * 64; w0 < Cw; w0 += kBlock) {
                    const int w = w0 + lane;
                    uint32_t z = 0;
                    if (w < Cw) {
                        const uint32_t y = cnt[cslot(pn) + w] ^ 0x11111111u;
                        z = ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;
                    }
                    const int mine = __popc(z);
                    const int incl = (int)wave_inclusive_scan((uint32_t)mine);
                    const int tot = __builtin_amdgcn_readlane(incl, 63);
                    if (tot == 0) continue;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&scal[R_TMP], tot);
                    int idx = __builtin_amdgcn_readfirstlane(base) + incl - mine;
                    while (z) {
                        const int k = (__ffs((int)z) - 1) >> 2;
                        z &= z - 1;
                        if (idx < qcap) qc[idx] = ((uint32_t)pn << 16) | (uint32_t)(w * 8 + k);
                        idx++;
                    }
                }
            }
            __syncthreads();
            const int tot = scal[R_TMP];
            if (tot <= qcap) { ncur = tot; scan = false; }                // else the full scan below finds them all
        }
        phi_prev = phi;
        int term = 0;
        for (int qq = posW; qq < qhi; qq++) term += pos_cnt[qq];          // erasures inside the window (BPW:791-809)
        for (;;) {
            uint32_t *qc = q[gen & 1], *qn = q[(gen + 1) & 1];
            int *push = &scal[R_PUSH + (gen + 1) % 3], *ovf = &scal[R_OVF + (gen + 1) % 3], *rem = &scal[R_REM + gen % 3];
            // the counters of the round after next: nobody reads or writes them before the next barrier
            if (tid == 0) { scal[R_PUSH + (gen + 2) % 3] = 0; scal[R_OVF + (gen + 2) % 3] = 0; scal[R_REM + (gen + 1) % 3] = 0; }
            int removed = 0;
            // CN (p, l) of the snapshot: release its lone erased neighbour unless that one is frozen
            // out[i] = 1 + [CN position | CN] of edge i if this release left that CN with one erased neighbour inside the window
            auto release = [&](int p, int l, uint32_t (&out)[4]) {
                if (p < posW) return;                                     // queued by the previous window for the position it left
                const uint4 s4 = *reinterpret_cast<const uint4 *>(crow + ((size_t)p * C + l) * DC);
                const uint32_t sk[8] = {s4.x & 0xFFFFu, s4.x >> 16, s4.y & 0xFFFFu, s4.y >> 16,
                                        s4.z & 0xFFFFu, s4.z >> 16, s4.w & 0xFFFFu, s4.w >> 16};
                int jq = -1, jt = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (sk[k] == 0xFFFFu) continue;
                    const int qq = p - (int)(sk[k] % DV), t = (int)(sk[k] / DV);
                    if ((unsigned)qq < (unsigned)L && ((S[sslot(qq) + (t >> 5)] >> (t & 31)) & 1u)) { jq = qq; jt = t; }
                }
                if (jq < posW) return;                                    // none left (released this round) or frozen (BPW:745)
                const uint2 r = vrow[(size_t)jq * V + jt];                // issued before the claim: overlaps its round trip
                const uint32_t bit = 1u << (jt & 31);
                if (!(atomicAnd(&S[sslot(jq) + (jt >> 5)], ~bit) & bit)) return;
                atomicSub(&pos_cnt[jq], 1);
                removed++;
                const uint32_t ll[4] = {r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16};
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < DV; i++)                              // the dv returning atomics go out back to back
                    o[i] = atomicSub(&cnt[cslot(jq + i) + (ll[i] >> 3)], 1u << ((ll[i] & 7) * 4));
#pragma unroll
                for (int i = 0; i < DV; i++)
                    if (((o[i] >> ((ll[i] & 7) * 4)) & 15u) == 2u && jq + i < phi)       // 2 -> 1 inside the window: fires next iteration
                        out[i] = 1u + (((uint32_t)(jq + i) << 16) | ll[i]);
            };
            // a wave appends its lanes' entries behind *push: one prefix scan + one LDS atomic per wave
            auto append = [&](const uint32_t (&out)[4]) {
                const int mine = (out[0] != 0u) + (out[1] != 0u) + (out[2] != 0u) + (out[3] != 0u);
                const int incl = (int)wave_inclusive_scan((uint32_t)mine);
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                if (tot == 0) return;
                int base = 0;
                if (lane == 0) base = atomicAdd(push, tot);
                int idx = __builtin_amdgcn_readfirstlane(base) + incl - mine;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (out[i]) { if (idx < qcap) qn[idx] = out[i] - 1u; else *ovf = 1; idx++; }
            };
            if (scan) {
                // snapshot {c in the window : count == 1} first: this round's releases must not promote CNs into it
                for (int p = posW; p < phi; p++)
                    for (int w = tid; w < Cw; w += kBlock) {
                        const uint32_t y = cnt[cslot(p) + w] ^ 0x11111111u;
                        uint32_t z = (~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u) >> 3;      // bit 4k: CN 8w + k
                        z = (z | (z >> 3)) & 0x03030303u;
                        z = (z | (z >> 6)) & 0x000F000Fu;
                        fbits[(p - posW) * Cw + w] = (uint8_t)((z | (z >> 12)) & 0xFFu);
                    }
                __syncthreads();
                for (int p = posW; p < phi; p++)
                    for (int w0 = (tid >> 6) * 64; w0 < Cw; w0 += kBlock) {          // wave-uniform trip count (append scans)
                        const int w = w0 + lane;
                        uint32_t z = w < Cw ? (uint32_t)fbits[(p - posW) * Cw + w] : 0u;
                        while (__any(z != 0u)) {
                            uint32_t out[4] = {0, 0, 0, 0};
                            if (z) {
                                const int k = __ffs((int)z) - 1;
                                z &= z - 1;
                                if (w * 8 + k < C) release(p, w * 8 + k, out);
                            }
                            append(out);
                        }
                    }
            } else {
                for (int k0 = (tid >> 6) * 64; k0 < ncur; k0 += kBlock) {
                    uint32_t out[4] = {0, 0, 0, 0};
                    if (k0 + lane < ncur) release((int)(qc[k0 + lane] >> 16), (int)(qc[k0 + lane] & 0xFFFFu), out);
                    append(out);
                }
            }
            removed = wave_sum(removed);
            if (lane == 0 && removed) atomicAdd(rem, removed);
            __syncthreads();                                              // end of the flooding iteration
            iters_total++;
            term -= *rem;
            scan = *ovf != 0;                                             // a full queue dropped CNs: find them by a scan
            ncur = min(*push, qcap);
            gen++;
            if (term == 0 || term == prec) break;                         // BPW:815-816
            prec = term;
            iter++;
            if (!(iter < cap)) break;                                     // BPW:839
        }
        carry_ok = !scan;
        carry_n = ncur;
        // position posW is decided (BPW:759-788): its S bits are VNerased from now on
        emit_erased(posW);
        if (posW + W < L) enter(posW + W);                                // the next window's new position
        else __syncthreads();
        if (posW - (DV - 1) >= 0) expurgate(posW - (DV - 1));             // CN positions up to posW are final now
        pw = posW + 1;
        cb = cb + 1 == R ? 0 : cb + 1;
        vb = vb + 1 == RV ? 0 : vb + 1;
    }
    __syncthreads();
    for (int qe = max(L - (DV - 1), 0); qe < L; qe++) expurgate(qe);       // the last positions: everything is final
    __syncthreads();

    if (tid == 0) {
        int ne = 0, be = 0, ee = 0, bee = 0, p1 = 0;
        const int ms = DV - 1;
        for (int pos = 0; pos < L; pos++) {
            const int c = pos_cnt[pos];
            ne += c;
            if (c > 0) be++;
            if (pos >= ms && pos <= W - 2) p1 += c;                       // NumErasuresP1 (BPW:846-847)
            const int e = c - pos_ss[pos];
            if (e > 0) { ee += e; bee++; }                                // every position (BPW:903-907)
        }
        int32_t *o = a.counters + (size_t)trial * SCLDPC_NCOUNTERS;
        o[SCLDPC_C_NUM_ERASURES] = ne;
        o[SCLDPC_C_NUM_BLOCKS_ERR] = be;
        o[SCLDPC_C_NUM_ERASURES_EXP] = ee;
        o[SCLDPC_C_NUM_BLOCKS_ERR_EXP] = bee;
        o[SCLDPC_C_NUM_ERASURES_P1] = p1;
        o[SCLDPC_C_ITERATIONS] = iters_total;
        o[SCLDPC_C_STATUS] = 0;
        o[SCLDPC_C_CHANNEL_ERASURES] = scal[R_NCH];
    }
}

// ---- CN -> socket table from the VN -> CN table: one workgroup per (trial, CN position) ------------------------------
struct IArgs {
    int dv, dc, L, V, C, n, nk;
    const uint16_t *vn_adj16;
    uint16_t *cn_sock16;
};

__global__ __launch_bounds__(256) void cn_sockets_kernel(const IArgs a)
{
    extern __shared__ uint32_t lds[];
    uint32_t *fill = lds;                                                 // C counters
    uint16_t *stage = reinterpret_cast<uint16_t *>(lds + ((a.C + 3) & ~3));      // C * dc sockets
    const int D = a.L + a.dv - 1;
    const int trial = blockIdx.x / D, p = blockIdx.x % D, tid = threadIdx.x;
    for (int i = tid; i < a.C; i += 256) fill[i] = 0;
    for (int i = tid; i < (a.C * a.dc + 1) / 2; i += 256) reinterpret_cast<uint32_t *>(stage)[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint16_t *rows = a.vn_adj16 + (size_t)trial * a.n * a.dv;
    for (int i = 0; i < a.dv; i++) {
        const int qq = p - i;
        if (qq < 0 || qq >= a.L) continue;
        for (int t = tid; t < a.V; t += 256) {
            const uint32_t l = rows[((size_t)qq * a.V + t) * a.dv + i];
            const uint32_t slot = atomicAdd(&fill[l], 1u);
            if (slot < (uint32_t)a.dc) stage[l * a.dc + slot] = (uint16_t)(a.dv * t + i);
        }
    }
    __syncthreads();
    uint16_t *dst = a.cn_sock16 + ((size_t)trial * a.nk + (size_t)p * a.C) * a.dc;
    for (int i = tid; i < a.C * a.dc; i += 256) dst[i] = stage[i];
}

int ring_args(const scldpc_code_params *p, int W, RArgs *a)
{
    a->dv = p->dv; a->dc = p->dc; a->L = p->L; a->V = p->vns_pos; a->C = p->cns_pos;
    a->n = scldpc::n_of(p); a->nk = scldpc::nk_of(p); a->W = W; a->nw = scldpc::nw_of(p);
    a->R = W + 2 * p->dv - 1; a->RV = W + p->dv;
    a->Cw = (p->cns_pos + 7) / 8; a->wpp = (p->vns_pos + 31) / 32;
    int off = 0;
    auto take = [&](int words) { int o = off; off += (words + 3) & ~3; return o; };
    take(a->R * a->Cw);
    a->off_S = take(a->RV * a->wpp);
    a->off_fb = take((W * a->Cw + 3) / 4);
    a->off_pos = take(2 * p->L);
    a->off_scal = take(R_N);
    int per_cu = 8;                                                     // aim at eight workgroups per CU (all 32 wave slots)
    if (const char *v = getenv("SCLDPC_DEBUG_RING_PER_CU")) per_cu = atoi(v) == 7 ? 7 : 8;
    int qcap = (scldpc::kMaxLdsBytes / per_cu / 4 - 64 - off) / 2;
    if (qcap < 512) qcap = 512;
    if (qcap > 4096) qcap = 4096;
    qcap &= ~3;
    a->qcap = qcap;
    a->off_q0 = take(qcap);
    a->off_q1 = take(qcap);
    a->total = off;
    return 4 * off <= scldpc::kMaxLdsBytes ? 0 : -1;
}

}  // namespace

// 1 when scldpc_sw_bp_ring_device takes (p, W): the square window of the (dv = 4, dc = 8) chain with 2-byte tables
extern "C" int scldpc_sw_bp_ring_supported(const scldpc_code_params *p, int32_t W)
{
    if (scldpc::check_params(p) || W < 1) return 0;
    RArgs a{};
    return p->dv == 4 && p->dc == 8 && p->cns_pos <= 65536 && (int64_t)p->vns_pos * p->dv <= 65535 && ring_args(p, W, &a) == 0;
}

extern "C" int scldpc_cn_sockets_device(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                        uint16_t *d_cn_sock16, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (ntrials < 0 || (ntrials > 0 && (!d_vn_adj16 || !d_cn_sock16)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_cn_sockets_device: null buffer or negative ntrials");
    if ((int64_t)p->vns_pos * p->dv > 65535 || p->cns_pos > 65536)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_cn_sockets_device: sockets and CN ids must fit 16 bits");
    if (ntrials == 0) return SCLDPC_OK;
    IArgs a{p->dv, p->dc, p->L, p->vns_pos, p->cns_pos, scldpc::n_of(p), scldpc::nk_of(p), d_vn_adj16, d_cn_sock16};
    const size_t lds_bytes = 4u * (size_t)((p->cns_pos + 3) & ~3) + 2u * (size_t)p->cns_pos * p->dc + 16;
    if (lds_bytes > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_cn_sockets_device: one CN position does not fit the LDS");
    const long long blocks = (long long)ntrials * (p->L + p->dv - 1);
    if (blocks > 0x7FFFFFFFll)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_cn_sockets_device: too many (trial, position) pairs for one launch");
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(cn_sockets_kernel))) return rc_;
    hipLaunchKernelGGL(cn_sockets_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_sw_bp_ring_device(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                        const uint16_t *d_cn_sock16, const uint32_t *d_chan_bits, int32_t W, int32_t max_it,
                                        int32_t init_it, int32_t *d_counters, uint32_t *d_erased_bits, void *stream)
{
    const char *who = "scldpc_sw_bp_ring_device";
    if (int rc = scldpc::check_params(p)) return rc;
    if (ntrials < 0 || (ntrials > 0 && (!d_counters || !d_vn_adj16 || !d_cn_sock16 || !d_chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: null buffer or negative ntrials", who);
    if (W < 1 || max_it < 0 || init_it < 0)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: need W >= 1, max_it >= 0, init_it >= 0", who);
    if (!scldpc_sw_bp_ring_supported(p, W))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: takes dv = 4, dc = 8, 16-bit sockets and a window that fits the LDS", who);
    if (ntrials == 0) return SCLDPC_OK;
    RArgs a{};
    ring_args(p, W, &a);
    a.max_it = max_it; a.init_it = init_it ? init_it : max_it;           // BPW:2101-2102
    a.vn_adj16 = d_vn_adj16; a.cn_sock16 = d_cn_sock16; a.chan = d_chan_bits;
    a.counters = d_counters; a.erased_out = d_erased_bits;
    void (*kern)(const RArgs) = sw_ring_kernel<4, 8>;
    const size_t lds_bytes = 4u * (size_t)a.total;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kBlock), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}
