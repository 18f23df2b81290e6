// Peeling to the fixpoint without a barrier per level (dv = 4, packed 16-bit CN words in LDS) — shared by
// full_bp_fixpoint_kernel (full_bp.hip) and the packed path of peel_sweep_kernel (peel_sweep.hip).
//
// The closure of "a CN with exactly one erased neighbour resolves it" does not depend on the order in which CNs fire, so:
//   phase A  barrier rounds over a shared queue while the frontier is wide (a scan of the CN words opens the run and
//            repairs a queue overflow);
//   phase B  once it is narrow every wave keeps the entries it is dealt and the ones its own releases create in a private
//            queue and runs them level after level on its own — no workgroup barrier, no shared counter.
// Without a barrier between a CN's updates and its next read the two-atomic packed word needs an order and a check:
//   * a release XORs its id out of the fold FIRST and decrements the count SECOND; every release claims its VN in U before
//     either.  The thread whose decrement returns count 2 owns the CN's follow-up; of the two neighbours not yet
//     decremented one is its own VN, so the fold it gets back is the other one's id X if X is unclaimed, and X or 0 if X
//     is being released elsewhere;
//   * the follow-up is taken only if the decoded VN really has this CN on the decoded edge AND its claim in U succeeds.
//     An unclaimed erased neighbour of the CN can only be X itself, so nothing is ever released wrongly, and if the claim
//     fails X is on its way out anyway.
#pragma once
#include "cn_words.h"

namespace scldpc_dev {

struct PeelCtx {                // everything the loop touches; LDS pointers unless said otherwise
    int L, V, C, n, cn_lim, qcap;
    const char *adj;            // this trial's VN→CN table (global)
    uint32_t *cn_state, *U, *fbits, *q0, *q1;
    int *push3, *ovf3;          // two rotating triples of counters (zeroed by the caller)
};

// may_fire(c, round, one): may CN c be started from a SCAN in this round?  `one` = c < cn_lim and c shows one erased
// neighbour.  Called by every thread of every 64-CN slice of a scan round, in wave-uniform control flow (it may ballot).
// Returns the number of barrier rounds; `removed` += the VNs this thread released.
template <bool A16, int BLOCK, class MayFire>
__device__ __forceinline__ int peel_to_fixpoint(const Geo &a, const PeelCtx &x, int &removed, MayFire may_fire)
{
    using ST = Packed;
    constexpr int DV = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int L = x.L, V = x.V, C = x.C, n = x.n, cn_lim = x.cn_lim, qcap = x.qcap;
    const char *adj = x.adj;
    uint32_t *cn_state = x.cn_state, *U = x.U, *fbits = x.fbits;
    uint32_t *q[2] = {x.q0, x.q1};
    (void)n;
    int rounds = 0, ncur = 0;
    bool overflow = false;
    // One release step.  `e` = [half-word : 16 | CN : 16]: CN c believed to have exactly one erased neighbour, and the
    // packed half-word saying so (cnt 1, fold).  Releases that neighbour if it checks out and returns in out[0..3] the
    // entries of the CNs this release left with one erased neighbour (0 = none).
    auto step = [&](uint32_t e, uint32_t (&out)[DV]) {
#pragma unroll
        for (int i = 0; i < DV; i++) out[i] = 0;
        const int c = (int)(e & 0xFFFFu);
        const uint32_t h = e >> 16;
        if ((h >> 12) != 1u) return;
        const uint32_t lid = h & 0xFFFu;
        const int i1 = (int)__umulhi(lid, a.magic_v);                       // edge index lid / V
        const int pos_c = (int)__umulhi((uint32_t)c, a.magic_c);
        const int pos = pos_c - i1, t = (int)lid - i1 * V;
        if (i1 >= DV || pos < 0 || pos >= L) return;                        // a fold caught between two updates
        const int j = pos * V + t;
        int32_t cc[8];
        load_adj<DV, A16>(adj, DV, j, pos, C, cc);
        if (cc[i1] != c) return;                                            // not this CN's neighbour: same reason
        const uint32_t bit = 1u << (j & 31);
        if (!(atomicAnd(&U[j >> 5], ~bit) & bit)) return;                   // already released, or being released elsewhere
        removed++;
#pragma unroll
        for (int i = 0; i < DV; i++) atomicXor(&cn_state[cc[i] >> 1], (uint32_t)(i * V + t) << ((cc[i] & 1) * 16));
#pragma unroll
        for (int i = 0; i < DV; i++) {
            const int sh = (cc[i] & 1) * 16;
            const uint32_t w = (atomicSub(&cn_state[cc[i] >> 1], 0x1000u << sh) >> sh) & 0xFFFFu;
            if ((w >> 12) == 2u && cc[i] < cn_lim) {
                out[i] = ((0x1000u | (w & 0xFFFu)) << 16) | (uint32_t)cc[i];
            }
        }
    };
    auto entry_of = [&](int c) { return (ST::half(cn_state, c) << 16) | (uint32_t)c; };

    // ---- phase A: barrier rounds over the shared queue while the frontier is wide --------------------------------
    // ---- phase B: once it is narrow every wave keeps the entries it gets and the ones its own releases create in a
    //      private queue and runs them level by level on its own: no workgroup barrier, no shared counter -------------
    // every wave takes part (measured: 16 waves with a few entries each beat 4 waves with many — the waves' dependent
    // steps overlap); the shared queue is dealt out round-robin
    constexpr int kWaves = BLOCK / 64, kSwitch = 16 * kWaves;
    const int wave = tid >> 6;
    const int wcap = (qcap / kWaves) & ~1;          // private queue space per wave: two halves of wcap/2 entries
    bool scan = true;
    for (;;) {
        const int g = rounds % 3, gn = (rounds + 1) % 3;
        uint32_t *qc = q[rounds & 1], *qn = q[(rounds + 1) & 1];
        if (tid == 0) { x.push3[gn] = 0; x.ovf3[gn] = 0; }
        auto push_shared = [&](const uint32_t (&out)[DV]) {
#pragma unroll
            for (int i = 0; i < DV; i++) {
                if (out[i]) {
                    const int idx = atomicAdd(&x.push3[g], 1);
                    if (idx < qcap) qn[idx] = out[i]; else overflow = true;
                }
            }
        };
        if (scan) {
            // every CN < cn_lim that shows one erased neighbour right now and may fire (a stale entry dies in step())
            for (int base = 0; base < cn_lim; base += BLOCK) {
                const int c = base + tid;
                const bool v = may_fire(c, rounds, c < cn_lim && ST::cnt(cn_state, c) == 1u);
                const unsigned long long m = __ballot(v);
                if (c - lane < cn_lim) {
                    if (lane == 0) fbits[c >> 5] = (uint32_t)m;
                    if (lane == 32) fbits[c >> 5] = (uint32_t)(m >> 32);
                }
            }
            __syncthreads();
            for (int base = 0; base < cn_lim; base += BLOCK) {
                const int c = base + tid;
                if (c < cn_lim && ((fbits[c >> 5] >> (c & 31)) & 1u)) { uint32_t out[DV]; step(entry_of(c), out); push_shared(out); }
            }
        } else if (ncur > kSwitch || wcap < 128) {
            for (int k = tid; k < ncur; k += BLOCK) { uint32_t out[DV]; step(qc[k], out); push_shared(out); }
        } else {
            // phase B.  Wave w takes entries w, w+16, ... of the shared queue into its private queue (in qn, which nobody
            // else touches now), then runs to exhaustion.
            uint32_t *mine = qn + wave * wcap;
            const int half_cap = wcap / 2;
            int cnt = (ncur - wave + kWaves - 1) / kWaves, cur = 0;         // entries wave, wave + kWaves, ...
            if (cnt < 0) cnt = 0;
            if (lane < cnt) mine[lane] = qc[wave + lane * kWaves];
            while (cnt > 0) {
                uint32_t *src = mine + cur * half_cap, *dst = mine + (cur ^ 1) * half_cap;
                int ncnt = 0;
                for (int base = 0; base < cnt; base += 64) {
                    uint32_t out[DV];
#pragma unroll
                    for (int i = 0; i < DV; i++) out[i] = 0;
                    if (base + lane < cnt) step(src[base + lane], out);
#pragma unroll
                    for (int i = 0; i < DV; i++) {                          // append: wave-synchronous, no atomics
                        const unsigned long long m = __ballot(out[i] != 0u);
                        if (out[i]) {
                            const int idx = ncnt + __popcll(m & ((1ull << lane) - 1ull));
                            if (idx < half_cap) dst[idx] = out[i]; else overflow = true;
                        }
                        ncnt += __popcll(m);
                    }
                }
                cnt = min(ncnt, half_cap);
                cur ^= 1;
            }
        }
        if (overflow) x.ovf3[g] = 1;
        __syncthreads();
        rounds++;
        overflow = false;
        scan = x.ovf3[g] != 0;            // an overflowing queue dropped CNs: find them by a scan
        ncur = scan ? 0 : x.push3[g];
        if (!scan && ncur == 0) break;
    }
    return rounds;
}

}  // namespace scldpc_dev
