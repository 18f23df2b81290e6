// The fixpoint of unlimited flooding BP (decodeBP, BPF:900-1140) with 4 bits of LDS per check node — gfx950.
//
// full_bp.hip keeps [count | fold of the erased neighbours' ids] per CN (16 bits) so that a CN with one erased neighbour
// names it without a lookup; the 52 KiB of CN words then limit a CU to two trials in flight, and the decoder is bound by
// the latency of its ~230 dependent levels.  Here a CN keeps ONLY the count (a nibble): when it drops to
// one, its dc neighbours are read from the CN -> VN table the second-generation sampler emits (sampler_v2.hip) and the
// one neighbour whose bit in the erased-VN bitmap U is still set is the one to resolve.  A trial needs 13 + 6 KiB of
// state + queues = 23 KiB, a 256-thread workgroup decodes it, and seven trials share a CU: 3.5 times the trials in
// flight for one more dependent gather per level (measured A/B of workgroup sizes 64 … 512, five to eight trials per
// CU and the switch-over width: DESIGN.md §5).
//
// Safe without a barrier per level because every release claims its VN in U (atomic test-and-clear) BEFORE it decrements
// the VN's CNs: whoever sees a CN's count reach one (its decrement returned two, or a scan read one) sees at most one
// neighbour with its U bit still set — the unclaimed one — and if that neighbour is being released elsewhere at that
// moment the bit is already clear and the entry is dropped (that release will take the count to zero).
//
// LEVEL = true is the same machine walked one flooding iteration per barrier round (scldpc_full_bp_device_cn16): a round
// releases exactly the CNs whose count was one when it began, so round t is the reference's iteration t (SURVEY.md §7.4 A;
// full_bp.hip does the same on 16-bit CN words with two trials per CU) — iteration count, the cap MaxNumIt (BPF:1065), the
// stop tests (BPF:1044-1045) and deg_1_iter's invariant (BPF:1035-1039) included.  Rounds whose frontier does not fit the
// queue (the first one or two) take it from a snapshot bitmap, one queue-full at a time; that bitmap costs a seventh trial
// per CU (six fit).
//
// Outputs: the counters of scldpc_full_bp_fixpoint_device (everything decodeBP reports except the iteration count), or
// with LEVEL all of scldpc_full_bp_device's counters.
// The size-2 stopping-set expurgation only looks at what the reference reports: the FIRST position with a positive
// expurgated count (is_first_printed, BPF:1074, 1126-1132), so only that position's erased VNs are examined.
#include "common.h"
#include "kernel_util.h"
#include <algorithm>
#include <cstdlib>

namespace {

using namespace scldpc_dev;

enum { SC_NE = 0, SC_REM, SC_N0, SC_N1, SC_OVF, SC_Q, SC_N = 8 };
// LEVEL: per-iteration counters rotated three ways, so that one barrier per iteration is enough (as in full_bp.hip)
enum { LV_VALID = 2, LV_EXTRA0 = 3, LV_PUSH = 6, LV_DROP = 9, LV_REM = 12, LV_OVF = 15, LV_N = 18 };

struct SmArgs {
    int L, V, C, n, nk, cn_lim, nw, ncw;            // ncw = words of 8 count nibbles
    uint32_t magic_v, magic_c;
    int ntrials;                                    // workgroup b decodes trials b, b + gridDim.x, …
    int kswitch;                                    // frontier width below which the waves go private
    int max_it;                                     // LEVEL: MaxNumIt, <= 0 = unlimited
    int rows_cap;                                   // TRAJ: rows kept per trial
    int32_t *rows;                                  // TRAJ: [T][rows_cap][3] = deg_1_iter, recovered, first erased position
    int off_U, off_q0, off_q1, off_pos, off_scal, off_fb, total, qcap;      // LDS offsets in 32-bit words; qcap in entries (u16)
    const uint16_t *vn_adj16;                       // [T][n][4]   CN index local to its position
    const uint16_t *cn_adj16;                       // [T][nk][8]  VNs of every CN (0xFFFF: none); SOCK: their sockets dv*t + i instead
    const uint32_t *chan;
    int32_t *counters;
    uint32_t *erased_out;
};

// Seven 4-wave workgroups per CU are 7 waves per SIMD: at most 96 SGPRs and 72 VGPRs per wave (MI355X_MICROARCH.md).
// SOCK: the CN -> VN table holds sockets (s = dv*t + i = edge i of VN t of position CNpos - i: scldpc_sample_philox_device_sock16's
// table, any chain length) instead of global VN ids (which need n < 65535).
// TRAJ (with LEVEL): the trajectory rows of the BPT build — per iteration deg_1_iter, the VNs recovered and the position of
// the first erased VN (BPT:988, 1037-1038, 1051), incl. iteration 0's count of degree-1 CNs whose only VN is known (BPF:973).
template <int BLOCK, bool LEVEL, bool PERSIST, bool SOCK, bool TRAJ = false>
__global__ __launch_bounds__(BLOCK, PERSIST ? 8 : 7) __attribute__((amdgpu_num_sgpr(96))) void full_bp_small_kernel(const SmArgs a)
{
    constexpr int kWaves = BLOCK / 64;
    extern __shared__ uint32_t lds[];
    uint32_t *cnt = lds;                                                 // nk nibbles
    uint32_t *U = lds + a.off_U;
    uint16_t *q[2] = {reinterpret_cast<uint16_t *>(lds + a.off_q0), reinterpret_cast<uint16_t *>(lds + a.off_q1)};
    int *pos_cnt = reinterpret_cast<int *>(lds + a.off_pos);
    int *pos_ss = pos_cnt + a.L;
    int *scal = reinterpret_cast<int *>(lds + a.off_scal);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto decode_trial = [&](const int trial) {
    STAMP_DECL
    const int n = a.n, nk = a.nk, cn_lim = a.cn_lim, nw = a.nw, V = a.V, C = a.C, L = a.L, qcap = a.qcap;
    const uint2 *vrow = reinterpret_cast<const uint2 *>(a.vn_adj16) + (size_t)trial * n;
    const uint4 *crow = reinterpret_cast<const uint4 *>(a.cn_adj16) + (size_t)trial * nk;
    const uint32_t *ch = a.chan + (size_t)trial * nw;

    // ---- channel bits, clear the counts -------------------------------------------------------------------------
    for (int c = tid; c < a.ncw; c += BLOCK) cnt[c] = 0;
    int ne_local = 0;
    for (int w = tid; w < nw; w += BLOCK) {
        uint32_t x = ch[w];
        if (w == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        U[w] = x;
        ne_local += __popc(x);
    }
    if (tid < LV_N) scal[tid] = 0;
    for (int i = tid; i < 2 * L; i += BLOCK) pos_cnt[i] = 0;
    __syncthreads();
    ne_local = wave_sum(ne_local);
    if (lane == 0 && ne_local) atomicAdd(&scal[SC_NE], ne_local);

    // ---- build: every erased VN counts itself into its 4 CNs; rows are loaded unconditionally (coalesced 8-B loads) --
    constexpr int NB = 8;                                                // rows in flight per thread
    for (int j0 = tid; j0 < n; j0 += NB * BLOCK) {
        uint2 r[NB];
        bool er[NB];
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int j = j0 + u * BLOCK;
            er[u] = false;
            if (j < n) { r[u] = vrow[j]; er[u] = (U[j >> 5] >> (j & 31)) & 1u; }
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int j = j0 + u * BLOCK;
            if (er[u]) {
                const int base = (int)__umulhi((uint32_t)j, a.magic_v) * C;
                const int c0 = base + (int)(r[u].x & 0xFFFFu), c1 = base + C + (int)(r[u].x >> 16);
                const int c2 = base + 2 * C + (int)(r[u].y & 0xFFFFu), c3 = base + 3 * C + (int)(r[u].y >> 16);
                atomicAdd(&cnt[c0 >> 3], 1u << ((c0 & 7) * 4));
                atomicAdd(&cnt[c1 >> 3], 1u << ((c1 & 7) * 4));
                atomicAdd(&cnt[c2 >> 3], 1u << ((c2 & 7) * 4));
                atomicAdd(&cnt[c3 >> 3], 1u << ((c3 & 7) * 4));
            }
        }
    }
    __syncthreads();
    STAMP(0);                                                            // channel + build
    const int nch = scal[SC_NE];

    // ---- one release step: CN c is believed to have exactly one erased neighbour -----------------------------------
    // out[i] = 1 + the CN on edge i of the released VN if this release left it with one erased neighbour, else 0
    int removed = 0, drops = 0;                                          // drops (LEVEL): counts taken from one to zero
    auto step = [&](int c, uint32_t (&out)[4]) {
        out[0] = out[1] = out[2] = out[3] = 0;
        const uint4 s4 = crow[c];
        uint32_t jk[8] = {s4.x & 0xFFFFu, s4.x >> 16, s4.y & 0xFFFFu, s4.y >> 16,
                          s4.z & 0xFFFFu, s4.z >> 16, s4.w & 0xFFFFu, s4.w >> 16};
        bool none[8];
#pragma unroll
        for (int k = 0; k < 8; k++) none[k] = jk[k] == 0xFFFFu;
        if constexpr (SOCK) {                                            // socket -> global VN index
            const int vbase = (int)__umulhi((uint32_t)c, a.magic_c) * V;                   // CN position * V
#pragma unroll
            for (int k = 0; k < 8; k++) jk[k] = (uint32_t)(vbase - (int)(jk[k] & 3u) * V) + (jk[k] >> 2);
        }
        uint32_t wd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) wd[k] = U[none[k] ? 0u : jk[k] >> 5];               // no VN: any word, masked below
        int j = -1;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (!none[k] && ((wd[k] >> (jk[k] & 31u)) & 1u)) j = (int)jk[k];
        if (j < 0) return;                                               // its last neighbour is being released elsewhere
        const uint2 r = vrow[j];                                         // issued before the claim: overlaps its round trip
        const uint32_t bit = 1u << (j & 31);
        if (!(atomicAnd(&U[j >> 5], ~bit) & bit)) return;
        removed++;
        const int base = (int)__umulhi((uint32_t)j, a.magic_v) * C;
        const int cc[4] = {base + (int)(r.x & 0xFFFFu), base + C + (int)(r.x >> 16),
                           base + 2 * C + (int)(r.y & 0xFFFFu), base + 3 * C + (int)(r.y >> 16)};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = atomicSub(&cnt[cc[i] >> 3], 1u << ((cc[i] & 7) * 4));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t old = (o[i] >> ((cc[i] & 7) * 4)) & 15u;
            if (old == 2u && cc[i] < cn_lim) out[i] = (uint32_t)cc[i] + 1u;
            if constexpr (LEVEL) drops += (old == 1u && cc[i] < cn_lim);
        }
    };
    // the CNs of a count word whose count is one, as a mask of the nibbles' top bits, CNs >= cn_lim dropped
    auto ones_of = [&](int w) {
        const uint32_t y = cnt[w] ^ 0x11111111u;                         // nibble == 1  <=>  zero nibble of y
        uint32_t z = ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;
        if (w * 8 + 8 > cn_lim) {                                        // the word that holds cn_lim
            const int keep = cn_lim - w * 8;
            z = keep <= 0 ? 0u : (z & ((1u << (4 * keep)) - 1u));
        }
        return z;
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

    int rounds = 0, ne = 0, status = 0;
    if constexpr (LEVEL) {
        // ---- one flooding iteration per barrier round (decodeBP's do-while, BPF:927-1065) ------------------------------
        uint8_t *fb = reinterpret_cast<uint8_t *>(lds + a.off_fb);       // snapshot of a scan round: one byte per count word
        int prec = n, iter = 0, ncur = 0, nfront = 0, first_word = 0;
        if constexpr (TRAJ) {
            // degree-1 CNs whose single VN is known count into iteration 0's deg_1_iter (BPF:969-978).  Only the first and the
            // last dv-1 CN positions of the chain hold CNs of degree below dc: their rows say how many neighbours they have.
            int extra = 0;
            const int head = 3 * C, tail0 = L * C;
            for (int i = tid; i < 2 * head; i += BLOCK) {
                const int c = i < head ? i : tail0 + (i - head);
                if (c >= cn_lim || c >= nk) continue;
                const uint4 s4 = crow[c];
                const int deg = ((s4.x & 0xFFFFu) != 0xFFFFu) + ((s4.x >> 16) != 0xFFFFu) + ((s4.y & 0xFFFFu) != 0xFFFFu) +
                                ((s4.y >> 16) != 0xFFFFu) + ((s4.z & 0xFFFFu) != 0xFFFFu) + ((s4.z >> 16) != 0xFFFFu) +
                                ((s4.w & 0xFFFFu) != 0xFFFFu) + ((s4.w >> 16) != 0xFFFFu);
                extra += deg == 1 && ((cnt[c >> 3] >> ((c & 7) * 4)) & 15u) == 0u;
            }
            extra = wave_sum(extra);
            if (lane == 0 && extra) atomicAdd(&scal[LV_EXTRA0], extra);
        }
        bool scan = true;                                                // iteration 0 has no queue yet
        ne = nch;
        for (;;) {
            const int g = iter % 3, gn = (iter + 1) % 3;
            uint16_t *qc = q[iter & 1], *qn = q[(iter + 1) & 1];
            if (tid == 0) { scal[LV_PUSH + gn] = 0; scal[LV_DROP + gn] = 0; scal[LV_REM + gn] = 0; scal[LV_OVF + gn] = 0; }
            int *push = &scal[LV_PUSH + g];                              // counts every 2 -> 1, queued or not
            bool overflow = false;
            removed = 0; drops = 0;
            auto run_queue = [&](int nq) {
                for (int k0 = wave * 64; k0 < nq; k0 += BLOCK) {
                    uint32_t out[4] = {0, 0, 0, 0};
                    if (k0 + lane < nq) step((int)qc[k0 + lane], out);
                    append(out, push, qn, overflow);
                }
            };
            if (scan) {
                // snapshot {c < cn_lim : count == 1} BEFORE any release of this round (releases must not promote CNs into it),
                // then one queue-full of it at a time
                int valid = 0;
                for (int w = tid; w < a.ncw; w += BLOCK) {
                    uint32_t y = ones_of(w) >> 3;                        // bit 4k: CN 8w+k
                    y = (y | (y >> 3)) & 0x03030303u;
                    y = (y | (y >> 6)) & 0x000F000Fu;
                    y = (y | (y >> 12)) & 0xFFu;                         // bit k: CN 8w+k
                    fb[w] = (uint8_t)y;
                    valid += __popc(y);
                }
                if (iter == 0) {
                    valid = wave_sum(valid);
                    if (lane == 0 && valid) atomicAdd(&scal[LV_VALID], valid);
                }
                if (tid == 0) scal[SC_Q] = 0;
                __syncthreads();
                if (iter == 0) nfront = scal[LV_VALID];
                for (;;) {
                    for (int w0 = wave * 64; w0 < a.ncw; w0 += BLOCK) {  // thread t owns bytes t, t + BLOCK, …
                        const int w = w0 + lane;
                        uint32_t y = w < a.ncw ? (uint32_t)fb[w] : 0u;
                        const int mine = __popc(y);
                        const int incl = (int)wave_inclusive_scan((uint32_t)mine);
                        const int tot = __builtin_amdgcn_readlane(incl, 63);
                        if (tot == 0) continue;
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&scal[SC_Q], tot);
                        int idx = __builtin_amdgcn_readfirstlane(base) + incl - mine;
                        uint32_t left = 0;
                        while (y) {
                            const int k = __ffs((int)y) - 1;
                            y &= y - 1;
                            if (idx < qcap) qc[idx] = (uint16_t)(w * 8 + k); else left |= 1u << k;
                            idx++;
                        }
                        if (mine) fb[w] = (uint8_t)left;
                    }
                    __syncthreads();
                    const int found = scal[SC_Q];
                    __syncthreads();
                    if (tid == 0) scal[SC_Q] = 0;
                    run_queue(min(found, qcap));
                    if (found <= qcap) break;
                    __syncthreads();                                     // this queue-full is done before qc is refilled
                }
            } else {
                run_queue(ncur);
            }
            STAMP(4);                                                    // (LEVEL) this wave's releases of the iteration
            {   // one reduction for both counts: a CN is queued once in its life and an iteration's entries are dealt to the four
                // waves, so a wave releases at most nk / 4 + 64 <= 16 448 VNs per iteration (15 bits) and zeroes at most four
                // times as many CNs (17 bits)
                const uint32_t both = (uint32_t)wave_sum((int)((uint32_t)removed | ((uint32_t)drops << 15)));
                removed = (int)(both & 0x7FFFu); drops = (int)(both >> 15);
            }
            if (lane == 0) {
                if (removed) atomicAdd(&scal[LV_REM + g], removed);
                if (drops) atomicAdd(&scal[LV_DROP + g], drops);
            }
            if (overflow) scal[LV_OVF + g] = 1;
            STAMP(5);                                                    // reductions
            __syncthreads();                                             // end of flooding iteration `iter`
            STAMP(6);                                                    // waiting for the other waves
            // ---- bookkeeping, identical in every thread (full_bp.hip)
            const int deg1 = nfront + ((TRAJ && iter == 0) ? scal[LV_EXTRA0] : 0);      // deg_1_iter, BPF:969-978
            ne -= scal[LV_REM + g];
            const int recovered = prec - ne;
            if constexpr (TRAJ) {
                if (wave == 0 && a.rows && rounds < a.rows_cap) {
                    // first erased VN (BPT:1037-1038): U only loses bits, so resume from the last hit
                    int fw = first_word, first = n;
                    while (fw < nw) {
                        const uint32_t w = (fw + lane < nw) ? U[fw + lane] : 0u;
                        const unsigned long long m = __ballot(w != 0u);
                        if (m) {
                            const int l0 = __ffsll((long long)m) - 1;
                            const uint32_t w0 = (uint32_t)__shfl((int)w, l0, 64);
                            fw += l0;
                            first = fw * 32 + (__ffs((int)w0) - 1);
                            break;
                        }
                        fw += 64;
                    }
                    first_word = fw;
                    if (lane == 0) {
                        int32_t *r = a.rows + ((size_t)trial * a.rows_cap + rounds) * 3;
                        r[0] = deg1; r[1] = recovered; r[2] = (int)__umulhi((uint32_t)first, a.magic_v);
                    }
                }
                __syncthreads();                                         // wave 0 read U above: the next round's releases stay behind it
            }
            rounds++;
            if (deg1 < recovered && iter > 0) { status = -1; break; }    // BPF:1035-1039
            if (ne == 0 || ne == prec) break;                            // BPF:1044-1045
            prec = ne;
            // next frontier: queued 2 -> 1 CNs minus those that went on to 0 within this round
            nfront = scal[LV_PUSH + g] - (scal[LV_DROP + g] - nfront);
            scan = scal[LV_OVF + g] != 0;
            ncur = scan ? 0 : scal[LV_PUSH + g];
            iter++;
            if (a.max_it > 0 && iter >= a.max_it) break;                 // BPF:1065
            STAMP(7);                                                    // bookkeeping
        }
        __syncthreads();
        STAMP(1);
    } else {
        // ---- peel: barrier rounds over a shared queue while the frontier is wide (a scan opens the run and repairs an
        //      overflow), then every wave runs the CNs its own releases create from a private queue, level after level ----
        const int kSwitch = a.kswitch;
        const int wcap = (qcap / kWaves) & ~1, half_cap = wcap / 2;
        int ncur = 0;
        bool scan = true;
        for (;;) {
            uint16_t *qc = q[rounds & 1], *qn = q[(rounds + 1) & 1];
            if (scan) {
                // every CN < cn_lim whose count is one right now, compacted into qc
                for (int w0 = wave * 64; w0 < a.ncw; w0 += BLOCK) {
                    const int w = w0 + lane;
                    uint32_t z = w < a.ncw ? ones_of(w) : 0u;
                    const int mine = __popc(z);
                    const int incl = (int)wave_inclusive_scan((uint32_t)mine);
                    const int tot = __builtin_amdgcn_readlane(incl, 63);
                    if (tot == 0) continue;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&scal[SC_Q], tot);
                    base = __builtin_amdgcn_readfirstlane(base);
                    int idx = base + incl - mine;
                    while (z) {
                        const int k = (__ffs((int)z) - 1) >> 2;
                        z &= z - 1;
                        if (idx < qcap) qc[idx] = (uint16_t)(w * 8 + k);
                        idx++;
                    }
                }
                __syncthreads();
                ncur = scal[SC_Q];
                if (ncur > qcap) { ncur = qcap; if (tid == 0) scal[SC_OVF] = 1; }      // the rest: next scan
                __syncthreads();
            }
            if (tid == 0) { scal[SC_Q] = 0; scal[SC_N0 + ((rounds + 1) & 1)] = 0; }
            int *push = &scal[SC_N0 + (rounds & 1)];
            bool overflow = false;
            if (ncur > kSwitch || half_cap < 64) {
                for (int k0 = wave * 64; k0 < ncur; k0 += BLOCK) {
                    uint32_t out[4] = {0, 0, 0, 0};
                    if (k0 + lane < ncur) step((int)qc[k0 + lane], out);
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
                        if (b0 + lane < cntw) step((int)src[b0 + lane], out);
                        {   // append: wave-synchronous, no atomics
                            const int mine_n = (out[0] != 0u) + (out[1] != 0u) + (out[2] != 0u) + (out[3] != 0u);
                            const int incl = (int)wave_inclusive_scan((uint32_t)mine_n);
                            int idx = ncnt + incl - mine_n;
    #pragma unroll
                            for (int i = 0; i < 4; i++)
                                if (out[i]) { if (idx < half_cap) dst[idx] = (uint16_t)(out[i] - 1u); else overflow = true; idx++; }
                            ncnt += __builtin_amdgcn_readlane(incl, 63);
                        }
                    }
                    cntw = min(ncnt, half_cap);
                    cur ^= 1;
                }
            }
            if (overflow) scal[SC_OVF] = 1;
            __syncthreads();
            rounds++;
            const int pushed = *push;
            scan = scal[SC_OVF] != 0;            // a full queue dropped CNs: find them by a scan
            __syncthreads();
            if (tid == 0) scal[SC_OVF] = 0;
            ncur = scan ? 0 : min(pushed, qcap);
            if (!scan && ncur == 0) break;
        }
        STAMP(1);                                                            // peeling
        removed = wave_sum(removed);
        if (lane == 0 && removed) atomicAdd(&scal[SC_REM], removed);
        __syncthreads();
        ne = nch - scal[SC_REM];
    }

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
                const uint2 r = vrow[j];
                const int base = q0 * C;
                const int cc[4] = {base + (int)(r.x & 0xFFFFu), base + C + (int)(r.x >> 16),
                                   base + 2 * C + (int)(r.y & 0xFFFFu), base + 3 * C + (int)(r.y >> 16)};
                bool pair = true;
#pragma unroll
                for (int i = 0; i < 4; i++) pair = pair && ((cnt[cc[i] >> 3] >> ((cc[i] & 7) * 4)) & 15u) == 2u;
                if (!pair) continue;
                int partner = -1;
                for (int i = 0; i < 4 && pair; i++) {                    // the other erased neighbour of each CN
                    const uint4 s4 = crow[cc[i]];
                    const uint32_t jk[8] = {s4.x & 0xFFFFu, s4.x >> 16, s4.y & 0xFFFFu, s4.y >> 16,
                                            s4.z & 0xFFFFu, s4.z >> 16, s4.w & 0xFFFFu, s4.w >> 16};
                    int other = -1;
                    for (int k = 0; k < 8; k++) {
                        if (jk[k] == 0xFFFFu) continue;
                        // (cc[i] lies in CN position q0 + i)
                        const int j2 = SOCK ? (q0 + i - (int)(jk[k] & 3u)) * V + (int)(jk[k] >> 2) : (int)jk[k];
                        if (j2 != j && ((U[j2 >> 5] >> (j2 & 31)) & 1u)) other = j2;
                    }
                    if (other < 0 || (i > 0 && other != partner)) pair = false;
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
        o[SCLDPC_C_ITERATIONS] = rounds;                                 // LEVEL: flooding iterations; else barrier rounds
        o[SCLDPC_C_STATUS] = status;
        o[SCLDPC_C_CHANNEL_ERASURES] = nch;
    }
    };
    // PERSIST: workgroup b decodes trials b, b + gridDim.x, …; otherwise exactly one
    if constexpr (PERSIST) {
        for (int trial = blockIdx.x; trial < a.ntrials; trial += gridDim.x) {
            decode_trial(trial);
            __syncthreads();                                             // LDS is re-used by the workgroup's next trial
        }
    } else {
        decode_trial((int)blockIdx.x);
    }
}

int make_args(const scldpc_code_params *p, int32_t is_term, SmArgs *a, int per_cu, bool level = false)
{
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    a->L = p->L; a->V = p->vns_pos; a->C = p->cns_pos; a->n = n; a->nk = nk;
    a->cn_lim = is_term ? nk : p->L * p->cns_pos;                        // BPT:944-948
    a->nw = (n + 31) / 32; a->ncw = (nk + 7) / 8;
    int off = 0;
    auto take = [&](int words) { int o = off; off += (words + 3) & ~3; return o; };
    take(a->ncw);
    a->off_U = take(a->nw);
    a->off_pos = take(2 * p->L);
    a->off_scal = take(LV_N);
    a->off_fb = level ? take((a->ncw + 3) / 4) : 0;                      // snapshot bytes of the scan rounds
    const int budget = scldpc::kMaxLdsBytes / per_cu / 4 - 128;           // words per workgroup
    int qwords = ((budget - off) / 2) & ~3;                              // per queue; two uint16 entries per word
    if (qwords > 2048) qwords = 2048;
    if (qwords < 128) return -1;
    a->qcap = 2 * qwords;
    a->off_q0 = take(qwords);
    a->off_q1 = take(qwords);
    a->total = off;
    return 0;
}

constexpr int kBlockSmall = 256;        // threads per trial
constexpr int kPerCu = 7;               // workgroups per CU the LDS carve aims at (SGPRs <= 96, VGPRs <= 72)
constexpr int kSwitchWidth = 128;       // frontier entries below which the waves go private
static_assert(kSwitchWidth <= 64 * (kBlockSmall / 64), "a wave takes at most one frontier entry per lane into its private queue");

}  // namespace

namespace {
bool small_shape(const scldpc_code_params *p)
{
    if (scldpc::check_params(p)) return false;
    SmArgs a{};
    uint32_t m;
    // queue entries are 16-bit CN ids; the per-trial state (4 bits per CN, one bit per VN) must fit the LDS
    return p->dv == 4 && p->dc == 8 && p->cns_pos <= 65536 && scldpc::nk_of(p) <= 65536 &&
           make_args(p, 1, &a, 1) == 0 && scldpc::magic_of(p->vns_pos, scldpc::n_of(p) + 32, &m) &&
           scldpc::magic_of(p->cns_pos, scldpc::nk_of(p), &m);
}
}  // namespace

// 1 when scldpc_full_bp_fixpoint_device_cn16 takes this ensemble (global VN ids in the CN -> VN table: n < 65535)
extern "C" int scldpc_full_bp_cn16_supported(const scldpc_code_params *p)
{
    return small_shape(p) && scldpc::n_of(p) < 65535;
}

// 1 when the _sock16 forms take this ensemble (sockets in the CN -> VN table: any n whose state fits the LDS)
extern "C" int scldpc_full_bp_sock16_supported(const scldpc_code_params *p)
{
    return small_shape(p) && (int64_t)p->vns_pos * p->dv <= 65535;
}

namespace {

int launch_small(const char *who, bool level, bool sock, const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                 const uint16_t *d_cn_adj16, const uint32_t *d_chan_bits, int32_t max_it, int32_t is_term,
                 int32_t *d_counters, uint32_t *d_erased_bits, void *stream, int32_t *d_rows = nullptr, int32_t rows_cap = 0)
{
    if (d_rows && (rows_cap <= 0 || !level))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: d_rows given but rows_cap <= 0", who);
    if (int rc = scldpc::check_params(p)) return rc;
    if (!(sock ? scldpc_full_bp_sock16_supported(p) : scldpc_full_bp_cn16_supported(p)))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: takes dv = 4, dc = 8, at most 65536 CNs per trial%s", who,
                                 sock ? " and 16-bit sockets" : " and fewer than 65535 VNs (use the _sock16 form beyond)");
    if (ntrials < 0 || (ntrials > 0 && (!d_counters || !d_vn_adj16 || !d_cn_adj16 || !d_chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: null buffer or negative ntrials", who);
    if (ntrials == 0) return SCLDPC_OK;
    SmArgs a{};
    int per_cu = kPerCu;                                               // workgroups per CU the LDS carve aims at
    while (per_cu > 1 && make_args(p, is_term, &a, per_cu, level) != 0) per_cu--;
    if (make_args(p, is_term, &a, per_cu, level) != 0)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: the CN counts and VN bits do not fit the LDS", who);
    scldpc::magic_of(p->vns_pos, a.n + 32, &a.magic_v);
    scldpc::magic_of(p->cns_pos, a.nk, &a.magic_c);
    a.vn_adj16 = d_vn_adj16; a.cn_adj16 = d_cn_adj16; a.chan = d_chan_bits;
    a.counters = d_counters; a.erased_out = d_erased_bits;
    a.kswitch = kSwitchWidth;
    // A/B only; a wave takes at most one entry per lane into its private queue, so the width is capped at 64 entries per wave
    if (const char *v = getenv("SCLDPC_DEBUG_DECODER_KSWITCH")) a.kswitch = std::min(atoi(v), 64 * (kBlockSmall / 64));
    a.ntrials = ntrials;
    a.rows = d_rows; a.rows_cap = d_rows ? rows_cap : 0;
    a.max_it = max_it;
    const int grid = scldpc::debug_grid("DECODER", ntrials);
    void (*kern)(const SmArgs) = sock ? (level ? full_bp_small_kernel<kBlockSmall, true, false, true> : full_bp_small_kernel<kBlockSmall, false, false, true>)
        : grid < ntrials ? (level ? full_bp_small_kernel<kBlockSmall, true, true, false> : full_bp_small_kernel<kBlockSmall, false, true, false>)
                         : (level ? full_bp_small_kernel<kBlockSmall, true, false, false> : full_bp_small_kernel<kBlockSmall, false, false, false>);
    if (sock && grid < ntrials) return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: no persistent form with the socket table", who);
    if (d_rows) kern = sock ? full_bp_small_kernel<kBlockSmall, true, false, true, true> : full_bp_small_kernel<kBlockSmall, true, false, false, true>;
    size_t lds_bytes = 4u * (size_t)a.total;
    lds_bytes = std::min(lds_bytes + scldpc::debug_lds_pad("DECODER"), (size_t)scldpc::kMaxLdsBytes);
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlockSmall), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

}  // namespace

extern "C" int scldpc_full_bp_fixpoint_device_cn16(const scldpc_code_params *p, int32_t ntrials,
                                                   const uint16_t *d_vn_adj16, const uint16_t *d_cn_adj16,
                                                   const uint32_t *d_chan_bits, int32_t is_term, int32_t *d_counters,
                                                   uint32_t *d_erased_bits, void *stream)
{
    return launch_small("scldpc_full_bp_fixpoint_device_cn16", false, false, p, ntrials, d_vn_adj16, d_cn_adj16, d_chan_bits, 0,
                        is_term, d_counters, d_erased_bits, stream);
}

// decodeBP with its iterations (count, cap MaxNumIt, stop tests): every counter of scldpc_full_bp_device, from both tables
extern "C" int scldpc_full_bp_device_cn16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                          const uint16_t *d_cn_adj16, const uint32_t *d_chan_bits, int32_t max_it,
                                          int32_t is_term, int32_t *d_counters, uint32_t *d_erased_bits, void *stream)
{
    return launch_small("scldpc_full_bp_device_cn16", true, false, p, ntrials, d_vn_adj16, d_cn_adj16, d_chan_bits, max_it,
                        is_term, d_counters, d_erased_bits, stream);
}

// The same two decoders reading the CN -> SOCKET table of scldpc_sample_philox_device_sock16 / scldpc_cn_sockets_device
// (position-local 16-bit sockets): no limit on the number of VNs per trial — e.g. the published L = 100, N = 1000 runs.
extern "C" int scldpc_full_bp_fixpoint_device_sock16(const scldpc_code_params *p, int32_t ntrials,
                                                     const uint16_t *d_vn_adj16, const uint16_t *d_cn_sock16,
                                                     const uint32_t *d_chan_bits, int32_t is_term, int32_t *d_counters,
                                                     uint32_t *d_erased_bits, void *stream)
{
    return launch_small("scldpc_full_bp_fixpoint_device_sock16", false, true, p, ntrials, d_vn_adj16, d_cn_sock16, d_chan_bits,
                        0, is_term, d_counters, d_erased_bits, stream);
}

extern "C" int scldpc_full_bp_device_sock16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                            const uint16_t *d_cn_sock16, const uint32_t *d_chan_bits, int32_t max_it,
                                            int32_t is_term, int32_t *d_counters, uint32_t *d_erased_bits, void *stream)
{
    return launch_small("scldpc_full_bp_device_sock16", true, true, p, ntrials, d_vn_adj16, d_cn_sock16, d_chan_bits, max_it,
                        is_term, d_counters, d_erased_bits, stream);
}

// decodeBP of the trajectory build (BPT:900-1140): the iterations with their rows — deg_1_iter, VNs recovered, position of the
// first erased VN (BPT:988, 1037-1038, 1051) — d_rows int32 [ntrials][rows_cap][3], d_counters[ITERATIONS] rows per trial.
extern "C" int scldpc_full_bp_traj_device_cn16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                               const uint16_t *d_cn_adj16, const uint32_t *d_chan_bits, int32_t max_it,
                                               int32_t is_term, int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                                               uint32_t *d_erased_bits, void *stream)
{
    if (!d_rows) return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_full_bp_traj_device_cn16: null d_rows");
    return launch_small("scldpc_full_bp_traj_device_cn16", true, false, p, ntrials, d_vn_adj16, d_cn_adj16, d_chan_bits, max_it,
                        is_term, d_counters, d_erased_bits, stream, d_rows, rows_cap);
}

extern "C" int scldpc_full_bp_traj_device_sock16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                                 const uint16_t *d_cn_sock16, const uint32_t *d_chan_bits, int32_t max_it,
                                                 int32_t is_term, int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                                                 uint32_t *d_erased_bits, void *stream)
{
    if (!d_rows) return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_full_bp_traj_device_sock16: null d_rows");
    return launch_small("scldpc_full_bp_traj_device_sock16", true, true, p, ntrials, d_vn_adj16, d_cn_sock16, d_chan_bits, max_it,
                        is_term, d_counters, d_erased_bits, stream, d_rows, rows_cap);
}
