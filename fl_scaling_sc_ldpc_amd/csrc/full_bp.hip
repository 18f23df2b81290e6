// Full flooding BP over the BEC for a batch of sampled SC-LDPC codes — gfx950 (MI355X) kernel.
//
// Replaces decodeBP of the reference (BPF:900-1140; BPT adds per-iteration rows and truncation,
// BPT:912-1051) bit for bit, but does O(E) work per trial instead of O(E·iterations):
// on the BEC, the a-posteriori erasure set after flooding iteration t equals the residual of
// level-synchronous parallel peeling after t rounds, so the kernel keeps
//   * one bit per VN  (U = currently erased VNs), and
//   * one small word per CN holding cnt = #erased neighbours and a fold (sum or xor) of their ids —
//     when cnt == 1 the fold IS the one erased neighbour, so only the VN→CN table is ever read
//     from HBM (no CN→VN lists, no per-edge messages),
// all of it in LDS (one workgroup = one trial), and walks a frontier of CNs with cnt == 1.
//
// Two CN-word layouts:
//   Wide   32 bit  [cnt:4 | deg:4 | Σ global VN id:24]   any ensemble that fits; needed for trajectories (deg)
//   Packed 16 bit  [cnt:4 | ⊕ local VN id:12]            when dv·vns_pos <= 4096 (local id = edge·vns_pos + t):
//                  half the LDS ⇒ two 1024-thread workgroups per CU hide each other's latencies.
//
// Iteration t of the reference == one frontier round (one barrier):
//   every CN with cnt == 1 at the start of the round releases its VN; the releasing thread fetches the
//   VN's dv CN ids (one 16-B load), decrements those CNs and queues the ones that drop 2 → 1.
//   deg_1_iter (BPF:969-978) needs no extra pass:  |F(t+1)| = pushes(t) − (drops_1→0(t) − |F(t)|).
// Afterwards the size-2 stopping-set expurgation (BPF:1067-1133) from the same CN words: an erased VN a
// belongs to such a pair iff every one of its CNs has cnt == 2 and the same partner, in a's position.
#include "common.h"
#include "kernel_util.h"
#include "cn_words.h"
#include "peel_fixpoint.h"

namespace {

using namespace scldpc_dev;

// per-iteration counters, rotated three ways so that one barrier per iteration is enough
enum { S_PUSH = 0, S_DROP = 3, S_REM = 6, S_OVF = 9, S_NE = 12, S_EXTRA0 = 13, S_VALID = 14, S_NSCAL = 16 };

struct Layout {             // offsets in 32-bit words into dynamic LDS
    int cn_state, U, fbits, q0, q1, pos_cnt, pos_ss, scal, total;
    int qcap, nw;
};

struct Args : Geo {             // Geo: vns_pos, magic_v, magic_c
    int dv, L, cns_pos, n, nk, cn_lim, max_it, rows_cap;
    int prebuilt;                   // WideG: the CN words were built by cn_build.hip (through LDS, not by global atomics)
    Layout lay;
    const void *vn_adj;             // int32 [T][n][dv] or uint16 [T][n][dv] (position-local ids)
    uint32_t *ws;                   // [T][nk] CN words in global memory (WideG only)
    const uint32_t *chan;
    int32_t *counters;
    int32_t *rows;
    uint32_t *erased_out;
};

template <class ST, bool TRAJ, int DV, bool A16, int BLOCK>
// Two 1024-thread workgroups per CU = 8 waves per SIMD, which the scalar file admits only up to 80 SGPRs per wave
// (⌊800 / (⌈sgpr/16⌉·16 + 16)⌋ waves): capped here, the overflow lives in VGPR lanes.
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(72))) void full_bp_kernel(const Args a)
{
    extern __shared__ uint32_t lds[];
    const int trial = blockIdx.x;
    uint32_t *cn_state;
    if constexpr (ST::kGlobal) cn_state = a.ws + (size_t)trial * a.nk;
    else                       cn_state = lds + a.lay.cn_state;
    uint32_t *U = lds + a.lay.U;
    uint32_t *fbits = lds + a.lay.fbits;
    uint32_t *q[2] = {lds + a.lay.q0, lds + a.lay.q1};
    int *pos_cnt = reinterpret_cast<int *>(lds + a.lay.pos_cnt);
    int *pos_ss = reinterpret_cast<int *>(lds + a.lay.pos_ss);
    int *scal = reinterpret_cast<int *>(lds + a.lay.scal);

    const int tid = threadIdx.x, lane = tid & 63;
    const int n = a.n, nk = a.nk, dv = (DV ? DV : a.dv), cn_lim = a.cn_lim, nw = a.lay.nw, qcap = a.lay.qcap;
    const int V = a.vns_pos;
    const char *adj = static_cast<const char *>(a.vn_adj) + (size_t)trial * n * dv * (A16 ? 2 : 4);
    const uint32_t *ch = a.chan + (size_t)trial * nw;
    auto make_vn = [&](int j) { Vn v; v.j = j; v.pos = (int)__umulhi((uint32_t)j, a.magic_v); v.t = j - v.pos * V; return v; };

    // ---- load channel bits, clear CN words -------------------------------------------------
    const bool prebuilt = ST::kGlobal && a.prebuilt;
    if (!prebuilt) for (int c = tid; c < ST::words(nk); c += BLOCK) cn_state[c] = 0;
    int ne_local = 0;
    for (int w = tid; w < nw; w += BLOCK) {
        uint32_t x = ch[w];
        if (w == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        U[w] = x;
        ne_local += __popc(x);
    }
    if (tid < S_NSCAL) scal[tid] = 0;
    for (int i = tid; i < a.L; i += BLOCK) { pos_cnt[i] = 0; pos_ss[i] = 0; }
    __syncthreads();
    STAMP_DECL
    STAMP(0);                                                       // clear + channel load
    ne_local = wave_sum(ne_local);
    if (lane == 0 && ne_local) atomicAdd(&scal[S_NE], ne_local);

    // Trajectory rows with packed words: deg_1_iter's iteration-0 quirk (BPF:969-978) asks for CNs of degree 1 whose only VN
    // is known.  Only the first and last dv-1 CN positions of the chain can have a degree below dc, so their degrees are
    // counted into a byte array that borrows queue 1 (first written by iteration 0's releases, after the scan has used it).
    constexpr bool kSideDeg = TRAJ && !ST::kHasDeg;
    uint32_t *degb = q[1];
    const int ms = dv - 1, tail0 = a.L * a.cns_pos;
    auto boundary = [&](int c) {                    // index into degb, or -1 for a CN of the chain's interior
        return c < ms * a.cns_pos ? c : c >= tail0 ? ms * a.cns_pos + (c - tail0) : -1;
    };
    if constexpr (kSideDeg) {
        for (int w = tid; w < (2 * ms * a.cns_pos + 3) / 4; w += BLOCK) degb[w] = 0;
        __syncthreads();
    }

    // ---- build: every erased VN adds itself to its dv CNs (TRAJ: every VN also adds to deg) ----
    // Loads are unconditional so that each wave instruction reads 1 KiB contiguous (dv = 4).
    for (int j0 = prebuilt ? n : tid; j0 < n; j0 += 4 * BLOCK) {
        int32_t c[4][8];
        bool er[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * BLOCK;
            er[u] = false;
            if (j < n) {
                load_adj<DV, A16>(adj, dv, j, (int)__umulhi((uint32_t)j, a.magic_v), a.cns_pos, c[u]);
                er[u] = (U[j >> 5] >> (j & 31)) & 1u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * BLOCK;
            if (j < n && (TRAJ || er[u])) {
                const Vn v = make_vn(j);
                for (int i = 0; i < dv; i++) {
                    ST::add(cn_state, c[u][i], v, i, V, er[u], TRAJ);
                    if constexpr (kSideDeg) {
                        const int b = boundary(c[u][i]);
                        if (b >= 0) atomicAdd(&degb[b >> 2], 1u << ((b & 3) * 8));
                    }
                }
            }
        }
    }
    __syncthreads();
    STAMP(1);                                                       // build

    int ne = scal[S_NE];
    const int nch = ne;
    int prec = n, iter = 0, ncur = 0, status = 0, rows_done = 0, nfront = 0;
    bool scan = true;                   // iteration 0 has no queue yet: frontier = all CNs with cnt == 1
    int first_word = 0;

    for (;;) {
        const int g = iter % 3, gn = (iter + 1) % 3;
        uint32_t *qc = q[iter & 1], *qn = q[(iter + 1) & 1];
        if (tid == 0) { scal[S_PUSH + gn] = 0; scal[S_DROP + gn] = 0; scal[S_REM + gn] = 0; scal[S_OVF + gn] = 0; }

        int removed = 0, drops = 0;
        auto release = [&](int c) {
            const int j = ST::lone_vn(cn_state, c, a);
            if (j < 0) return;                                      // its VN was just released via another CN
            const Vn v = make_vn(j);
            int32_t cc[8];
            load_adj<DV, A16>(adj, dv, j, v.pos, a.cns_pos, cc);    // issued before the ownership test: its latency
            const uint32_t bit = 1u << (j & 31);                    // overlaps the LDS round trip below
            const uint32_t old = atomicAnd(&U[j >> 5], ~bit);
            if (!(old & bit)) return;                               // lost the race for VN j
            removed++;
            uint32_t o[8];
#pragma unroll
            for (int i = 0; i < (DV ? DV : 8); i++)                 // the dv returning atomics go out back to back
                if (i < dv) o[i] = ST::remove_cnt(cn_state, cc[i], v, i, V);
#pragma unroll
            for (int i = 0; i < (DV ? DV : 8); i++)
                if (i < dv) ST::remove_fold(cn_state, cc[i], v, i, V);
#pragma unroll
            for (int i = 0; i < (DV ? DV : 8); i++) {
                if (i < dv && cc[i] < cn_lim) {
                    drops += (o[i] == 1u);
                    if (o[i] == 2u) {                               // 2 → 1: candidate for the next round
                        const int idx = atomicAdd(&scal[S_PUSH + g], 1);
                        if (idx < qcap) qn[idx] = (uint32_t)cc[i]; else scal[S_OVF + g] = 1;
                    }
                }
            }
        };
        if (scan) {
            // snapshot {c < cn_lim : cnt == 1} first: this round's releases must not promote CNs into it
            int valid = 0, extra = 0;
            for (int base = 0; base < cn_lim; base += BLOCK) {
                const int c = base + tid;
                bool v = false;
                if (c < cn_lim) {
                    const uint32_t k = ST::cnt(cn_state, c);
                    v = k == 1u;
                    if (TRAJ && iter == 0) {                        // degree-1 CN whose only VN is known (BPF:973)
                        uint32_t d;
                        if constexpr (kSideDeg) { const int b = boundary(c); d = b < 0 ? 0xFFu : (degb[b >> 2] >> ((b & 3) * 8)) & 0xFFu; }
                        else                    d = ST::deg(cn_state, c);
                        extra += (k == 0u && d == 1u);
                    }
                }
                const unsigned long long m = __ballot(v);
                if (c - lane < cn_lim) {                            // this wave's 64-CN slice starts inside the range
                    if (lane == 0) fbits[c >> 5] = (uint32_t)m;
                    if (lane == 32) fbits[c >> 5] = (uint32_t)(m >> 32);
                }
                valid += v;
            }
            if (iter == 0) {
                valid = wave_sum(valid);
                if (lane == 0 && valid) atomicAdd(&scal[S_VALID], valid);
                if (TRAJ) {
                    extra = wave_sum(extra);
                    if (lane == 0 && extra) atomicAdd(&scal[S_EXTRA0], extra);
                }
            }
            __syncthreads();
            if (iter == 0) nfront = scal[S_VALID];
            for (int base = 0; base < cn_lim; base += BLOCK) {
                const int c = base + tid;
                if (c < cn_lim && ((fbits[c >> 5] >> (c & 31)) & 1u)) release(c);
            }
        } else {
            for (int k = tid; k < ncur; k += BLOCK) release((int)qc[k]);
        }
        STAMP(2);                                                   // release
        {   // one DPP reduction for both counts (each < 2^16 per wave)
            const uint32_t both = wave_inclusive_scan(((uint32_t)removed << 16) | (uint32_t)drops);
            if (lane == 63 && both) {
                if (both >> 16) atomicAdd(&scal[S_REM + g], (int)(both >> 16));
                if (both & 0xFFFFu) atomicAdd(&scal[S_DROP + g], (int)(both & 0xFFFFu));
            }
        }
        STAMP(3);                                                   // wave reductions
        __syncthreads();                                            // end of flooding iteration `iter`
        STAMP(4);                                                   // barrier wait

        // ---- bookkeeping, identical in every thread ----------------------------------------
        const int deg1 = nfront + ((TRAJ && iter == 0) ? scal[S_EXTRA0] : 0);
        ne -= scal[S_REM + g];
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
                r[0] = deg1; r[1] = recovered; r[2] = first / V;
            }
        }
        if (TRAJ) __syncthreads();      // wave 0 read U above: keep the next round's releases behind it
        rows_done++;
        if (deg1 < recovered && iter > 0) { status = -1; break; }   // BPF:1035-1039
        if (ne == 0 || ne == prec) break;                           // BPF:1044-1045
        prec = ne;
        // frontier of the next round: queued 2→1 CNs minus those that went on to 0 within this round
        nfront = scal[S_PUSH + g] - (scal[S_DROP + g] - nfront);
        scan = scal[S_OVF + g] != 0;
        ncur = scan ? 0 : scal[S_PUSH + g];
        iter++;
        if (a.max_it > 0 && iter >= a.max_it) break;                // BPF:1065
        STAMP(5);                                                   // bookkeeping
    }
    __syncthreads();
    STAMP(5);

    // ---- per-position erasure counts + size-2 stopping sets (BPF:1067-1133) -----------------
    if (ne > 0) {
        for (int w = tid; w < nw; w += BLOCK) {
            uint32_t x = U[w];
            while (x) {
                const int b = __ffs((int)x) - 1;
                x &= x - 1;
                const Vn v = make_vn(w * 32 + b);
                atomicAdd(&pos_cnt[v.pos], 1);
                int32_t cc[8];
                load_adj<DV, A16>(adj, dv, v.j, v.pos, a.cns_pos, cc);
                bool pair = true;
                int partner = -1;
                for (int i = 0; i < dv; i++) {
                    const int b2 = ST::partner(cn_state, cc[i], v, i, a);
                    if (b2 < 0 || (i > 0 && b2 != partner)) { pair = false; break; }
                    partner = b2;
                }
                if (pair && (int)__umulhi((uint32_t)partner, a.magic_v) == v.pos) atomicAdd(&pos_ss[v.pos], 1);
            }
        }
    }
    __syncthreads();
    STAMP(6);                                                       // final counts + expurgation
    STAMP_FLUSH();
    if (a.erased_out)
        for (int w = tid; w < nw; w += BLOCK) a.erased_out[(size_t)trial * nw + w] = U[w];
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

// =================================================================================================
// Fixpoint variant: the residual of UNLIMITED flooding without walking its iterations one barrier at a time.
//
// On the BEC the set flooding BP converges to is the closure of "a CN with exactly one erased neighbour resolves it" and
// does not depend on the order in which CNs fire — only the iteration COUNT does.  When nobody asks
// for the count (no iteration cap, no trajectory rows) a lane that has released a VN may therefore go straight on with a
// CN that this very release left with one erased neighbour, instead of queueing it for the next barrier round: the
// ~230 dependent levels of a trial at eps = 0.48 become ~230 steps of (one row gather + two rounds of LDS atomics) in a
// row, with a barrier only when chains end and work is redistributed.  Everything reported except
// SCLDPC_C_ITERATIONS (here: the number of barrier rounds) is identical to full_bp_kernel's — tests compare them.
//
// The barrier-free loop and the update protocol that makes it safe live in peel_fixpoint.h.
// =================================================================================================
template <bool A16, int BLOCK>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(72))) void full_bp_fixpoint_kernel(const Args a)
{
    using ST = Packed;
    constexpr int DV = 4;
    extern __shared__ uint32_t lds[];
    const int trial = blockIdx.x;
    uint32_t *cn_state = lds + a.lay.cn_state;
    uint32_t *U = lds + a.lay.U;
    uint32_t *fbits = lds + a.lay.fbits;
    uint32_t *q[2] = {lds + a.lay.q0, lds + a.lay.q1};
    int *pos_cnt = reinterpret_cast<int *>(lds + a.lay.pos_cnt);
    int *pos_ss = reinterpret_cast<int *>(lds + a.lay.pos_ss);
    int *scal = reinterpret_cast<int *>(lds + a.lay.scal);

    const int tid = threadIdx.x, lane = tid & 63;
    const int n = a.n, nk = a.nk, cn_lim = a.cn_lim, nw = a.lay.nw, qcap = a.lay.qcap;
    const int V = a.vns_pos, C = a.cns_pos, L = a.L;
    const char *adj = static_cast<const char *>(a.vn_adj) + (size_t)trial * n * DV * (A16 ? 2 : 4);
    const uint32_t *ch = a.chan + (size_t)trial * nw;
    auto make_vn = [&](int j) { Vn v; v.j = j; v.pos = (int)__umulhi((uint32_t)j, a.magic_v); v.t = j - v.pos * V; return v; };

    const bool prebuilt = ST::kGlobal && a.prebuilt;
    if (!prebuilt) for (int c = tid; c < ST::words(nk); c += BLOCK) cn_state[c] = 0;
    int ne_local = 0;
    for (int w = tid; w < nw; w += BLOCK) {
        uint32_t x = ch[w];
        if (w == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        U[w] = x;
        ne_local += __popc(x);
    }
    if (tid < S_NSCAL) scal[tid] = 0;
    for (int i = tid; i < L; i += BLOCK) { pos_cnt[i] = 0; pos_ss[i] = 0; }
    __syncthreads();
    ne_local = wave_sum(ne_local);
    if (lane == 0 && ne_local) atomicAdd(&scal[S_NE], ne_local);
    for (int j0 = prebuilt ? n : tid; j0 < n; j0 += 4 * BLOCK) {   // build, as in full_bp_kernel
        int32_t c[4][8];
        bool er[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * BLOCK;
            er[u] = false;
            if (j < n) {
                load_adj<DV, A16>(adj, DV, j, (int)__umulhi((uint32_t)j, a.magic_v), C, c[u]);
                er[u] = (U[j >> 5] >> (j & 31)) & 1u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * BLOCK;
            if (j < n && er[u]) {
                const Vn v = make_vn(j);
#pragma unroll
                for (int i = 0; i < DV; i++) ST::add(cn_state, c[u][i], v, i, V, true, false);
            }
        }
    }
    __syncthreads();
    STAMP_DECL
    STAMP(0);                                                       // channel + build
    const int nch = scal[S_NE];

    int removed = 0;
    PeelCtx x{L, V, C, n, cn_lim, qcap, adj, cn_state, U, fbits, q[0], q[1], &scal[S_PUSH], &scal[S_OVF]};
    const int rounds = peel_to_fixpoint<A16, BLOCK>(a, x, removed, [](int, int, bool one) { return one; });
    STAMP(1);                                                       // peeling (barrier rounds + barrier-free phase)
    removed = wave_sum(removed);
    if (lane == 0 && removed) atomicAdd(&scal[S_REM], removed);
    __syncthreads();
    const int ne = nch - scal[S_REM];

    // ---- per-position erasure counts + size-2 stopping sets (BPF:1067-1133), as in full_bp_kernel ----
    if (ne > 0) {
        for (int w = tid; w < nw; w += BLOCK) {
            uint32_t x = U[w];
            while (x) {
                const int b = __ffs((int)x) - 1;
                x &= x - 1;
                const Vn v = make_vn(w * 32 + b);
                atomicAdd(&pos_cnt[v.pos], 1);
                int32_t cc[8];
                load_adj<DV, A16>(adj, DV, v.j, v.pos, C, cc);
                bool pair = true;
                int partner = -1;
                for (int i = 0; i < DV; i++) {
                    const int b2 = ST::partner(cn_state, cc[i], v, i, a);
                    if (b2 < 0 || (i > 0 && b2 != partner)) { pair = false; break; }
                    partner = b2;
                }
                if (pair && (int)__umulhi((uint32_t)partner, a.magic_v) == v.pos) atomicAdd(&pos_ss[v.pos], 1);
            }
        }
    }
    __syncthreads();
    STAMP(2);                                                       // final counts + expurgation
    STAMP_FLUSH();
    if (a.erased_out)
        for (int w = tid; w < nw; w += BLOCK) a.erased_out[(size_t)trial * nw + w] = U[w];
    if (tid == 0) {
        int be = 0, ee = 0, bee = 0;
        for (int pos = 0; pos < L; pos++) {
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
        o[SCLDPC_C_ITERATIONS] = rounds;                            // barrier rounds, NOT flooding iterations
        o[SCLDPC_C_STATUS] = 0;
        o[SCLDPC_C_CHANNEL_ERASURES] = nch;
    }
}

// LDS carve.  `budget` = bytes this workgroup may take (160 KiB alone on the CU, 80 KiB when two share it).
template <class ST>
int make_layout(const scldpc_code_params *p, int budget_bytes, Layout *lay)
{
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    int off = 0;
    auto take = [&](int words) { int o = off; off += (words + 3) & ~3; return o; };   // keep 16-B alignment
    lay->nw = (n + 31) / 32;
    lay->cn_state = take(ST::lds_words(nk));
    lay->U = take(lay->nw);
    lay->fbits = take(((nk + 63) / 64) * 2);
    lay->pos_cnt = take(p->L);
    lay->pos_ss = take(p->L);
    lay->scal = take(S_NSCAL);
    const int left = budget_bytes / 4 - off;
    int qcap = (left / 2) & ~3;
    if (qcap > 8192) qcap = 8192;
    if (qcap < 256) return -1;
    lay->qcap = qcap;
    lay->q0 = take(qcap);
    lay->q1 = take(qcap);
    lay->total = off;
    return 0;
}

bool packed_ok(const scldpc_code_params *p)
{
    return (int64_t)p->dv * p->vns_pos <= 4096 && p->dc <= 15 && p->dv <= 8;
}

}  // namespace

extern "C" int64_t scldpc_full_bp_lds_bytes(const scldpc_code_params *p)
{
    if (int rc = scldpc::check_params(p)) return rc;
    Layout lay;
    if (packed_ok(p) && make_layout<Packed>(p, scldpc::kMaxLdsBytes / 2 - 1024, &lay) == 0) return 4ll * lay.total;
    if (make_layout<Wide>(p, scldpc::kMaxLdsBytes, &lay) == 0) return 4ll * lay.total;
    if (make_layout<WideG>(p, scldpc::kMaxLdsBytes, &lay) == 0) return 4ll * lay.total;   // CN words in the workspace
    return 4ll * scldpc::nw_of(p) + (64 << 10);
}

static int launch_full_bp(const scldpc_code_params *p, int32_t ntrials, const void *d_vn_adj, bool adj16,
                          const uint32_t *d_chan_bits, int32_t max_it, int32_t is_term,
                          int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                          uint32_t *d_erased_bits, const scldpc::Scratch &scratch, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (scratch.query) *scratch.query = 0;
    if (!scratch.query && (ntrials < 0 || (ntrials > 0 && (!d_counters || !d_vn_adj || !d_chan_bits))))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_full_bp_device: null buffer or negative ntrials");
    if (d_rows && rows_cap <= 0)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_full_bp_device: d_rows given but rows_cap <= 0");
    if (ntrials <= 0) return SCLDPC_OK;
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    if (p->dc > 15 || p->dv > 8 || (int64_t)p->dc * n >= (1ll << kDegShift))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE,
                                 "scldpc_full_bp_device: needs dc <= 15, dv <= 8, dc*n < 2^24 (got dc=%d dv=%d n=%d)",
                                 p->dc, p->dv, n);
    const bool traj = d_rows != nullptr;
    Args a{};
    // two workgroups per CU with the packed CN words when the ensemble allows it
    // (trajectory rows: the side array of boundary degrees must fit queue 1)
    const bool packed = packed_ok(p) && make_layout<Packed>(p, scldpc::kMaxLdsBytes / 2 - 1024, &a.lay) == 0 &&
                        (!traj || 2 * (p->dv - 1) * p->cns_pos <= 4 * a.lay.qcap);
    bool global_ws = false;
    // Wide words: two workgroups per CU when everything (with queues of >= 1024 entries) fits half the LDS, else one
    auto fits = [&](auto tag, Layout *lay) {
        using ST = decltype(tag);
        if (make_layout<ST>(p, scldpc::kMaxLdsBytes / 2 - 1024, lay) == 0 && lay->qcap >= 1024) return true;
        return make_layout<ST>(p, scldpc::kMaxLdsBytes, lay) == 0;
    };
    if (!packed && !fits(Wide{}, &a.lay)) {
        // CN words to a global workspace; the VN bitmap, scan bitmap and queues must still fit the LDS
        if (!fits(WideG{}, &a.lay))
            return scldpc::set_error(SCLDPC_ERR_TOO_LARGE,
                                     "scldpc_full_bp_device: n=%d VN bits + nk=%d scan bits do not fit 160 KiB of LDS", n, nk);
        global_ws = true;
        const size_t need = (size_t)ntrials * nk * sizeof(uint32_t);
        if (scratch.query) { *scratch.query = need; return SCLDPC_OK; }
        void *ws = nullptr;
        if (int rc = scldpc::take_scratch("scldpc_full_bp_device", scratch, need, &ws)) return rc;
        a.ws = static_cast<uint32_t *>(ws);
    }
    if (scratch.query) return SCLDPC_OK;
    a.dv = p->dv; a.L = p->L; a.vns_pos = p->vns_pos; a.cns_pos = p->cns_pos; a.n = n; a.nk = nk;
    a.cn_lim = is_term ? nk : p->L * p->cns_pos;                    // BPT:944-948
    a.max_it = max_it; a.rows_cap = traj ? rows_cap : 0;
    if (!scldpc::magic_of(p->vns_pos, n, &a.magic_v) || !scldpc::magic_of(p->cns_pos, nk, &a.magic_c))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_full_bp_device: reciprocal division inexact for this size");
    a.vn_adj = d_vn_adj; a.chan = d_chan_bits; a.counters = d_counters; a.rows = d_rows;
    a.erased_out = d_erased_bits;

    void (*kern)(const Args) = nullptr;
    int block = 1024;
    const bool d4 = p->dv == 4;
#define PICK(ST, TRAJ, BLK) (d4 ? (adj16 ? full_bp_kernel<ST, TRAJ, 4, true, BLK> : full_bp_kernel<ST, TRAJ, 4, false, BLK>) \
                                : (adj16 ? full_bp_kernel<ST, TRAJ, 0, true, BLK> : full_bp_kernel<ST, TRAJ, 0, false, BLK>))
    if (packed) kern = traj ? PICK(Packed, true, 1024) : PICK(Packed, false, 1024);
    else if (global_ws) kern = traj ? PICK(WideG, true, 1024) : PICK(WideG, false, 1024);
    else if (traj) kern = PICK(Wide, true, 1024);
    else kern = PICK(Wide, false, 1024);
#undef PICK
    // the workspace's CN words through an LDS ring (cn_build.hip) instead of one global atomic per edge
    if (global_ws && adj16) {
        bool pre = true;
        if (const char *v = getenv("SCLDPC_DEBUG_FULLBP_PREBUILD")) pre = atoi(v) != 0;                     // A/B, tests
        a.prebuilt = pre && scldpc::cn_build_launch(p, ntrials, static_cast<const uint16_t *>(d_vn_adj), d_chan_bits, a.ws, traj, stream) ? 1 : 0;
    }
    const size_t lds_bytes = 4u * (size_t)a.lay.total;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(block), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_full_bp_device(const scldpc_code_params *p, int32_t ntrials,
                                     const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                                     int32_t max_it, int32_t is_term,
                                     int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                                     uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    return launch_full_bp(p, ntrials, d_vn_adj, false, d_chan_bits, max_it, is_term, d_counters, d_rows, rows_cap,
                          d_erased_bits, scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

extern "C" int scldpc_full_bp_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                           const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                           int32_t max_it, int32_t is_term,
                                           int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                                           uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream)
{
    return launch_full_bp(p, ntrials, d_vn_adj16, true, d_chan_bits, max_it, is_term, d_counters, d_rows, rows_cap,
                          d_erased_bits, scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

// workspace of scldpc_full_bp_device(_adj16) / scldpc_full_bp_fixpoint_device(_adj16) for ntrials trials (rows: trajectory mode)
int64_t scldpc_full_bp_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t want_rows)
{
    uint64_t need = 0;
    int32_t dummy_rows = 0;
    const int rc = launch_full_bp(p, ntrials, nullptr, true, nullptr, 0, 1, nullptr, want_rows ? &dummy_rows : nullptr,
                                  want_rows ? 1 : 0, nullptr, scldpc::Scratch{nullptr, 0, &need}, nullptr);
    return rc ? (int64_t)rc : (int64_t)need;
}

// The fixpoint of unlimited flooding by chain-following peeling (full_bp_fixpoint_kernel).  Ensembles the packed words do
// not cover (dv != 4, dv*vns_pos > 4096, LDS) take the level-synchronous kernel: same results, ITERATIONS = flooding iterations.
static int launch_full_bp_fixpoint(const scldpc_code_params *p, int32_t ntrials, const void *d_vn_adj, bool adj16,
                                   const uint32_t *d_chan_bits, int32_t is_term, int32_t *d_counters,
                                   uint32_t *d_erased_bits, const scldpc::Scratch &scratch, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    Args a{};
    const bool ok = p->dv == 4 && packed_ok(p) && make_layout<Packed>(p, scldpc::kMaxLdsBytes / 2 - 1024, &a.lay) == 0;
    if (!ok)
        return launch_full_bp(p, ntrials, d_vn_adj, adj16, d_chan_bits, 0, is_term, d_counters, nullptr, 0, d_erased_bits,
                              scratch, stream);
    if (ntrials < 0 || (ntrials > 0 && (!d_counters || !d_vn_adj || !d_chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_full_bp_fixpoint_device: null buffer or negative ntrials");
    if (ntrials == 0) return SCLDPC_OK;
    const int n = scldpc::n_of(p), nk = scldpc::nk_of(p);
    a.dv = p->dv; a.L = p->L; a.vns_pos = p->vns_pos; a.cns_pos = p->cns_pos; a.n = n; a.nk = nk;
    a.cn_lim = is_term ? nk : p->L * p->cns_pos;                    // BPT:944-948
    if (!scldpc::magic_of(p->vns_pos, n > 4096 ? n : 4096, &a.magic_v) || !scldpc::magic_of(p->cns_pos, nk, &a.magic_c))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_full_bp_fixpoint_device: reciprocal division inexact for this size");
    a.vn_adj = d_vn_adj; a.chan = d_chan_bits; a.counters = d_counters; a.erased_out = d_erased_bits;
    void (*kern)(const Args) = adj16 ? full_bp_fixpoint_kernel<true, 1024> : full_bp_fixpoint_kernel<false, 1024>;
    const size_t lds_bytes = 4u * (size_t)a.lay.total;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(1024), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_full_bp_fixpoint_device(const scldpc_code_params *p, int32_t ntrials,
                                              const int32_t *d_vn_adj, const uint32_t *d_chan_bits, int32_t is_term,
                                              int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace,
                                              uint64_t workspace_bytes, void *stream)
{
    return launch_full_bp_fixpoint(p, ntrials, d_vn_adj, false, d_chan_bits, is_term, d_counters, d_erased_bits,
                                   scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

extern "C" int scldpc_full_bp_fixpoint_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                                    const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits, int32_t is_term,
                                                    int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace,
                                                    uint64_t workspace_bytes, void *stream)
{
    return launch_full_bp_fixpoint(p, ntrials, d_vn_adj16, true, d_chan_bits, is_term, d_counters, d_erased_bits,
                                   scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}
