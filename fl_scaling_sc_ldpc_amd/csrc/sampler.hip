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
// LDS ring, and emits VN position p-dv+1 as whole rows (16 B as int32 CN ids, or 8 B as uint16
// position-local CN ids), so HBM sees only full-line stores.
// The exact-replay sampler (glibc stream, identical seeds) is glibc_sampler.cpp.
#include "common.h"
#include "kernel_util.h"
#include "philox.h"
#include <cstdlib>

namespace {

constexpr int kBuckets = 2048;            // big kernel, fused ranking: straddling buckets listed per position (1200 at N = 10000)
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxDoped = 32;

struct SArgs {
    int force_wide;             // diagnostics / tests: rank every position of the big kernel by the 16-bit-counter fallback
    int fine;                   // big kernel: nb counter words (4 * nb nibble buckets) instead of nb / 2
    int fused;                  // big kernel: the ranking with the stage of S sockets in LDS (rank_fused)
    int dv, dc, L, cns_pos, vns_pos, n, S, D, nb, shift, lgchunk, dc_shift, nw;
    int ens, wrapL, pbits;      // ensemble: 0 Olmos chain, 1 tail-biting (stream and CN position wrap at wrapL = L), 2 protograph
    int ndoped;
    int doped[kMaxDoped];
    uint32_t seed_lo, seed_hi;
    unsigned long long trial0;
    uint32_t thresh;            // erased iff (draw >> 1) < thresh
    int off_gkey, off_gidx, off_win, off_wsum;   // LDS offsets in 32-bit words
    int32_t *vn_adj;            // int32 [T][n][dv]  (or)
    uint16_t *vn_adj16;         // uint16 [T][n][dv], CN index local to its position
    uint32_t *chan;
};

using scldpc_dev::philox4x32_10;

using scldpc_dev::wave_inclusive_scan;

// KMAX = Philox calls per thread per permutation (4 sockets each); ROWS = 64-counter rows each wave scans
// ENS = 2: protograph chain (sc_ldpc_protograph.py:6-20).  A pass then ranks the S = dv*vns_pos sockets of ONE VN position,
// socket s = (portion*dv + i)*cns_pos + u; the top pbits bits of its key are its permutation id s / cns_pos, so the one
// ranking orders all dc permutations of cns_pos elements at once (rank - id*cns_pos = perm value), and the position's
// rows go out right after its own pass (no ring).  ENS = 0 also serves the tail-biting closure (sc_ldpc.py:41-45): the
// stream index and the emitted CN position wrap at wrapL.
// FINE: four byte-wide bucket counters per 32-bit word of `hist` — four times as many buckets in the same LDS, so that a
// bucket holds 0.24 keys on average instead of 1 and the ranking loop below runs ~3 steps per wave instead of ~7.  After the
// scan a word holds [exclusive prefix:14 | c0:4 | c1:4 | c2:4 | c3:4].  A bucket with 16 keys (never, for Philox keys: the
// mean is 0.24) traps instead of corrupting its neighbour.
// Two of these 1024-thread workgroups share a CU only if a wave's SGPR allocation lets 8 waves sit on a SIMD: the 800-entry
// scalar file admits ⌊800 / (⌈sgpr/16⌉·16 + 16)⌋ waves, i.e. at most 80 SGPRs per wave.  Left alone the compiler takes 106
// (one workgroup per CU, half the throughput); capped, the few extra uniforms live in VGPR lanes.
template <int KMAX, int ROWS, bool ADJ16, int ENS, bool FINE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_num_sgpr(72))) void sample_philox_kernel(const SArgs a)
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
    const int kshift = FINE ? a.shift - 2 : a.shift;                // key >> kshift = bucket

    STAMP_DECL
    for (int b = tid; b < nb; b += kThreads) hist[b] = 0;
    __syncthreads();
    for (int p = 0; p < a.D; p++) {
        STAMP(0);                                   // (emit of the previous position)

        // ---- keys + bucket histogram; the atomic's return value is the arrival slot in the bucket
        uint32_t key[KMAX * 4], slot[KMAX * 4], crowded = 0;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const int q = tid + k * kThreads;
            if (q < ncalls) {
                uint32_t r[4];
                philox4x32_10((uint32_t)q, (uint32_t)(p >= a.wrapL ? p - a.wrapL : p), t_lo, t_hi, a.seed_lo, a.seed_hi, r);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (ENS == 2) r[u] = ((uint32_t)((q * 4 + u) / a.cns_pos) << (32 - a.pbits)) | (r[u] >> a.pbits);
                    key[k * 4 + u] = r[u];
                    if (q * 4 + u < S) {
                        if (FINE) {
                            const uint32_t b = r[u] >> kshift, sh = (b & 3u) * 8u;
                            slot[k * 4 + u] = (atomicAdd(&hist[b >> 2], 1u << sh) >> sh) & 0xFFu;
                            crowded = max(crowded, slot[k * 4 + u]);
                        } else {
                            slot[k * 4 + u] = atomicAdd(&hist[r[u] >> kshift], 1u);
                        }
                    }
                }
            }
        }
        if (FINE && crowded >= 15u) __builtin_trap();                // a bucket count must fit its nibble (never happens)
        __syncthreads();
        STAMP(1);                                   // keys + histogram

        // ---- exclusive scan of the bucket counts.  Each wave owns nb/16 contiguous words = ROWS rows of 64; the rows are
        //      scanned independently (DPP) and chained by their totals; after the barrier every wave scans the 16 slice
        //      totals itself and folds its own slice base into its words, so a reader needs ONE word per key:
        //      FINE word = [global exclusive prefix:14 | c0:4 | c1:4 | c2:4 | c3:4]  (base of bucket k = prefix + c0..c(k-1)).
        uint32_t mine[ROWS];
        {
            const int base = wave * (ROWS * 64) + lane;
            uint32_t v[ROWS], inc[ROWS], raw[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                raw[r] = hist[base + r * 64];
                v[r] = FINE ? (raw[r] * 0x01010101u) >> 24 : raw[r];        // FINE: keys in the word's four buckets
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) inc[r] = wave_inclusive_scan(v[r]);
            uint32_t carry = 0;
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const uint32_t excl = carry + inc[r] - v[r];
                if (FINE) {     // byte counts (each < 16, checked above) → nibbles
                    const uint32_t x = raw[r];
                    mine[r] = excl | ((x & 0xFu) << 14) | (((x >> 8) & 0xFu) << 18) | (((x >> 16) & 0xFu) << 22) | ((x >> 24) << 26);
                } else {
                    mine[r] = excl;
                }
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc[r], 63);
            }
            if (lane == 0) wsum[wave] = carry;
        }
        __syncthreads();
        STAMP(2);                                   // scan
        {
            const uint32_t t = lane < kWaves ? wsum[lane] : 0u;
            const uint32_t inc = wave_inclusive_scan(t);
            const uint32_t slice = (uint32_t)__builtin_amdgcn_readlane((int)(inc - t), wave);
            const int base = wave * (ROWS * 64) + lane;
#pragma unroll
            for (int r = 0; r < ROWS; r++) hist[base + r * 64] = mine[r] + slice;
        }
        __syncthreads();

        // ---- group (key, socket) by bucket.  The bucket words of the thread's keys are read together (indices clamped
        //      instead of branched on), one wait, then the stores.
        uint32_t g0s[KMAX * 4], g1s[KMAX * 4];
        {
            constexpr int E = KMAX * 4;
            const uint32_t nbk = FINE ? 4u * (uint32_t)nb : (uint32_t)nb;
            uint32_t b0[E], h0[E], h1[E];
            bool ok[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int q = tid + (e >> 2) * kThreads, sck = q * 4 + (e & 3);
                ok[e] = q < ncalls && sck < S;
                b0[e] = ok[e] ? key[e] >> kshift : 0u;
                h0[e] = hist[FINE ? b0[e] >> 2 : b0[e]];
                if (!FINE) h1[e] = hist[min(b0[e] + 1u, nbk - 1u)];
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (FINE) {
                    const uint32_t k = b0[e] & 3u, h = h0[e];
                    const uint32_t c0 = (h >> 14) & 15u, c1 = (h >> 18) & 15u, c2 = (h >> 22) & 15u;
                    g0s[e] = (h & 0x3FFFu) + (k > 0 ? c0 : 0u) + (k > 1 ? c1 : 0u) + (k > 2 ? c2 : 0u);
                    g1s[e] = g0s[e] + ((h >> (14u + 4u * k)) & 15u);
                } else {
                    g0s[e] = h0[e];
                    g1s[e] = b0[e] + 1u >= nbk ? (uint32_t)S : h1[e];
                }
                if (!ok[e]) g0s[e] = g1s[e] = 0;
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (ok[e] && g1s[e] - g0s[e] > 1u) {                // a key alone in its bucket has rank g0: nothing to compare
                    const uint32_t g = g0s[e] + slot[e];
                    gkey[g] = key[e];
                    gidx[g] = (uint16_t)((tid + (e >> 2) * kThreads) * 4 + (e & 3));
                }
            }
        }
        __syncthreads();
        STAMP(3);                                   // group

        // ---- rank = bucket base + #(smaller (key, socket) pairs in the bucket); CN-local id = rank / dc.
        //      Step k reads bucket-mate k of each of the thread's keys: the reads go out together (unconditionally, with
        //      a harmless index when the bucket is exhausted), one wait per step; the socket id only on a key tie.
        uint16_t *wp = win + (size_t)(p % dv) * S;
        for (int b = tid; b < nb; b += kThreads) hist[b] = 0;        // the counters are dead: clear them for the next position
        {
            constexpr int E = KMAX * 4;
            uint32_t rank[E], span = 0;
#pragma unroll
            for (int e = 0; e < E; e++) { rank[e] = g0s[e]; span = max(span, g1s[e] - g0s[e]); }
            for (uint32_t step = 0; step < span; step++) {
                uint32_t k2[E];
                bool on[E], tie = false;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const uint32_t g = g0s[e] + step;
                    on[e] = g < g1s[e] && step != slot[e];          // slot[e] is this key's own place in the bucket
                    k2[e] = gkey[on[e] ? g : 0u];
                }
#pragma unroll
                for (int e = 0; e < E; e++) {
                    rank[e] += on[e] && k2[e] < key[e];
                    tie |= on[e] && k2[e] == key[e];
                }
                if (tie) {
#pragma unroll
                    for (int e = 0; e < E; e++)
                        if (on[e] && k2[e] == key[e])
                            rank[e] += gidx[g0s[e] + step] < (uint16_t)((tid + (e >> 2) * kThreads) * 4 + (e & 3));
                }
            }
#pragma unroll
            for (int k = 0; k < KMAX; k++) {
                const int q = tid + k * kThreads, s0 = q * 4;
                uint32_t id[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t r = rank[k * 4 + u];
                    if (ENS == 2) id[u] = r - (uint32_t)(((s0 + u) / a.cns_pos) * a.cns_pos);
                    else          id[u] = a.dc_shift >= 0 ? r >> a.dc_shift : r / (uint32_t)a.dc;
                }
                if (q < ncalls && s0 + 3 < S && (S & 3) == 0) {     // the usual case: one 8-byte store for 4 sockets
                    uint2 v; v.x = id[0] | (id[1] << 16); v.y = id[2] | (id[3] << 16);
                    *reinterpret_cast<uint2 *>(wp + s0) = v;
                } else if (q < ncalls) {
#pragma unroll
                    for (int u = 0; u < 4; u++) if (s0 + u < S) wp[s0 + u] = (uint16_t)id[u];
                }
            }
        }
        __syncthreads();
        STAMP(4);                                   // rank

        // ---- VN position q = p-dv+1 now has all its dv permutations in the ring (BPF:1703-1716)
        const int qpos = ENS == 2 ? p : p - (dv - 1);
        if (ENS == 2) {
            for (int t = tid; t < a.vns_pos; t += kThreads) {
                const size_t j = (size_t)blockIdx.x * a.n + (size_t)qpos * a.vns_pos + t;
                const int portion = t / a.cns_pos, u = t - portion * a.cns_pos;
                for (int i = 0; i < dv; i++) {
                    const uint32_t l = wp[(size_t)(portion * dv + i) * a.cns_pos + u];
                    if (ADJ16) a.vn_adj16[j * dv + i] = (uint16_t)l;
                    else       a.vn_adj[j * dv + i] = (qpos + i) * a.cns_pos + (int)l;
                }
            }
        } else if (qpos >= 0) {
            // tail-biting: CN position (qpos+i) mod L (global ids only; the position-local layout has no wrap)
            auto cpos = [&](int i) { const int c = qpos + i; return c >= a.wrapL ? c - a.wrapL : c; };
            for (int t = tid; t < a.vns_pos; t += kThreads) {
                const size_t j = (size_t)blockIdx.x * a.n + (size_t)qpos * a.vns_pos + t;
                if (dv == 4) {
                    const uint32_t l0 = win[(size_t)((qpos + 0) & 3) * S + 4 * t + 0];
                    const uint32_t l1 = win[(size_t)((qpos + 1) & 3) * S + 4 * t + 1];
                    const uint32_t l2 = win[(size_t)((qpos + 2) & 3) * S + 4 * t + 2];
                    const uint32_t l3 = win[(size_t)((qpos + 3) & 3) * S + 4 * t + 3];
                    if (ADJ16) {
                        uint2 v;
                        v.x = l0 | (l1 << 16); v.y = l2 | (l3 << 16);
                        reinterpret_cast<uint2 *>(a.vn_adj16)[j] = v;
                    } else {
                        int4 v;
                        v.x = cpos(0) * a.cns_pos + (int)l0; v.y = cpos(1) * a.cns_pos + (int)l1;
                        v.z = cpos(2) * a.cns_pos + (int)l2; v.w = cpos(3) * a.cns_pos + (int)l3;
                        reinterpret_cast<int4 *>(a.vn_adj)[j] = v;
                    }
                } else {
                    for (int i = 0; i < dv; i++) {
                        const uint32_t l = win[(size_t)((qpos + i) % dv) * S + dv * t + i];
                        if (ADJ16) a.vn_adj16[j * dv + i] = (uint16_t)l;
                        else       a.vn_adj[j * dv + i] = cpos(i) * a.cns_pos + (int)l;
                    }
                }
            }
        }
        STAMP(5);                                   // emit
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
    STAMP(6);                                       // channel
    STAMP_FLUSH();
}

// Ensembles beyond 8192 sockets per position (the notebook's N = 10000 is 40000): the bucket counters (16 bits, two per
// word: every bucket's first rank after the scan), the sockets' arrival slots (one byte each) stay in LDS — 72 KB at
// S = 40000, two workgroups per CU — and the keys are drawn again by each of the three passes instead of being stored
// (Philox is pure VALU).  Only the keys of straddling buckets (stream_bp.hip's fallback has the same ranking) and the ring of dv
// permutations live in a per-trial slice of the caller's workspace (L2-resident): the straddlers are ranked from a dense
// worklist, one trip to the L2 per lane, not one per key and wave.
template <int ROWS, bool ADJ16, bool FUSED>
__global__ __launch_bounds__(kThreads, FUSED ? 4 : 8) __attribute__((amdgpu_num_sgpr(72))) void sample_philox_big_kernel(const SArgs a, char *ws, size_t ws_stride)
{
    extern __shared__ uint32_t lds[];
    uint32_t *hist = lds;                                               // nb / 2 words
    uint32_t *wsum = lds + a.off_wsum;                                  // 16 wave totals, [16] = worklist length
    uint8_t *tsl = reinterpret_cast<uint8_t *>(wsum + 32);              // [S]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long trial = a.trial0 + blockIdx.x;
    const uint32_t t_lo = (uint32_t)trial, t_hi = (uint32_t)(trial >> 32);
    const int S = a.S, nb = a.nb, dv = a.dv, ncalls = (S + 3) >> 2;
    char *base = ws + (size_t)blockIdx.x * ws_stride;
    uint2 *gkey = reinterpret_cast<uint2 *>(base);                      // [S] (key, socket) of straddling buckets' keys, by rank slot
    uint2 *wlist = gkey + S;                                            // [S] the same as a dense list: (key, socket | first rank << 16)
    uint16_t *win = reinterpret_cast<uint16_t *>(wlist + S);            // [dv][S] CN-local id of every socket, by CN position % dv
    // a row is kept by edge — socket s = dv*t + i at i * vns_pos + t — so that the wiring of a VN position reads vns_pos consecutive
    // entries of each of its dv rows instead of every dv-th entry of whole rows (a quarter of the lines)
    const uint32_t Sdv = (uint32_t)(S / dv);
    auto tp = [&](uint32_t sck) { return dv == 4 ? (sck & 3u) * Sdv + (sck >> 2) : (sck % (uint32_t)dv) * Sdv + sck / (uint32_t)dv; };
    auto cn_of = [&](uint32_t rank) { return a.dc_shift >= 0 ? rank >> a.dc_shift : rank / (uint32_t)a.dc; };
    auto bucket_base = [&](uint32_t b) -> uint32_t {
        return b >= (uint32_t)nb ? (uint32_t)S : (hist[b >> 1] >> ((b & 1u) * 16u)) & 0xFFFFu;
    };
    // one past the last rank of a non-empty bucket that starts at g0: the next bucket's first rank, which as 16 bits reads
    // 0 instead of 65536 when S = 65536 and only empty buckets follow
    auto bucket_end = [&](uint32_t b, uint32_t g0) -> uint32_t {
        const uint32_t g1 = bucket_base(b + 1u);
        return g1 < g0 ? g1 + 0x10000u : g1;
    };
    // CN = rank / dc: only buckets whose ranks straddle a multiple of dc need their keys ordered (sampler_v2.hip)
    auto straddles = [&](uint32_t g0, uint32_t g1) {
        return g1 - g0 > 1u && (a.dc_shift >= 0 ? (g0 >> a.dc_shift) != ((g1 - 1u) >> a.dc_shift)
                                                : g0 / (uint32_t)a.dc != (g1 - 1u) / (uint32_t)a.dc);
    };

    // Ranking of CN position p with the 16-bit bucket counters: round 3's fallback for a position in which sixteen keys meet in
    // one of the nibble-wide counters of rank_nib below (never on real draws).
    auto rank_wide = [&](int p) {
        for (int b = tid; b < nb / 2; b += kThreads) hist[b] = 0;
        if (tid == 0) wsum[kWaves] = 0;
        __syncthreads();
        uint32_t crowded = 0;
        for (int q = tid; q < ncalls; q += kThreads) {
            uint32_t r[4];
            philox4x32_10((uint32_t)q, (uint32_t)p, t_lo, t_hi, a.seed_lo, a.seed_hi, r);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = q * 4 + u;
                if (s < S) {
                    const uint32_t b = r[u] >> a.shift, sh = (b & 1u) * 16u;
                    const uint32_t sl = (atomicAdd(&hist[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
                    tsl[s] = (uint8_t)sl;
                    crowded |= sl;
                }
            }
        }
        if (crowded >= 256u) __builtin_trap();              // arrival slots are kept in a byte (a bucket holds 1-4 keys on average)
        __syncthreads();
        // wave w scans buckets [w, w+1) * nb/16 = ROWS * 32 words of two counters: exclusive prefix inside the chunk, then
        // (second barrier) plus the chunks before it — every bucket's first rank, 16 bits
        constexpr int R2 = ROWS / 2;
        const int w0 = wave * (ROWS * 32) + lane;
        {
            uint32_t v[R2], ps[R2], inc[R2];
#pragma unroll
            for (int r = 0; r < R2; r++) { v[r] = hist[w0 + r * 64]; ps[r] = (v[r] & 0xFFFFu) + (v[r] >> 16); }
#pragma unroll
            for (int r = 0; r < R2; r++) inc[r] = wave_inclusive_scan(ps[r]);
            uint32_t carry = 0;
#pragma unroll
            for (int r = 0; r < R2; r++) {
                const uint32_t ex = carry + inc[r] - ps[r];
                hist[w0 + r * 64] = (ex & 0xFFFFu) | ((ex + (v[r] & 0xFFFFu)) << 16);
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc[r], 63);
            }
            if (lane == 0) wsum[wave] = carry;
        }
        __syncthreads();
        {
            const uint32_t t = lane < kWaves ? wsum[lane] : 0u;
            const uint32_t inc = wave_inclusive_scan(t);
            const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(inc - t), wave);
#pragma unroll
            for (int r = 0; r < R2; r++) {                                  // (a first rank of 65536 = S wraps to 0: see bucket_end)
                const uint32_t w = hist[w0 + r * 64];
                hist[w0 + r * 64] = ((w + before) & 0xFFFFu) | (((w >> 16) + before) << 16);
            }
        }
        __syncthreads();
        uint16_t *wp = win + (size_t)(p % dv) * S;
        for (int q = tid; q < ncalls; q += kThreads) {
            uint32_t r[4], c4[4] = {0, 0, 0, 0};            // (a straddler's entry is written from the worklist)
            philox4x32_10((uint32_t)q, (uint32_t)p, t_lo, t_hi, a.seed_lo, a.seed_hi, r);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = q * 4 + u;
                if (s >= S) continue;
                const uint32_t k = r[u], b = k >> a.shift, g0 = bucket_base(b), g1 = bucket_end(b, g0);
                if (!straddles(g0, g1)) { c4[u] = cn_of(g0); continue; }
                gkey[g0 + tsl[s]] = make_uint2(k, (uint32_t)s);
                wlist[atomicAdd(&wsum[kWaves], 1u)] = make_uint2(k, (uint32_t)s | (g0 << 16));
            }
            for (int u = 0; u < 4; u++) if (q * 4 + u < S) wp[tp((uint32_t)(q * 4 + u))] = (uint16_t)c4[u];
        }
        __syncthreads();
        {
            const int nwl = (int)wsum[kWaves];
            for (int w = tid; w < nwl; w += kThreads) {
                const uint2 e = wlist[w];
                const uint32_t k = e.x, s = e.y & 0xFFFFu, g0 = e.y >> 16, g1 = bucket_end(k >> a.shift, g0);
                uint32_t rank = g0;                                     // mates fetched four at a time; a key's own record
                for (uint32_t g = g0; g < g1; g += 4) {                 // compares false with itself
                    uint2 m[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) m[i] = g + i < g1 ? gkey[g + i] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
                    for (int i = 0; i < 4; i++) rank += (m[i].x < k) || (m[i].x == k && m[i].y < s);
                }
                wp[tp(s)] = (uint16_t)cn_of(rank);
            }
        }
        __syncthreads();
    };

    // Round 3 (stream_bp.hip's generation went one step further: its straddlers never leave the LDS): four nibble-wide
    // counters per word plus the word's 16-bit first rank — twice as many buckets in the same LDS (nb / 2 words: 32768 buckets at N = 10000), so half as many keys sit in buckets that
    // straddle two CNs (15 % instead of 30 %) and half as many straddler records travel through the workspace; the arrival
    // slots are nibbles (one 16-bit store per Philox call).  Returns false (for every thread) when a bucket met a sixteenth key.
    auto rank_nib = [&](int p) -> bool {
        // a.fine: nb counter words (4 * nb buckets, twice the LDS: one workgroup per CU at N = 10000) instead of nb / 2
        const int rn = a.fine ? ROWS : ROWS / 2;                            // counter words per thread
        const int nbw = a.fine ? nb : nb / 2, bshift = a.fine ? a.shift - 2 : a.shift - 1;      // key >> bshift = fine bucket
        uint16_t *tsl16 = reinterpret_cast<uint16_t *>(tsl);
        uint32_t k_lo = a.seed_lo, k_hi = a.seed_hi;
        asm volatile("" : "+s"(k_lo), "+s"(k_hi));                          // (no Philox round keys hoisted and spilled)
        for (int b = tid; b < nbw; b += kThreads) hist[b] = 0;
        if (tid == 0) { wsum[kWaves] = 0; wsum[kWaves + 1] = 0; }
        __syncthreads();
        uint32_t crowded = 0;
        for (int q = tid; q < ncalls; q += kThreads) {
            uint32_t r[4], pk = 0;
            philox4x32_10((uint32_t)q, (uint32_t)p, t_lo, t_hi, k_lo, k_hi, r);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (q * 4 + u < S) {
                    const uint32_t b = r[u] >> bshift, sh = (b & 3u) * 4u;
                    const uint32_t sl = (atomicAdd(&hist[b >> 2], 1u << sh) >> sh) & 15u;
                    pk |= sl << (4 * u);
                    crowded |= sl + 1u;
                }
            }
            tsl16[q] = (uint16_t)pk;
        }
        if (crowded & 16u) wsum[kWaves + 1] = 1u;                           // a nibble wrapped
        __syncthreads();
        if (wsum[kWaves + 1] || a.force_wide) { __syncthreads(); return false; }
        {
            uint32_t x[ROWS], v[ROWS], tot = 0;
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                x[r] = 0; v[r] = 0;
                if (r < rn) {
                    x[r] = hist[tid * rn + r];
                    const uint32_t sb = (x[r] & 0x0F0Fu) + ((x[r] >> 4) & 0x0F0Fu);
                    v[r] = (sb + (sb >> 8)) & 0xFFu;
                    tot += v[r];
                }
            }
            const uint32_t inc = wave_inclusive_scan(tot);
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            const uint32_t wt = lane < kWaves ? wsum[lane] : 0u;
            const uint32_t winc = wave_inclusive_scan(wt);
            uint32_t pre = inc - tot + (uint32_t)__builtin_amdgcn_readlane((int)(winc - wt), wave);
#pragma unroll
            for (int r = 0; r < ROWS; r++) if (r < rn) { hist[tid * rn + r] = x[r] | (pre << 16); pre += v[r]; }
        }
        __syncthreads();
        auto bucket_of = [&](uint32_t k, uint32_t &g0, uint32_t &cnt) {     // first rank and size of a key's bucket
            const uint32_t b = k >> bshift, sh = (b & 3u) * 4u, x = hist[b >> 2], below = x & ((1u << sh) - 1u);
            g0 = ((x >> 16) + (below & 0xFu) + ((below >> 4) & 0xFu) + ((below >> 8) & 0xFu)) & 0xFFFFu;
            cnt = (x >> sh) & 0xFu;
        };
        uint16_t *wp = win + (size_t)(p % dv) * S;
        for (int q = tid; q < ncalls; q += kThreads) {
            uint32_t r[4], c4[4] = {0, 0, 0, 0};                            // (a straddler's entry is written from the worklist)
            philox4x32_10((uint32_t)q, (uint32_t)p, t_lo, t_hi, k_lo, k_hi, r);
            const uint32_t slots = tsl16[q];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = q * 4 + u;
                if (s >= S) continue;
                uint32_t g0, cnt;
                bucket_of(r[u], g0, cnt);
                if (!straddles(g0, g0 + cnt)) { c4[u] = cn_of(g0); continue; }
                gkey[g0 + ((slots >> (4 * u)) & 15u)] = make_uint2(r[u], (uint32_t)s);
                wlist[atomicAdd(&wsum[kWaves], 1u)] = make_uint2(r[u], (uint32_t)s | (g0 << 16));
            }
            for (int u = 0; u < 4; u++) if (q * 4 + u < S) wp[tp((uint32_t)(q * 4 + u))] = (uint16_t)c4[u];
        }
        __syncthreads();
        {
            const int nwl = (int)wsum[kWaves];
            for (int w = tid; w < nwl; w += kThreads) {
                const uint2 e = wlist[w];
                const uint32_t k = e.x, s = e.y & 0xFFFFu;
                uint32_t g0, cnt;
                bucket_of(k, g0, cnt);
                uint32_t rank = g0;
                for (uint32_t g = g0; g < g0 + cnt; g += 4) {
                    uint2 m[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) m[i] = g + i < g0 + cnt ? gkey[g + i] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
                    for (int i = 0; i < 4; i++) rank += (m[i].x < k) || (m[i].x == k && m[i].y < s);
                }
                wp[tp(s)] = (uint16_t)cn_of(rank);
            }
        }
        __syncthreads();
        return true;
    };


    // Round 3, the streaming generator's form (stream_bp.hip, rank_fused) for whole trials: the arrival slots stay in registers
    // (one nibble per key), so the LDS holds a stage of the position's S sockets beside the 4 * nb nibble counters; first rank +
    // arrival slot is a rank slot of the key's own: the socket goes to stage[rank slot], the buckets that straddle two CNs are
    // listed by their first arrivals and ordered one bucket per lane (the keys drawn again from their sockets, once each), and
    // the socket -> CN row is the inverse of the stage, built over the counters half a row at a time and written out as whole
    // lines.  Nothing but that row goes through global memory: no records of straddling keys, no partial-line fix-ups.
    // Returns false (for every thread) when a bucket met a sixteenth key or a list overflowed: rank_wide then ranks the position.
    auto rank_fused = [&](int p) -> bool {
        const int nbw = nb, bshift = a.shift - 2;
        uint16_t *stage = reinterpret_cast<uint16_t *>(tsl);
        uint32_t *bl = reinterpret_cast<uint32_t *>(stage + ((S + 1) & ~1));      // [kBuckets] first rank | size << 16
        uint32_t k_lo = a.seed_lo, k_hi = a.seed_hi;
        asm volatile("" : "+s"(k_lo), "+s"(k_hi));
        for (int b = tid; b < nbw; b += kThreads) hist[b] = 0;
        if (tid == 0) { wsum[kWaves] = 0; wsum[kWaves + 1] = 0; wsum[kWaves + 2] = 0; }
        __syncthreads();
        // arrival slots of call tid + k * kThreads: bits 16 (k & 1) of pk[k >> 1]; loops over k are not unrolled (uniform selects)
        uint32_t pk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, crowded = 0;
#pragma unroll 1
        for (int kk = 0; kk * kThreads < ncalls; kk++) {
            const int q = tid + kk * kThreads;
            uint32_t mine = 0;
            if (q < ncalls) {
                uint32_t r[4];
                philox4x32_10((uint32_t)q, (uint32_t)p, t_lo, t_hi, k_lo, k_hi, r);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (q * 4 + u < S) {
                        const uint32_t b = r[u] >> bshift, sh = (b & 3u) * 4u;
                        const uint32_t sl = (atomicAdd(&hist[b >> 2], 1u << sh) >> sh) & 15u;
                        mine |= sl << (4 * u);
                        crowded |= sl + 1u;
                    }
                }
            }
            mine <<= (kk & 1) * 16;
#pragma unroll
            for (int h = 0; h < 8; h++) pk[h] |= (kk >> 1) == h ? mine : 0u;
        }
        if (crowded & 16u) wsum[kWaves + 1] = 1u;
        __syncthreads();
        if (wsum[kWaves + 1] || a.force_wide) { __syncthreads(); return false; }
        {   // exclusive scan, bank-conflict free: wave w owns words [w, w + 1) * nbw / 16, 64 at a time; first ranks into the high halves
            const int cw = nbw / kWaves, w0 = wave * cw;
            uint32_t carry = 0;
            for (int i0 = 0; i0 < cw; i0 += 64) {
                const bool on = i0 + lane < cw;
                const uint32_t x = on ? hist[w0 + i0 + lane] : 0u;
                const uint32_t sb = (x & 0x0F0Fu) + ((x >> 4) & 0x0F0Fu), ps = (sb + (sb >> 8)) & 0xFFu;
                const uint32_t inc = wave_inclusive_scan(ps);
                if (on) hist[w0 + i0 + lane] = x | ((carry + inc - ps) << 16);
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
            if (lane == 0) wsum[wave] = carry;
            __syncthreads();
            const uint32_t t = lane < kWaves ? wsum[lane] : 0u;
            const uint32_t tinc = wave_inclusive_scan(t);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(tinc - t), wave) << 16;
            for (int i0 = 0; i0 < cw; i0 += 64)
                if (i0 + lane < cw) hist[w0 + i0 + lane] += base;
        }
        __syncthreads();
        auto bucket_of = [&](uint32_t key, uint32_t &g0, uint32_t &cnt) {
            const uint32_t b = key >> bshift, sh = (b & 3u) * 4u, x = hist[b >> 2], below = x & ((1u << sh) - 1u);
            g0 = ((x >> 16) + (below & 0xFu) + ((below >> 4) & 0xFu) + ((below >> 8) & 0xFu)) & 0xFFFFu;
            cnt = (x >> sh) & 0xFu;
        };
        bool spill = false;
#pragma unroll 1
        for (int kk = 0; kk * kThreads < ncalls; kk++) {
            const int q = tid + kk * kThreads;
            uint32_t slots = 0;
#pragma unroll
            for (int h = 0; h < 8; h++) slots |= (kk >> 1) == h ? pk[h] : 0u;
            slots >>= (kk & 1) * 16;
            if (q < ncalls) {
                uint32_t r[4];
                philox4x32_10((uint32_t)q, (uint32_t)p, t_lo, t_hi, k_lo, k_hi, r);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int sck = q * 4 + u;
                    if (sck >= S) continue;
                    uint32_t g0, cnt;
                    bucket_of(r[u], g0, cnt);
                    const uint32_t sl = (slots >> (4 * u)) & 15u;
                    stage[g0 + sl] = (uint16_t)sck;
                    if (sl != 0u || !straddles(g0, g0 + cnt)) continue;
                    const uint32_t at = atomicAdd(&wsum[kWaves], 1u);        // the first key to arrive lists its straddling bucket
                    if (at < (uint32_t)kBuckets) bl[at] = g0 | (cnt << 16); else spill = true;
                }
            }
        }
        if (spill) wsum[kWaves + 1] = 1u;
        __syncthreads();
        if (wsum[kWaves + 1]) { __syncthreads(); return false; }
        {   // the straddling buckets, one per lane: keys drawn again (once each) into a scratch over the counters, ordered, the
            // sockets put back in rank order; a lane owns its bucket's slots of the stage
            const int nbl = (int)wsum[kWaves];
            constexpr int kScr = 8192;                                            // 48 KB of the 64 KB of counters
            uint32_t *kscr = hist;
            uint16_t *sscr = reinterpret_cast<uint16_t *>(hist + kScr);
            bool over = false;
            for (int b = tid; b < nbl; b += kThreads) {
                const uint32_t g0 = bl[b] & 0xFFFFu, cnt = bl[b] >> 16;
                const uint32_t base = atomicAdd(&wsum[kWaves + 2], cnt);
                if (base + cnt > (uint32_t)min(kScr, (nbw * 4 / 6) & ~1)) { over = true; continue; }
                for (uint32_t m = 0; m < cnt; m++) {
                    const uint32_t s2 = stage[g0 + m];
                    uint32_t r2[4];
                    philox4x32_10(s2 >> 2, (uint32_t)p, t_lo, t_hi, k_lo, k_hi, r2);
                    kscr[base + m] = (s2 & 2u) ? ((s2 & 1u) ? r2[3] : r2[2]) : ((s2 & 1u) ? r2[1] : r2[0]);
                    sscr[base + m] = (uint16_t)s2;
                }
                for (uint32_t m = 0; m < cnt; m++) {
                    const uint32_t km = kscr[base + m], sm = sscr[base + m];
                    uint32_t rank = g0;
                    for (uint32_t m2 = 0; m2 < cnt; m2++) {
                        const uint32_t k2 = kscr[base + m2], s2 = sscr[base + m2];
                        rank += (k2 < km) || (k2 == km && s2 < sm);
                    }
                    stage[rank] = (uint16_t)sm;
                }
            }
            if (over) wsum[kWaves + 1] = 1u;
        }
        __syncthreads();
        if (wsum[kWaves + 1]) { __syncthreads(); return false; }
        // the socket -> CN row (by edge: tp) = the inverse of the stage, half a row at a time over the counters, out as whole lines
        uint16_t *irow = reinterpret_cast<uint16_t *>(hist);
        uint16_t *wp = win + (size_t)(p % dv) * S;
        const uint32_t half = (uint32_t)((S / 2 + 1) & ~1);
        for (uint32_t h0 = 0; h0 < (uint32_t)S; h0 += half) {
            for (int r = tid; r < S; r += kThreads) {
                const uint32_t t = tp(stage[r]) - h0;
                if (t < half) irow[t] = (uint16_t)cn_of((uint32_t)r);
            }
            __syncthreads();
            const uint32_t cnt16 = min(half, (uint32_t)S - h0);
            if ((h0 & 1u) == 0u && (cnt16 & 1u) == 0u) {
                uint32_t *d32 = reinterpret_cast<uint32_t *>(wp + h0);
                const uint32_t *s32 = reinterpret_cast<const uint32_t *>(irow);
                for (uint32_t w = tid; w < cnt16 / 2; w += kThreads) d32[w] = s32[w];
            } else {
                for (uint32_t w = tid; w < cnt16; w += kThreads) wp[h0 + w] = irow[w];
            }
            __syncthreads();
        }
        return true;
    };

    for (int p = 0; p < a.D; p++) {
        bool ranked;
        if constexpr (FUSED) ranked = rank_fused(p); else ranked = rank_nib(p);
        if (!ranked) rank_wide(p);
        const int qpos = p - (dv - 1);
        if (qpos >= 0) {
            for (int t = tid; t < a.vns_pos; t += kThreads) {
                const size_t j = (size_t)blockIdx.x * a.n + (size_t)qpos * a.vns_pos + t;
                uint32_t l[8];
                if (dv == 4) {                              // four independent loads in flight (a run-time dv loop serialises them)
#pragma unroll
                    for (int i = 0; i < 4; i++) l[i] = win[(size_t)((qpos + i) & 3) * S + (size_t)i * Sdv + t];
                } else {
                    for (int i = 0; i < dv; i++) l[i] = win[(size_t)((qpos + i) % dv) * S + (size_t)i * Sdv + t];
                }
                if (ADJ16 && dv == 4) {
                    *reinterpret_cast<uint2 *>(a.vn_adj16 + j * 4) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
                } else {
                    for (int i = 0; i < dv; i++) {
                        if (ADJ16) a.vn_adj16[j * dv + i] = (uint16_t)l[i];
                        else       a.vn_adj[j * dv + i] = (qpos + i) * a.cns_pos + (int)l[i];
                    }
                }
            }
        }
    }
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
        for (int d = 0; d < a.ndoped; d++) {
            const int lo = max(a.doped[d] * a.vns_pos, j0) - j0, hi = min((a.doped[d] + 1) * a.vns_pos, j0 + 32) - j0;
            if (lo < hi) word &= ~(((hi - lo) == 32 ? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u)) << lo);
        }
        chan[w] = word;
    }
}

template <bool ADJ16>
int launch(const scldpc_code_params *p, int ensemble, uint64_t seed, uint64_t trial0, int32_t ntrials, double eps,
           int32_t ndoped, const int32_t *doped_positions, void *d_adj, uint32_t *d_chan_bits,
           const scldpc::Scratch &scratch, void *stream, const char *who)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (scratch.query) *scratch.query = 0;
    if (!scratch.query && (ntrials < 0 || (ntrials > 0 && (!d_adj || !d_chan_bits))))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: null buffer or negative ntrials", who);
    if (ndoped < 0 || ndoped > kMaxDoped || (ndoped > 0 && !doped_positions))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: 0 <= ndoped <= %d", who, kMaxDoped);
    if (!(eps >= 0.0 && eps <= 1.0))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: eps=%g outside [0,1]", who, eps);
    if (ntrials <= 0) return SCLDPC_OK;

    SArgs a{};
    a.dv = p->dv; a.dc = p->dc; a.L = p->L; a.cns_pos = p->cns_pos; a.vns_pos = p->vns_pos;
    a.n = scldpc::n_of(p); a.S = p->cns_pos * p->dc; a.D = p->L + p->dv - 1; a.nw = scldpc::nw_of(p);
    if (a.S > 65536 || p->dv > 8)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: %d sockets per position > 65536", who, a.S);
    const bool big = a.S > 8192;
    if (ensemble < SCLDPC_ENS_OLMOS || ensemble > SCLDPC_ENS_PROTOGRAPH)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: unknown ensemble %d", who, ensemble);
    a.ens = ensemble; a.wrapL = ensemble == SCLDPC_ENS_TAIL_BITING ? p->L : 0x7FFFFFFF;
    while ((1 << a.pbits) < p->dc) a.pbits++;
    if (ensemble != SCLDPC_ENS_OLMOS && big)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: the tail-biting / protograph samplers take at most 8192 sockets "
                                 "per position (got %d)", who, a.S);
    if (ensemble == SCLDPC_ENS_TAIL_BITING && (ADJ16 || p->L < p->dv))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: tail-biting needs global CN ids (int32 adjacency) and L >= dv", who);
    if (ensemble == SCLDPC_ENS_PROTOGRAPH) {
        if (p->dc % p->dv != 0)
            return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: the protograph ensemble needs dv | dc (vns_pos = (dc/dv) * cns_pos)", who);
        a.D = p->L;                                     // one pass per VN position
    }
    if (ADJ16 && p->cns_pos > 65536)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: position-local CN ids need cns_pos <= 65536", who);
    int lg = 10;                                    // nb = power of two >= max(S, kThreads), at most 16384
    while ((1 << lg) < a.S && lg < 14) lg++;
    a.nb = 1 << lg; a.shift = 32 - lg; a.lgchunk = lg - 4;            // 16 waves
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
    // nb counter words (4 * nb nibble buckets: 7.6 % of the keys in straddling buckets at N = 10000 instead of 15 %) at one
    // workgroup per CU beat nb / 2 words at two: 105 against 165 ms per 8192 trials (profiles/r03_ab_c3_sampler.txt)
    a.fine = 1;
    if (const char *v = getenv("SCLDPC_SAMPLER_BIG_FINE")) a.fine = atoi(v) != 0;
    int off = big ? (a.fine ? a.nb : a.nb / 2) : (a.nb + 3) & ~3;
    if (!big) {
        a.off_gkey = off; off += (a.S + 3) & ~3;
        a.off_gidx = off; off += ((a.S + 1) / 2 + 3) & ~3;
        a.off_win = off;  off += (((size_t)p->dv * a.S + 1) / 2 + 3) & ~3;
    }
    // fused ranking (big, fine): behind the scan scratch a stage of S sockets (2 bytes each; the fallback's byte-wide arrival
    // slots use the same room) and the list of straddling buckets — 152 KB at N = 10000
    a.fused = big && a.fine && (a.S + 3) / 4 <= 16 * kThreads;
    if (const char *v = getenv("SCLDPC_DEBUG_SAMPLER_FUSED")) a.fused = a.fused && atoi(v) != 0;      // A/B, tests
    if (a.fused && 4u * (size_t)(off + 32 + ((a.S + 1) & ~1) / 2 + kBuckets) > (size_t)scldpc::kMaxLdsBytes) a.fused = 0;
    a.off_wsum = off; off += big ? (a.fused ? 32 + ((a.S + 1) & ~1) / 2 + kBuckets : 32 + (a.S + 15) / 16 * 4) : 32 + kWaves * kWaves;
    if (const char *v = getenv("SCLDPC_DEBUG_SAMPLER_WIDE")) a.force_wide = atoi(v);          // diagnostics / tests only
    const size_t lds_bytes = 4u * (size_t)off;
    if (lds_bytes > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: needs %zu B of LDS", who, lds_bytes);
    a.vn_adj = ADJ16 ? nullptr : static_cast<int32_t *>(d_adj);
    a.vn_adj16 = ADJ16 ? static_cast<uint16_t *>(d_adj) : nullptr;
    a.chan = d_chan_bits;

    if (big) {
        const size_t stride = (((size_t)a.S * (16 + 2 * p->dv)) + 255) & ~(size_t)255;
        if (scratch.query) { *scratch.query = stride * (size_t)ntrials; return SCLDPC_OK; }
        void *ws = nullptr;
        if (int rc = scldpc::take_scratch(who, scratch, stride * (size_t)ntrials, &ws)) return rc;
        void (*kb)(const SArgs, char *, size_t) = a.fused ? (a.nb == 16384 ? sample_philox_big_kernel<16, ADJ16, true> : sample_philox_big_kernel<8, ADJ16, true>)
                                                          : (a.nb == 16384 ? sample_philox_big_kernel<16, ADJ16, false> : sample_philox_big_kernel<8, ADJ16, false>);
        if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kb))) return rc_;
        hipLaunchKernelGGL(kb, dim3(ntrials), dim3(kThreads), lds_bytes, static_cast<hipStream_t>(stream), a,
                           static_cast<char *>(ws), stride);
        SCLDPC_HIP_CHECK(hipGetLastError());
        return SCLDPC_OK;
    }
    if (scratch.query) return SCLDPC_OK;
    const int kmax = ((a.S + 3) / 4 + kThreads - 1) / kThreads;     // 1 or 2
    const int rows = a.nb / kThreads;                               // 1, 2, 4 or 8 rows of 64 per wave
    void (*kern)(const SArgs) = nullptr;
    if (ensemble == SCLDPC_ENS_PROTOGRAPH) {
        if (kmax == 1) kern = rows == 1 ? sample_philox_kernel<1, 1, ADJ16, 2, false> : rows == 2 ? sample_philox_kernel<1, 2, ADJ16, 2, false>
                                                                                          : sample_philox_kernel<1, 4, ADJ16, 2, false>;
        else           kern = sample_philox_kernel<2, 8, ADJ16, 2, false>;
        if (kmax == 1 && rows > 4) kern = sample_philox_kernel<2, 8, ADJ16, 2, false>;
    } else {
        // byte-wide bucket counters (FINE) for the chain ensembles
        if (kmax == 1) kern = rows == 1 ? sample_philox_kernel<1, 1, ADJ16, 0, true> : rows == 2 ? sample_philox_kernel<1, 2, ADJ16, 0, true>
                                                                                                : sample_philox_kernel<1, 4, ADJ16, 0, true>;
        else           kern = sample_philox_kernel<2, 8, ADJ16, 0, true>;
        if (kmax == 1 && rows > 4) kern = sample_philox_kernel<2, 8, ADJ16, 0, true>;
    }
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kThreads), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

}  // namespace

extern "C" int scldpc_sample_philox_device(const scldpc_code_params *p, uint64_t seed, uint64_t trial0,
                                           int32_t ntrials, double eps, int32_t ndoped, const int32_t *doped_positions,
                                           int32_t *d_vn_adj, uint32_t *d_chan_bits, void *d_workspace,
                                           uint64_t workspace_bytes, void *stream)
{
    return launch<false>(p, SCLDPC_ENS_OLMOS, seed, trial0, ntrials, eps, ndoped, doped_positions, d_vn_adj, d_chan_bits,
                         scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream, "scldpc_sample_philox_device");
}

extern "C" int scldpc_sample_philox_device_adj16(const scldpc_code_params *p, uint64_t seed, uint64_t trial0,
                                                 int32_t ntrials, double eps, int32_t ndoped,
                                                 const int32_t *doped_positions, uint16_t *d_vn_adj16,
                                                 uint32_t *d_chan_bits, void *d_workspace, uint64_t workspace_bytes,
                                                 void *stream)
{
    return launch<true>(p, SCLDPC_ENS_OLMOS, seed, trial0, ntrials, eps, ndoped, doped_positions, d_vn_adj16, d_chan_bits,
                        scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream, "scldpc_sample_philox_device_adj16");
}

extern "C" int scldpc_sample_philox_ensemble_device(const scldpc_code_params *p, int32_t ensemble, uint64_t seed,
                                                    uint64_t trial0, int32_t ntrials, double eps, int32_t ndoped,
                                                    const int32_t *doped_positions, int32_t *d_vn_adj,
                                                    uint32_t *d_chan_bits, void *stream)
{
    return launch<false>(p, ensemble, seed, trial0, ntrials, eps, ndoped, doped_positions, d_vn_adj, d_chan_bits,
                         scldpc::Scratch{nullptr, 0, nullptr}, stream, "scldpc_sample_philox_ensemble_device");
}

// workspace of scldpc_sample_philox_device(_adj16) for ntrials trials (ensembles beyond 8192 sockets per position)
int64_t scldpc_sample_workspace_query(const scldpc_code_params *p, int32_t ntrials)
{
    uint64_t need = 0;
    const int rc = launch<true>(p, SCLDPC_ENS_OLMOS, 0, 0, ntrials, 0.5, 0, nullptr, nullptr, nullptr,
                                scldpc::Scratch{nullptr, 0, &need}, nullptr, "scldpc_workspace_bytes");
    return rc ? (int64_t)rc : (int64_t)need;
}
