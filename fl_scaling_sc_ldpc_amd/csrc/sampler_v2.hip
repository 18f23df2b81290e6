// Throughput-mode sampler, second generation (gfx950) — the (dv = 4, dc = 8) Olmos chain with up to 4096 sockets per
// CN position, i.e. the BASELINE ensemble (4,8,L,N <= 1024).  Same law and same Philox keys as sampler.hip
// (generate_code / channel_doped, BPF:1656-1761, 1547-1574): its vn_adj16 and channel words are bit for bit those of
// scldpc_sample_philox_device_adj16 and of the CPU twin.  Two differences:
//
//  * Ranking only where it matters.  The CN of a socket is rank / dc, so a bucket of the key histogram whose rank range
//    [g0, g1) lies inside one block of dc ranks gives all its keys the same CN whatever their order: CN = g0 / dc, no
//    comparison, no grouping.  Only the keys of buckets that straddle a multiple of dc (3 % of the keys at 0.24 keys per
//    bucket) are written out (packed [key's low bits | socket], one word) and compared with their bucket mates.
//    sampler.hip groups and ranks every key: 61 % of its cycles (profiles/r01_stamps.txt).
//  * The CN -> VN table comes with it: cn_adj16[trial][CN][dc] = the VNs attached to every CN (global VN index, which
//    needs n < 65535), which is just the inverse of the ranking: rank r holds socket s = dv*t + i, i.e. edge i of VN t
//    of position CNpos - i; at the chain ends sockets of positions outside [0, L) stand for no VN (the reference leaves
//    such CNs with a lower degree, BPF:1703-1716) and read 0xFFFF.  Within a CN the order is unspecified (arrival order
//    of the histogram atomics for non-straddling buckets): consumers treat the dc entries as a set.
//    full_bp_small.hip decodes from (vn_adj16, cn_adj16) with 4 bits of LDS per CN.
#include "common.h"
#include "kernel_util.h"
#include "philox.h"
#include <algorithm>
#include <cstdlib>

namespace {

using scldpc_dev::philox4x32_10;
using scldpc_dev::wave_inclusive_scan;

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxDoped = 32;
constexpr int kWorkCap = 1024;          // keys of buckets that span two CNs, per position (expected: 0.03 * S <= 250)

struct S2Args {
    int L, cns_pos, vns_pos, n, S, D, nb, shift, sbits, nw;
    int ntrials;                        // workgroup b samples trials b, b + gridDim.x, … (gridDim.x == ntrials unless persistent)
    int force_exact;                    // diagnostics: rank this CN position by the exact fallback (-1: none, -2: every position)
    int ndoped;
    int doped[kMaxDoped];
    uint32_t seed_lo, seed_hi;
    unsigned long long trial0;
    uint32_t thresh;                    // erased iff (draw >> 1) < thresh
    int off_gpk, off_fix, off_stage, off_wsum;      // LDS offsets in 32-bit words
    uint16_t *vn_adj16;                 // uint16 [T][n][4], CN index local to its position
    uint16_t *cn_adj16;                 // uint16 [T][D*cns_pos][8] VNs of every CN (0xFFFF: none), or null
    uint32_t *chan;
};

// KMAX = Philox calls (4 sockets each) per thread and position: 1 up to 4096 sockets per position, 2 up to 8192
// ROWS = histogram words per thread (nb / 1024)
// CNMODE: what the rank-ordered stage holds — 0 nothing, 1 the sockets' VNs (global VN index, scldpc_sample_philox_device_cn16),
//         2 the sockets themselves (s = dv*t + i: the table scldpc_sw_bp_ring_device reads; any n)
template <int KMAX, int ROWS, int CNMODE, bool PERSIST = false>
__global__ __launch_bounds__(kThreads, KMAX == 1 ? 8 : 4) __attribute__((amdgpu_num_sgpr(72))) void sample_philox_v2_kernel(const S2Args a)
{
    constexpr int DV = 4, DC_SHIFT = 3, E = 4 * KMAX;
    extern __shared__ uint32_t lds[];
    uint32_t *hist = lds;                                               // nb words of four nibble-wide bucket counters
    uint32_t *gpk = lds + a.off_gpk;                                    // S words: packed keys of straddling buckets
    uint16_t *fix = reinterpret_cast<uint16_t *>(lds + a.off_fix);      // S CN-local ids of this position's sockets
    uint16_t *stage = reinterpret_cast<uint16_t *>(lds + a.off_stage);  // the S sockets (or their VNs) in rank order
    uint32_t *wsum = lds + a.off_wsum;                                  // 16 wave totals + the worklist counter
    uint32_t *wl = wsum + 32;                                           // worklist: 2 words per key of a straddling bucket

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = a.S, nb = a.nb;
    const int ncalls = S >> 2;                                          // S % 4 == 0 (checked on the host)
    const int kshift = a.shift - 2;                                     // key >> kshift = fine bucket
    const uint32_t lowmask = (1u << kshift) - 1u;
    bool own[KMAX];                                                     // call k of this thread: sockets 4*(tid + 1024 k) .. +3
#pragma unroll
    for (int k = 0; k < KMAX; k++) own[k] = tid + k * kThreads < ncalls;
    auto sock = [&](int e) { return (uint32_t)((tid + (e >> 2) * kThreads) * 4 + (e & 3)); };
    // what the stage holds for socket sck = 4*t + u at CN position p: edge u of VN t of position p - u (BPF:1712)
    auto stage_entry = [&](int p, uint32_t sck) -> uint16_t {
        const int u = (int)(sck & 3u), t = (int)(sck >> 2);
        if ((unsigned)(p - u) >= (unsigned)a.L) return (uint16_t)0xFFFFu;
        return CNMODE == 1 ? (uint16_t)((p - u) * a.vns_pos + t) : (uint16_t)sck;
    };

    // hist word = [exclusive prefix:16 | n3:4 | n2:4 | n1:4 | n0:4]: four nibble-wide bucket counters in the low half
    // (the atomic's return value is the key's arrival slot), the scan's prefix ORed into the high half.  Thread t owns
    // words t*ROWS .. t*ROWS+ROWS-1 (wide LDS accesses).
    auto nib_sum = [](uint32_t x) {                                     // sum of the four nibbles of the low half
        const uint32_t v = (x & 0x0F0Fu) + ((x >> 4) & 0x0F0Fu);
        return (v + (v >> 8)) & 0xFFu;
    };
    auto load_words = [&](uint32_t (&x)[ROWS]) {
        if constexpr (ROWS % 4 == 0) {
#pragma unroll
            for (int r = 0; r < ROWS / 4; r++) {
                const uint4 q = reinterpret_cast<const uint4 *>(hist)[tid * (ROWS / 4) + r];
                x[4 * r] = q.x; x[4 * r + 1] = q.y; x[4 * r + 2] = q.z; x[4 * r + 3] = q.w;
            }
        } else if constexpr (ROWS == 2) { const uint2 q = reinterpret_cast<const uint2 *>(hist)[tid]; x[0] = q.x; x[1] = q.y; }
        else x[0] = hist[tid];
    };
    auto store_words = [&](const uint32_t (&x)[ROWS]) {
        if constexpr (ROWS % 4 == 0) {
#pragma unroll
            for (int r = 0; r < ROWS / 4; r++)
                reinterpret_cast<uint4 *>(hist)[tid * (ROWS / 4) + r] = make_uint4(x[4 * r], x[4 * r + 1], x[4 * r + 2], x[4 * r + 3]);
        } else if constexpr (ROWS == 2) reinterpret_cast<uint2 *>(hist)[tid] = make_uint2(x[0], x[1]);
        else hist[tid] = x[0];
    };
    STAMP_DECL
    for (int b = tid; b < nb; b += kThreads) hist[b] = 0;
    if (tid == 0) { wsum[kWaves] = 0; wsum[kWaves + 1] = 0; }
    __syncthreads();
    auto sample_trial = [&](const int tr) {
    const unsigned long long trial = a.trial0 + (unsigned long long)tr;
    const uint32_t t_lo = (uint32_t)trial, t_hi = (uint32_t)(trial >> 32);
    // The keys of position p+1 are drawn (pure VALU) while the few worklist lanes of position p chase their bucket mates
    // through the LDS: nxt[] carries them across the barrier.
    // Thread t draws the keys of sockets 4t .. 4t+3 of every CN position, and socket 4t+i of CN position q+i is edge i of
    // VN (q, t) (BPF:1712): the row of a VN is assembled in its thread's registers over dv consecutive positions.
    // Entering step p: rowP = [edge 0 @ p-3 | edge 1 @ p-2], rowQ = [edge 0 @ p-2 | edge 1 @ p-1], rowR = [edge 0 @ p-1 | edge 2 @ p-1]
    uint32_t rowP[KMAX], rowQ[KMAX], rowR[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) rowP[k] = rowQ[k] = rowR[k] = 0;
    uint32_t nxt[E];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        uint32_t r[4] = {0, 0, 0, 0};
        if (own[k]) philox4x32_10((uint32_t)(tid + k * kThreads), 0u, t_lo, t_hi, a.seed_lo, a.seed_hi, r);
#pragma unroll
        for (int u = 0; u < 4; u++) nxt[4 * k + u] = r[u];
    }
    for (int p = 0; p < a.D; p++) {
        STAMP(0);
        // ---- bucket histogram of this position's keys
        uint32_t key[E], slot[E], crowded = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            key[e] = nxt[e];                                            // (threads without sockets hold zero keys and add zero)
            const uint32_t b = key[e] >> kshift, sh = (b & 3u) * 4u;
            slot[e] = (atomicAdd(&hist[b >> 2], (own[e >> 2] ? 1u : 0u) << sh) >> sh) & 0xFu;
            crowded = max(crowded, slot[e]);
        }
        // A bucket count must fit its nibble (0.24 keys per bucket on average: 15 in one never happens in practice) and the
        // straddlers their worklist: when either fails the position is ranked again by the exact fallback below — same
        // CN ids by construction, only slower — instead of trapping the process.
        if (crowded >= 15u) wsum[kWaves + 1] = 1u;
        __syncthreads();
        STAMP(1);

        // ---- exclusive scan of the bucket counts: every thread scans its ROWS words, the wave scans the thread totals
        //      (DPP), the 16 wave totals meet in wsum; the global prefix goes into the words' high halves
        {
            uint32_t x[ROWS], v[ROWS], tot = 0;
            load_words(x);
#pragma unroll
            for (int r = 0; r < ROWS; r++) v[r] = 0;
            if constexpr (ROWS % 2 == 0) {                              // two words' counters (16 bits each) per 32-bit lane
#pragma unroll
                for (int r = 0; r < ROWS; r += 2) {
                    const uint32_t y = x[r] | (x[r + 1] << 16);
                    const uint32_t sb = (y & 0x0F0F0F0Fu) + ((y >> 4) & 0x0F0F0F0Fu);      // four byte sums <= 30
                    v[r] = (sb & 0xFFu) + ((sb >> 8) & 0xFFu);
                    v[r + 1] = ((sb >> 16) & 0xFFu) + (sb >> 24);
                    tot += v[r] + v[r + 1];
                }
            } else {
                static_assert(ROWS == 1 || ROWS % 2 == 0, "ROWS is 1, 2, 4 or 8");
                v[0] = nib_sum(x[0]);
                tot = v[0];
            }
            const uint32_t inc = wave_inclusive_scan(tot);
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            const uint32_t wt = lane < kWaves ? wsum[lane] : 0u;
            const uint32_t winc = wave_inclusive_scan(wt);
            uint32_t pre = inc - tot + (uint32_t)__builtin_amdgcn_readlane((int)(winc - wt), wave);
#pragma unroll
            for (int r = 0; r < ROWS; r++) { x[r] |= pre << 16; pre += v[r]; }
            store_words(x);
        }
        __syncthreads();
        STAMP(2);

        // ---- classify: every key gets rank g0 + arrival slot — any bijection onto its bucket's ranks gives the right CN
        //      (g0 / dc) when the bucket lies inside one block of dc ranks.  CN ids go into the ring, sockets into the
        //      rank-ordered stage.  Keys of buckets that span two CNs (3 %) are also put on a worklist for their true rank.
        if (wsum[kWaves + 1] == 0u) {                                    // (an overflowed histogram has no ranks worth scattering by)
            uint32_t h[E], rk[E], g0a[E], cnta[E], smask = 0;
#pragma unroll
            for (int e = 0; e < E; e++) h[e] = hist[own[e >> 2] ? (key[e] >> kshift) >> 2 : 0u];
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t k4 = ((key[e] >> kshift) & 3u) * 4u, x = h[e];
                const uint32_t below = x & ((1u << k4) - 1u);            // the counters of the word's lower buckets
                g0a[e] = (x >> 16) + (below & 0xFu) + ((below >> 4) & 0xFu) + ((below >> 8) & 0xFu);
                cnta[e] = (x >> k4) & 0xFu;
                rk[e] = g0a[e] + slot[e];
                if (own[e >> 2] && ((g0a[e] + cnta[e] - 1u) >> DC_SHIFT) != (g0a[e] >> DC_SHIFT)) smask |= 1u << e;
            }
            while (smask) {                                             // rare (3 % of the keys): one short divergent loop
                const uint32_t e = (uint32_t)__ffs((int)smask) - 1u;
                smask &= smask - 1u;
                uint32_t ky = key[0], sl = slot[0], g0 = g0a[0], cnt = cnta[0];
#pragma unroll
                for (int f = 1; f < E; f++)
                    if (e == (uint32_t)f) { ky = key[f]; sl = slot[f]; g0 = g0a[f]; cnt = cnta[f]; }
                const uint32_t pk = ((ky & lowmask) << a.sbits) | (uint32_t)((tid + (int)(e >> 2) * kThreads) * 4 + (int)(e & 3u));
                gpk[g0 + sl] = pk;
                const int w = atomicAdd(reinterpret_cast<int *>(&wsum[kWaves]), 1);
                if (w < kWorkCap) { wl[2 * w] = g0 | (cnt << 16) | (sl << 20); wl[2 * w + 1] = pk; }
            }
#pragma unroll
            for (int k = 0; k < KMAX; k++) {                            // provisional CN ids (final unless on the worklist)
                if (!own[k]) continue;
                uint2 v;
                v.x = (rk[4 * k] >> DC_SHIFT) | ((rk[4 * k + 1] >> DC_SHIFT) << 16);
                v.y = (rk[4 * k + 2] >> DC_SHIFT) | ((rk[4 * k + 3] >> DC_SHIFT) << 16);
                reinterpret_cast<uint2 *>(fix)[tid + k * kThreads] = v;
            }
            if constexpr (CNMODE != 0) {
#pragma unroll
                for (int k = 0; k < KMAX; k++) {
                    if (!own[k]) continue;
#pragma unroll
                    for (int u = 0; u < 4; u++) stage[rk[4 * k + u]] = stage_entry(p, sock(4 * k + u));
                }
            }
        }
        __syncthreads();
        STAMP(3);

        // ---- the worklist: true rank among the bucket mates; counters cleared for the next position
        {
            uint32_t z[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) z[r] = 0;
            store_words(z);
            int nwork = (int)wsum[kWaves];                              // 3 % of S on average (<= 250 keys); the list holds 1024
            const bool exact = wsum[kWaves + 1] != 0u || nwork > kWorkCap || a.force_exact == p || a.force_exact == -2;
            if (exact) {
                // every key's true rank among all S keys, ties by socket (what the histogram path computes where it matters):
                // S comparisons per key — never taken in a real run, see above
                nwork = 0;
#pragma unroll
                for (int e = 0; e < E; e++) if (own[e >> 2]) gpk[sock(e)] = key[e];
                __syncthreads();
                uint32_t xr[E];
#pragma unroll
                for (int e = 0; e < E; e++) xr[e] = 0;
                for (int s2 = 0; s2 < S; s2++) {
                    const uint32_t k2 = gpk[s2];
#pragma unroll
                    for (int e = 0; e < E; e++) xr[e] += (k2 < key[e]) || (k2 == key[e] && (uint32_t)s2 < sock(e));
                }
#pragma unroll
                for (int k = 0; k < KMAX; k++) {
                    if (!own[k]) continue;
                    uint2 v;
                    v.x = (xr[4 * k] >> DC_SHIFT) | ((xr[4 * k + 1] >> DC_SHIFT) << 16);
                    v.y = (xr[4 * k + 2] >> DC_SHIFT) | ((xr[4 * k + 3] >> DC_SHIFT) << 16);
                    reinterpret_cast<uint2 *>(fix)[tid + k * kThreads] = v;
                    if constexpr (CNMODE != 0) {
#pragma unroll
                        for (int u = 0; u < 4; u++) stage[xr[4 * k + u]] = stage_entry(p, sock(4 * k + u));
                    }
                }
            }
            for (int w = tid; w < nwork; w += kThreads) {
                const uint32_t ea = wl[2 * w], pk = wl[2 * w + 1];
                const uint32_t g0 = ea & 0xFFFFu, cnt = (ea >> 16) & 0xFu, sl = ea >> 20;
                uint32_t r = g0;
                for (uint32_t m = 0; m < cnt; m++)
                    if (m != sl) r += gpk[g0 + m] < pk;
                const uint32_t sck = pk & ((1u << a.sbits) - 1u);
                if constexpr (CNMODE != 0) stage[r] = stage_entry(p, sck);
                fix[sck] = (uint16_t)(r >> DC_SHIFT);
            }
        }
        if (p + 1 < a.D) {
            // the round keys are recomputed from the seed here (twenty scalar adds) rather than kept in twenty SGPRs across
            // the whole loop, which the 72-SGPR budget of two workgroups per CU would spill into VGPR lanes
            uint32_t k_lo = a.seed_lo, k_hi = a.seed_hi;
            asm volatile("" : "+s"(k_lo), "+s"(k_hi));
#pragma unroll
            for (int k = 0; k < KMAX; k++) {
                uint32_t r[4] = {0, 0, 0, 0};
                if (own[k]) philox4x32_10((uint32_t)(tid + k * kThreads), (uint32_t)(p + 1), t_lo, t_hi, k_lo, k_hi, r);
#pragma unroll
                for (int u = 0; u < 4; u++) nxt[4 * k + u] = r[u];
            }
        }
        __syncthreads();
        if (tid == 0) { wsum[kWaves] = 0; wsum[kWaves + 1] = 0; }        // read again only after the next two barriers
        STAMP(4);

        // ---- VN position q = p-dv+1 now has all its dv edges (BPF:1703-1716); CN position p its sockets
        const int qpos = p - (DV - 1);
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            // this step drew edge 0 of VN position p, edge 1 of p-1, edge 2 of p-2 and edge 3 of p-3 = qpos
            uint2 c = make_uint2(0u, 0u);                               // [edge 0 | edge 1], [edge 2 | edge 3] of this step
            if (own[k]) c = reinterpret_cast<const uint2 *>(fix)[tid + k * kThreads];
            if (qpos >= 0 && own[k]) {
                const size_t j = (size_t)tr * a.n + (size_t)qpos * a.vns_pos + (size_t)(tid + k * kThreads);
                uint2 v;
                v.x = rowP[k];
                v.y = (rowR[k] >> 16) | (c.y & 0xFFFF0000u);
                reinterpret_cast<uint2 *>(a.vn_adj16)[j] = v;
            }
            rowP[k] = rowQ[k];
            rowQ[k] = (rowR[k] & 0xFFFFu) | (c.x & 0xFFFF0000u);
            rowR[k] = (c.x & 0xFFFFu) | (c.y << 16);
        }
        if constexpr (CNMODE != 0) {
            uint2 *dst = reinterpret_cast<uint2 *>(a.cn_adj16) + ((size_t)tr * a.D + p) * (size_t)(S >> 2);
            const uint2 *src = reinterpret_cast<const uint2 *>(stage);
            for (int w = tid; w < (S >> 2); w += kThreads) dst[w] = src[w];
        }
        STAMP(5);
    }

    // ---- channel: 32 VNs per output word, 8 Philox calls
    uint32_t *chan = a.chan + (size_t)tr * a.nw;
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
        for (int d = 0; d < a.ndoped; d++) {                            // doped positions are never erased (BPF:1566-1573)
            const int lo = max(a.doped[d] * a.vns_pos, j0) - j0, hi = min((a.doped[d] + 1) * a.vns_pos, j0 + 32) - j0;
            if (lo < hi) word &= ~(((hi - lo) == 32 ? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u)) << lo);
        }
        chan[w] = word;
    }
    STAMP(6);
    };
    // PERSIST (diagnostics): workgroup b samples trials b, b + gridDim.x, …; otherwise exactly one
    if constexpr (PERSIST) {
        for (int tr = blockIdx.x; tr < a.ntrials; tr += gridDim.x) {
            sample_trial(tr);
            __syncthreads();                            // the stage of the last position has been copied out
        }
    } else {
        sample_trial((int)blockIdx.x);
    }
    STAMP_FLUSH();
}


// ------------------------------------------------------------------------------------------------------------------------
// Third generation of the same sampler (same law, same Philox keys, same tables bit for bit), re-cut along what the gfx950
// vector unit and the barriers charge for (tools/calib/valu.hip, profiles/r03_valu_classes.txt):
//  * add / sub / and / or / xor / mov / right shifts / v_bitop3 issue at one wave-instruction per 2 cycles and SIMD; left
//    shifts, multiplies, every other three-operand form, v_cndmask, DPP, and ANY operation with a scalar-register source
//    take 4.  So the bucket geometry is a template parameter (shifts and masks become immediates, no SGPR operands), byte
//    addresses and nibble shifts are cut out of the key by right shifts and masks, and selects are avoided.
//  * three barriers per CN position instead of five: the histogram is double-buffered, so position p+1 is counted while the
//    straddlers of p are ranked (phase A); the prefix scan is wave-local — the 16 wave totals are added at look-up time
//    through a ds_bpermute instead of a second pass over the counters (phase B, beside the copy-out of position p-1); the
//    scan leaves INCLUSIVE nibble prefixes in the word, so a key's first rank and its bucket's size are two shifts and two
//    masks (phase C).
// hist word after the scan: [wave-local prefix:12 | i3 i2 i1 i0 : inclusive prefixes of the four bucket counters | 0:4].
// CNMODE 1: the VN that socket sck = 4*t + u of CN position p stands for (edge u of VN t of position p - u)
__device__ __forceinline__ uint32_t wl_entry_vn(const S2Args &a, int p, uint32_t sck)
{
    return (uint32_t)((p - (int)(sck & 3u)) * a.vns_pos + (int)(sck >> 2));
}

template <int LG, int CNMODE>
__global__ __launch_bounds__(kThreads, 8) __attribute__((amdgpu_num_sgpr(72))) void sample_philox_v3_kernel(const S2Args a)
{
    constexpr int DV = 4, DC_SHIFT = 3, E = 4;
    constexpr int NB = 1 << LG, ROWS = NB / kThreads;                   // histogram words (4 buckets each); words per thread
    constexpr int KSHIFT = 32 - LG - 2;                                 // key >> KSHIFT = fine bucket
    constexpr uint32_t LOWMASK = (1u << KSHIFT) - 1u;
    constexpr int SBITS = LG;
    static_assert(ROWS == 1 || ROWS == 2 || ROWS == 4, "one Philox call per thread: at most 4096 sockets per position");
    extern __shared__ uint32_t lds[];
    uint32_t *hist0 = lds;                                              // two buffers of NB words
    uint32_t *gpk = lds + a.off_gpk;
    uint16_t *fix = reinterpret_cast<uint16_t *>(lds + a.off_fix);
    uint16_t *stage = reinterpret_cast<uint16_t *>(lds + a.off_stage);
    uint32_t *wsum = lds + a.off_wsum;                                  // [0,16) wave totals, [16] worklist count, [18 + parity] overflow flags
    uint32_t *wl = wsum + 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = a.S;
    const bool own = tid < (S >> 2);                                    // sockets 4*tid .. 4*tid + 3
    const unsigned long long trial = a.trial0 + blockIdx.x;
    const uint32_t t_lo = (uint32_t)trial, t_hi = (uint32_t)(trial >> 32);
    const int tr = blockIdx.x;

    STAMP_DECL
    for (int b = tid; b < 2 * NB; b += kThreads) hist0[b] = 0;
    if (tid < 32) wsum[tid] = 0;
    __syncthreads();

    uint32_t rowP = 0, rowQ = 0, rowR = 0;
    uint32_t key[E] = {0, 0, 0, 0}, slot[E] = {0, 0, 0, 0};
    // what the rank-ordered stage holds for this thread's socket 4*tid + u at the CN position being classified: the VN
    // (p - u) * V + tid (kept as one running register and V in a VGPR: subtractions with a scalar operand cost double)
    uint32_t ent0 = CNMODE == 1 ? (uint32_t)tid : (uint32_t)(4 * tid);
    uint32_t vV = (uint32_t)a.vns_pos;
    asm volatile("" : "+v"(vV));

    // Positions flow through three phases, one barrier each; iteration p runs A(p), B(p), C(p):
    //   A(p): count the keys of p into hist[p & 1]; rank the straddlers of p-1 (patching fix / stage of p-1); clear hist[(p-1) & 1]
    //   B(p): wave-local scan of hist[p & 1]; copy position p-1 out (VN rows from registers, CN rows from the stage)
    //   C(p): look every key of p up: CN ids into fix, sockets into the stage, straddlers onto the worklist
    for (int p = 0; p <= a.D; p++) {
        uint32_t *hc = hist0 + (p & 1) * NB, *hp = hist0 + ((p & 1) ^ 1) * NB;
        const bool live = p < a.D;
        STAMP(0);
        // ================================================ phase A ================================================
        if (p > 0) {
            // the worklist of p-1: true rank among the bucket mates (or, never in a real run, the exact fallback)
            int nwork = (int)wsum[16];
            const bool exact = wsum[18 + ((p - 1) & 1)] != 0u || nwork > kWorkCap || a.force_exact == p - 1 || a.force_exact == -2;
            if (exact) {
                uint32_t k2[4] = {0, 0, 0, 0};
                uint32_t f_lo = a.seed_lo, f_hi = a.seed_hi;
                asm volatile("" : "+s"(f_lo), "+s"(f_hi));              // (no round keys hoisted out of the position loop)
                if (own) philox4x32_10((uint32_t)tid, (uint32_t)(p - 1), t_lo, t_hi, f_lo, f_hi, k2);
                if (own) { gpk[4 * tid] = k2[0]; gpk[4 * tid + 1] = k2[1]; gpk[4 * tid + 2] = k2[2]; gpk[4 * tid + 3] = k2[3]; }
                __syncthreads();
                uint32_t xr[4] = {0, 0, 0, 0};
                for (int s2 = 0; s2 < S; s2++) {
                    const uint32_t kk = gpk[s2];
#pragma unroll
                    for (int u = 0; u < 4; u++) xr[u] += (kk < k2[u]) || (kk == k2[u] && s2 < 4 * tid + u);
                }
                if (own) {
                    uint2 v;
                    v.x = (xr[0] >> DC_SHIFT) | ((xr[1] >> DC_SHIFT) << 16);
                    v.y = (xr[2] >> DC_SHIFT) | ((xr[3] >> DC_SHIFT) << 16);
                    reinterpret_cast<uint2 *>(fix)[tid] = v;
                    if constexpr (CNMODE != 0) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int pu = p - 1 - u;
                            stage[xr[u]] = (unsigned)pu < (unsigned)a.L ? (uint16_t)(CNMODE == 1 ? pu * a.vns_pos + tid : 4 * tid + u) : (uint16_t)0xFFFFu;
                        }
                    }
                }
                nwork = 0;
            }
            for (int w = tid; w < nwork; w += kThreads) {
                const uint32_t ea = wl[2 * w], pk = wl[2 * w + 1];
                const uint32_t g0 = ea & 0xFFFFu, cnt = (ea >> 16) & 0xFu, sl = (ea >> 20) & 0xFu;
                uint32_t r = g0;
                for (uint32_t m = 0; m < cnt; m++)
                    if (m != sl) r += gpk[g0 + m] < pk;
                const uint32_t sck = pk & ((1u << SBITS) - 1u);
                if constexpr (CNMODE != 0) stage[r] = (uint16_t)(ea >> 24 ? 0xFFFFu : (CNMODE == 1 ? wl_entry_vn(a, p - 1, sck) : sck));
                fix[sck] = (uint16_t)(r >> DC_SHIFT);
            }
            {   // the counters of p-1 are free: every look-up of phase C(p-1) lies behind the last barrier
                if constexpr (ROWS == 4) reinterpret_cast<uint4 *>(hp)[tid] = make_uint4(0u, 0u, 0u, 0u);
                else if constexpr (ROWS == 2) reinterpret_cast<uint2 *>(hp)[tid] = make_uint2(0u, 0u);
                else hp[tid] = 0u;
            }
        }
        STAMP(1);                                                       // A: worklist of p-1 + clear
        if (live) {
            {   // the keys of p (threads without sockets hold zero keys and add zero); the round keys are recomputed by
                // scalar adds rather than held in twenty SGPRs across the loop
                uint32_t k_lo = a.seed_lo, k_hi = a.seed_hi;
                asm volatile("" : "+s"(k_lo), "+s"(k_hi));
                uint32_t r[4] = {0, 0, 0, 0};
                if (own) philox4x32_10((uint32_t)tid, (uint32_t)p, t_lo, t_hi, k_lo, k_hi, r);
#pragma unroll
                for (int u = 0; u < 4; u++) key[u] = r[u];
            }
            uint32_t crowded = 0;
#pragma unroll
            for (int e = 0; e < E; e++) {
                // byte address of the word = (bucket >> 2) * 4, nibble shift = (bucket & 3) * 4: right shifts and masks only
                const uint32_t boff = (key[e] >> KSHIFT) & ~3u, sh = (key[e] >> (KSHIFT - 2)) & 12u;
                const uint32_t old = atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(hc) + boff), (own ? 16u : 0u) << sh);
                slot[e] = (old >> (sh + 4u)) & 0xFu;
                crowded |= slot[e] + 1u;
            }
            if (crowded & 16u) wsum[18 + (p & 1)] = 1u;                 // a bucket met its 16th key: its nibble wrapped
        }
        STAMP(2);                                                       // A: keys + histogram
        __syncthreads();
        STAMP(3);                                                       // barrier wait
        // ================================================ phase B ================================================
        if (tid == 0) { wsum[16] = 0; wsum[18 + ((p & 1) ^ 1)] = 0; }  // worklist count and the flag of p-1: consumed in phase A
        if (live) {
            uint32_t x[ROWS], tot = 0, over = 0;
            if constexpr (ROWS == 4) { const uint4 q = reinterpret_cast<const uint4 *>(hc)[tid]; x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w; }
            else if constexpr (ROWS == 2) { const uint2 q = reinterpret_cast<const uint2 *>(hc)[tid]; x[0] = q.x; x[1] = q.y; }
            else x[0] = hc[tid];
            uint32_t v[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                // counters sit in bits [4, 20): c0..c3.  Inclusive prefixes by one multiply (valid while the word's total <= 15)
                const uint32_t y = x[r] * 0x1111u;                      // nibble k of bits [4, 20) = c0 + .. + ck
                const uint32_t sb = (x[r] & 0x0F0F0u) + ((x[r] >> 4) & 0x0F0F0u);
                v[r] = ((sb + (sb >> 8)) >> 4) & 0xFFu;                  // the true total (two byte sums <= 30 each)
                over |= v[r];
                x[r] = y & 0xFFFF0u;
                tot += v[r];
            }
            if (over & ~15u) wsum[18 + (p & 1)] = 1u;                   // sixteen keys in four adjacent buckets: exact fallback
            const uint32_t inc = wave_inclusive_scan(tot);
            uint32_t pre = inc - tot;
#pragma unroll
            for (int r = 0; r < ROWS; r++) { x[r] |= pre << 20; pre += v[r]; }
            if constexpr (ROWS == 4) reinterpret_cast<uint4 *>(hc)[tid] = make_uint4(x[0], x[1], x[2], x[3]);
            else if constexpr (ROWS == 2) reinterpret_cast<uint2 *>(hc)[tid] = make_uint2(x[0], x[1]);
            else hc[tid] = x[0];
            if (lane == 63) wsum[wave] = inc;
        }
        if (p > 0) {
            // ---- VN position q = p-1-dv+1 now has all its dv edges (BPF:1703-1716); CN position p-1 its sockets
            const int pp = p - 1, qpos = pp - (DV - 1);
            uint2 c = make_uint2(0u, 0u);                               // [edge 0 | edge 1], [edge 2 | edge 3] of step pp
            if (own) c = reinterpret_cast<const uint2 *>(fix)[tid];
            if (qpos >= 0 && own) {
                const size_t j = (size_t)tr * a.n + (size_t)qpos * a.vns_pos + (size_t)tid;
                uint2 vv;
                vv.x = rowP;
                vv.y = (rowR >> 16) | (c.y & 0xFFFF0000u);
                reinterpret_cast<uint2 *>(a.vn_adj16)[j] = vv;
            }
            rowP = rowQ;
            rowQ = (rowR & 0xFFFFu) | (c.x & 0xFFFF0000u);
            rowR = (c.x & 0xFFFFu) | (c.y << 16);
            if constexpr (CNMODE != 0) {
                uint2 *dst = reinterpret_cast<uint2 *>(a.cn_adj16) + ((size_t)tr * a.D + pp) * (size_t)(S >> 2);
                const uint2 *src = reinterpret_cast<const uint2 *>(stage);
                if (own) dst[tid] = src[tid];
            }
        }
        STAMP(4);                                                       // B: scan + copy-out
        __syncthreads();
        STAMP(5);                                                       // barrier wait
        // ================================================ phase C ================================================
        if (live && wsum[18 + (p & 1)] == 0u) {
            // exclusive prefix of the 16 wave totals, in lanes 0..15 of every wave; a key's wave is its top four bits
            const uint32_t wt = lane < kWaves ? wsum[lane] : 0u;
            uint32_t wi = wt;
            wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x111, 0xF, 0xF, false);
            wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x112, 0xF, 0xF, false);
            wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x114, 0xF, 0xF, false);
            wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x118, 0xF, 0xF, false);
            const uint32_t wex = wi - wt;
            uint32_t h[E], wb[E], rk[E], g0a[E], cnta[E];
#pragma unroll
            for (int e = 0; e < E; e++) h[e] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(hc) + ((key[e] >> KSHIFT) & ~3u));
#pragma unroll
            for (int e = 0; e < E; e++) wb[e] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((key[e] >> 26) & 0x3Cu), (int)wex);
            bool st[E];
            bool anyst = false;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t t1 = h[e] >> ((key[e] >> (KSHIFT - 2)) & 12u);   // [.. | incl | excl]
                const uint32_t excl = t1 & 15u, incl = (t1 >> 4) & 15u;
                g0a[e] = (h[e] >> 20) + wb[e] + excl;
                cnta[e] = incl - excl;
                rk[e] = g0a[e] + slot[e];
                st[e] = own && ((g0a[e] ^ (g0a[e] + cnta[e] - 1u)) >> DC_SHIFT) != 0u;
                anyst = anyst || st[e];
            }
            if (own) {
                uint2 vv;                                                // provisional CN ids (final unless on the worklist)
                vv.x = (rk[0] >> DC_SHIFT) | ((rk[1] >> DC_SHIFT) << 16);
                vv.y = (rk[2] >> DC_SHIFT) | ((rk[3] >> DC_SHIFT) << 16);
                reinterpret_cast<uint2 *>(fix)[tid] = vv;
            }
            // chain ends: sockets of VN positions outside [0, L) stand for no VN (BPF:1703-1716)
            const bool edge = p < DV - 1 || p >= a.L;
            if constexpr (CNMODE != 0) {
                if (own) {
                    uint32_t ent[E];
                    ent[0] = ent0;
#pragma unroll
                    for (int u = 1; u < 4; u++) ent[u] = CNMODE == 1 ? ent[u - 1] - vV : ent[u - 1] + 1u;
                    if (!edge) {
#pragma unroll
                        for (int u = 0; u < 4; u++) stage[rk[u]] = (uint16_t)ent[u];
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; u++) stage[rk[u]] = (unsigned)(p - u) < (unsigned)a.L ? (uint16_t)ent[u] : (uint16_t)0xFFFFu;
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(anyst)) {
                // the wave's straddlers take consecutive worklist entries: one LDS atomic per wave, no select chains
                unsigned long long vote[E];
                int total = 0;
#pragma unroll
                for (int e = 0; e < E; e++) { vote[e] = __builtin_amdgcn_ballot_w64(st[e]); total += __builtin_popcountll(vote[e]); }
                int base = 0;
                if (lane == 0) base = atomicAdd(reinterpret_cast<int *>(&wsum[16]), total);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if (st[e]) {
                        const int w = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(vote[e] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote[e], 0u));
                        const uint32_t pk = ((key[e] & LOWMASK) << SBITS) | (uint32_t)(4 * tid + e);
                        gpk[g0a[e] + slot[e]] = pk;
                        const uint32_t none = CNMODE != 0 && edge && !((unsigned)(p - e) < (unsigned)a.L) ? 1u : 0u;
                        if (w < kWorkCap) { wl[2 * w] = g0a[e] | (cnta[e] << 16) | (slot[e] << 20) | (none << 24); wl[2 * w + 1] = pk; }
                    }
                    base += __builtin_popcountll(vote[e]);
                }
            }
        }
        if constexpr (CNMODE == 1) ent0 += vV;                           // the VN of socket 0 moves one position on
        STAMP(6);                                                       // C: classify
        __syncthreads();
        STAMP(7);                                                       // barrier wait
    }

    // ---- channel: 32 VNs per output word, 8 Philox calls
    uint32_t *chan = a.chan + (size_t)tr * a.nw;
    for (int w = tid; w < a.nw; w += kThreads) {
        uint32_t word = 0;
        uint32_t c_lo = a.seed_lo, c_hi = a.seed_hi;
        asm volatile("" : "+s"(c_lo), "+s"(c_hi));
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint32_t r[4];
            philox4x32_10((uint32_t)(w * 8 + c), 0x80000000u, t_lo, t_hi, c_lo, c_hi, r);
#pragma unroll
            for (int u = 0; u < 4; u++) word |= (uint32_t)((r[u] >> 1) < a.thresh) << (c * 4 + u);
        }
        const int j0 = w * 32;
        if (j0 + 32 > a.n) word &= (1u << (a.n - j0)) - 1u;
        for (int d = 0; d < a.ndoped; d++) {                            // doped positions are never erased (BPF:1566-1573)
            const int lo = max(a.doped[d] * a.vns_pos, j0) - j0, hi = min((a.doped[d] + 1) * a.vns_pos, j0 + 32) - j0;
            if (lo < hi) word &= ~(((hi - lo) == 32 ? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u)) << lo);
        }
        chan[w] = word;
    }
    STAMP(8);                                                           // channel
    STAMP_FLUSH();
}

int launch_v2(const char *who, int cnmode, const scldpc_code_params *p, uint64_t seed, uint64_t trial0, int32_t ntrials,
              double eps, int32_t ndoped, const int32_t *doped_positions, uint16_t *d_vn_adj16, uint16_t *d_table,
              uint32_t *d_chan_bits, void *stream)
{
    if (ntrials < 0 || (ntrials > 0 && (!d_vn_adj16 || !d_chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: null buffer or negative ntrials", who);
    if (ndoped < 0 || ndoped > kMaxDoped || (ndoped > 0 && !doped_positions))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: 0 <= ndoped <= %d", who, kMaxDoped);
    if (!(eps >= 0.0 && eps <= 1.0))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: eps=%g outside [0,1]", who, eps);
    if (ntrials == 0) return SCLDPC_OK;
    if (!d_table) cnmode = 0;

    S2Args a{};
    a.L = p->L; a.cns_pos = p->cns_pos; a.vns_pos = p->vns_pos;
    a.n = scldpc::n_of(p); a.S = p->cns_pos * p->dc; a.D = p->L + p->dv - 1; a.nw = scldpc::nw_of(p);
    int lg = 10;                                    // nb = power of two >= max(S, kThreads): the histogram of sampler.hip
    while ((1 << lg) < a.S) lg++;
    a.nb = 1 << lg; a.shift = 32 - lg; a.sbits = lg;
    a.ndoped = ndoped;
    for (int d = 0; d < ndoped; d++) {
        if (doped_positions[d] < 0 || doped_positions[d] >= p->L)
            return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "doped position %d outside [0,%d)", doped_positions[d], p->L);
        a.doped[d] = doped_positions[d];
    }
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.trial0 = trial0;
    {   // erased iff r/RAND_MAX < eps with r = 31-bit draw  ⇔  r < ceil(eps * RAND_MAX)   (BPF:370,1554-1562)
        const double x = eps * 2147483647.0;
        double c = (double)(uint64_t)x;
        if (c < x) c += 1.0;
        a.thresh = (uint32_t)c;
    }
    // Third generation (one Philox call per thread: at most 4096 sockets per position; two histogram buffers): measured equal to
    // the second within +-4 % at N = 1000 (faster without the CN table and at N <= 512, slower with it: DESIGN.md §5), so
    // it is opt-in (SCLDPC_SAMPLER_GEN=3) and the second generation stays the default.
    const char *gen_env = getenv("SCLDPC_SAMPLER_GEN");
    const bool v3 = a.nb <= 4 * kThreads && gen_env && atoi(gen_env) == 3;
    int off = ((v3 ? 2 : 1) * a.nb + 3) & ~3;
    a.off_gpk = off;   off += (a.S + 3) & ~3;
    a.off_fix = off;   off += (a.S / 2 + 3) & ~3;           // S uint16
    a.off_stage = off; off += cnmode ? (a.S / 2 + 3) & ~3 : 0;
    a.off_wsum = off;  off += 32 + 2 * kWorkCap;
    size_t lds_bytes = 4u * (size_t)off;
    lds_bytes = std::min(lds_bytes + scldpc::debug_lds_pad("SAMPLER"), std::max(lds_bytes, (size_t)scldpc::kMaxLdsBytes));
    if (lds_bytes > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: %zu bytes of LDS per trial", who, lds_bytes);
    a.vn_adj16 = d_vn_adj16; a.cn_adj16 = d_table; a.chan = d_chan_bits;
    a.ntrials = ntrials;
    a.force_exact = -1;
    if (const char *v = getenv("SCLDPC_DEBUG_SAMPLER_EXACT_POS")) a.force_exact = atoi(v);     // diagnostics / tests only

    using Kern = void (*)(const S2Args);
    static const Kern table[4][3] = {
        {sample_philox_v2_kernel<1, 1, 0>, sample_philox_v2_kernel<1, 1, 1>, sample_philox_v2_kernel<1, 1, 2>},
        {sample_philox_v2_kernel<1, 2, 0>, sample_philox_v2_kernel<1, 2, 1>, sample_philox_v2_kernel<1, 2, 2>},
        {sample_philox_v2_kernel<1, 4, 0>, sample_philox_v2_kernel<1, 4, 1>, sample_philox_v2_kernel<1, 4, 2>},
        {sample_philox_v2_kernel<2, 8, 0>, sample_philox_v2_kernel<2, 8, 1>, sample_philox_v2_kernel<2, 8, 2>},
    };
    const int rows = a.nb / kThreads;                       // 1, 2, 4 (one Philox call per thread) or 8 (two)
    static const Kern table3[3][3] = {
        {sample_philox_v3_kernel<10, 0>, sample_philox_v3_kernel<10, 1>, sample_philox_v3_kernel<10, 2>},
        {sample_philox_v3_kernel<11, 0>, sample_philox_v3_kernel<11, 1>, sample_philox_v3_kernel<11, 2>},
        {sample_philox_v3_kernel<12, 0>, sample_philox_v3_kernel<12, 1>, sample_philox_v3_kernel<12, 2>},
    };
    const int grid = v3 ? ntrials : scldpc::debug_grid("SAMPLER", ntrials);
    Kern kern = v3 ? table3[lg - 10][cnmode] : table[rows == 1 ? 0 : rows == 2 ? 1 : rows == 4 ? 2 : 3][cnmode];
    if (grid < ntrials && rows == 4 && cnmode == 1) kern = sample_philox_v2_kernel<1, 4, 1, true>;     // diagnostics: persistent launch
    else if (grid < ntrials) return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "SCLDPC_DEBUG_GRID_SAMPLER: only the <1,4,1> kernel has a persistent form");
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

bool v2_shape(const scldpc_code_params *p)
{
    if (scldpc::check_params(p)) return false;
    const int S = p->cns_pos * p->dc;
    return p->dv == 4 && p->dc == 8 && (S & 3) == 0 && S <= 8192 && p->vns_pos * p->dv == S;
}

}  // namespace

// 1 when scldpc_sample_philox_device_cn16 takes this ensemble, else 0 (callers then use scldpc_sample_philox_device_adj16)
extern "C" int scldpc_sample_philox_cn16_supported(const scldpc_code_params *p)
{
    return v2_shape(p) && scldpc::n_of(p) < 65535;
}

// 1 when scldpc_sample_philox_device_sock16 takes this ensemble (any chain length: the table holds sockets, not VNs)
extern "C" int scldpc_sample_philox_sock16_supported(const scldpc_code_params *p)
{
    return v2_shape(p);
}

extern "C" int scldpc_sample_philox_device_cn16(const scldpc_code_params *p, uint64_t seed, uint64_t trial0,
                                                int32_t ntrials, double eps, int32_t ndoped,
                                                const int32_t *doped_positions, uint16_t *d_vn_adj16,
                                                uint16_t *d_cn_adj16, uint32_t *d_chan_bits, void *stream)
{
    const char *who = "scldpc_sample_philox_device_cn16";
    if (int rc = scldpc::check_params(p)) return rc;
    if (!scldpc_sample_philox_cn16_supported(p))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: takes dv = 4, dc = 8, at most 8192 sockets per position and "
                                 "fewer than 65535 VNs (got dv=%d dc=%d cns_pos=%d n=%d)", who, p->dv, p->dc, p->cns_pos,
                                 scldpc::n_of(p));
    return launch_v2(who, 1, p, seed, trial0, ntrials, eps, ndoped, doped_positions, d_vn_adj16, d_cn_adj16, d_chan_bits,
                     stream);
}

extern "C" int scldpc_sample_philox_device_sock16(const scldpc_code_params *p, uint64_t seed, uint64_t trial0,
                                                  int32_t ntrials, double eps, int32_t ndoped,
                                                  const int32_t *doped_positions, uint16_t *d_vn_adj16,
                                                  uint16_t *d_cn_sock16, uint32_t *d_chan_bits, void *stream)
{
    const char *who = "scldpc_sample_philox_device_sock16";
    if (int rc = scldpc::check_params(p)) return rc;
    if (!scldpc_sample_philox_sock16_supported(p))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: takes dv = 4, dc = 8 and at most 8192 sockets per position "
                                 "(got dv=%d dc=%d cns_pos=%d)", who, p->dv, p->dc, p->cns_pos);
    return launch_v2(who, 2, p, seed, trial0, ntrials, eps, ndoped, doped_positions, d_vn_adj16, d_cn_sock16, d_chan_bits,
                     stream);
}
