// Streaming (circular-buffer) window decoding of a doped SC-LDPC stream — gfx950 kernels.
//
// Replaces the body of main_streaming (BPF:1934-2054; BPF = simulators_sc_ldpc/bp_decoding/
// SC_LDPC_Simulator_BPDecoder_BEC_full_BP_LimIter_OlmosRandomEnsemble.c, a mode the shipped source compiles out
// with `#undef CIRCULAR`, BPF:33-34): per step
//     count the generated bits / blocks of the positions being decided (BPF:2017-2028),
//     decodeBP_SW_circular(pos)      (BPF:1403-1500: classical window, CNs [pos, pos+W), VNs [pos-ms, pos+W),
//                                     flooding to the window's fixpoint, decision on position pos-ms,
//                                     size-2 stopping-set expurgation of position pos-2dv+1),
//     generate_stream_pos(gen_pos++) (BPF:1927-1932: shuffle CN position gen_pos+dv-1, wire VN position gen_pos,
//                                     draw its channel unless it is doped, reset its messages).
// One workgroup = one independent stream; the unit of work is one decoded position.  Two kernels share a stream's blob:
//
//  * GENERATE (1024 threads, the ranking's LDS: two streams per CU) draws a chunk of positions ahead: the socket permutation
//    of a CN position by ranking Philox keys (as sampler.hip), the VN -> CN rows of a VN position, its channel bits, and —
//    round 3 — the CN -> socket rows of the CN position (every CN's dc sockets s = dv*t + i = edge i of VN t of position
//    CNpos - i), staged in the LDS the ranking has finished with and written out whole.
//  * DECODE (256 threads) keeps the window's state — and nothing else — in LDS, as sw_ring.hip does for the square window:
//    4-bit CN counts of positions [pos-2dv+1, pos+W+dv-2] and the S bits (what the CNs still see as erased) of VN positions
//    [pos-2dv+1, pos+W-1].  A VN position ENTERS when the window first reaches it (its channel bits are read, its erased VNs
//    count themselves into their dv CN positions); a CN whose count is one finds its lone erased neighbour in its
//    CN -> socket row as the socket whose S bit is still set.  Until round 3 the CN state ([cnt | sum of ids] words), the S
//    bits and the per-position counts lived in the blob and every release made six round trips to the L2 / memory (row,
//    S-bit claim, dv returning atomics): 14 500 requests per decoded position at N = 5000, 40 G requests/s over 2048
//    streams — the memory system's random-request ceiling, not latency, was the bound (profiles/r03_stream_*.txt).  Now a
//    release makes two (CN row, VN row), all atomics are LDS atomics, and generation makes none.
//
// Generating position g earlier than main_streaming does (it alternates decode / generate, BPF:2015-2046) changes nothing the
// decoder can see: decodeBP_SW_circular(pos) touches positions pos-2dv+1 .. pos+W+dv-2 only, and a chunk is kept short
// enough that no ring slot still in use is overwritten.  Window semantics: only CNs of the window fire
// (CNs to the right have never been updated and send erasures; CNs to the left are frozen and, with exact counts, have
// nothing left to say), every VN a window CN is left with lies inside the VN window.  A window reaches its fixpoint before
// the next one opens, so the frontier of a new window is just the degree-1 CNs of the CN position that entered it.
// Sampling is keyed like sampler.hip: permutation of CN position c = rank of the Philox words with counter
// (socket>>2, c, stream id), channel of VN position q = counter (t>>2, 2^31 | q, stream id).
#include "common.h"
#include "kernel_util.h"
#include "philox.h"
#include <algorithm>
#include <cstdlib>

namespace {

using namespace scldpc_dev;

constexpr int kGenThreads = 1024, kDecThreads = 256, kMaxDoped = 32, kMaxL = 256;
constexpr int kQCap = 512;                          // frontier-queue entries (an overflow falls back to a scan of the window's CNs)
constexpr int kScratch = 2048;                       // GENERATE: keys of straddling buckets ordered at a time (beyond: the fallback ranking)
constexpr int kFrozen = 8;                          // slots of the frozen-position rings in the blob (> 2dv - 1 - (dv - 1) positions)
enum { C_NE = 0, C_BE, C_EE, C_BEE, C_GB, C_GBL, C_GBE, C_GBLE, C_POS, C_GEN, C_NCOUNT = 16 };
enum { S_PUSH = 0, S_OVF = 3, S_REM = 6, S_ACC = 9, S_WL = 10, S_BAD = 11, S_NSCAL = 16 };

struct StateLayout {        // byte offsets inside one stream's blob
    size_t adj, cnsock, inter, sbits, ring_cnt, ring_s, fz_cnt, fz_s, poscnt, gkey, wlist, tslg, counters, total;
    int wpp;                // 32-bit words of S bits per position
    int Cw;                 // 32-bit words of count nibbles per CN position
    int R, RV;              // ring slots: CN positions / VN positions the decoder keeps in LDS
    int Lp;                 // L rounded up to four (per-slot counts in LDS)
};

struct Args {
    int dv, dc, L, C, V, S, W, nb, shift, lgchunk, dc_shift, npos;
    int gen_ahead;              // GENERATE: until gen == pos + L/2 + gen_ahead (and the first L/2 positions of a new stream)
    int force_wide;             // diagnostics / tests: 1 = rank every position by the 16-bit-counter fallback, 2 = and report its overflow
    int wlcap;                  // GENERATE, fused ranking: the list of straddling buckets holds wlcap / 2 entries (<= kGenThreads: one per lane)
    int ndoped, doped[kMaxDoped];
    uint32_t seed_lo, seed_hi, thresh;
    unsigned long long sid0;
    StateLayout lay;
    // same-input mode (scldpc_stream_run_device_inputs): per-position inputs replayed on the host from the reference's
    // own glibc stream instead of Philox draws — ext_inter uint16 [nstreams][ext_npos + dv - 1][S] = CN-local id of
    // every socket of CN position c (fill_interleaver_pos, BPF:1763-1787), ext_chan uint32 [nstreams][ext_npos][wpp] =
    // erasure bits of VN position g (generate_channel_doped_circular, BPF:1621-1654)
    const uint16_t *ext_inter;
    const uint32_t *ext_chan;
    int ext_npos;
    long long ext_pos0;         // first generated position the two arrays hold
    char *state;
    long long *counters_out;    // [nstreams][10]
    int32_t *trace;             // optional [nstreams][trace_stride][10]; this launch writes rows trace_off .. trace_off + npos - 1
    int trace_stride, trace_off;
};

using scldpc_dev::philox4x32_10;

// a value every lane holds alike (loaded by a vector load), moved to scalar registers: what is derived from it — buffer
// slots, row pointers, loop bounds — then costs scalar arithmetic and no vector registers
__device__ __forceinline__ long long uniform64(long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ bool position_is_doped(const Args &a, long long pos)     // BPF:1589-1612
{
    if (a.ndoped == 0) return false;
    const int left = a.doped[0], period = a.doped[a.ndoped - 1] + 1, m = (int)(pos % period);
    if (m < left) return false;
    for (int i = 0; i < a.ndoped; i++) if (a.doped[i] == m) return true;
    return false;
}

// =================================================== GENERATE ============================================================
// LDS, FUSED (at most 8 Philox calls per thread, everything within a CU's LDS): [scan scratch | scalars | stage: S sockets of
// 2 bytes | hist: a.nb / 2 words | the straddlers' worklist]; once the ranks are final, what follows the stage holds the
// socket -> CN row (S entries of 2 bytes).  Otherwise [scan scratch | scalars | hist | fill counters of cn_rows] and the stage
// lies over hist.  78 KB at N = 5000: two workgroups per CU (64 VGPRs, 72 SGPRs).
template <bool FUSED>
__device__ __forceinline__ void stream_gen_body(const Args &a)
{
    constexpr int kThreads = kGenThreads, kWaves = kThreads / 64;
    extern __shared__ uint32_t lds[];
    uint32_t *wsum = lds;                                               // scan scratch
    int *scal = reinterpret_cast<int *>(wsum + 32);
    uint32_t *base = reinterpret_cast<uint32_t *>(scal + S_NSCAL);
    // a.nb / 2 words: four nibble-wide (fused) or two 16-bit (wide) bucket counters each
    uint32_t *hist = FUSED ? base + ((a.S + 1) & ~1) / 2 : base;
    uint32_t *aux = FUSED ? base : hist + a.nb / 2;                     // FUSED: the stage; else the fill counters

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L, C = a.C, V = a.V, S = a.S, dv = a.dv, wpp = a.lay.wpp;
    char *st = a.state + (size_t)blockIdx.x * a.lay.total;
    // the blob's arrays, addressed from `st` where they are used: kept as seven pointers they would sit in scalar registers the
    // kernel does not have (72 per wave for two workgroups per CU) and end up in scratch memory
    auto blob = [&](size_t off) { char *p = st; asm volatile("" : "+s"(p)); return p + off; };
    auto adj_p = [&] { return reinterpret_cast<uint16_t *>(blob(a.lay.adj)); };          // [L][V][dv] position-local CN ids
    auto cnsock_p = [&] { return reinterpret_cast<uint16_t *>(blob(a.lay.cnsock)); };    // [L][C][dc] sockets of every CN
    auto inter_p = [&] { return reinterpret_cast<uint16_t *>(blob(a.lay.inter)); };      // [dv][S] CN-local id of socket (at tp(socket): by edge, then VN), by CN position % dv
    auto Sb_p = [&] { return reinterpret_cast<uint32_t *>(blob(a.lay.sbits)); };         // [L][wpp] channel bits of generated positions
    auto gkey_p = [&] { return reinterpret_cast<uint2 *>(blob(a.lay.gkey)); };           // [S] (key, socket) of straddling buckets' keys, by rank slot (rank_wide)
    auto wlist_p = [&] { return reinterpret_cast<uint2 *>(blob(a.lay.wlist)); };         // [S] the same keys as a dense list: (key, socket | first rank << 16)
    auto cnt64_p = [&] { return reinterpret_cast<long long *>(blob(a.lay.counters)); };
    // a socket -> CN row is kept by edge: entry of socket s = dv*t + i at i * V + t, so that the wiring of a VN position reads
    // V consecutive entries of each of its dv rows (a quarter of a row each) instead of every dv-th entry of whole rows
    const uint32_t S4 = (uint32_t)(a.S >> 2);
    auto tp = [&](uint32_t sck) { return (sck & 3u) * S4 + (sck >> 2); };
    const unsigned long long sid = a.sid0 + blockIdx.x;
    const uint32_t s_lo = (uint32_t)sid, s_hi = (uint32_t)(sid >> 32);

    STAMP_DECL
    if (tid < S_NSCAL) scal[tid] = 0;
    __syncthreads();
    const long long pos = uniform64(cnt64_p()[C_POS]);
    long long gen = uniform64(cnt64_p()[C_GEN]);
    if (gen < 0) return;                                    // a stream marked unusable (see rank_wide) stays so

    // ---- socket permutation of CN position cpos → inter[cpos % dv] (fill_interleaver_pos, BPF:1763-1787) ----
    // Ranking with 16-bit bucket counters (a.nb buckets, two per word, arrival slots as bytes in the blob): the fallback of
    // rank_nib below for a position in which sixteen keys meet in one of its fine buckets — never on real draws.
    auto rank_wide = [&](long long cpos) {
        uint8_t *tsl = reinterpret_cast<uint8_t *>(blob(a.lay.tslg));
        uint2 *gkey = gkey_p(), *wlist = wlist_p();
        // Keys are never stored: Philox is pure VALU, so the three passes (count, classify, rank the straddlers) draw them
        // again; what a pass hands to the next lives in LDS (16-bit prefix per bucket, one byte of arrival slot per socket)
        // except the straddling buckets' keys, which are grouped in the stream's blob (3-15 % of the sockets).
        const int ncalls = (S + 3) / 4;
        uint32_t k_lo = a.seed_lo, k_hi = a.seed_hi;
        asm volatile("" : "+s"(k_lo), "+s"(k_hi));          // (no Philox round keys hoisted out of the position loop and spilled)
        for (int b = tid; b < a.nb / 2; b += kThreads) hist[b] = 0;
        if (tid == 0) scal[S_WL] = 0;
        __syncthreads();
        uint32_t crowded = 0;
        for (int q = tid; q < ncalls; q += kThreads) {
            uint32_t r[4];
            philox4x32_10((uint32_t)q, (uint32_t)cpos, s_lo, s_hi, k_lo, k_hi, r);
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
        if (crowded >= 256u || a.force_wide == 2) scal[S_BAD] = 1;               // arrival slots are kept in a byte (a bucket holds 1-4 keys on average): the stream is
                                                            // marked unusable (gen < 0) instead of being ranked wrongly — 256 of 2^16 keys in one of >= 1024 buckets
        __syncthreads();
        STAMP(7);
        // wave w scans words [w, w + 1) * nb / 32 of two counters, 64 at a time: exclusive prefix inside the chunk, then (second
        // barrier) plus the chunks before it — every bucket's first rank, 16 bits
        {
            const int cw = a.nb / 2 / kWaves, w0 = wave * cw;
            uint32_t carry = 0;
            for (int i0 = 0; i0 < cw; i0 += 64) {
                const bool on = i0 + lane < cw;
                const uint32_t v = on ? hist[w0 + i0 + lane] : 0u, ps = (v & 0xFFFFu) + (v >> 16);
                const uint32_t inc = wave_inclusive_scan(ps), ex = carry + inc - ps;
                if (on) hist[w0 + i0 + lane] = (ex & 0xFFFFu) | ((ex + (v & 0xFFFFu)) << 16);
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
            if (lane == 0) wsum[wave] = carry;
            __syncthreads();
            const uint32_t t = lane < kWaves ? wsum[lane] : 0u;
            const uint32_t tinc = wave_inclusive_scan(t);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(tinc - t), wave);
            for (int i0 = 0; i0 < cw; i0 += 64)
                if (i0 + lane < cw) {                       // (a first rank of 65536 = S wraps to 0: see bucket_end)
                    const uint32_t w = hist[w0 + i0 + lane];
                    hist[w0 + i0 + lane] = ((w + base) & 0xFFFFu) | (((w >> 16) + base) << 16);
                }
        }
        __syncthreads();
        STAMP(8);
        auto bucket_base = [&](uint32_t b) -> uint32_t {
            return b >= (uint32_t)a.nb ? (uint32_t)S : (hist[b >> 1] >> ((b & 1u) * 16u)) & 0xFFFFu;
        };
        // one past the last rank of a non-empty bucket that starts at g0: the next bucket's first rank, which as 16 bits
        // reads 0 instead of 65536 when S = 65536 and only empty buckets follow
        auto bucket_end = [&](uint32_t b, uint32_t g0) -> uint32_t {
            const uint32_t g1 = bucket_base(b + 1u);
            return g1 < g0 ? g1 + 0x10000u : g1;
        };
        // CN = rank / dc, so a bucket whose ranks [g0, g1) lie inside one block of dc ranks gives all its keys the same CN
        // whatever their order: only the keys of buckets that straddle a multiple of dc are grouped and compared.
        auto straddles = [&](uint32_t g0, uint32_t g1) {
            return g1 - g0 > 1u && (a.dc_shift >= 0 ? (g0 >> a.dc_shift) != ((g1 - 1u) >> a.dc_shift)
                                                    : g0 / (uint32_t)a.dc != (g1 - 1u) / (uint32_t)a.dc);
        };
        auto cn_of = [&](uint32_t rank) { return (uint16_t)(a.dc_shift >= 0 ? rank >> a.dc_shift : rank / (uint32_t)a.dc); };
        uint16_t *dst = inter_p() + (size_t)(cpos % dv) * S;
        for (int q = tid; q < ncalls; q += kThreads) {
            uint32_t r[4];
            philox4x32_10((uint32_t)q, (uint32_t)cpos, s_lo, s_hi, k_lo, k_hi, r);
            uint32_t c4[4] = {0, 0, 0, 0};                  // (a straddler's entry is written by the third pass)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = q * 4 + u;
                if (s >= S) continue;
                const uint32_t k = r[u], b = k >> a.shift, g0 = bucket_base(b), g1 = bucket_end(b, g0);
                if (!straddles(g0, g1)) { c4[u] = cn_of(g0); continue; }
                const uint32_t at = g0 + tsl[s];
                gkey[at] = make_uint2(k, (uint32_t)s);
                wlist[atomicAdd(&scal[S_WL], 1)] = make_uint2(k, (uint32_t)s | (g0 << 16));
            }
            for (int u = 0; u < 4; u++) if (q * 4 + u < S) dst[tp((uint32_t)(q * 4 + u))] = (uint16_t)c4[u];
        }
        __syncthreads();
        STAMP(9);
        // the straddlers as a dense list: a wave that met them where they stand would stop at nearly every key for a
        // trip to the L2 (14 % of the lanes, 20 keys per lane: 20 % of a position's time); here every lane has one
        {
            const int nwl = scal[S_WL];
            for (int w = tid; w < nwl; w += kThreads) {
                const uint2 e = wlist[w];
                const uint32_t k = e.x, s = e.y & 0xFFFFu, g0 = e.y >> 16, g1 = bucket_end(k >> a.shift, g0);
                // rank among the bucket mates: their records are fetched four at a time (independent loads in flight),
                // a key's own record compares false with itself
                uint32_t rank = g0;
                for (uint32_t g = g0; g < g1; g += 4) {
                    uint2 m[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) m[i] = g + i < g1 ? gkey[g + i] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
                    for (int i = 0; i < 4; i++) rank += (m[i].x < k) || (m[i].x == k && m[i].y < s);
                }
                dst[tp(s)] = cn_of(rank);
            }
        }
        __syncthreads();
    };


    // Round 3, FUSED: ranking and both tables of the position in LDS.  Four nibble-wide counters per word (2 * a.nb buckets)
    // plus the word's 16-bit first rank; a key's arrival slot in its bucket stays in a register of the thread that drew it (one
    // nibble per key).  first rank + arrival slot is a rank slot of the key's own, so the socket goes straight to
    // stage[rank slot]; only the keys of buckets that straddle two CNs (7.6 % at N = 5000) must be ordered: the buckets are
    // listed by their first arrivals, and a lane per bucket draws its keys again from their sockets (Philox is pure
    // arithmetic), orders them and puts the sockets back in rank order.  The stage then IS the CN -> socket rows (CN = rank /
    // dc), and its inverse, built over the counters, is the socket -> CN row: both leave for the blob as whole lines, nothing
    // else goes through global memory.  Returns false (for every thread, nothing usable written) when a bucket met a sixteenth
    // key or a list overflowed: the caller ranks the position again with rank_wide + cn_rows.
    auto rank_fused = [&](long long cpos) -> bool {
        const int ncalls = (S + 3) / 4, nbw = a.nb / 2, bshift = a.shift - 1;
        uint16_t *stage = reinterpret_cast<uint16_t *>(aux);
        uint32_t *bl = hist + nbw;                               // the straddling buckets: first rank | size << 16 (a.wlcap / 2 entries)
        uint16_t *irow = reinterpret_cast<uint16_t *>(hist);     // the socket -> CN row, once the counters and the worklist are done with
        uint32_t k_lo = a.seed_lo, k_hi = a.seed_hi;
        asm volatile("" : "+s"(k_lo), "+s"(k_hi));          // (no Philox round keys hoisted out of the position loop and spilled)
        for (int b = tid; b < nbw; b += kThreads) hist[b] = 0;
        if (tid == 0) { scal[S_WL] = 0; scal[S_OVF] = 0; scal[S_REM] = 0; }
        __syncthreads();
        // the arrival slots (a nibble each) of the keys this thread draws: call tid + k * kThreads in bits 16 (k & 1) of pk[k >> 1];
        // the loops over k are not unrolled (their trip count is uniform: the selects are scalar), or eight inlined Philox
        // bodies in flight spill half a kilobyte per lane
        uint32_t pk[4] = {0, 0, 0, 0}, crowded = 0;
#pragma unroll 1
        for (int k = 0; k * kThreads < ncalls; k++) {
            const int q = tid + k * kThreads;
            uint32_t mine = 0;
            if (q < ncalls) {
                uint32_t r[4];
                philox4x32_10((uint32_t)q, (uint32_t)cpos, s_lo, s_hi, k_lo, k_hi, r);
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
            mine <<= (k & 1) * 16;
#pragma unroll
            for (int h = 0; h < 4; h++) pk[h] |= (k >> 1) == h ? mine : 0u;
        }
        if (crowded & 16u) scal[S_OVF] = 1;                 // a nibble wrapped
        __syncthreads();
        STAMP(7);
        if (scal[S_OVF] || a.force_wide) { __syncthreads(); return false; }
        // exclusive scan of the bucket counts, bank-conflict free: wave w owns words [w, w + 1) * nbw / 16, its lanes take
        // them 64 at a time; the word's first rank goes into its high half, the waves before it are added in a second pass
        {
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
        STAMP(8);
        auto cn_of = [&](uint32_t rank) { return (uint16_t)(a.dc_shift >= 0 ? rank >> a.dc_shift : rank / (uint32_t)a.dc); };
        auto straddles = [&](uint32_t g0, uint32_t cnt) {
            return cnt > 1u && (a.dc_shift >= 0 ? (g0 >> a.dc_shift) != ((g0 + cnt - 1u) >> a.dc_shift)
                                                : g0 / (uint32_t)a.dc != (g0 + cnt - 1u) / (uint32_t)a.dc);
        };
        // first rank and size of a key's bucket from its word
        auto bucket_of = [&](uint32_t k, uint32_t &g0, uint32_t &cnt) {
            const uint32_t b = k >> bshift, sh = (b & 3u) * 4u, x = hist[b >> 2], below = x & ((1u << sh) - 1u);
            g0 = ((x >> 16) + (below & 0xFu) + ((below >> 4) & 0xFu) + ((below >> 8) & 0xFu)) & 0xFFFFu;
            cnt = (x >> sh) & 0xFu;
        };
        bool spill = false;
#pragma unroll 1
        for (int k = 0; k * kThreads < ncalls; k++) {
            const int q = tid + k * kThreads;
            uint32_t slots = 0;
#pragma unroll
            for (int h = 0; h < 4; h++) slots |= (k >> 1) == h ? pk[h] : 0u;
            slots >>= (k & 1) * 16;
            if (q < ncalls) {
                uint32_t r[4];
                philox4x32_10((uint32_t)q, (uint32_t)cpos, s_lo, s_hi, k_lo, k_hi, r);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int s = q * 4 + u;
                    if (s >= S) continue;
                    uint32_t g0, cnt;
                    bucket_of(r[u], g0, cnt);
                    stage[g0 + ((slots >> (4 * u)) & 15u)] = (uint16_t)s;
                    if (((slots >> (4 * u)) & 15u) != 0u || !straddles(g0, cnt)) continue;
                    const int at = atomicAdd(&scal[S_WL], 1);               // the first key to arrive lists its straddling bucket
                    if (at < a.wlcap / 2) bl[at] = g0 | (cnt << 16); else spill = true;
                }
            }
        }
        if (spill) scal[S_OVF] = 1;
        __syncthreads();
        STAMP(9);
        if (scal[S_OVF]) { __syncthreads(); return false; }
        // the straddling buckets, one per lane: the bucket's keys are drawn again from their sockets — once each — into a
        // scratch over the counters (done with: the list carries first rank and size), ordered there, and the sockets put back
        // in rank order.  A lane owns its bucket's slots of the stage: no other lane reads or writes them.
        {
            const int nbl = scal[S_WL];
            const int scap = min(kScratch, ((nbw * 4) / 6) & ~1);              // 6 bytes an entry, inside the counters' words
            uint32_t *kscr = hist;                                              // [scap] keys
            uint16_t *sscr = reinterpret_cast<uint16_t *>(hist + scap);         // [scap] their sockets
            bool over = false;
            for (int b = tid; b < nbl; b += kThreads) {
                const uint32_t g0 = bl[b] & 0xFFFFu, cnt = bl[b] >> 16;
                const int base = atomicAdd(&scal[S_REM], (int)cnt);
                if (base + (int)cnt > scap) { over = true; continue; }
                for (uint32_t m = 0; m < cnt; m++) {
                    const uint32_t s2 = stage[g0 + m];
                    uint32_t r2[4];
                    philox4x32_10(s2 >> 2, (uint32_t)cpos, s_lo, s_hi, k_lo, k_hi, r2);
                    kscr[base + m] = (s2 & 2u) ? ((s2 & 1u) ? r2[3] : r2[2]) : ((s2 & 1u) ? r2[1] : r2[0]);
                    sscr[base + m] = (uint16_t)s2;
                }
                for (uint32_t m = 0; m < cnt; m++) {
                    const uint32_t km = kscr[base + m], sm = sscr[base + m];
                    uint32_t rank = g0;
                    for (uint32_t m2 = 0; m2 < cnt; m2++) {                     // (a key compares false with itself)
                        const uint32_t k2 = kscr[base + m2], s2 = sscr[base + m2];
                        rank += (k2 < km) || (k2 == km && s2 < sm);
                    }
                    stage[rank] = (uint16_t)sm;
                }
            }
            if (over) scal[S_OVF] = 1;
        }
        __syncthreads();
        if (scal[S_OVF]) { __syncthreads(); return false; }
        // the stage is the position's sockets in rank order = the CN -> socket rows; its inverse is the socket -> CN row, built
        // over the counters (done with) so that both leave for the blob as whole lines
        for (int r = tid; r < S; r += kThreads) irow[tp(stage[r])] = cn_of((uint32_t)r);
        __syncthreads();
        STAMP(4);
        {
            uint16_t *rows = cnsock_p() + (size_t)(cpos % L) * S, *dst = inter_p() + (size_t)(cpos % dv) * S;
            if ((S & 1) == 0) {
                uint32_t *d32 = reinterpret_cast<uint32_t *>(rows), *i32 = reinterpret_cast<uint32_t *>(dst);
                const uint32_t *s32 = reinterpret_cast<const uint32_t *>(stage), *r32 = reinterpret_cast<const uint32_t *>(irow);
                for (int w = tid; w < S / 2; w += kThreads) { d32[w] = s32[w]; i32[w] = r32[w]; }
            } else {
                for (int w = tid; w < S; w += kThreads) { rows[w] = stage[w]; dst[w] = irow[w]; }
            }
        }
        __syncthreads();
        return true;
    };

    // ---- the CN -> socket rows of CN position cpos from its finished socket -> CN row (the fallback ranking, the same-input
    //      mode): staged in LDS (a nibble-wide fill counter per CN hands out the dc places of a row: which place a socket
    //      gets is immaterial, consumers treat a row as a set) and written out whole, `chunk` CNs at a time -------------------
    auto cn_rows = [&](long long cpos) {
        const uint16_t *row = inter_p() + (size_t)(cpos % dv) * S;
        uint16_t *stage = reinterpret_cast<uint16_t *>(FUSED ? aux : hist);
        uint32_t *fill = FUSED ? hist : aux;
        const int chunk = FUSED ? C : std::min(C, (a.nb / a.dc) & ~7);         // CNs per pass: the stage over hist holds a.nb sockets
        uint16_t *dst = cnsock_p() + (size_t)(cpos % L) * C * a.dc;
        for (int c0 = 0; c0 < C; c0 += chunk) {
            const int c1 = std::min(C, c0 + chunk);
            for (int w = tid; w < (chunk + 7) / 8; w += kThreads) fill[w] = 0;
            __syncthreads();
            for (int s = tid; s < S; s += kThreads) {
                const int c = (int)row[tp((uint32_t)s)] - c0;
                if ((unsigned)c < (unsigned)(c1 - c0)) {   // (c: the CN of socket s)
                    const uint32_t sh = (uint32_t)(c & 7) * 4u;
                    const uint32_t k = (atomicAdd(&fill[c >> 3], 1u << sh) >> sh) & 15u;
                    stage[c * a.dc + (int)k] = (uint16_t)s;
                }
            }
            __syncthreads();
            if ((a.dc & 1) == 0) {
                uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + (size_t)c0 * a.dc);
                const uint32_t *s32 = reinterpret_cast<const uint32_t *>(stage);
                for (int w = tid; w < (c1 - c0) * a.dc / 2; w += kThreads) d32[w] = s32[w];
            } else {
                for (int w = tid; w < (c1 - c0) * a.dc; w += kThreads) dst[(size_t)c0 * a.dc + w] = stage[w];
            }
            __syncthreads();
        }
    };

    // socket -> CN row and CN -> socket rows of CN position cpos (every lambda has ONE call site: the kernel holds one copy of
    // each ranking, and what is live across them fits the 64 VGPRs two workgroups per CU allow)
    auto rank_position = [&](long long cpos) {
        bool rows_done = false;
        if (a.ext_inter) {                                  // same-input mode: the permutation was drawn on the host
            const uint16_t *src = a.ext_inter + ((size_t)blockIdx.x * (size_t)(a.ext_npos + dv - 1) + (size_t)(cpos - a.ext_pos0)) * S;
            uint16_t *dst = inter_p() + (size_t)(cpos % dv) * S;
            for (int s = tid; s < S; s += kThreads) dst[tp((uint32_t)s)] = src[s];
            __syncthreads();
        } else {
            if constexpr (FUSED) rows_done = rank_fused(cpos);
            if (!rows_done) rank_wide(cpos);
        }
        if (!rows_done) {
            STAMP(4);
            cn_rows(cpos);
        }
    };

    // ---- generate_stream_pos(g) (BPF:1927-1932): CN position g+dv-1 ranked, VN position g wired, its channel drawn ----------
    auto generate = [&](long long g) {
        rank_position(g + dv - 1);
        if (g < 0) return;                                  // a new stream's CN positions 0 .. dv-2 (initialize_arrays_circular, BPF:1808-1813)
        const int slot = (int)(g % L);
        const bool doped = position_is_doped(a, g);
        uint32_t c_lo = a.seed_lo, c_hi = a.seed_hi;
        asm volatile("" : "+s"(c_lo), "+s"(c_hi));
        // channel (BPF:1621-1654): Philox call w * 8 + c8 gives bits 4 c8 .. 4 c8 + 3 of word w — one call per lane, the eight
        // nibbles of a word meet through three lane exchanges (a word's calls are eight consecutive lanes of one wave)
        uint32_t *Sb = Sb_p() + slot * wpp;
        if (a.ext_chan || doped) {
            for (int w = tid; w < wpp; w += kThreads)
                Sb[w] = a.ext_chan ? a.ext_chan[((size_t)blockIdx.x * a.ext_npos + (size_t)(g - a.ext_pos0)) * wpp + w] : 0u;   // doped: all known
        } else {
            const int n8 = wpp * 8;
            for (int it0 = tid - lane; it0 < n8; it0 += kThreads) {             // wave-uniform trip count
                const int it = it0 + lane;
                uint32_t v = 0;
                if (it < n8) {
                    uint32_t r[4];
                    philox4x32_10((uint32_t)it, 0x80000000u | (uint32_t)g, s_lo, s_hi, c_lo, c_hi, r);
#pragma unroll
                    for (int u = 0; u < 4; u++) v |= (uint32_t)((r[u] >> 1) < a.thresh) << u;
                    v <<= (it & 7) * 4;
                }
                v |= (uint32_t)__shfl_xor((int)v, 1, 64);
                v |= (uint32_t)__shfl_xor((int)v, 2, 64);
                v |= (uint32_t)__shfl_xor((int)v, 4, 64);
                if (it < n8 && (it & 7) == 0) {
                    const int w = it >> 3;
                    if (w * 32 + 32 > V) v &= (1u << (V - w * 32)) - 1u;
                    Sb[w] = v;
                }
            }
        }
        STAMP(5);
        const uint16_t *inter = inter_p();
        uint16_t *vrows = adj_p() + (size_t)slot * V * dv;
        for (int t = tid; t < V; t += kThreads) {                                // wiring (BPF:1841-1854)
            if (dv == 4) {                                  // four independent loads, one 8-byte row store
                uint32_t loc[4];
#pragma unroll
                for (int i = 0; i < 4; i++) loc[i] = inter[(size_t)((g + i) & 3) * S + (size_t)i * S4 + t];
                *reinterpret_cast<uint2 *>(vrows + (size_t)t * 4) = make_uint2(loc[0] | (loc[1] << 16), loc[2] | (loc[3] << 16));
            } else {
                for (int i = 0; i < dv; i++) vrows[(size_t)t * dv + i] = inter[(size_t)((g + i) % dv) * S + tp((uint32_t)(dv * t + i))];
            }
        }
        __syncthreads();
        STAMP(6);
    };

    // a new stream starts with its first dv - 1 CN positions and L/2 positions (BPF:2003-2012); then what the next DECODE
    // launches will have consumed (BPF:2036-2045)
    const long long target = pos + L / 2 + a.gen_ahead;
    for (long long g = (gen == 0 && pos == 0) ? -(long long)(dv - 1) : gen; g < target; g++) generate(g);
    gen = gen > target ? gen : target;
    STAMP_FLUSH();
    __syncthreads();
    if (scal[S_BAD]) gen = -1;
    if (tid == 0) {
        cnt64_p()[C_GEN] = gen;
        if (a.counters_out) a.counters_out[(size_t)blockIdx.x * 10 + 9] = gen;
    }
}

template <bool FUSED>
__global__ __launch_bounds__(kGenThreads, 8) __attribute__((amdgpu_num_sgpr(80))) void stream_gen_kernel(const Args a)
{
    stream_gen_body<FUSED>(a);
}

// ==================================================== DECODE =============================================================
// 256 threads per stream; the window's state in LDS (sw_ring.hip's layout, here for the classical window on a circular
// buffer with unlimited iterations per position).  Ring slots are addressed relative to the decoder's position: every
// position a release can touch while it stands at `pos` lies within [pos - dv + 1, pos + W + dv - 2], less than a ring apart.
// What lies further left is frozen: VN position pos - dv + 1 is decided in this step and CN position pos - dv + 1 is out of
// reach of the next window's VNs, so both leave the LDS at the end of the step for a small ring in the blob, where the
// size-2 stopping-set test of position pos - 2dv + 1 reads them (only when that position still holds erasures).  That keeps
// the LDS at W + 2dv - 2 CN positions and W + dv - 1 VN positions: 50 KB at N = 5000, W = 20 — three streams per CU.
template <int DV>
__device__ __forceinline__ void stream_dec_body(const Args &a)
{
    constexpr int kThreads = kDecThreads;
    extern __shared__ uint32_t lds[];
    const int L = a.L, C = a.C, V = a.V, W = a.W, wpp = a.lay.wpp, Cw = a.lay.Cw, R = a.lay.R, RV = a.lay.RV, dc = a.dc;
    constexpr int ms = DV - 1;
    long long *acc = reinterpret_cast<long long *>(lds);                // the eight running totals: thread 0's alone
    uint32_t *cnt = lds + 16;                                           // [R][Cw] words of 8 count nibbles
    uint32_t *Sr = cnt + R * Cw;                                        // [RV][wpp] S bits
    uint32_t *q0 = Sr + RV * wpp, *q1 = q0 + kQCap;                     // frontier queues: [CN position offset + 2dv | CN]
    int *pos_cnt = reinterpret_cast<int *>(q1 + kQCap);                 // [L] erased VNs per buffer slot (position % L)
    int *scal = pos_cnt + a.lay.Lp;

    const int tid = threadIdx.x, lane = tid & 63;
    char *st = a.state + (size_t)blockIdx.x * a.lay.total;
    const uint16_t *adj = reinterpret_cast<const uint16_t *>(st + a.lay.adj);
    const uint16_t *cnsock = reinterpret_cast<const uint16_t *>(st + a.lay.cnsock);
    const uint32_t *Sb = reinterpret_cast<const uint32_t *>(st + a.lay.sbits);
    uint32_t *ring_cnt = reinterpret_cast<uint32_t *>(st + a.lay.ring_cnt);
    uint32_t *ring_s = reinterpret_cast<uint32_t *>(st + a.lay.ring_s);
    uint32_t *fz_cnt = reinterpret_cast<uint32_t *>(st + a.lay.fz_cnt);            // [kFrozen][Cw] counts of frozen CN positions, by position % kFrozen
    uint32_t *fz_s = reinterpret_cast<uint32_t *>(st + a.lay.fz_s);                // [kFrozen][wpp] S bits of frozen VN positions
    int *pos_cnt_g = reinterpret_cast<int *>(st + a.lay.poscnt);
    long long *cnt64 = reinterpret_cast<long long *>(st + a.lay.counters);

    STAMP_DECL
    long long pos = uniform64(cnt64[C_POS]);
    if (uniform64(cnt64[C_GEN]) < 0) return;                           // generation marked the stream unusable
    // ---- the rings and the per-slot counts come from the blob (zero for a new stream) ------------------------------------
    if (pos == 0) {
        for (int i = tid; i < R * Cw + RV * wpp; i += kThreads) cnt[i] = 0;        // (Sr follows cnt)
        for (int i = tid; i < L; i += kThreads) pos_cnt[i] = 0;
    } else {
        for (int i = tid; i < R * Cw; i += kThreads) cnt[i] = ring_cnt[i];
        for (int i = tid; i < RV * wpp; i += kThreads) Sr[i] = ring_s[i];
        for (int i = tid; i < L; i += kThreads) pos_cnt[i] = pos_cnt_g[i];
    }
    if (tid < S_NSCAL) scal[tid] = 0;
    if (tid == 0) for (int k = C_NE; k <= C_GBLE; k++) acc[k] = cnt64[k];
    __syncthreads();

    // slot of a position in a ring of n slots (p may be negative near the start of a stream: those positions hold nothing)
    auto slot_of = [](long long p, int n) { int r = (int)(p % n); return r < 0 ? r + n : r; };
    int cb = slot_of(pos, R), vb = slot_of(pos, RV), lb = slot_of(pos, L);           // slots of the decoder's own position
    auto cslot = [&](int d) {                                           // word base of CN position pos + d
        int sl = cb + d;
        sl -= sl >= R ? R : 0;
        sl += sl < 0 ? R : 0;
        return sl * Cw;
    };
    auto sslot = [&](int d) {                                           // word base of the S bits of VN position pos + d
        int sl = vb + d;
        sl -= sl >= RV ? RV : 0;
        sl += sl < 0 ? RV : 0;
        return sl * wpp;
    };
    auto lslot = [&](int d) {                                           // buffer slot (position % L) of position pos + d
        int sl = lb + d;
        sl -= sl >= L ? L : 0;
        sl += sl < 0 ? L : 0;
        return sl;
    };

    // ---- VN position pos + d enters the ring: channel bits, per-position count, its erased VNs into their CN positions.
    //      Its right-most CN position is a fresh slot. ----------------------------------------------------------------------
    auto enter = [&](int d) {
        for (int i = tid; i < Cw; i += kThreads) cnt[cslot(d + DV - 1) + i] = 0;
        const int ls = lslot(d), sb = sslot(d);
        int mine = 0;
        for (int w = tid; w < wpp; w += kThreads) {
            const uint32_t x = Sb[ls * wpp + w];
            Sr[sb + w] = x;
            mine += __popc(x);
        }
        mine = wave_sum(mine);
        if (tid == 0) pos_cnt[ls] = 0;
        __syncthreads();
        if (lane == 0 && mine) atomicAdd(&pos_cnt[ls], mine);
        // rows are loaded unconditionally, four per thread in flight (coalesced 8-byte loads), then the erased ones count
        const uint2 *vrow = reinterpret_cast<const uint2 *>(adj) + (size_t)ls * V;
        for (int t0 = tid; t0 < V; t0 += 4 * kThreads) {
            uint2 r[4];
            bool er[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + u * kThreads;
                er[u] = false;
                if (t < V) { r[u] = vrow[t]; er[u] = (Sr[sb + (t >> 5)] >> (t & 31)) & 1u; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!er[u]) continue;
                const uint32_t l[4] = {r[u].x & 0xFFFFu, r[u].x >> 16, r[u].y & 0xFFFFu, r[u].y >> 16};
#pragma unroll
                for (int i = 0; i < DV; i++) atomicAdd(&cnt[cslot(d + i) + (l[i] >> 3)], 1u << ((l[i] & 7) * 4));
            }
        }
        __syncthreads();
    };

    int genq = 0;                                           // queue generation counter
    for (int step = 0; step < a.npos; step++, pos++) {
        STAMP(0);
        const long long pd = pos - ms, pe = pos - 2 * DV + 1;
        if (tid == 0) {                                                         // BPF:2017-2028
            if (pd >= 0 && !position_is_doped(a, pd)) { acc[C_GB] += V; acc[C_GBL] += 1; }
            if (pe >= 0 && !position_is_doped(a, pe)) { acc[C_GBE] += V; acc[C_GBLE] += 1; }
        }
        // the VN window [pos - ms, pos + W) takes in position pos + W - 1 (a new stream: positions 0 .. W-1)
        for (int d = (pos == 0 ? 0 : W - 1); d < W; d++) enter(d);

        // ---- decodeBP_SW_circular(pos): the frontier of a new window = the degree-1 CNs of the position(s) that entered it
        if (tid == 0) { scal[S_PUSH + genq % 3] = 0; scal[S_OVF + genq % 3] = 0; }
        __syncthreads();
        {
            uint32_t *qc = (genq & 1) ? q1 : q0;
            for (int d = (pos == 0 ? 0 : W - 1); d < W; d++) {
                const int cs = cslot(d);
                for (int w0 = (tid >> 6) * 64; w0 < Cw; w0 += kThreads) {
                    const int w = w0 + lane;
                    uint32_t z = 0;
                    if (w < Cw) {
                        const uint32_t y = cnt[cs + w] ^ 0x11111111u;    // nibble == 1  <=>  zero nibble of y
                        z = ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;
                        if (w * 8 + 8 > C) z &= (1u << (4 * (C - w * 8))) - 1u;
                    }
                    const int minen = __popc(z);
                    const int incl = (int)wave_inclusive_scan((uint32_t)minen);
                    const int tot = __builtin_amdgcn_readlane(incl, 63);
                    if (tot == 0) continue;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&scal[S_PUSH + genq % 3], tot);
                    int idx = __builtin_amdgcn_readfirstlane(base) + incl - minen;
                    while (z) {
                        const int k = (__ffs((int)z) - 1) >> 2;
                        z &= z - 1;
                        if (idx < kQCap) qc[idx] = ((uint32_t)(d + 2 * DV) << 16) | (uint32_t)(w * 8 + k); else scal[S_OVF + genq % 3] = 1;
                        idx++;
                    }
                }
            }
        }
        __syncthreads();
        STAMP(1);
        int ncur = min(scal[S_PUSH + genq % 3], kQCap);
        bool rescan = scal[S_OVF + genq % 3] != 0;
        for (;;) {
            uint32_t *qc = (genq & 1) ? q1 : q0, *qn = (genq & 1) ? q0 : q1;
            int *push_cnt = &scal[S_PUSH + (genq + 1) % 3], *push_ovf = &scal[S_OVF + (genq + 1) % 3];
            if (tid == 0) { scal[S_PUSH + (genq + 2) % 3] = 0; scal[S_OVF + (genq + 2) % 3] = 0; }
            // CN l of position pos + d is believed to hold one erased neighbour: release it unless it is frozen.  Two trips
            // to memory: the CN's sockets, then the VN's row (issued before the claim of its S bit); every atomic is in LDS.
            // out[i] = 1 + [CN position offset + 2dv | CN] of edge i if this release left that CN with one erased neighbour
            // inside the CN window.
            auto release = [&](int d, int l, uint32_t (&out)[4]) {
                const uint16_t *row = cnsock + ((size_t)lslot(d) * C + l) * dc;
                uint32_t sk[8];
                if (dc == 8) {
                    const uint4 s4 = *reinterpret_cast<const uint4 *>(row);
                    sk[0] = s4.x & 0xFFFFu; sk[1] = s4.x >> 16; sk[2] = s4.y & 0xFFFFu; sk[3] = s4.y >> 16;
                    sk[4] = s4.z & 0xFFFFu; sk[5] = s4.z >> 16; sk[6] = s4.w & 0xFFFFu; sk[7] = s4.w >> 16;
                }
                int jd = 1 << 20, jt = 0;                                // offset of the neighbour's position from pos
                for (int k0 = 0; k0 < dc; k0 += 8) {
                    if (dc != 8) for (int k = 0; k < 8; k++) sk[k] = k0 + k < dc ? row[k0 + k] : 0xFFFFFFFFu;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (sk[k] == 0xFFFFFFFFu) continue;
                        const int i = (int)(sk[k] % DV), t = (int)(sk[k] / DV), dq = d - i;
                        if (pos + dq < 0) continue;                      // the stream has no VN position before 0
                        if ((Sr[sslot(dq) + (t >> 5)] >> (t & 31)) & 1u) { jd = dq; jt = t; }
                    }
                }
                if (jd == (1 << 20) || jd < -ms) return;                 // none left (released this round) or frozen (BPF:1285-1310)
                const uint2 r = reinterpret_cast<const uint2 *>(adj)[(size_t)lslot(jd) * V + jt];
                const uint32_t bit = 1u << (jt & 31);
                if (!(atomicAnd(&Sr[sslot(jd) + (jt >> 5)], ~bit) & bit)) return;
                atomicSub(&pos_cnt[lslot(jd)], 1);
                const uint32_t ll[4] = {r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16};
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < DV; i++)                             // the dv returning atomics go out back to back
                    o[i] = atomicSub(&cnt[cslot(jd + i) + (ll[i] >> 3)], 1u << ((ll[i] & 7) * 4));
#pragma unroll
                for (int i = 0; i < DV; i++)
                    if (((o[i] >> ((ll[i] & 7) * 4)) & 15u) == 2u && jd + i >= 0 && jd + i < W)    // 2 -> 1 inside the CN window
                        out[i] = 1u + (((uint32_t)(jd + i + 2 * DV) << 16) | ll[i]);
            };
            // a wave appends its lanes' entries behind *push_cnt: one prefix scan + one LDS atomic per wave
            auto append = [&](const uint32_t (&out)[4]) {
                const int mine = (out[0] != 0u) + (out[1] != 0u) + (out[2] != 0u) + (out[3] != 0u);
                const int incl = (int)wave_inclusive_scan((uint32_t)mine);
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                if (tot == 0) return;
                int base = 0;
                if (lane == 0) base = atomicAdd(push_cnt, tot);
                int idx = __builtin_amdgcn_readfirstlane(base) + incl - mine;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (out[i]) { if (idx < kQCap) qn[idx] = out[i] - 1u; else *push_ovf = 1; idx++; }
            };
            if (rescan) {
                // queue overflow: walk every CN of the window (a VN released here may promote further CNs into THIS round —
                // allowed: the window runs to its fixpoint and only the fixpoint is observable)
                for (int d = 0; d < W; d++) {
                    const int cs = cslot(d);
                    for (int w0 = (tid >> 6) * 64; w0 < Cw; w0 += kThreads) {        // wave-uniform trip count (append scans)
                        const int w = w0 + lane;
                        uint32_t z = 0;
                        if (w < Cw) {
                            const uint32_t y = cnt[cs + w] ^ 0x11111111u;
                            z = ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;
                            if (w * 8 + 8 > C) z &= (1u << (4 * (C - w * 8))) - 1u;
                        }
                        while (__any(z != 0u)) {
                            uint32_t out[4] = {0, 0, 0, 0};
                            if (z) {
                                const int k = (__ffs((int)z) - 1) >> 2;
                                z &= z - 1;
                                release(d, w * 8 + k, out);
                            }
                            append(out);
                        }
                    }
                }
            } else {
                for (int k0 = (tid >> 6) * 64; k0 < ncur; k0 += kThreads) {
                    uint32_t out[4] = {0, 0, 0, 0};
                    if (k0 + lane < ncur) release((int)(qc[k0 + lane] >> 16) - 2 * DV, (int)(qc[k0 + lane] & 0xFFFFu), out);
                    append(out);
                }
            }
            __syncthreads();
            rescan = *push_ovf != 0;
            ncur = min(*push_cnt, kQCap);
            genq++;
            if (ncur == 0 && !rescan) break;                                      // nothing left to fire (BPF:1454-1455)
        }
        STAMP(2);
        // decision on position pos - ms (BPF:1445-1449): its S bits are VNerased from now on (frozen)
        const int nep = pd >= 0 ? pos_cnt[lslot(-ms)] : 0;
        if (tid == 0) { acc[C_NE] += nep; if (nep > 0) acc[C_BE] += 1; }           // BPF:1480-1483
        // size-2 stopping-set expurgation of position pos - 2dv + 1 (get_deg_two_ss, BPF:1227-1283, 1485-1497): an erased VN
        // whose dv CNs all hold exactly two erased neighbours, the other one the same VN of the same position each time
        if (pe >= 0) {
            const int de = -(2 * DV - 1), ls = lslot(de);
            const uint32_t *fs = fz_s + (size_t)(pe & (kFrozen - 1)) * wpp;     // frozen since the end of step pe + dv - 1
            int mine = 0;
            if (pos_cnt[ls] > 0) {
                for (int w = tid; w < wpp; w += kThreads) {
                    uint32_t x = fs[w];
                    while (x) {
                        const int b = __ffs((int)x) - 1;
                        x &= x - 1;
                        const int t = w * 32 + b;
                        const uint2 r = reinterpret_cast<const uint2 *>(adj)[(size_t)ls * V + t];
                        const uint32_t l[4] = {r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16};
                        bool pair = true;
#pragma unroll
                        for (int i = 0; i < DV; i++)
                            pair = pair && ((fz_cnt[(size_t)((pe + i) & (kFrozen - 1)) * Cw + (l[i] >> 3)] >> ((l[i] & 7) * 4)) & 15u) == 2u;
                        int partner = -1;
                        for (int i = 0; i < DV && pair; i++) {            // the other erased neighbour, if it is in position pe
                            const uint16_t *row = cnsock + ((size_t)lslot(de + i) * C + l[i]) * dc;
                            int other = -1;
                            for (int k = 0; k < dc; k++) {
                                const uint32_t s = row[k];
                                const int i2 = (int)(s % DV), t2 = (int)(s / DV);
                                if (i2 == i && t2 != t && ((fs[t2 >> 5] >> (t2 & 31)) & 1u)) other = t2;
                            }
                            if (other < 0 || (i > 0 && other != partner)) pair = false;
                            partner = other;
                        }
                        mine += pair ? 0 : 1;
                    }
                }
            }
            const uint32_t tot = wave_inclusive_scan((uint32_t)mine);
            if (lane == 63 && tot) atomicAdd(&scal[S_ACC], (int)tot);
            __syncthreads();
            const int cexp = scal[S_ACC];
            __syncthreads();
            if (tid == 0) scal[S_ACC] = 0;
            if (tid == 0 && cexp > 0) { acc[C_EE] += cexp; acc[C_BEE] += 1; }
        }
        if (a.trace && tid == 0) {
            int32_t *tr = a.trace + ((size_t)blockIdx.x * a.trace_stride + a.trace_off + step) * 10;
            tr[0] = (int32_t)pos; tr[1] = nep;
            tr[2] = (int32_t)acc[C_NE]; tr[3] = (int32_t)acc[C_BE]; tr[4] = (int32_t)acc[C_EE]; tr[5] = (int32_t)acc[C_BEE];
            tr[6] = (int32_t)acc[C_GB]; tr[7] = (int32_t)acc[C_GBL]; tr[8] = (int32_t)acc[C_GBE]; tr[9] = (int32_t)acc[C_GBLE];
        }
        // VN position pos - ms (decided above) and CN position pos - ms (its VNs are all frozen now) leave the LDS
        if (pd >= 0) {
            const int cs = cslot(-ms), sb = sslot(-ms);
            uint32_t *gc = fz_cnt + (size_t)(pd & (kFrozen - 1)) * Cw, *gs = fz_s + (size_t)(pd & (kFrozen - 1)) * wpp;
            for (int i = tid; i < Cw; i += kThreads) gc[i] = cnt[cs + i];
            for (int w = tid; w < wpp; w += kThreads) gs[w] = Sr[sb + w];
        }
        __syncthreads();
        STAMP(3);
        cb = cb + 1 == R ? 0 : cb + 1;                      // the decoder moves on: every relative offset shifts by one
        vb = vb + 1 == RV ? 0 : vb + 1;
        lb = lb + 1 == L ? 0 : lb + 1;
    }
    STAMP_FLUSH();

    // ---- the window's state goes back to the blob until the next launch ---------------------------------------------------
    __syncthreads();
    for (int i = tid; i < R * Cw; i += kThreads) ring_cnt[i] = cnt[i];
    for (int i = tid; i < RV * wpp; i += kThreads) ring_s[i] = Sr[i];
    for (int i = tid; i < L; i += kThreads) pos_cnt_g[i] = pos_cnt[i];
    if (tid == 0) {
        for (int k = C_NE; k <= C_GBLE; k++) cnt64[k] = acc[k];
        cnt64[C_POS] = pos;
        if (a.counters_out) {
            long long *o = a.counters_out + (size_t)blockIdx.x * 10;
            for (int k = C_NE; k <= C_GBLE; k++) o[k] = acc[k];
            o[8] = pos;
        }
    }
}

template <int DV>
__global__ __launch_bounds__(kDecThreads, 4) __attribute__((amdgpu_num_sgpr(96))) void stream_dec_kernel(const Args a)
{
    stream_dec_body<DV>(a);
}

int make_state_layout(const scldpc_code_params *p, int W, StateLayout *lay)
{
    const size_t L = p->L, V = p->vns_pos, C = p->cns_pos, S = (size_t)p->cns_pos * p->dc, dv = p->dv;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    lay->wpp = (int)((V + 31) / 32);
    lay->Cw = (int)((C + 7) / 8);
    lay->R = W + 2 * p->dv - 2;                     // CN positions pos - dv + 1 .. pos + W + dv - 2
    lay->RV = W + p->dv - 1;                        // VN positions pos - dv + 1 .. pos + W - 1
    lay->Lp = (p->L + 3) & ~3;
    lay->adj = take(L * V * dv * 2);
    lay->cnsock = take(L * S * 2);
    lay->inter = take(dv * S * 2);
    lay->sbits = take(L * lay->wpp * 4);
    lay->ring_cnt = take((size_t)lay->R * lay->Cw * 4);
    lay->ring_s = take((size_t)lay->RV * lay->wpp * 4);
    lay->fz_cnt = take((size_t)kFrozen * lay->Cw * 4);
    lay->fz_s = take((size_t)kFrozen * lay->wpp * 4);
    lay->poscnt = take(L * 4);
    lay->gkey = take(S * 8); lay->wlist = take(S * 8);
    lay->tslg = take(S);                            // arrival slots of the 16-bit-counter fallback ranking
    lay->counters = take(C_NCOUNT * 8);
    lay->total = off;
    return 0;
}

size_t dec_lds_bytes(const StateLayout &lay)
{
    return 4u * ((size_t)16 + (size_t)lay.R * lay.Cw + (size_t)lay.RV * lay.wpp + 2 * kQCap + lay.Lp + S_NSCAL);
}

int check_stream(const scldpc_code_params *p, int W, const char *who)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (p->L > kMaxL || p->L < 2 * p->dv)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: buffer length L=%d outside [%d, %d]", who, p->L, 2 * p->dv, kMaxL);
    // the stream is generated L/2 positions ahead (BPF:2001): the window and the CNs of its VNs must exist already
    if (W < 1 || W + p->dv - 1 > p->L / 2)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: need 1 <= W and W + dv - 1 <= L/2 (W=%d, L=%d)", who, W, p->L);
    if (p->dv != 4 || (int64_t)p->cns_pos * p->dc > 65536 || p->dc > 15)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: the streaming kernels take dv = 4, dc <= 15 and at most 65536 "
                                 "sockets per position", who);
    StateLayout lay;
    make_state_layout(p, W, &lay);
    if (dec_lds_bytes(lay) > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: the window's state (%zu bytes) does not fit the LDS", who, dec_lds_bytes(lay));
    return SCLDPC_OK;
}

}  // namespace

extern "C" int64_t scldpc_stream_state_bytes(const scldpc_code_params *p, int32_t W)
{
    if (int rc = check_stream(p, W, "scldpc_stream_state_bytes")) return rc;
    StateLayout lay;
    make_state_layout(p, W, &lay);
    return (int64_t)lay.total;
}

static int stream_run(const scldpc_code_params *p, int32_t nstreams, uint64_t seed, uint64_t stream0,
                      double eps, int32_t W, int32_t ndoped, const int32_t *doped_positions,
                      int32_t npos, void *d_state, int64_t *d_counters, int32_t *d_trace,
                      const uint16_t *d_ext_inter, const uint32_t *d_ext_chan, int32_t ext_npos, int64_t ext_pos0, void *stream)
{
    if (int rc = check_stream(p, W, "scldpc_stream_run_device")) return rc;
    if (nstreams < 0 || npos < 0 || (nstreams > 0 && !d_state))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_stream_run_device: null state or negative count");
    if (ndoped < 0 || ndoped > kMaxDoped || (ndoped > 0 && !doped_positions))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_stream_run_device: 0 <= ndoped <= %d", kMaxDoped);
    if (!(eps >= 0.0 && eps <= 1.0))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_stream_run_device: eps=%g outside [0,1]", eps);
    if (nstreams == 0) return SCLDPC_OK;
    Args a{};
    a.dv = p->dv; a.dc = p->dc; a.L = p->L; a.C = p->cns_pos; a.V = p->vns_pos; a.S = p->cns_pos * p->dc; a.W = W;
    a.npos = npos;
    int lg = 10;
    while ((1 << lg) < a.S && lg < 14) lg++;
    a.nb = 1 << lg; a.shift = 32 - lg; a.lgchunk = lg - 4;
    a.dc_shift = -1;
    for (int k = 0; k < 8; k++) if ((1 << k) == p->dc) a.dc_shift = k;
    a.ndoped = ndoped;
    for (int d = 0; d < ndoped; d++) {
        if (doped_positions[d] < 0 || (d > 0 && doped_positions[d] <= doped_positions[d - 1]))
            return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "doped positions must be non-negative and ascending (BPF:1585-1587)");
        a.doped[d] = doped_positions[d];
    }
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.sid0 = stream0;
    {
        const double x = eps * 2147483647.0;
        double c = (double)(uint64_t)x;
        if (c < x) c += 1.0;
        a.thresh = (uint32_t)c;
    }
    make_state_layout(p, W, &a.lay);
    a.state = static_cast<char *>(d_state); a.counters_out = reinterpret_cast<long long *>(d_counters); a.trace = d_trace;
    a.ext_inter = d_ext_inter; a.ext_chan = d_ext_chan; a.ext_npos = ext_npos; a.ext_pos0 = ext_pos0;
    const size_t lds_dec = dec_lds_bytes(a.lay);
    // GENERATE: the fused ranking (stage of S sockets + worklist beside the counters) where a thread draws at most eight
    // Philox calls per position (S <= 32768) and it all fits a CU; else the counters, the stage over them, fill counters
    const int kc = ((a.S + 3) / 4 + kGenThreads - 1) / kGenThreads;
    a.wlcap = std::min(2 * kGenThreads, std::max(64, (a.S / 8 + 63) & ~63));
    const size_t lds_head = 4u * ((size_t)32 + S_NSCAL + (size_t)a.nb / 2);
    const size_t s_even = (size_t)((a.S + 1) & ~1);
    const size_t lds_fused = 4u * ((size_t)32 + S_NSCAL) + 2u * s_even + std::max(2u * (size_t)a.nb + 2u * (size_t)a.wlcap, 2u * s_even);
    bool fused = kc <= 8 && lds_fused <= (size_t)scldpc::kMaxLdsBytes;
    if (const char *v = getenv("SCLDPC_DEBUG_STREAM_LEGACY")) fused = fused && atoi(v) == 0;     // tests: the other LDS layout
    const size_t lds_gen = fused ? lds_fused : lds_head + 4u * (size_t)(a.C / 8 + 4);
    if (const char *v = getenv("SCLDPC_DEBUG_STREAM_WIDE")) a.force_wide = atoi(v);         // diagnostics / tests only
    if (lds_gen > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_stream_run_device: %zu bytes of LDS per stream", lds_gen);
    void (*gen_kern)(const Args) = fused ? stream_gen_kernel<true> : stream_gen_kernel<false>;
    void (*dec_kern)(const Args) = stream_dec_kernel<4>;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(gen_kern))) return rc_;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(dec_kern))) return rc_;
    // Cycles of GENERATE (run `ahead` positions beyond the reference's lag of L/2) and DECODE (as many positions as are
    // then generated far enough: decodeBP_SW_circular(pos) needs positions up to pos + W + dv - 2).  `ahead` is bounded by the
    // buffer: generating position g re-uses the slots of VN position g - L and CN position g + dv - 1 - L, and the decoder
    // still reads CN position pos - 2dv + 1 (expurgation): ahead <= ceil(L/2) - 3dv + 2.  The call ends as main_streaming
    // leaves a stream: generated = decoded + L/2.
    const int half = p->L / 2, ahead_max = std::max(0, (p->L - half) - 3 * p->dv + 2);
    const hipStream_t hs = static_cast<hipStream_t>(stream);
    a.trace_stride = npos;
    for (int done = 0; done < npos;) {
        const int ahead = std::min(ahead_max, npos - done);
        const int c = std::min(npos - done, ahead + half - W - p->dv + 2);
        a.gen_ahead = ahead; a.npos = 0;
        hipLaunchKernelGGL(gen_kern, dim3(nstreams), dim3(kGenThreads), lds_gen, hs, a);
        a.npos = c; a.trace_off = done;
        hipLaunchKernelGGL(dec_kern, dim3(nstreams), dim3(kDecThreads), lds_dec, hs, a);
        done += c;
    }
    a.gen_ahead = 0; a.npos = 0;                                    // a new stream's first L/2 positions, or catching up
    hipLaunchKernelGGL(gen_kern, dim3(nstreams), dim3(kGenThreads), lds_gen, hs, a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_stream_run_device(const scldpc_code_params *p, int32_t nstreams, uint64_t seed, uint64_t stream0,
                                        double eps, int32_t W, int32_t ndoped, const int32_t *doped_positions,
                                        int32_t npos, void *d_state, int64_t *d_counters, int32_t *d_trace, void *stream)
{
    return stream_run(p, nstreams, seed, stream0, eps, W, ndoped, doped_positions, npos, d_state, d_counters, d_trace,
                      nullptr, nullptr, 0, 0, stream);
}

extern "C" int scldpc_stream_run_device_inputs_at(const scldpc_code_params *p, int32_t nstreams, int32_t W, int32_t ndoped,
                                                  const int32_t *doped_positions, int32_t npos, void *d_state,
                                                  int64_t *d_counters, int32_t *d_trace, const uint16_t *d_inter,
                                                  const uint32_t *d_chan_bits, int64_t inputs_pos0, int32_t inputs_npos,
                                                  int64_t positions_done, void *stream)
{
    const char *who = "scldpc_stream_run_device_inputs";
    if (!d_inter || !d_chan_bits || inputs_npos < 0 || positions_done < 0 || inputs_pos0 < 0)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: null inputs or negative count", who);
    if (p && npos > 0) {
        // a stream is generated L/2 positions ahead and one more per decoded position (BPF:2001-2012, 2036-2045)
        const int64_t lo = positions_done == 0 ? 0 : (int64_t)p->L / 2 + positions_done;
        const int64_t hi = (int64_t)p->L / 2 + positions_done + npos;       // one past the last generated position needed
        if (lo < inputs_pos0 || hi > inputs_pos0 + inputs_npos)
            return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: decoding positions %lld .. %lld needs the generated positions "
                                     "%lld .. %lld, the inputs hold %lld .. %lld", who, (long long)positions_done,
                                     (long long)(positions_done + npos - 1), (long long)lo, (long long)(hi - 1),
                                     (long long)inputs_pos0, (long long)(inputs_pos0 + inputs_npos - 1));
    }
    return stream_run(p, nstreams, 0, 0, 0.0, W, ndoped, doped_positions, npos, d_state, d_counters, d_trace,
                      d_inter, d_chan_bits, inputs_npos, inputs_pos0, stream);
}

extern "C" int scldpc_stream_run_device_inputs(const scldpc_code_params *p, int32_t nstreams, int32_t W, int32_t ndoped,
                                               const int32_t *doped_positions, int32_t npos, void *d_state,
                                               int64_t *d_counters, int32_t *d_trace, const uint16_t *d_inter,
                                               const uint32_t *d_chan_bits, int32_t inputs_npos, int64_t positions_done,
                                               void *stream)
{
    return scldpc_stream_run_device_inputs_at(p, nstreams, W, ndoped, doped_positions, npos, d_state, d_counters, d_trace,
                                              d_inter, d_chan_bits, 0, inputs_npos, positions_done, stream);
}
