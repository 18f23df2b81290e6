// Streaming (circular-buffer) window decoding of a doped SC-LDPC stream — gfx950 kernel.
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
// One workgroup = one independent stream; the unit of work is one decoded position.  Same node-level machine as
// sw_bp.hip: S = what the CNs see, one [cnt | Σ id] word per CN slot kept exact, only CNs of the
// window fire (CNs to the right have never been updated and send erasures; CNs to the left are frozen and, with
// exact counts, have nothing left to say), every VN a window CN is left with lies inside the VN window.  A window
// reaches its fixpoint before the next one opens, so the frontier of a new window is just the degree-1 CNs of the
// position that entered it.  All per-stream state lives in a global-memory blob that persists between launches
// (ring of L positions: 2-byte adjacency rows, S / VNerased bits, CN words, the last dv socket permutations,
// counters); LDS holds the frontier queues, the bucket counters of the permutation ranking and per-slot counts.
// Two kernels share the blob (round 3): GENERATE (1024 threads, the ranking's 71 KB of LDS: two streams per CU) draws a chunk
// of positions ahead, DECODE (256 threads, 18 KB of LDS: eight streams per CU) then decodes as many — the window rounds are
// waits on L2 / memory round trips, so four times the streams in flight per CU is what they want, while the ranking wants
// the threads and the LDS.  Generating position g earlier than main_streaming does (it alternates decode / generate,
// BPF:2015-2046) changes nothing the decoder can see: decodeBP_SW_circular(pos) touches positions pos-2dv+1 .. pos+W+dv-2
// only, and a chunk is kept short enough that no ring slot still in use is overwritten.
// Sampling is keyed like sampler.hip: permutation of CN position c = rank of the Philox words with counter
// (socket>>2, c, stream id), channel of VN position q = counter (t>>2, 2^31 | q, stream id).
#include "common.h"
#include "kernel_util.h"
#include "philox.h"
#include <algorithm>

namespace {

using namespace scldpc_dev;

constexpr int kGenThreads = 1024, kDecThreads = 256, kMaxDoped = 32, kMaxL = 256;
constexpr int kQCap = 2048;                         // frontier-queue entries (an overflow falls back to a scan of the window's CNs)
enum { C_NE = 0, C_BE, C_EE, C_BEE, C_GB, C_GBL, C_GBE, C_GBLE, C_POS, C_GEN, C_NCOUNT = 16 };
enum { S_PUSH = 0, S_OVF = 3, S_REM = 6, S_ACC = 9, S_WL = 10, S_NSCAL = 16 };

struct StateLayout {        // byte offsets inside one stream's blob
    size_t adj, inter, sbits, ebits, cn, poscnt, gkey, wlist, counters, total;
    int wpp;                // 32-bit words of S / VNerased per position
};

struct Args {
    int dv, dc, L, C, V, S, W, nb, shift, lgchunk, dc_shift, npos;
    int gen_ahead;              // GEN: generate until gen == pos + L/2 + gen_ahead (and the first L/2 positions of a new stream)
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

__device__ __forceinline__ bool position_is_doped(const Args &a, long long pos)     // BPF:1589-1612
{
    if (a.ndoped == 0) return false;
    const int left = a.doped[0], period = a.doped[a.ndoped - 1] + 1, m = (int)(pos % period);
    if (m < left) return false;
    for (int i = 0; i < a.ndoped; i++) if (a.doped[i] == m) return true;
    return false;
}

// ROWS <= 4 (at most 4096 counter words, N <= 1024 at (4,8)): the LDS part is 34 KB, so two workgroups fit a CU if the
// compiler keeps to 64 VGPRs and 80 SGPRs (see full_bp.hip) — the kernel waits on L2 round trips most of the time and a second
// stream on the CU hides them.  Larger ensembles need the whole LDS for the counters and keep the registers they want.
// GEN: this launch generates positions (up to a.gen_ahead beyond the decoder's position + L/2); DEC: it decodes a.npos
// positions.  The product launches GEN-only (1024 threads) and DEC-only (256 threads) kernels in turn.
template <int ROWS, bool GEN, bool DEC, int kThreads>
__device__ __forceinline__ void stream_bp_body(const Args &a)
{
    constexpr int kWaves = kThreads / 64;
    extern __shared__ uint32_t lds[];
    // LDS: [queues | scan scratch | pos_cnt | scalars | totals] for both kernels; GEN adds [bucket counters | arrival slots]
    uint32_t *q0 = lds, *q1 = q0 + kQCap;                   // frontier queues
    uint32_t *wsum = q1 + kQCap;                            // scan scratch
    int *pos_cnt = reinterpret_cast<int *>(wsum + 32);      // [L] erased VNs per ring slot
    int *scal = pos_cnt + kMaxL;
    long long *acc = reinterpret_cast<long long *>(scal + S_NSCAL);      // the eight running totals: thread 0's alone
    uint32_t *hist = reinterpret_cast<uint32_t *>(acc + 8);             // nb 16-bit bucket counters, two per word (ranking)
    uint8_t *tsl = reinterpret_cast<uint8_t *>(hist + a.nb / 2);         // [S] arrival slot of every socket's key in its bucket

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L, C = a.C, V = a.V, S = a.S, dv = a.dv, ms = a.dv - 1, W = a.W, wpp = a.lay.wpp;
    char *st = a.state + (size_t)blockIdx.x * a.lay.total;
    uint16_t *adj = reinterpret_cast<uint16_t *>(st + a.lay.adj);          // [L][V][dv] position-local CN ids
    uint16_t *inter = reinterpret_cast<uint16_t *>(st + a.lay.inter);      // [dv][S] CN-local id of socket, by CN position % dv
    uint32_t *Sb = reinterpret_cast<uint32_t *>(st + a.lay.sbits);         // [L][wpp]
    uint32_t *Eb = reinterpret_cast<uint32_t *>(st + a.lay.ebits);         // [L][wpp] VNerased
    uint32_t *cn = reinterpret_cast<uint32_t *>(st + a.lay.cn);            // [L*C]
    int *pos_cnt_g = reinterpret_cast<int *>(st + a.lay.poscnt);
    uint2 *gkey = reinterpret_cast<uint2 *>(st + a.lay.gkey);             // [S] (key, socket) of straddling buckets' keys, by rank slot
    uint2 *wlist = reinterpret_cast<uint2 *>(st + a.lay.wlist);           // [S] the same keys as a dense list: (key, socket | first rank << 16)
    long long *cnt64 = reinterpret_cast<long long *>(st + a.lay.counters);
    const unsigned long long sid = a.sid0 + blockIdx.x;
    const uint32_t s_lo = (uint32_t)sid, s_hi = (uint32_t)(sid >> 32);
    auto ldcn = [&](int c) { return __hip_atomic_load(&cn[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    STAMP_DECL
    for (int i = tid; i < L; i += kThreads) pos_cnt[i] = pos_cnt_g[i];
    if (tid < S_NSCAL) scal[tid] = 0;
    __syncthreads();
    if (tid == 0) for (int k = C_NE; k <= C_GBLE; k++) acc[k] = cnt64[k];
    long long pos = cnt64[C_POS], gen = cnt64[C_GEN];

    // ---- socket permutation of CN position cpos → inter[cpos % dv] (fill_interleaver_pos, BPF:1763-1787) ----
    auto rank_position = [&](long long cpos) {
        if (a.ext_inter) {                                  // same-input mode: the permutation was drawn on the host
            const uint16_t *src = a.ext_inter + ((size_t)blockIdx.x * (size_t)(a.ext_npos + dv - 1) + (size_t)(cpos - a.ext_pos0)) * S;
            uint16_t *dst = inter + (size_t)(cpos % dv) * S;
            for (int s = tid; s < S; s += kThreads) dst[s] = src[s];
            __syncthreads();
            return;
        }
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
        if (crowded >= 256u) __builtin_trap();              // arrival slots are kept in a byte (a bucket holds 1-4 keys on average)
        __syncthreads();
        STAMP(7);
        // wave w scans buckets [w, w+1) * nb/16 = ROWS * 32 words of two counters: exclusive prefix inside the chunk, then
        // (second barrier) plus the chunks before it — every bucket's first rank, 16 bits
        constexpr int R2 = ROWS >= 2 ? ROWS / 2 : 1;
        const bool on = ROWS >= 2 || lane < 32;
        const int w0 = wave * (ROWS * 32) + lane;
        {
            uint32_t v[R2], ps[R2], inc[R2];
#pragma unroll
            for (int r = 0; r < R2; r++) { v[r] = on ? hist[w0 + r * 64] : 0u; ps[r] = (v[r] & 0xFFFFu) + (v[r] >> 16); }
#pragma unroll
            for (int r = 0; r < R2; r++) inc[r] = wave_inclusive_scan(ps[r]);
            uint32_t carry = 0;
#pragma unroll
            for (int r = 0; r < R2; r++) {
                const uint32_t ex = carry + inc[r] - ps[r];
                if (on) hist[w0 + r * 64] = (ex & 0xFFFFu) | ((ex + (v[r] & 0xFFFFu)) << 16);
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc[r], 63);
            }
            if (lane == 0) wsum[wave] = carry;
        }
        __syncthreads();
        {
            const uint32_t t = lane < kWaves ? wsum[lane] : 0u;
            const uint32_t inc = wave_inclusive_scan(t);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(inc - t), wave);
#pragma unroll
            for (int r = 0; r < R2; r++)
                if (on) {                                   // (a first rank of 65536 = S wraps to 0: see bucket_end)
                    const uint32_t w = hist[w0 + r * 64];
                    hist[w0 + r * 64] = ((w + base) & 0xFFFFu) | (((w >> 16) + base) << 16);
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
        uint16_t *dst = inter + (size_t)(cpos % dv) * S;
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
            if (q * 4 + 3 < S && (S & 3) == 0) {
                *reinterpret_cast<uint2 *>(dst + (size_t)q * 4) = make_uint2(c4[0] | (c4[1] << 16), c4[2] | (c4[3] << 16));
            } else {
                for (int u = 0; u < 4; u++) if (q * 4 + u < S) dst[q * 4 + u] = (uint16_t)c4[u];
            }
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
                dst[s] = cn_of(rank);
            }
        }
        __syncthreads();
    };

    // ---- generate_stream_pos(g) + initialize_messages_circular(g) (BPF:1927-1932, 1149-1166) ------------------
    auto generate = [&](long long g) {
        rank_position(g + dv - 1);
        STAMP(4);
        const int slot = (int)(g % L), cslot_new = (int)((g + dv - 1) % L);
        for (int k = tid; k < C; k += kThreads) cn[cslot_new * C + k] = 0;      // a fresh CN position (BPF:1832-1837)
        const bool doped = position_is_doped(a, g);
        uint32_t c_lo = a.seed_lo, c_hi = a.seed_hi;
        asm volatile("" : "+s"(c_lo), "+s"(c_hi));
        int erased_here = 0;
        for (int w = tid; w < wpp; w += kThreads) {                              // channel (BPF:1621-1654)
            uint32_t word = 0;
            if (a.ext_chan) {
                word = a.ext_chan[((size_t)blockIdx.x * a.ext_npos + (size_t)(g - a.ext_pos0)) * wpp + w];     // doped positions: all zero
            } else if (!doped) {
#pragma unroll
                for (int c8 = 0; c8 < 8; c8++) {
                    uint32_t r[4];
                    philox4x32_10((uint32_t)(w * 8 + c8), 0x80000000u | (uint32_t)g, s_lo, s_hi, c_lo, c_hi, r);
#pragma unroll
                    for (int u = 0; u < 4; u++) word |= (uint32_t)((r[u] >> 1) < a.thresh) << (c8 * 4 + u);
                }
                if (w * 32 + 32 > V) word &= (1u << (V - w * 32)) - 1u;
            }
            Sb[slot * wpp + w] = word;
            erased_here += __popc(word);
        }
        {
            const uint32_t tot = wave_inclusive_scan((uint32_t)erased_here);
            if (lane == 63 && tot) atomicAdd(&scal[S_ACC], (int)tot);
        }
        __syncthreads();
        STAMP(5);
        if (tid == 0) { pos_cnt[slot] = scal[S_ACC]; scal[S_ACC] = 0; }
        for (int t = tid; t < V; t += kThreads) {                                // wiring (BPF:1841-1854)
            uint16_t loc[8];
            for (int i = 0; i < dv; i++) {
                loc[i] = inter[(size_t)((g + i) % dv) * S + dv * t + i];
                adj[((size_t)slot * V + t) * dv + i] = loc[i];
            }
            if ((Sb[slot * wpp + (t >> 5)] >> (t & 31)) & 1u)
                for (int i = 0; i < dv; i++)
                    atomicAdd(&cn[(int)((g + i) % L) * C + loc[i]], kCntOne + (uint32_t)(slot * V + t));
        }
        __syncthreads();
    };

    if constexpr (GEN) {
        if (gen == 0 && pos == 0) {                         // a new stream (BPF:2003-2012)
            for (int i = tid; i < L * wpp; i += kThreads) { Sb[i] = 0; Eb[i] = 0; }
            for (int i = tid; i < L * C; i += kThreads) cn[i] = 0;
            for (int i = tid; i < L; i += kThreads) pos_cnt[i] = 0;
            __syncthreads();
            for (int c = 0; c < dv - 1; c++) rank_position(c);  // initialize_arrays_circular (BPF:1808-1813)
            for (; gen < L / 2; gen++) generate(gen);
        }
        if constexpr (!DEC)                                 // the positions the next DECODE launch will have consumed (BPF:2036-2045)
            for (; gen < pos + L / 2 + a.gen_ahead; gen++) generate(gen);
    }

    int genq = 0;                                           // queue generation counter
    for (int step = 0; DEC && step < a.npos; step++, pos++) {
        STAMP(0);
        const long long pd = pos - ms, pe = pos - 2 * dv + 1;
        if (tid == 0) {                                                         // BPF:2017-2028
            if (pd >= 0 && !position_is_doped(a, pd)) { acc[C_GB] += V; acc[C_GBL] += 1; }
            if (pe >= 0 && !position_is_doped(a, pe)) { acc[C_GBE] += V; acc[C_GBLE] += 1; }
        }

        // ---- decodeBP_SW_circular(pos) --------------------------------------------------------------------
        // frontier of a new window: degree-1 CNs of the position(s) that entered it
        if (tid == 0) { scal[S_PUSH + genq % 3] = 0; scal[S_OVF + genq % 3] = 0; }
        __syncthreads();
        {
            uint32_t *qc = (genq & 1) ? q1 : q0;
            for (long long qq = (pos == 0 ? 0 : pos + W - 1); qq < pos + W; qq++) {
                const int cs = (int)(qq % L) * C;
                for (int k = tid; k < C; k += kThreads) {
                    const uint32_t w = ldcn(cs + k);
                    if ((w >> kCntShift) == 1u) {                                 // the queues hold the VN to release
                        const int idx = atomicAdd(&scal[S_PUSH + genq % 3], 1);
                        if (idx < kQCap) qc[idx] = w & kSumMask; else scal[S_OVF + genq % 3] = 1;
                    }
                }
            }
        }
        __syncthreads();
        STAMP(1);
        int ncur = min(scal[S_PUSH + genq % 3], kQCap);
        bool rescan = scal[S_OVF + genq % 3] != 0;
        const int basemod = (int)(((pos - ms) % L + L) % L);
        for (;;) {
            uint32_t *qc = (genq & 1) ? q1 : q0, *qn = (genq & 1) ? q0 : q1;
            int *push_cnt = &scal[S_PUSH + (genq + 1) % 3], *push_ovf = &scal[S_OVF + (genq + 1) % 3];
            if (tid == 0) { scal[S_PUSH + (genq + 2) % 3] = 0; scal[S_OVF + (genq + 2) % 3] = 0; }
            // Release VN j, the lone erased neighbour of some window CN.  A level costs two dependent trips to the L2
            // instead of four: the queue names the VN (a CN whose count drops 2 -> 1 is left with Σ ids − j: no read of the
            // CN word next round), and the VN's row is fetched beside the claim of its S bit, not after it.  The claim is
            // the guard: an entry whose VN was released meanwhile (by another CN, or named twice) finds the bit clear.
            // (Letting a thread follow its chain past the barrier was tried: one thread then walks what a round spreads
            // over the workgroup — 1.75 times the time.)
            auto release = [&](int j) {
                const int slot_j = j / V, t = j - slot_j * V;
                const uint32_t bit = 1u << (t & 31);
                uint16_t loc[8];
                if (dv == 4) {
                    const uint2 r = *reinterpret_cast<const uint2 *>(adj + (size_t)j * 4);
                    loc[0] = (uint16_t)r.x; loc[1] = (uint16_t)(r.x >> 16); loc[2] = (uint16_t)r.y; loc[3] = (uint16_t)(r.y >> 16);
                } else {
                    for (int i = 0; i < dv; i++) loc[i] = adj[(size_t)j * dv + i];
                }
                if (!(atomicAnd(&Sb[slot_j * wpp + (t >> 5)], ~bit) & bit)) return;
                atomicSub(&pos_cnt[slot_j], 1);
                const long long qj = pos - ms + ((slot_j - basemod + L) % L);     // absolute position of VN j
                for (int i = 0; i < dv; i++) {
                    const long long qc2 = qj + i;
                    const int c2 = (int)(qc2 % L) * C + loc[i];
                    const uint32_t old = atomicSub(&cn[c2], kCntOne + (uint32_t)j);
                    if ((old >> kCntShift) == 2u && qc2 >= pos && qc2 < pos + W) { // a window CN is left with one VN
                        const int idx = atomicAdd(push_cnt, 1);
                        if (idx < kQCap) qn[idx] = (old & kSumMask) - (uint32_t)j; else *push_ovf = 1;
                    }
                }
            };
            if (rescan) {
                // queue overflow: walk every CN of the window (each CN is the lone holder of at most one VN, and a
                // VN released here may promote further CNs into THIS round — allowed: the window runs to its
                // fixpoint and only the fixpoint is observable)
                for (long long qq = pos; qq < pos + W; qq++) {
                    const int cs = (int)(qq % L) * C;
                    for (int k = tid; k < C; k += kThreads) {
                        const uint32_t w = ldcn(cs + k);
                        if ((w >> kCntShift) == 1u) release((int)(w & kSumMask));
                    }
                }
            } else {
                for (int k = tid; k < ncur; k += kThreads) release((int)qc[k]);
            }
            __syncthreads();
            rescan = *push_ovf != 0;
            ncur = min(*push_cnt, kQCap);
            genq++;
            if (ncur == 0 && !rescan) break;                                      // nothing left to fire (BPF:1454-1455)
        }
        STAMP(2);
        // decision on position pos-ms (BPF:1445-1449), VNerased := S there
        int nep = 0;
        if (pd >= 0) {
            const int slot = (int)(pd % L);
            nep = pos_cnt[slot];
            for (int w = tid; w < wpp; w += kThreads) Eb[slot * wpp + w] = Sb[slot * wpp + w];
        }
        if (tid == 0) { acc[C_NE] += nep; if (nep > 0) acc[C_BE] += 1; }           // BPF:1480-1483
        __syncthreads();
        // size-2 stopping-set expurgation of position pos-2dv+1 (get_deg_two_ss, BPF:1227-1283, 1485-1497)
        if (pe >= 0) {
            const int slot = (int)(pe % L);
            int mine = 0;
            for (int w = tid; w < wpp; w += kThreads) {
                uint32_t x = Eb[slot * wpp + w];
                while (x) {
                    const int b = __ffs((int)x) - 1;
                    x &= x - 1;
                    const int t = w * 32 + b, va = slot * V + t;
                    bool pair = true;
                    int partner = -1;
                    for (int i = 0; i < dv; i++) {
                        const uint32_t s = ldcn((int)((pe + i) % L) * C + adj[(size_t)va * dv + i]);
                        const int b2 = (int)((s & kSumMask) - (uint32_t)va);
                        if ((s >> kCntShift) != 2u || (i > 0 && b2 != partner)) { pair = false; break; }
                        partner = b2;
                    }
                    mine += 1 - (pair && partner / V == slot ? 1 : 0);
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
        __syncthreads();
        STAMP(3);
        if constexpr (GEN) { generate(gen); gen++; }                              // BPF:2036-2045 (one-kernel form)
        STAMP(6);
    }
    STAMP_FLUSH();

    __syncthreads();
    for (int i = tid; i < L; i += kThreads) pos_cnt_g[i] = pos_cnt[i];
    if (tid == 0) {
        for (int k = C_NE; k <= C_GBLE; k++) cnt64[k] = acc[k];
        cnt64[C_POS] = pos; cnt64[C_GEN] = gen;
        if (a.counters_out) {
            long long *o = a.counters_out + (size_t)blockIdx.x * 10;
            for (int k = C_NE; k <= C_GBLE; k++) o[k] = acc[k];
            o[8] = pos; o[9] = gen;
        }
    }
}

template <int ROWS>
__global__ __launch_bounds__(kGenThreads, 8) __attribute__((amdgpu_num_sgpr(72))) void stream_gen_kernel(const Args a)
{
    stream_bp_body<ROWS, true, false, kGenThreads>(a);
}

__global__ __launch_bounds__(kDecThreads, 8) __attribute__((amdgpu_num_sgpr(80))) void stream_dec_kernel(const Args a)
{
    stream_bp_body<1, false, true, kDecThreads>(a);
}


int make_state_layout(const scldpc_code_params *p, StateLayout *lay)
{
    const size_t L = p->L, V = p->vns_pos, C = p->cns_pos, S = (size_t)p->cns_pos * p->dc, dv = p->dv;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    lay->wpp = (int)((V + 31) / 32);
    lay->adj = take(L * V * dv * 2);
    lay->inter = take(dv * S * 2);
    lay->sbits = take(L * lay->wpp * 4);
    lay->ebits = take(L * lay->wpp * 4);
    lay->cn = take(L * C * 4);
    lay->poscnt = take(L * 4);
    lay->gkey = take(S * 8); lay->wlist = take(S * 8);
    lay->counters = take(C_NCOUNT * 8);
    lay->total = off;
    return 0;
}

int check_stream(const scldpc_code_params *p, int W, const char *who)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (p->L > kMaxL || p->L < 2 * p->dv)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: buffer length L=%d outside [%d, %d]", who, p->L, 2 * p->dv, kMaxL);
    // the stream is generated L/2 positions ahead (BPF:2001): the window and the CNs of its VNs must exist already
    if (W < 1 || W + p->dv - 1 > p->L / 2)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "%s: need 1 <= W and W + dv - 1 <= L/2 (W=%d, L=%d)", who, W, p->L);
    if ((int64_t)p->cns_pos * p->dc > 65536 || p->dc > 15 || p->dv > 8 ||
        (int64_t)p->dc * p->L * p->vns_pos >= (1ll << kDegShift))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "%s: ensemble too large for the streaming kernel", who);
    return SCLDPC_OK;
}

}  // namespace

extern "C" int64_t scldpc_stream_state_bytes(const scldpc_code_params *p, int32_t W)
{
    if (int rc = check_stream(p, W, "scldpc_stream_state_bytes")) return rc;
    StateLayout lay;
    make_state_layout(p, &lay);
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
    make_state_layout(p, &a.lay);
    a.state = static_cast<char *>(d_state); a.counters_out = reinterpret_cast<long long *>(d_counters); a.trace = d_trace;
    a.ext_inter = d_ext_inter; a.ext_chan = d_ext_chan; a.ext_npos = ext_npos; a.ext_pos0 = ext_pos0;
    const int rows = a.nb / kGenThreads;
    const size_t lds_dec = 4u * ((size_t)2 * kQCap + 32 + kMaxL + S_NSCAL + 2 * 8);
    const size_t lds_gen = lds_dec + 4u * ((size_t)a.nb / 2) + (((size_t)a.S + 15) & ~(size_t)15);
    if (lds_gen > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_stream_run_device: %zu bytes of LDS per stream", lds_gen);
    void (*gen_kern)(const Args) = rows == 1 ? stream_gen_kernel<1> : rows == 2 ? stream_gen_kernel<2>
                                   : rows == 4 ? stream_gen_kernel<4> : rows == 8 ? stream_gen_kernel<8> : stream_gen_kernel<16>;
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(gen_kern))) return rc_;
    // Cycles of GENERATE (run `ahead` positions beyond the reference's lag of L/2) and DECODE (as many positions as are
    // then generated far enough: decodeBP_SW_circular(pos) needs positions up to pos + W + dv - 2).  `ahead` is bounded by the
    // ring: generating position g re-uses the slots of VN position g - L and CN position g + dv - 1 - L, and the decoder
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
        hipLaunchKernelGGL(stream_dec_kernel, dim3(nstreams), dim3(kDecThreads), lds_dec, hs, a);
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
