// Random-pick peeling with the degree-1-CN trajectory — gfx950 kernel.
//
// Replaces, per trial, the body of the `for o in trange(num_repeats)` loop of simulate_peeling_decoder_ldpc
// (simulators_sc_ldpc/peeling_decoding/peeling_decoding.py = PD, PD:740-785):
//   r[t] = #erased VNs on CN t for t < total_size (PD:756-757; CNs beyond a non-terminated chain's L positions stay in
//   the graph but are never pickable);  r1[0] = #{t : r[t] == 1};  then num_pd_steps times:
//     m = random.choice(np.flatnonzero(r == 1))   (PD:1022-1026: ascending index order; CPython's _randbelow:
//         k = n.bit_length(); draw getrandbits(k) = MT19937 word >> (32-k) until < n)
//     remove the single VN of m from its l CNs, r[s] -= 1 for s < total_size (PD:769-777);  r1[step+1] = #{r == 1}.
//   With no degree-1 CN left the count is copied forward and no number is drawn (PD:765-767).
// This is a chain of dependent steps: the parallelism is across trials (one workgroup each, stepped by one wave)
// and inside a step (rank-select of the x-th set bit of the degree-1 bitmap by popcount + DPP prefix scan; the l
// CN updates on l lanes).  State per trial in LDS: one [cnt:4 | Σ VN id:24] word per CN, the degree-1 bitmap, and
// the generator: either the caller's MT19937 state (exact replay of the Python `random` stream; read and written
// back so that a host loop can chain trials like the reference's single stream) or a per-trial Philox counter.
#include "common.h"
#include "kernel_util.h"
#include "philox.h"

#ifndef SCLDPC_PICK_SGPRS
#define SCLDPC_PICK_SGPRS 80
#endif

namespace {

using namespace scldpc_dev;

constexpr int kBlock = 64;

struct Args {
    int dv, vns_pos, cns_pos, n, ncn, total_size, steps, nw, nd1;     // nd1 = 64-bit words of the degree-1 bitmap
    int rng_mode;                   // 0 = MT19937 state in d_mt, 1 = Philox keyed by (seed, trial0 + trial)
    int ntrials;                    // (multi-trial kernel: the last wave may hold fewer than TPW trials)
    int prebuilt;                   // (multi-trial kernel) the CN words were built by cn_build.hip
    int bshift;                     // a rank-select block covers 2^bshift CNs (64 * 2^(bshift-12) bitmap words)
    uint32_t magic_v, seed_lo, seed_hi;
    unsigned long long trial0;
    int off_d1, off_blk, off_mt, off_sc;   // LDS offsets (32-bit words) behind the CN words
    const void *vn_adj;
    const uint32_t *chan;
    uint32_t *ws;                   // [T][ncn] CN words in global memory (G only)
    unsigned long long *ws_d1;      // [T][nd1] degree-1 bitmaps in global memory (D1G)
    unsigned long long *moments;    // optional [3][steps+1]: #(r1 != 0), Σ r1, Σ r1² over the trials (atomic adds)
    uint32_t *mt;                   // [T][625]: 624 state words + index
    int32_t *r1;                    // [T][steps+1] or null
    int32_t *out;                   // [T][4]: #erased, #picked, last r1, steps executed with a pick
};

using scldpc_dev::philox4x32_10;

// G: CN words in the global workspace (ensembles beyond the LDS budget, e.g. the notebook's N = 10000)
// Throughput = trials in flight, one wave each: the scalar file admits eight waves per SIMD only up to 80 SGPRs per wave
// (⌊800 / (⌈sgpr/16⌉·16 + 16)⌋; uncapped the kernel takes 112 and six waves fit), the overflow lives in VGPR lanes.
template <int DV, bool A16, bool G, bool D1G>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(SCLDPC_PICK_SGPRS))) void peel_pick_kernel(const Args a)
{
    extern __shared__ uint32_t lds[];
    uint32_t *cn;                                                         // ncn words
    if constexpr (G) cn = a.ws + (size_t)blockIdx.x * a.ncn;
    else             cn = lds;
    auto ldw = [&](int c) -> uint32_t {
        if constexpr (G) return __hip_atomic_load(&cn[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else             return cn[c];
    };
    int *blk = reinterpret_cast<int *>(lds + a.off_blk);                  // #degree-1 CNs per block of 64 bitmap words
    // the degree-1 bitmap: nd1 words of 64 bits, in LDS or (D1G) behind the trial's CN words in the workspace
    unsigned long long *d1;
    if constexpr (D1G) d1 = a.ws_d1 + (size_t)blockIdx.x * a.nd1;
    else               d1 = reinterpret_cast<unsigned long long *>(lds + a.off_d1);
    uint32_t *mt = lds + a.off_mt;                                        // 624 words
    int *sc = reinterpret_cast<int *>(lds + a.off_sc);                   // [0] #erased, [1] r1[0]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int trial = blockIdx.x;
    const int n = a.n, dv = (DV ? DV : a.dv), ts = a.total_size;
    const char *adj = static_cast<const char *>(a.vn_adj) + (size_t)trial * n * dv * (A16 ? 2 : 4);
    auto pos_of = [&](int j) { return (int)__umulhi((uint32_t)j, a.magic_v); };

    for (int c = tid; c < a.ncn; c += kBlock) cn[c] = 0;
    for (int w = tid; w < a.nd1; w += kBlock) d1[w] = 0ull;
    if (tid == 0) { sc[0] = 0; sc[1] = 0; }
    if (tid < 64) blk[tid] = 0;
    int ne_local = 0;
    for (int w = tid; w < a.nw; w += kBlock) {
        uint32_t x = a.chan[(size_t)trial * a.nw + w];
        if (w == a.nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
        ne_local += __popc(x);
    }
    if (a.rng_mode == 0)
        for (int i = tid; i < 624; i += kBlock) mt[i] = a.mt[(size_t)trial * 625 + i];
    __syncthreads();
    {
        const uint32_t tot = wave_inclusive_scan((uint32_t)ne_local);
        if (lane == 63 && tot) atomicAdd(&sc[0], (int)tot);
    }
    for (int j = tid; j < n; j += kBlock) {
        if ((a.chan[(size_t)trial * a.nw + (j >> 5)] >> (j & 31)) & 1u) {      // (j < n: no padding bit is ever tested)
            int32_t cc[8];
            load_adj<DV, A16>(adj, dv, j, pos_of(j), a.cns_pos, cc);
            for (int i = 0; i < dv; i++) atomicAdd(&cn[cc[i]], kCntOne + (uint32_t)j);
        }
    }
    __syncthreads();
    // degree-1 bitmap over the pickable CNs (PD:756-757) and r1[0] (PD:758)
    int n1_local = 0;
    for (int base = 0; base < ts; base += kBlock) {
        const int c = base + tid;
        const bool one = c < ts && (ldw(c) >> kCntShift) == 1u;
        const unsigned long long m = __ballot(one);
        if (lane == 0 && c < ts) { d1[c >> 6] = m; if (m) atomicAdd(&blk[c >> a.bshift], __popcll(m)); }
        n1_local += one;
    }
    {
        const uint32_t tot = wave_inclusive_scan((uint32_t)n1_local);
        if (lane == 63 && tot) atomicAdd(&sc[1], (int)tot);
    }
    __syncthreads();
    if (wave != 0) return;                     // the step chain runs on one wave; no barriers below

    int n1 = sc[1], picked = 0, with_pick = 0;
    int32_t *r1 = a.r1 ? a.r1 + (size_t)trial * (a.steps + 1) : nullptr;
    if (lane == 0) {
        if (r1) r1[0] = n1;
        if (a.moments && n1) {
            atomicAdd(&a.moments[0], 1ull);
            atomicAdd(&a.moments[(size_t)(a.steps + 1)], (unsigned long long)n1);
            atomicAdd(&a.moments[2 * (size_t)(a.steps + 1)], (unsigned long long)n1 * (unsigned long long)n1);
        }
    }
    uint32_t mti = a.rng_mode == 0 ? a.mt[(size_t)trial * 625 + 624] : 0u;      // MT index, or Philox draw counter
    const unsigned long long gtrial = a.trial0 + (unsigned long long)trial;

    uint32_t prc[4] = {0u, 0u, 0u, 0u};         // the current Philox call's four words
    auto next_u32 = [&]() -> uint32_t {        // wave-uniform result
        if (a.rng_mode == 0) {
            if (mti >= 624u) {                 // twist, in lockstep batches of 64 == the sequential recurrence
                for (int b = 0; b < 624; b += 64) {
                    const int i = b + lane;
                    uint32_t y = 0, src = 0;
                    if (i < 624) {
                        const uint32_t cur = mt[i], nxt = mt[i == 623 ? 0 : i + 1];
                        src = mt[i < 227 ? i + 397 : i - 227];
                        y = (cur & 0x80000000u) | (nxt & 0x7FFFFFFFu);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (i < 624) mt[i] = src ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
                    __builtin_amdgcn_wave_barrier();
                }
                mti = 0;
            }
            uint32_t y = mt[mti++];
            y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
            return y;
        }
        // draw d is word d & 3 of Philox call d >> 2: one call serves four draws (a call per draw was a quarter of a pick's
        // instructions; the chain waits on memory, so it bought 3 %)
        if ((mti & 3u) == 0u)
            philox4x32_10(mti >> 2, 0x90000000u, (uint32_t)gtrial, (uint32_t)(gtrial >> 32), a.seed_lo, a.seed_hi, prc);
        const uint32_t y = (mti & 2u) ? ((mti & 1u) ? prc[3] : prc[2]) : ((mti & 1u) ? prc[1] : prc[0]);
        mti++;
        return y;
    };

    STAMP_DECL
    int s = 0;
    for (; s < a.steps && n1 > 0; s++) {
        STAMP(0);                                           // loop end: r1 store, counters
        // ---- x = _randbelow(n1) ----------------------------------------------------------------
        const int k = 32 - __clz(n1);
        uint32_t x;
        do { x = next_u32() >> (32 - k); } while (x >= (uint32_t)n1);
        STAMP(1);                                           // the draw
        // ---- m = x-th set bit of the degree-1 bitmap, ascending: block of 64 words, word, bit ------------
        const uint32_t bc = (uint32_t)blk[lane];
        const uint32_t binc = wave_inclusive_scan(bc);
        const int B0 = __ffsll((long long)__ballot(binc > x)) - 1;
        uint32_t r = x - (uint32_t)__builtin_amdgcn_readlane((int)(binc - bc), B0);
        // block B0 = 64 * kw bitmap words; lane l holds its words l*kw .. l*kw + kw - 1 (kw = 1 up to 262144 pickable CNs)
        const int kw = 1 << (a.bshift - 12);
        unsigned long long word[4] = {0ull, 0ull, 0ull, 0ull};
        uint32_t wc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int widx = (B0 * 64 + lane) * kw + q;
            if (q < kw && widx < a.nd1) {
                if constexpr (D1G) word[q] = __hip_atomic_load(&d1[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else               word[q] = d1[widx];
            }
            wc += (uint32_t)__popcll(word[q]);
        }
        const uint32_t winc = wave_inclusive_scan(wc);
        const int W0 = __ffsll((long long)__ballot(winc > r)) - 1;
        r -= (uint32_t)__builtin_amdgcn_readlane((int)(winc - wc), W0);
        int m = -1;
        if (lane == W0) {
            int q = 0;
            unsigned long long w = word[0];
#pragma unroll
            for (int qq = 1; qq < 4; qq++)                              // the word of this lane that holds the r-th set bit
                if (q == qq - 1 && r >= (uint32_t)__popcll(w)) { r -= (uint32_t)__popcll(w); w = word[qq]; q = qq; }
            while (r--) w &= w - 1;
            m = ((B0 * 64 + lane) * kw + q) * 64 + (__ffsll((long long)w) - 1);
        }
        m = __builtin_amdgcn_readlane(m, W0);
        STAMP(2);                                           // rank-select incl. the bitmap words' trip
        // ---- remove its single VN from all its CNs (PD:769-777) ------------------------------------
        const int j = (int)(ldw(m) & kSumMask);
        STAMP(3);                                           // the CN word's trip
        int32_t cc[8];
        load_adj<DV, A16>(adj, dv, j, pos_of(j), a.cns_pos, cc);
        STAMP(4);                                           // the VN row's trip
        bool plus = false, minus = false;
        if (lane < dv) {
            const int c = cc[lane];
            const uint32_t nc = (atomicSub(&cn[c], kCntOne + (uint32_t)j) >> kCntShift) - 1u;
            if (c < ts) {
                minus = nc == 0u;                       // was 1
                plus = nc == 1u;                        // became 1
                if (plus || minus) {
                    atomicXor(&d1[c >> 6], 1ull << (c & 63));
                    atomicAdd(&blk[c >> a.bshift], plus ? 1 : -1);
                }
            }
        }
        n1 += __popcll(__ballot(plus)) - __popcll(__ballot(minus));
        STAMP(5);                                           // the returning atomics' trip + bitmap updates
        picked++; with_pick++;
        if (lane == 0) {
            if (r1) r1[s + 1] = n1;
            if (a.moments && n1) {
                atomicAdd(&a.moments[s + 1], 1ull);
                atomicAdd(&a.moments[(size_t)(a.steps + 1) + s + 1], (unsigned long long)n1);
                atomicAdd(&a.moments[2 * (size_t)(a.steps + 1) + s + 1], (unsigned long long)n1 * (unsigned long long)n1);
            }
        }
    }
    STAMP_FLUSH();
    // no degree-1 CN left: the count (0) is copied forward, nothing is drawn (PD:765-767)
    if (r1)
        for (int t = s + 1 + lane; t <= a.steps; t += 64) r1[t] = n1;
    if (a.rng_mode == 0) {
        for (int i = lane; i < 624; i += 64) a.mt[(size_t)trial * 625 + i] = mt[i];
        if (lane == 0) a.mt[(size_t)trial * 625 + 624] = mti;
    }
    if (lane == 0) {
        int32_t *o = a.out + (size_t)trial * 4;
        o[0] = sc[0]; o[1] = picked; o[2] = n1; o[3] = with_pick;
    }
}


// ---- several trials per wave ------------------------------------------------------------------------------------------------
// A pick is a chain of four dependent trips to DRAM (bitmap words, the CN's word, the VN's row, dv returning atomics: the
// stamps say 4.4 us with every wave slot of the chip holding a trial), so throughput = trials in flight — and a wave's 64 lanes
// have little to do for one trial.  Here a wave steps TPW trials in lockstep, 64 / TPW lanes each: every instruction works
// for all of them (the draw, the scans — segmented: a DPP row is 16 lanes — the loads), the trips of TPW chains overlap, and
// twice or four times as many trials are in flight.  The rank-select has three levels (64 blocks, 8 sub-blocks, 8 words; the
// counts of the first two in LDS, 1.25 KB per trial so that 32 waves still share a CU): a pick reads 64 bytes of the bitmap, not
// 512.  For the layout of BASELINE config 3: CN words and degree-1 bitmap in the
// workspace, 2-byte rows, dv = 4, Philox draws, at most 262 144 pickable CNs (one bitmap word per lane and block).  The draws,
// the ascending-order selection and the trajectory are those of peel_pick_kernel, value for value (tests/test_gpu_pd.py).
template <int TPW>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(SCLDPC_PICK_SGPRS))) void peel_pick_multi_kernel(const Args a)
{
    constexpr int LPT = 64 / TPW, Q = 64 / LPT;                           // lanes per trial; rank-select blocks (and block words) per lane
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x, seg = lane / LPT, sl = lane % LPT, sbase = seg * LPT;
    const int trial = blockIdx.x * TPW + seg;
    const bool valid = trial < a.ntrials;
    int *blk = reinterpret_cast<int *>(lds) + seg * (64 + 256);           // #degree-1 CNs per block of 64 bitmap words
    uint32_t *sub = reinterpret_cast<uint32_t *>(blk + 64);               // ... and per sub-block of 8 words (64 bytes of the bitmap), 16 bits each
    auto sub_add = [&](int i, int d) { atomicAdd(&sub[i >> 1], (uint32_t)d << ((i & 1) * 16)); };      // (counts <= 512: no carry, no borrow)
    const int n = a.n, ts = a.total_size;
    const size_t tix = valid ? (size_t)trial : 0;                         // (lanes of an absent trial idle on trial 0's addresses, predicated off)
    uint32_t *cn = a.ws + tix * a.ncn;
    unsigned long long *d1 = a.ws_d1 + tix * a.nd1;
    const unsigned long long *rows = reinterpret_cast<const unsigned long long *>(a.vn_adj) + tix * n;     // 4 x uint16 per VN
    const uint32_t *chan = a.chan + tix * a.nw;
    auto pos_of = [&](int j) { return (int)__umulhi((uint32_t)j, a.magic_v); };
    auto ldw = [&](int c) { return __hip_atomic_load(&cn[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // inclusive scan over the LPT lanes of a segment: DPP row_shr inside rows of 16 lanes, row_bcast:15 joins two rows
    auto seg_scan = [](uint32_t x) {
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
        if constexpr (LPT == 32) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
        return x;
    };
    auto seg_bits = [&](unsigned long long m) { return (uint32_t)((m >> sbase) & ((1ull << LPT) - 1ull)); };
    auto from = [&](uint32_t v, int l) { return (uint32_t)__shfl((int)v, sbase + l, 64); };      // lane l of the own segment

    // ---- per trial: clear, count the erased VNs into their CNs, the degree-1 bitmap --------------------------------------
    for (int i = sl; i < 64 + 256; i += LPT) blk[i] = 0;
    int ne = 0, n1 = 0;
    if (valid) {
        if (!a.prebuilt) for (int c = sl; c < a.ncn; c += LPT) cn[c] = 0;
        for (int w = sl; w < a.nw; w += LPT) {
            uint32_t x = chan[w];
            if (w == a.nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1u;
            ne += __popc(x);
        }
    }
    __syncthreads();                                                      // (one wave: orders the stores above before the atomics)
    if (valid && !a.prebuilt) {
        for (int j = sl; j < n; j += LPT) {
            if ((chan[j >> 5] >> (j & 31)) & 1u) {
                const unsigned long long r = rows[j];
                const int base = pos_of(j) * a.cns_pos;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    atomicAdd(&cn[base + i * a.cns_pos + (int)((r >> (16 * i)) & 0xFFFFull)], kCntOne + (uint32_t)j);
            }
        }
    }
    __syncthreads();
    {
        unsigned long long acc = 0ull;                                    // lane 0 of the segment assembles the bitmap words
        for (int base = 0; base < ((ts + 63) & ~63); base += LPT) {
            const int c = base + sl;
            const bool one = valid && c < ts && (ldw(c) >> kCntShift) == 1u;
            const uint32_t bits = seg_bits(__ballot(one));
            n1 += __popc(bits);
            acc |= (unsigned long long)bits << (base & 63);
            if (((base + LPT) & 63) == 0) {
                if (valid && sl == 0) {
                    if ((base >> 6) < a.nd1) d1[base >> 6] = acc;
                    if (acc) { atomicAdd(&blk[base >> a.bshift], __popcll(acc)); sub_add(base >> 9, __popcll(acc)); }
                }
                acc = 0ull;
            }
        }
        if (valid && sl == 0) for (int w = ((ts + 63) >> 6); w < a.nd1; w++) d1[w] = 0ull;
    }
    ne = (int)from(seg_scan((uint32_t)ne), LPT - 1);
    __syncthreads();

    int picked = 0;
    int32_t *r1 = a.r1 ? a.r1 + tix * (a.steps + 1) : nullptr;
    if (valid && sl == 0 && r1) r1[0] = n1;
    const unsigned long long gtrial = a.trial0 + (unsigned long long)tix;
    uint32_t mti = 0, prc[4] = {0u, 0u, 0u, 0u};
    int s = 0;
    bool alive = valid && a.steps > 0 && n1 > 0;
    while (__any(alive)) {
        // ---- x = _randbelow(n1): draws until one is below n1; a trial that has its x waits for the others of the wave ----
        const int k = 32 - __clz(n1 | 1);
        uint32_t x = 0;
        bool need = alive;
        while (__any(need)) {
            if (need) {
                if ((mti & 3u) == 0u)
                    philox4x32_10(mti >> 2, 0x90000000u, (uint32_t)gtrial, (uint32_t)(gtrial >> 32), a.seed_lo, a.seed_hi, prc);
                const uint32_t y = (mti & 2u) ? ((mti & 1u) ? prc[3] : prc[2]) : ((mti & 1u) ? prc[1] : prc[0]);
                mti++;
                x = y >> (32 - k);
                need = x >= (uint32_t)n1;
            }
        }
        // ---- m = x-th set bit of the degree-1 bitmap, ascending: block, word, bit -------------------------------------------
        uint32_t bc[Q], bt = 0;
#pragma unroll
        for (int q = 0; q < Q; q++) { bc[q] = (uint32_t)blk[sl * Q + q]; bt += bc[q]; }
        const uint32_t binc = seg_scan(bt);
        const int L0 = __ffs((int)seg_bits(__ballot(alive && binc > x))) - 1;            // (alive: n1 > x, so some lane qualifies)
        const int L0s = L0 < 0 ? 0 : L0;
        uint32_t r = x - from(binc - bt, L0s);
        int B0 = L0s * Q;
#pragma unroll
        for (int q = 0; q < Q - 1; q++) {
            const uint32_t cq = from(bc[q], L0s);
            if (B0 == L0s * Q + q && r >= cq) { r -= cq; B0++; }
        }
        // the block's eight sub-blocks of eight words, then the sub-block's eight words (64 bytes of the bitmap, one request)
        const uint32_t sc = sl < 8 ? (sub[(B0 * 8 + sl) >> 1] >> ((sl & 1) * 16)) & 0xFFFFu : 0u;
        const uint32_t sinc = seg_scan(sc);
        const int S0 = __ffs((int)seg_bits(__ballot(alive && sinc > r))) - 1;
        const int S0s = S0 < 0 ? 0 : S0;
        r -= from(sinc - sc, S0s);
        const int wbase = (B0 * 8 + S0s) * 8;
        unsigned long long word = 0ull;
        if (alive && sl < 8 && wbase + sl < a.nd1) word = __hip_atomic_load(&d1[wbase + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t wc = (uint32_t)__popcll(word);
        const uint32_t winc = seg_scan(wc);
        const int W0 = __ffs((int)seg_bits(__ballot(alive && winc > r))) - 1;
        const int W0s = W0 < 0 ? 0 : W0;
        r -= from(winc - wc, W0s);
        int m = 0;
        if (sl == W0s) {
            unsigned long long w = word;
            for (int g = 0; g < 63 && r; g++, r--) w &= w - 1;
            m = (wbase + sl) * 64 + (__ffsll((long long)w) - 1);
        }
        m = (int)from((uint32_t)m, W0s);
        if (!alive || (unsigned)m >= (unsigned)ts) m = 0;                  // (m < ts whenever the counts and the bitmap agree)
        // ---- remove its single VN from its four CNs (PD:769-777) ---------------------------------------------------------------
        const int j = (int)(ldw(m) & kSumMask);
        const unsigned long long row = rows[alive ? j : 0];
        bool plus = false, minus = false;
        if (alive && sl < 4) {
            const int c = (pos_of(j) + sl) * a.cns_pos + (int)((row >> (16 * sl)) & 0xFFFFull);
            const uint32_t nc = (atomicSub(&cn[c], kCntOne + (uint32_t)j) >> kCntShift) - 1u;
            if (c < ts) {
                minus = nc == 0u;                                         // was 1
                plus = nc == 1u;                                          // became 1
                if (plus || minus) {
                    atomicXor(&d1[c >> 6], 1ull << (c & 63));
                    atomicAdd(&blk[c >> a.bshift], plus ? 1 : -1);
                    sub_add(c >> 9, plus ? 1 : -1);
                }
            }
        }
        n1 += __popc(seg_bits(__ballot(plus))) - __popc(seg_bits(__ballot(minus)));
        if (alive) {
            picked++;
            s++;
            if (sl == 0 && r1) r1[s] = n1;
        }
        alive = alive && s < a.steps && n1 > 0;
    }
    // no degree-1 CN left: the count (0) is copied forward, nothing is drawn (PD:765-767)
    if (valid && r1)
        for (int t = s + 1 + sl; t <= a.steps; t += LPT) r1[t] = n1;
    if (valid && sl == 0) {
        int32_t *o = a.out + tix * 4;
        o[0] = ne; o[1] = picked; o[2] = n1; o[3] = picked;
    }
}

}  // namespace

static int launch_peel_pick(const scldpc_code_params *p, int32_t ntrials, const void *d_vn_adj, bool adj16,
                            const uint32_t *d_chan_bits, int32_t total_size, int32_t num_steps,
                            uint32_t *d_mt_state, uint64_t seed, uint64_t trial0,
                            int32_t *d_r1, int64_t *d_moments, int32_t *d_out, const scldpc::Scratch &scratch, void *stream)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (scratch.query) *scratch.query = 0;
    if (!scratch.query && (ntrials < 0 || (ntrials > 0 && (!d_out || !d_vn_adj || !d_chan_bits))))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_peel_pick_device: null buffer or negative ntrials");
    const int n = scldpc::n_of(p), ncn = scldpc::nk_of(p);
    if (total_size < 0 || total_size > ncn || num_steps < 0)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_peel_pick_device: total_size outside [0,%d] or negative steps", ncn);
    if (ntrials <= 0) return SCLDPC_OK;
    if (p->dc > 15 || p->dv > 8 || (int64_t)p->dc * n >= (1ll << kDegShift))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_peel_pick_device: needs dc <= 15, dv <= 8, dc*n < 2^24");
    Args a{};
    a.dv = p->dv; a.vns_pos = p->vns_pos; a.cns_pos = p->cns_pos; a.n = n; a.ncn = ncn; a.total_size = total_size;
    a.steps = num_steps; a.nw = (n + 31) / 32; a.nd1 = (total_size + 63) / 64 + 1;
    a.rng_mode = d_mt_state ? 0 : 1;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.trial0 = trial0;
    a.magic_v = (uint32_t)((1ull << 32) / (uint32_t)p->vns_pos) + 1u;
    for (int64_t q = 0; q <= p->L; q++) {
        const uint64_t x0 = (uint64_t)q * p->vns_pos, x1 = x0 ? x0 - 1 : 0;
        if (((x0 * a.magic_v) >> 32) != (uint64_t)q || ((x1 * a.magic_v) >> 32) != x1 / (uint64_t)p->vns_pos)
            return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_peel_pick_device: reciprocal division inexact");
    }
    // two-level rank-select over the degree-1 bitmap: 64 blocks of 64 * kw words, kw = 1, 2 or 4 words per lane
    a.bshift = 12;
    while (a.bshift < 14 && a.nd1 > (64 << (a.bshift - 6))) a.bshift++;
    if (a.nd1 > (64 << (a.bshift - 6)))
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_peel_pick_device: more than 1048576 pickable CNs");
    // One wave steps one trial through a chain of dependent picks, so throughput = trials in flight.  A workgroup is one
    // wave (up to 32 per CU); what it keeps in LDS decides how many fit: the CN words go to the workspace unless the
    // LDS-resident layout already allows 16 workgroups per CU, and so does the degree-1 bitmap when it is the next obstacle.
    auto layout = [&](bool cn_global, bool d1_global) {
        int off = cn_global ? 0 : (ncn + 3) & ~3;
        a.off_d1 = off; off += d1_global ? 0 : (2 * a.nd1 + 3) & ~3;
        a.off_blk = off; off += 64;
        a.off_mt = off; off += a.rng_mode == 0 ? 624 : 0;
        a.off_sc = off; off += 4;
        return 4u * (size_t)off;
    };
    const size_t want = (size_t)scldpc::kMaxLdsBytes / 16;
    bool gws = false, d1g = false;
    size_t lds_bytes = layout(false, false);
    if (lds_bytes > want) { gws = true; lds_bytes = layout(true, false); }
    if (gws && lds_bytes > want) { d1g = true; lds_bytes = layout(true, true); }
    if (lds_bytes + 64 > (size_t)scldpc::kMaxLdsBytes)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_peel_pick_device: needs %zu B of LDS", lds_bytes);
    if (gws) {
        const size_t cn_bytes = (((size_t)ntrials * ncn * sizeof(uint32_t)) + 255) & ~(size_t)255;
        const size_t need = cn_bytes + (d1g ? (size_t)ntrials * a.nd1 * 8 : 0);
        if (scratch.query) { *scratch.query = need; return SCLDPC_OK; }
        void *ws = nullptr;
        if (int rc = scldpc::take_scratch("scldpc_peel_pick_device", scratch, need, &ws)) return rc;
        a.ws = static_cast<uint32_t *>(ws);
        a.ws_d1 = reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + cn_bytes);
    }
    if (scratch.query) return SCLDPC_OK;
    a.moments = reinterpret_cast<unsigned long long *>(d_moments);
    a.vn_adj = d_vn_adj; a.chan = d_chan_bits; a.mt = d_mt_state; a.r1 = d_r1; a.out = d_out;
    void (*kern)(const Args);
#define PICK(G, D) (p->dv == 4 ? (adj16 ? peel_pick_kernel<4, true, G, D> : peel_pick_kernel<4, false, G, D>) \
                               : (adj16 ? peel_pick_kernel<0, true, G, D> : peel_pick_kernel<0, false, G, D>))
    kern = d1g ? PICK(true, true) : gws ? PICK(true, false) : PICK(false, false);
#undef PICK
    // BASELINE config 3's layout: several trials per wave (the chain waits on DRAM: more chains in flight)
    int tpw = (d1g && adj16 && p->dv == 4 && a.rng_mode == 1 && !d_moments && a.bshift == 12) ? 2 : 1;
    if (const char *v = getenv("SCLDPC_DEBUG_PICK_TPW")) tpw = tpw > 1 ? atoi(v) : 1;                 // A/B, tests: 1, 2 or 4
    if (tpw == 2 || tpw == 4) {
        a.ntrials = ntrials;
        // the CN words through a ring of dv CN positions in LDS (cn_build.hip) where that ring fits
        bool pre = true;
        if (const char *v = getenv("SCLDPC_DEBUG_PICK_PREBUILD")) pre = atoi(v) != 0;                      // A/B, tests
        a.prebuilt = pre && scldpc::cn_build_launch(p, ntrials, static_cast<const uint16_t *>(d_vn_adj), d_chan_bits, a.ws, false, stream) ? 1 : 0;
        kern = tpw == 2 ? peel_pick_multi_kernel<2> : peel_pick_multi_kernel<4>;
        hipLaunchKernelGGL(kern, dim3((ntrials + tpw - 1) / tpw), dim3(kBlock), (size_t)tpw * (64 + 256) * 4, static_cast<hipStream_t>(stream), a);
        SCLDPC_HIP_CHECK(hipGetLastError());
        return SCLDPC_OK;
    }
    if (int rc_ = scldpc::allow_max_lds(reinterpret_cast<const void *>(kern))) return rc_;
    hipLaunchKernelGGL(kern, dim3(ntrials), dim3(kBlock), lds_bytes, static_cast<hipStream_t>(stream), a);
    SCLDPC_HIP_CHECK(hipGetLastError());
    return SCLDPC_OK;
}

extern "C" int scldpc_peel_pick_device(const scldpc_code_params *p, int32_t ntrials,
                                       const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                                       int32_t total_size, int32_t num_steps,
                                       uint32_t *d_mt_state, uint64_t seed, uint64_t trial0,
                                       int32_t *d_r1, int64_t *d_moments, int32_t *d_out, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_peel_pick(p, ntrials, d_vn_adj, false, d_chan_bits, total_size, num_steps, d_mt_state, seed, trial0,
                            d_r1, d_moments, d_out, scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

extern "C" int scldpc_peel_pick_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                             const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                             int32_t total_size, int32_t num_steps,
                                             uint32_t *d_mt_state, uint64_t seed, uint64_t trial0,
                                             int32_t *d_r1, int64_t *d_moments, int32_t *d_out, void *d_workspace,
        uint64_t workspace_bytes, void *stream)
{
    return launch_peel_pick(p, ntrials, d_vn_adj16, true, d_chan_bits, total_size, num_steps, d_mt_state, seed, trial0,
                            d_r1, d_moments, d_out, scldpc::Scratch{d_workspace, workspace_bytes, nullptr}, stream);
}

// workspace of scldpc_peel_pick_device(_adj16) for ntrials trials; rng_mt: the MT19937 state is staged in LDS too
int64_t scldpc_peel_pick_workspace_query(const scldpc_code_params *p, int32_t ntrials, int32_t total_size, int32_t rng_mt)
{
    uint64_t need = 0;
    uint32_t dummy = 0;
    const int rc = launch_peel_pick(p, ntrials, nullptr, true, nullptr, total_size, 0, rng_mt ? &dummy : nullptr, 0, 0,
                                    nullptr, nullptr, nullptr, scldpc::Scratch{nullptr, 0, &need}, nullptr);
    return rc ? (int64_t)rc : (int64_t)need;
}
