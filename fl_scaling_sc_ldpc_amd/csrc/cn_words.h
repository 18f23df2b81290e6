// CN-word policies shared by the flooding decoders (full_bp.hip, sw_bp.hip): how one check node's residual state —
// cnt = #erased neighbours and a fold of their ids (== the id when cnt == 1) — is stored and updated.
//   WideT<G> 32 bit [cnt:4 | deg:4 | Σ global VN id:24]; G = true: the words live in a global-memory workspace
//   Packed   16 bit [cnt:4 | ⊕ local VN id:12], two CNs per 32-bit word; needs dv·vns_pos <= 4096
#pragma once
#include "kernel_util.h"

namespace scldpc_dev {

// what the policies need to know about the ensemble
struct Geo {
    int vns_pos;
    uint32_t magic_v, magic_c;      // floor(2^32/d)+1: x/d == umulhi(x, magic) for the ranges used (checked on the host)
};

struct Vn { int j, pos, t; };

// ---- CN-word policies --------------------------------------------------------------------------
// G = false: the words live in LDS.  G = true: they live in a global-memory workspace (one nk-word slice per trial,
// L2-resident while hot) for ensembles whose CN words exceed the LDS — N >= 2500 at (4,8), e.g. bp_traj's shipped
// Def_M = 2500.  Same algorithm with global atomics; plain reads go past the CU's L1 (agent-scope loads) so that they
// see what the atomics did in L2.  U, the frontier queues and the scan bitmap stay in LDS either way.
template <bool G>
struct WideT {
    static constexpr bool kGlobal = G;
    static constexpr bool kHasDeg = true;          // the CN degree rides in the word (trajectory mode)
    static constexpr bool kPacked16 = false;
    static __host__ __device__ int lds_words(int nk) { return G ? 0 : nk; }
    static __host__ __device__ int words(int nk) { return nk; }
    static __device__ __forceinline__ uint32_t ld(const uint32_t *st, int c)
    {
        if constexpr (G) return __hip_atomic_load(&st[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else             return st[c];
    }
    static __device__ __forceinline__ void add(uint32_t *st, int c, const Vn &v, int, int, bool erased, bool deg)
    {
        atomicAdd(&st[c], (deg ? kDegOne : 0u) + (erased ? kCntOne + (uint32_t)v.j : 0u));
    }
    static __device__ __forceinline__ uint32_t cnt(const uint32_t *st, int c) { return ld(st, c) >> kCntShift; }
    static __device__ __forceinline__ uint32_t deg(const uint32_t *st, int c) { return (ld(st, c) >> kDegShift) & kDegMask; }
    // the single erased neighbour of c (valid only if cnt == 1 in the word that was read); -1 otherwise
    static __device__ __forceinline__ int lone_vn(const uint32_t *st, int c, const Geo &)
    {
        const uint32_t w = ld(st, c);
        return (w >> kCntShift) == 1u ? (int)(w & kSumMask) : -1;
    }
    static __device__ __forceinline__ uint32_t remove_cnt(uint32_t *st, int c, const Vn &v, int, int)   // returns old cnt
    {
        return atomicSub(&st[c], kCntOne + (uint32_t)v.j) >> kCntShift;
    }
    static __device__ __forceinline__ void remove_fold(uint32_t *, int, const Vn &, int, int) {}
    // partner of v at CN c if cnt == 2, else -1
    static __device__ __forceinline__ int partner(const uint32_t *st, int c, const Vn &v, int, const Geo &)
    {
        const uint32_t w = ld(st, c);
        return (w >> kCntShift) == 2u ? (int)((w & kSumMask) - (uint32_t)v.j) : -1;
    }
};
using Wide = WideT<false>;
using WideG = WideT<true>;

struct Packed {     // two CNs per 32-bit word; CN c lives in half (c & 1) of word c >> 1
    static constexpr bool kGlobal = false;
    static constexpr bool kHasDeg = false;         // trajectory mode keeps the few degrees it needs in a side array
    static constexpr bool kPacked16 = true;        // peel_fixpoint.h applies
    static __host__ __device__ int lds_words(int nk) { return (nk + 1) / 2; }
    static __host__ __device__ int words(int nk) { return (nk + 1) / 2; }
    static __device__ __forceinline__ void add(uint32_t *st, int c, const Vn &v, int i, int V, bool erased, bool)
    {
        if (!erased) return;
        const int sh = (c & 1) * 16;
        atomicAdd(&st[c >> 1], 0x1000u << sh);
        atomicXor(&st[c >> 1], (uint32_t)(i * V + v.t) << sh);
    }
    static __device__ __forceinline__ uint32_t half(const uint32_t *st, int c) { return (st[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu; }
    static __device__ __forceinline__ uint32_t cnt(const uint32_t *st, int c) { return half(st, c) >> 12; }
    static __device__ __forceinline__ uint32_t deg(const uint32_t *, int) { return 0; }
    static __device__ __forceinline__ int lid_to_vn(int c, uint32_t lid, const Geo &a)
    {
        const int V = a.vns_pos;
        int i = 0;
        while ((i + 1) * V <= (int)lid) i++;                         // < dv steps
        const int pos_c = (int)__umulhi((uint32_t)c, a.magic_c);
        return (pos_c - i) * V + ((int)lid - i * V);
    }
    static __device__ __forceinline__ int lone_vn(const uint32_t *st, int c, const Geo &a)
    {
        // cnt and fold are updated by two atomics (count first, fold second).  The only update a
        // CN of the CURRENT frontier can see in its round is the removal of its own VN, which drops cnt to 0
        // before it touches the fold — so a frontier CN read with cnt == 1 carries exactly its one neighbour.
        const uint32_t h = half(st, c);
        return (h >> 12) == 1u ? lid_to_vn(c, h & 0xFFFu, a) : -1;
    }
    static __device__ __forceinline__ uint32_t remove_cnt(uint32_t *st, int c, const Vn &, int, int)
    {
        const int sh = (c & 1) * 16;
        return (atomicSub(&st[c >> 1], 0x1000u << sh) >> (sh + 12)) & 0xFu;
    }
    static __device__ __forceinline__ void remove_fold(uint32_t *st, int c, const Vn &v, int i, int V)
    {
        atomicXor(&st[c >> 1], (uint32_t)(i * V + v.t) << ((c & 1) * 16));
    }
    static __device__ __forceinline__ int partner(const uint32_t *st, int c, const Vn &v, int i, const Geo &a)
    {
        const uint32_t h = half(st, c);
        return (h >> 12) == 2u ? lid_to_vn(c, (h & 0xFFFu) ^ (uint32_t)(i * a.vns_pos + v.t), a) : -1;
    }
};


}  // namespace scldpc_dev
