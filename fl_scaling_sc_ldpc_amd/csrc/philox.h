// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) — the one
// counter-based generator of the throughput-mode samplers (sampler*.hip, stream_bp.hip, peel_pick.hip).  Not reference
// arithmetic: the reference draws from one sequential glibc / MT19937 stream (glibc_sampler.cpp replays that exactly);
// its known-answer vectors are tested (tests/test_abi.py) and every device sampler has a CPU twin in oracle/.
#pragma once
#include <cstdint>

namespace scldpc_dev {

// The round's two 32x32 -> 64 products are written as 64-bit multiplies: hipcc then emits ONE v_mad_u64_u32 per
// product instead of a v_mul_hi_u32 + v_mul_lo_u32 pair (18 instead of 36 quarter-rate multiplies per call), and the
// three-way XORs are gfx950's v_bitop3_b32 (truth table 0x96), one instruction instead of two.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96); c1 = (uint32_t)p1;
        c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96); c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // namespace scldpc_dev
