// Device helpers shared by the decoder kernels (gfx950, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace scldpc_dev {

// One 32-bit word per CN: [cnt:4 | deg:4 | idsum:24].  cnt = #erased neighbours, deg = CN degree
// (trajectory mode only), idsum = Σ ids of the erased neighbours (== the id when cnt == 1).
constexpr uint32_t kCntShift = 28, kDegShift = 24;
constexpr uint32_t kCntOne = 1u << kCntShift, kDegOne = 1u << kDegShift;
constexpr uint32_t kSumMask = (1u << kDegShift) - 1, kDegMask = 0xFu;
constexpr uint32_t kInvalid = 0xFFFFFFFFu;

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// CN ids of the dv edges of VN j (VNdegree[j][1..dv], BPF:87).  DV = 4: one 16-byte load.
template <int DV>
__device__ __forceinline__ void load_adj(const int32_t *adj, int dv, int j, int32_t (&c)[8])
{
    if constexpr (DV == 4) {
        const int4 v = reinterpret_cast<const int4 *>(adj)[j];
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    } else {
        for (int i = 0; i < dv; i++) c[i] = adj[(size_t)j * dv + i];
    }
}

}  // namespace scldpc_dev
