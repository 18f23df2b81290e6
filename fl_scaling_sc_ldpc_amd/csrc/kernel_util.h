// Device helpers shared by the decoder kernels (gfx950, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// In-kernel phase stamps — diagnostic build only (make -C csrc stamps).  The product build compiles them out.
#ifdef SCLDPC_STAMPS
extern __device__ long long *g_scldpc_stamps;          // [block][16] cycle sums, written by lane 0 of wave 0
#define STAMP_DECL long long st_acc[16] = {0}; long long st_last = (long long)__builtin_amdgcn_s_memtime();
#define STAMP(k) do { if (threadIdx.x == 0) { long long t_ = (long long)__builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_last; st_last = t_; } } while (0)
#define STAMP_FLUSH() do { if (threadIdx.x == 0 && g_scldpc_stamps) for (int k_ = 0; k_ < 16; k_++) g_scldpc_stamps[(size_t)blockIdx.x * 16 + k_] += st_acc[k_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(k) do {} while (0)
#define STAMP_FLUSH() do {} while (0)
#endif

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

// Inclusive scan over the 64 lanes of a wave with DPP moves (no LDS round trips): row_shr:1,2,4,8 inside
// each row of 16 lanes, then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.  Lane 63 = total.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

// CN ids of the dv edges of VN j (VNdegree[j][1..dv], BPF:87).
//   A16 = false: int32 [n][dv] global CN ids; dv = 4: one 16-byte load.
//   A16 = true : uint16 [n][dv] CN index local to its position; edge i of a VN at position pos lands in CN
//                position pos+i (BPF:1712), so the global id is (pos+i)*cns_pos + local; dv = 4: one 8-byte load.
template <int DV, bool A16>
__device__ __forceinline__ void load_adj(const void *adj, int dv, int j, int pos, int cns_pos, int32_t (&c)[8])
{
    if constexpr (A16) {
        if constexpr (DV == 4) {
            const uint2 v = reinterpret_cast<const uint2 *>(adj)[j];
            c[0] = (pos + 0) * cns_pos + (int)(v.x & 0xFFFFu); c[1] = (pos + 1) * cns_pos + (int)(v.x >> 16);
            c[2] = (pos + 2) * cns_pos + (int)(v.y & 0xFFFFu); c[3] = (pos + 3) * cns_pos + (int)(v.y >> 16);
        } else {
            const uint16_t *r = reinterpret_cast<const uint16_t *>(adj) + (size_t)j * dv;
            for (int i = 0; i < dv; i++) c[i] = (pos + i) * cns_pos + (int)r[i];
        }
    } else {
        if constexpr (DV == 4) {
            const int4 v = reinterpret_cast<const int4 *>(adj)[j];
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        } else {
            const int32_t *r = reinterpret_cast<const int32_t *>(adj) + (size_t)j * dv;
            for (int i = 0; i < dv; i++) c[i] = r[i];
        }
    }
}

}  // namespace scldpc_dev
