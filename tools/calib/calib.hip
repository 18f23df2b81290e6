// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access patterns (diagnostic tool, not part
// of libscldpc_hip.so): known byte counts moved as (a) a wide coalesced stream, 16 B per lane, (b) random 8-byte row
// gathers (the decoders' VN rows), (c) random 16-byte row gathers (the CN rows), (d) 8-byte-per-lane coalesced stores
// (the sampler's rows).  MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of the bytes of (a); "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  tools/calib/run.sh runs it under
// `rocprofv3 --pmc` and tools/summarize_prof.py turns the counters into bytes-per-access factors.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void calib_stream16(const uint4 *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every lane makes `count` dependent-free random reads of W-byte rows (W = 8 or 16) out of `nrows`
template <class T>
__global__ void calib_gather(const T *p, uint32_t nrows, int count, uint32_t *sink)
{
    uint32_t acc = 0, h = mix(blockIdx.x * 1024u + threadIdx.x + 1u);
    for (int k = 0; k < count; k++) {
        h = mix(h + 0x9E3779B9u);
        const T v = p[h % nrows];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// every lane reads `count` random ALIGNED 128-byte rows in full (8 x 16 B): one L2 line each — does the L2 ask the
// fabric for it in one request or in two 64-byte ones?
__global__ void calib_gather_line(const uint4 *p, uint32_t nlines, int count, uint32_t *sink)
{
    uint32_t acc = 0, h = mix(blockIdx.x * 1024u + threadIdx.x + 77u);
    for (int k = 0; k < count; k++) {
        h = mix(h + 0x9E3779B9u);
        const uint4 *row = p + (size_t)(h % nlines) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint4 v = row[i]; acc ^= v.x ^ v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void calib_store8(uint2 *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint2((uint32_t)i, 7u);
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)8 << 30;                 // 8 GiB: far beyond L2 (32 MiB) and the Infinity Cache (256 MiB)
    void *buf; uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const int blocks = 2048, threads = 256, count = 256;
    // (a) 4 GiB streamed at 16 B per lane
    hipLaunchKernelGGL(calib_stream16, dim3(blocks), dim3(threads), 0, 0, (const uint4 *)buf, ((size_t)4 << 30) / 16, sink);
    // (b) 2048*256*256 = 134 217 728 random 8-byte rows = 1 GiB of requested bytes
    hipLaunchKernelGGL(calib_gather<uint2>, dim3(blocks), dim3(threads), 0, 0, (const uint2 *)buf, (uint32_t)(bytes / 8 - 1), count, sink);
    // (c) as many random 16-byte rows = 2 GiB of requested bytes
    hipLaunchKernelGGL(calib_gather<uint4>, dim3(blocks), dim3(threads), 0, 0, (const uint4 *)buf, (uint32_t)(bytes / 16 - 1), count, sink);
    // (c') 2048*256*32 = 16 777 216 random full 128-byte lines = 2 GiB
    hipLaunchKernelGGL(calib_gather_line, dim3(blocks), dim3(threads), 0, 0, (const uint4 *)buf, (uint32_t)(bytes / 128 - 1), 32, sink);
    // (d) 4 GiB stored at 8 B per lane
    hipLaunchKernelGGL(calib_store8, dim3(blocks), dim3(threads), 0, 0, (uint2 *)buf, ((size_t)4 << 30) / 8);
    CHECK(hipDeviceSynchronize());
    printf("calib: stream16 bytes=%zu gather8 n=%zu gather16 n=%zu store8 bytes=%zu\n", (size_t)4 << 30,
           (size_t)blocks * threads * count, (size_t)blocks * threads * count, (size_t)4 << 30);
    return 0;
}
