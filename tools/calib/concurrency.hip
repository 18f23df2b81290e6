// Diagnostic: do two kernels launched on two HIP streams run side by side on this box?  Each kernel is a small grid of
// workgroups that spin for a fixed wall time (s_memrealtime, 100 MHz); if the pair takes one spin, the streams overlap; if it
// takes two, something serialises them (then no co-residency experiment means anything).  Also reports, per CU, how many
// workgroups of each kernel it hosted (HW_REG_HW_ID), i.e. how the dispatcher places two grids that could share every CU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long ticks, uint32_t *where, int tag)
{
    extern __shared__ uint32_t pad[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x] = (hw & 0xFFFFFu) | ((xcc & 0xFu) << 20) | ((uint32_t)tag << 28);
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (pad[0] == 0x12345u && ticks == 1) where[0] = 1;
}

int main()
{
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    uint32_t *w1, *w2;
    const int g1 = 256, g2 = 1024;
    hipMalloc(&w1, g1 * 4); hipMalloc(&w2, g2 * 4);
    const unsigned long long ticks = 500000;      // 5 ms at 100 MHz
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 2; rep++) {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipStreamWaitEvent(s1, e0, 0); hipStreamWaitEvent(s2, e0, 0);
        hipLaunchKernelGGL(spin, dim3(g1), dim3(1024), 56 * 1024, s1, ticks, w1, 1);     // "sampler": 16 waves, 56 KB
        hipLaunchKernelGGL(spin, dim3(g2), dim3(256), 23 * 1024, s2, ticks, w2, 2);      // "decoder": 4 waves, 23 KB, four per CU
        hipEvent_t d1, d2;
        hipEventCreate(&d1); hipEventCreate(&d2);
        hipEventRecord(d1, s1); hipEventRecord(d2, s2);
        hipStreamWaitEvent(0, d1, 0); hipStreamWaitEvent(0, d2, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("two kernels of 5 ms each on two streams: %.2f ms  (%s)\n", ms, ms < 7.5f ? "side by side" : "one after the other");
    }
    std::vector<uint32_t> h1(g1), h2(g2);
    hipMemcpy(h1.data(), w1, g1 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), w2, g2 * 4, hipMemcpyDeviceToHost);
    // HW_ID: [11:8] CU id, [15:13] SE id (gfx9 layout); together with the XCC id a key per CU
    auto cu_key = [](uint32_t v) { return ((v >> 20) & 0xF) * 4096u + ((v >> 13) & 0x7) * 64u + ((v >> 8) & 0xF) + ((v >> 12) & 1) * 16u; };
    std::vector<int> c1(65536, 0), c2(65536, 0);
    for (auto v : h1) c1[cu_key(v)]++;
    for (auto v : h2) c2[cu_key(v)]++;
    int hist[16][16] = {};
    int cus = 0;
    for (int k = 0; k < 65536; k++) if (c1[k] || c2[k]) { cus++; hist[c1[k] > 15 ? 15 : c1[k]][c2[k] > 15 ? 15 : c2[k]]++; }
    printf("CUs seen: %d;  #CUs by (sampler-like workgroups, decoder-like workgroups) hosted in the last run:\n", cus);
    for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) if (hist[a][b]) printf("   (%d, %d): %d CUs\n", a, b, hist[a][b]);
    return 0;
}
