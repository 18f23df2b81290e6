// Diagnostic: vector-issue rate of the integer instructions the sampler is made of, at 1 .. 8 waves per SIMD (gfx950).
// Each wave runs N independent chains of one operation; reported: wave-instructions per CU-cycle equivalent (at the wall
// clock) and Ginstr/s chip-wide.  Tells whether "VALU bound" means one wave64 instruction per 4 or per 2 cycles per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    extern __shared__ uint32_t pad[];
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 9u + i;
    uint32_t m = seed | 1u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // inline asm so that the chains are not folded: the instruction named is the instruction issued
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 1) asm volatile("v_lshrrev_b32 %0, 3, %0\n\tv_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 4) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (OP == 5) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
                if (OP == 6) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == 8) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 9) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i];
    if (x == 0x12345u) out[0] = x + pad[0];
}

template <int OP>
void run(const char *name, int instr_per_step)
{
    uint32_t *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        const size_t lds = 160 * 1024 / wg_per_cu - 512;                        // forces the residency
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int grid = 256 * wg_per_cu;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), lds, 0, d, 10, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), lds, 0, d, iters, 3u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)grid * 4 * iters * 64.0 * instr_per_step;  // wave-instructions
        printf("%-28s %d wave/SIMD: %8.3f ms  %8.1f G wave-instr/s  = %.3f per SIMD-cycle @2.4GHz\n", name, wg_per_cu, ms,
               winstr / ms / 1e6, winstr / ms / 1e6 / (1024 * 2.4));
    }
    hipFree(d);
}

int main()
{
    run<0>("v_add_u32", 1);
    run<1>("lshr+xor", 2);
    run<2>("v_mul_hi_u32", 1);
    run<3>("v_mul_lo_u32", 1);
    run<4>("v_and_or_b32", 1);
    run<5>("v_bfe_u32", 1);
    run<6>("v_lshl_add_u32", 1);
    run<7>("v_cndmask_b32", 1);
    run<8>("v_mov_b32_dpp row_shr", 1);
    run<9>("v_pk_add_u16", 1);
    return 0;
}
