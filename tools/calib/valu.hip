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
    uint32_t m = seed | 1u, m2 = seed * 3u + 5u, m3 = (seed & 3u) + 1u, sd32 = 0;
    unsigned long long smask = 0x5555555555555555ull + seed, sdum = 0;
    asm volatile("" : "+s"(smask));
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
                if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(smask));
                if (OP == 8) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 9) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 10) { unsigned long long p_; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p_), "=s"(sdum) : "v"(a[i]), "v"(m)); a[i] = (uint32_t)p_; }
                if (OP == 11) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(m), "v"(m2));
                if (OP == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 13) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 14) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 16) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == 17) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(m));
                if (OP == 18) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t)smask));
                if (OP == 19) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(m2));
                if (OP == 20) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(m2));
                if (OP == 21) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(m2));
                if (OP == 22) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t)smask));
                if (OP == 23) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(m));
                if (OP == 24) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(m2));
                if (OP == 25) asm volatile("v_cmp_lt_u32 %1, %0, %2\n\tv_addc_co_u32 %0, %1, %0, %2, %1" : "+v"(a[i]), "=s"(sdum) : "v"(m));
                if (OP == 26) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 27) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 28) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (OP == 29) asm volatile("v_and_b32 %0, 0xf0f0f0f, %0" : "+v"(a[i]));
                if (OP == 30) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 31) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 32) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m3));
                if (OP == 33) asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 34) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == 35) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t)smask));
                if (OP == 36) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t)smask));
                if (OP == 37) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
                if (OP == 38) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 39) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 40) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
                if (OP == 41) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (OP == 42) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
                if (OP == 43) asm volatile("v_add_u32 %0, 7, %0" : "+v"(a[i]));
                if (OP == 44) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 45) asm volatile("v_readlane_b32 %1, %0, 3\n\tv_add_u32 %0, %1, %0" : "+v"(a[i]), "=s"(sd32));
                if (OP == 46) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(m), "v"(m2));
                if (OP == 47) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == 48) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == 49) asm volatile("v_and_b32 %0, 15, %0\n\tv_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i];
    if (x == 0x12345u) out[0] = x + pad[0] + (uint32_t)sdum + sd32;
}

// LDS instruction rates at 8 waves per SIMD: every lane issues N independent operations at pseudo-random word addresses
template <int OP>
__global__ __launch_bounds__(256) void kl(uint32_t *out, int iters, uint32_t seed)
{
    extern __shared__ uint32_t sm[];
    const int words = 4096;                                                    // 16 KB, as the sampler's histogram
    for (int i = threadIdx.x; i < words; i += 256) sm[i] = i;
    __syncthreads();
    uint32_t h = seed + threadIdx.x * 2654435761u, acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            h = h * 1664525u + 1013904223u;
            const uint32_t w = (h >> 12) & (words - 1);
            if (OP == 0) acc += sm[w];                                          // ds_read_b32
            if (OP == 1) acc += atomicAdd(&sm[w], 1u << (h & 12));              // ds_add_rtn_u32
            if (OP == 2) reinterpret_cast<uint16_t *>(sm)[w * 2 + (h & 1)] = (uint16_t)h;   // ds_write_b16
            if (OP == 3) acc += reinterpret_cast<uint16_t *>(sm)[w * 2 + (h & 1)];           // ds_read_u16
            if (OP == 4) { const uint4 q = reinterpret_cast<uint4 *>(sm)[w >> 2]; acc += q.x ^ q.w; }      // ds_read_b128
            if (OP == 5) atomicAdd(&sm[w], 1u);                                 // ds_add_u32 (no return)
            if (OP == 6) { const uint2 q = reinterpret_cast<uint2 *>(sm)[w >> 1]; acc += q.x ^ q.y; }      // ds_read_b64
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int OP>
void runl(const char *name)
{
    uint32_t *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500, wg_per_cu = 8;
    const size_t lds = 16384;
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(kl<OP>, dim3(grid), dim3(256), lds, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kl<OP>, dim3(grid), dim3(256), lds, 0, d, iters, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)grid * 4 * iters * 8.0;
    printf("LDS %-24s 8 wave/SIMD (random words of 16 KB): %8.3f ms  %8.1f G wave-instr/s = %.3f per CU-cycle @2.4GHz (incl. ~3 VALU per op)\n",
           name, ms, winstr / ms / 1e6, winstr / ms / 1e6 / (256 * 2.4));
    hipFree(d);
}

void lds_bench()
{
    runl<0>("ds_read_b32");
    runl<1>("ds_add_rtn_u32");
    runl<2>("ds_write_b16");
    runl<3>("ds_read_u16");
    runl<4>("ds_read_b128");
    runl<5>("ds_add_u32");
    runl<6>("ds_read_b64");
}

template <int OP>
void run(const char *name, int instr_per_step)
{
    uint32_t *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wg_per_cu = 2; wg_per_cu <= 8; wg_per_cu *= 4) {
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

int main(int argc, char **argv)
{
    const bool all = argc > 1;
    if (all) run<0>("v_add_u32", 1);
    if (all) run<1>("lshr+xor", 2);
    if (all) run<2>("v_mul_hi_u32", 1);
    if (all) run<3>("v_mul_lo_u32", 1);
    if (all) run<4>("v_and_or_b32", 1);
    if (all) run<5>("v_bfe_u32", 1);
    if (all) run<6>("v_lshl_add_u32", 1);
    if (all) run<7>("v_cndmask_b32", 1);
    if (all) run<8>("v_mov_b32_dpp row_shr", 1);
    if (all) run<9>("v_pk_add_u16", 1);
    if (all) run<10>("v_mad_u64_u32", 1);
    if (all) run<11>("v_bitop3_b32", 1);
    if (all) run<12>("v_mul_u32_u24", 1);
    if (all) run<13>("v_mul_hi_u32_u24", 1);
    if (all) run<14>("v_add_u32_dpp row_shr", 1);
    if (all) run<15>("v_and_b32", 1);
    if (all) run<16>("v_lshlrev_b32 (inline const)", 1);
    if (all) run<17>("v_add_u32_sdwa WORD_1", 1);
    if (all) run<18>("v_mbcnt_lo_u32_b32", 1);
    if (all) run<19>("v_or3_b32", 1);
    if (all) run<20>("v_add3_u32", 1);
    if (all) run<21>("v_mad_u32_u24", 1);
    if (all) run<22>("v_xor_b32 (sgpr src)", 1);
    if (all) run<23>("v_alignbit_b32", 1);
    if (all) run<24>("v_perm_b32", 1);
    if (all) run<25>("v_cmp + v_addc_co", 2);
    if (all) run<26>("v_sub_u32", 1);
    if (all) run<27>("v_max_u32", 1);
    if (all) run<28>("v_lshrrev_b32 (vgpr shift)", 1);
    if (all) run<29>("v_and_b32 (literal)", 1);
    run<30>("v_mov_b32", 1);
    run<31>("v_or_b32", 1);
    run<32>("v_lshlrev_b32 (vgpr shift)", 1);
    run<33>("v_lshlrev 1 + v_xor", 2);
    run<34>("v_cmp(vcc) + v_cndmask(vcc)", 2);
    run<35>("v_and_b32 (sgpr src)", 1);
    run<36>("v_add_u32 (sgpr src)", 1);
    run<37>("v_not_b32", 1);
    run<38>("v_min_u32", 1);
    run<39>("v_lshl_or_b32", 1);
    run<40>("v_ashrrev_i32 const", 1);
    run<41>("v_subrev_u32", 1);
    run<42>("v_add_u32 (literal)", 1);
    run<43>("v_add_u32 (inline const)", 1);
    run<44>("v_xor_b32", 1);
    run<45>("v_readlane + v_add(sgpr)", 2);
    run<46>("v_bfi_b32", 1);
    run<47>("v_lshrrev_b32 const", 1);
    run<48>("v_cmp_ne + v_addc_co", 2);
    run<49>("v_and const + v_add", 2);
    lds_bench();
    return 0;
}
