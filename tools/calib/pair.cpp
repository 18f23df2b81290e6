// Diagnostic: the headline's two kernels as PERSISTENT launches on two NATIVE HIP streams (no PyTorch in the process) — does a
// sampler workgroup (56 KB, 16 waves) beside k decoder workgroups (23 KB, 4 waves each) on every CU deliver more than one
// kernel after the other?  tools/calib/concurrency.hip shows that two such grids do run side by side and are dealt evenly
// (one + four per CU); this measures what the real kernels make of it.  Uses the C-ABI only.
//   hipcc -O2 -I include tools/calib/pair.cpp -L fl_scaling_sc_ldpc_amd -lscldpc_hip -o pair
#include "scldpc.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { if ((x) != 0) { fprintf(stderr, "fail %s: %s\n", #x, scldpc_last_error()); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32768;
    scldpc_code_params P = {4, 8, 50, 500, 1000};
    const size_t n = 50000, nk = 26500, nw = (n + 31) / 32;
    uint16_t *adj[2], *cn[2]; uint32_t *ch[2]; int32_t *cnt;
    for (int b = 0; b < 2; b++) {
        hipMalloc((void **)&adj[b], (size_t)B * n * 4 * 2); hipMalloc((void **)&cn[b], (size_t)B * nk * 8 * 2);
        hipMalloc((void **)&ch[b], (size_t)B * nw * 4);
    }
    hipMalloc((void **)&cnt, (size_t)B * 8 * 4);
    hipStream_t sa, sb;
    hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto setgrid = [](int gs, int gd) {
        if (gs) { char v[32]; snprintf(v, 32, "%d", gs); setenv("SCLDPC_DEBUG_GRID_SAMPLER", v, 1); } else unsetenv("SCLDPC_DEBUG_GRID_SAMPLER");
        if (gd) { char v[32]; snprintf(v, 32, "%d", gd); setenv("SCLDPC_DEBUG_GRID_DECODER", v, 1); } else unsetenv("SCLDPC_DEBUG_GRID_DECODER");
    };
    auto sample = [&](int b, uint64_t t0, hipStream_t s) { CK(scldpc_sample_philox_device_cn16(&P, 1, t0, B, 0.48, 0, nullptr, adj[b], cn[b], ch[b], s)); };
    auto decode = [&](int b, hipStream_t s) { CK(scldpc_full_bp_fixpoint_device_cn16(&P, B, adj[b], cn[b], ch[b], 1, cnt, nullptr, s)); };
    auto timed = [&](const char *name, auto fn) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipStreamWaitEvent(sa, e0, 0); hipStreamWaitEvent(sb, e0, 0);
            fn();
            hipEvent_t da, db;
            hipEventCreate(&da); hipEventCreate(&db);
            hipEventRecord(da, sa); hipEventRecord(db, sb);
            hipStreamWaitEvent(0, da, 0); hipStreamWaitEvent(0, db, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-72s %8.3f ms / %d trials = %.3f M trials/s\n", name, best, B, B / best / 1e3);
        return best;
    };
    setgrid(0, 0);
    sample(0, 0, sa); sample(1, B, sa);
    hipDeviceSynchronize();
    const float s0 = timed("sampler alone (default launch)", [&] { sample(1, 2 * B, sa); });
    const float d0 = timed("decoder alone (default launch)", [&] { decode(0, sa); });
    timed("default launches, two streams", [&] { decode(0, sa); sample(1, 3 * B, sb); });
    printf("serial: %.2f ms\n", s0 + d0);
    for (int gs = 1; gs <= 2; gs++)
        for (int gd = 2; gd <= 6; gd++) {
            if (gs * 56 + gd * 23 > 160 || gs * 16 + gd * 4 > 32) continue;
            setgrid(256 * gs, 256 * gd);
            char name[128];
            snprintf(name, sizeof name, "persistent pair: %d sampler + %d decoders per CU, decoder launched first", gs, gd);
            const float t = timed(name, [&] { decode(0, sa); sample(1, 4 * B, sb); });
            snprintf(name, sizeof name, "persistent pair: %d sampler + %d decoders per CU, sampler launched first", gs, gd);
            const float t2 = timed(name, [&] { sample(1, 5 * B, sb); decode(0, sa); });
            printf("%-72s x%.3f of serial\n", "", (s0 + d0) / (t < t2 ? t : t2));
        }
    return 0;
}
