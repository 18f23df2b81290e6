/*
 * A C caller of libscldpc_hip.so: the frame loop of main_terminated (BPF:2117-2144) for one ε point, batched on the
 * device — generate_code + channel_doped (scldpc_sample_philox_device), decodeBP (scldpc_full_bp_device),
 * plr_computation + willIstop (scldpc_accumulate_run_device) — exactly as INTEGRATION.md §2a shows it to a maintainer
 * of the reference.  Plain C99 + the HIP runtime API; built and run by tests/test_gpu_c_caller.py, which compares the
 * line it prints with the same loop driven through Python.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_caller/frame_loop.c \
 *       -L fl_scaling_sc_ldpc_amd -lscldpc_hip -L /opt/rocm/lib -lamdhip64 -o frame_loop
 *   ./frame_loop L N eps max_it numero_frame numero_frame_err seed [batch]
 */
#include "scldpc.h"
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_LIB(x) do { if ((x) != SCLDPC_OK) { fprintf(stderr, "%s: %s\n", #x, scldpc_last_error()); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s L N eps max_it numero_frame numero_frame_err seed [batch]\n", argv[0]); return 1; }
    const int Def_dv = 4, Def_dc = 8;
    const int L = atoi(argv[1]), VNsPos = atoi(argv[2]);
    const double epsilon = atof(argv[3]);
    const int MaxNumIt = atoi(argv[4]), numero_frame = atoi(argv[5]), numero_frame_err = atoi(argv[6]);
    const uint64_t seed = strtoull(argv[7], 0, 10);
    const int B = argc > 8 ? atoi(argv[8]) : 256;
    const int sim = 0;

    scldpc_code_params P = { Def_dv, Def_dc, L, VNsPos * Def_dv / Def_dc, VNsPos };
    const size_t n = (size_t)VNsPos * L, nw = (n + 31) / 32;
    if (scldpc_abi_version() != SCLDPC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    /* caller-owned scratch: only ensembles beyond the LDS budget need any (0 at this size) */
    const int64_t ws_sample = scldpc_workspace_bytes(SCLDPC_WS_SAMPLE, &P, B, 0, 0);
    const int64_t ws_decode = scldpc_workspace_bytes(SCLDPC_WS_FULL_BP, &P, B, 0, 0);
    if (ws_sample < 0 || ws_decode < 0) { fprintf(stderr, "%s\n", scldpc_last_error()); return 3; }
    const size_t ws_bytes = (size_t)(ws_sample > ws_decode ? ws_sample : ws_decode);

    int32_t *d_adj, *d_cnt; uint32_t *d_ch; int64_t *d_run; void *d_ws = NULL;
    int64_t run[SCLDPC_NRUN];
    CHECK_HIP(hipMalloc((void **)&d_adj, (size_t)B * n * Def_dv * 4));
    CHECK_HIP(hipMalloc((void **)&d_ch, (size_t)B * nw * 4));
    CHECK_HIP(hipMalloc((void **)&d_cnt, (size_t)B * SCLDPC_NCOUNTERS * 4));
    CHECK_HIP(hipMalloc((void **)&d_run, sizeof run));
    if (ws_bytes) CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemset(d_run, 0, sizeof run));

    int f;
    for (f = 0; f < numero_frame; f += B) {                       /* loop on simulated frames, B at a time */
        const int nb = numero_frame - f < B ? numero_frame - f : B;
        CHECK_LIB(scldpc_sample_philox_device(&P, seed, ((uint64_t)sim << 40) + (uint64_t)f, nb, epsilon, 0, NULL,
                                              d_adj, d_ch, d_ws, ws_bytes, NULL));
        CHECK_LIB(scldpc_full_bp_device(&P, nb, d_adj, d_ch, MaxNumIt, /*is_term*/ 1, d_cnt, NULL, 0, NULL,
                                        d_ws, ws_bytes, NULL));
        CHECK_LIB(scldpc_accumulate_run_device(nb, d_cnt, numero_frame_err, d_run, NULL));
        CHECK_HIP(hipMemcpy(run, d_run, sizeof run, hipMemcpyDeviceToHost));
        if (run[SCLDPC_R_FRAME_ERR] >= numero_frame_err) break;   /* willIstop: the counters stop at the tripping frame */
    }
    /* what risultati() prints for the point (BPF:499-515) */
    printf("%f %e %e %e %e %e %e %zu %d %lld %lld %lld %lld %lld %lld %lld\n", epsilon,
           (double)run[SCLDPC_R_USERS_ERR] / n / run[SCLDPC_R_FRAMES], (double)run[SCLDPC_R_FRAME_ERR] / run[SCLDPC_R_FRAMES],
           (double)run[SCLDPC_R_BLOCK_ERR] / L / run[SCLDPC_R_FRAMES],
           (double)run[SCLDPC_R_USERS_ERR_EXP] / n / run[SCLDPC_R_FRAMES],
           (double)run[SCLDPC_R_FRAME_ERR_EXP] / run[SCLDPC_R_FRAMES],
           (double)run[SCLDPC_R_BLOCK_ERR_EXP] / L / run[SCLDPC_R_FRAMES], n, L, (long long)run[SCLDPC_R_FRAMES],
           (long long)run[SCLDPC_R_USERS_ERR], (long long)run[SCLDPC_R_FRAME_ERR], (long long)run[SCLDPC_R_BLOCK_ERR],
           (long long)run[SCLDPC_R_USERS_ERR_EXP], (long long)run[SCLDPC_R_FRAME_ERR_EXP], (long long)run[SCLDPC_R_BLOCK_ERR_EXP]);

    /* struct / enum layout as this translation unit sees it (compared with the Python binding's view) */
    printf("layout %zu %zu %zu %d %d %d %d\n", sizeof(scldpc_code_params), offsetof(scldpc_code_params, cns_pos),
           offsetof(scldpc_code_params, vns_pos), SCLDPC_NCOUNTERS, SCLDPC_NRUN, SCLDPC_C_CHANNEL_ERASURES, SCLDPC_R_FRAMES);
    hipFree(d_adj); hipFree(d_ch); hipFree(d_cnt); hipFree(d_run); if (d_ws) hipFree(d_ws);
    return 0;
}
