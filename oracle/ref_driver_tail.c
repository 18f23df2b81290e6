/*
 * TEST INFRASTRUCTURE — per-trial deterministic driver around the *real* reference
 * BP simulators (BPF / BPW / BPT of SURVEY.md).  This text is appended, on a pipe,
 * behind the reference translation unit (see oracle/Makefile: `sed … ref.c ; cat
 * ref_driver_tail.c | gcc -x c -Dmain=ref_main -`), so it sees the reference's
 * file-scope globals (VNdegree, CNdegree, LLRsChannel, VNerased, sim, MaxNumIt …).
 * No reference source is copied to disk; the only output is a binary in oracle/_ref/.
 *
 * What it replays (BPF:2111-2144): for trial t
 *     sim=0; inizio_sim();            -- BPF:286-318, resets counters AND perm_code
 *     srandom(seed0+t);               -- self-contained trial
 *     generate_code(); channel_doped(); decodeBP*();
 * and prints one text record per trial on stdout, parsed by oracle/make_golden.py.
 *
 * Variant is chosen at compile time: -DVARIANT_BPF, -DVARIANT_BPW, -DVARIANT_BPT, or
 * -DVARIANT_BPFSW (the BPF source, but calling its classical-window decodeBP_SW).
 *
 * usage: ref_xxx T seed0 eps max_it init_it W is_term dump [ndoped d0 d1 …]
 *   dump=1 → also print VNdegree / LLRsChannel / VNerased of every trial
 *   whole-run mode (T<0): -T frames WITHOUT re-seeding/inizio_sim between frames
 *   (srandom(seed0) once) — replays the carry-over of perm_code and of the RNG stream.
 */
#undef main
#include <stdint.h>

static uint64_t drv_fnv(const void *p, size_t nbytes, uint64_t h)
{
    const unsigned char *c = (const unsigned char *)p;
    for (size_t i = 0; i < nbytes; i++) { h ^= c[i]; h *= 1099511628211ULL; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 9) {
        fprintf(stderr, "usage: %s T seed0 eps max_it init_it W is_term dump [ndoped d0 ...]\n", argv[0]);
        return 2;
    }
    int T = atoi(argv[1]);
    unsigned seed0 = (unsigned)strtoul(argv[2], 0, 10);
    double eps = atof(argv[3]);
    int max_it = atoi(argv[4]);
    int init_it = atoi(argv[5]);
    int W = atoi(argv[6]);
    int is_term = atoi(argv[7]);
    int dump = atoi(argv[8]);
    int ndoped = argc > 9 ? atoi(argv[9]) : 0;
    int doped[32] = {0};
    for (int i = 0; i < ndoped && i < 32; i++) doped[i] = atoi(argv[10 + i]);
    (void)init_it; (void)is_term;

    int whole_run = 0;
    if (T < 0) { whole_run = 1; T = -T; }

    int n, nk; double rate, ShLm; int L = Def_L;
    initialize_variables(&n, &nk, L, &rate, &ShLm);
    MaxNumIt = max_it;
#ifdef VARIANT_BPW
    InitNumIt = init_it ? init_it : max_it;   /* BPW:2101-2102 */
#endif
    printf("HDR dv=%d dc=%d L=%d CNsPos=%d VNsPos=%d n=%d nk=%d T=%d seed0=%u eps=%.17g max_it=%d init_it=%d W=%d is_term=%d whole_run=%d ndoped=%d",
           Def_dv, Def_dc, Def_L, Def_CNsPos, Def_VNsPos, n, nk, T, seed0, eps, max_it, init_it, W, is_term, whole_run, ndoped);
    for (int i = 0; i < ndoped; i++) printf(" d%d=%d", i, doped[i]);
    printf("\n");

    if (whole_run) { sim = 0; inizio_sim(); srandom(seed0); }

    for (int t = 0; t < T; t++) {
        if (!whole_run) { sim = 0; inizio_sim(); srandom(seed0 + (unsigned)t); }
        generate_code(L, Def_VNsPos, Def_CNsPos, n, nk);
        channel_doped(n, eps, Def_VNsPos, ndoped, doped);

        uint64_t hg = 14695981039346656037ULL, hc = hg, he = hg;
        for (int j = 0; j < n; j++) hg = drv_fnv(&VNdegree[j][1], sizeof(int) * Def_dv, hg);
        hc = drv_fnv(LLRsChannel, sizeof(int) * (size_t)n, hc);
        int nch = 0; for (int j = 0; j < n; j++) nch += LLRsChannel[j];

        int be = 0, ee = 0, bee = 0, p1 = 0, ne;
        char *traj = NULL; size_t trajlen = 0;
#if defined(VARIANT_BPT)
        FILE *ft = open_memstream(&traj, &trajlen);
        ne = decodeBP(n, nk, L, W, Def_VNsPos, Def_CNsPos, &be, &ee, &bee, ft, is_term);
        fclose(ft);
#elif defined(VARIANT_BPW) || defined(VARIANT_BPFSW)
        /* BPW: square window (BPW:628-912).  BPFSW: the classical window kept in BPF:627-897. */
        ne = decodeBP_SW(n, nk, L, W, Def_VNsPos, Def_CNsPos, &p1, &be, &ee, &bee);
#else
        ne = decodeBP(n, nk, L, W, Def_VNsPos, Def_CNsPos, &be, &ee, &bee);
#endif
        he = drv_fnv(VNerased, (size_t)n, he);
        printf("TRIAL t=%d seed=%u nch=%d ne=%d p1=%d be=%d ee=%d bee=%d hg=%016llx hc=%016llx he=%016llx\n",
               t, seed0 + (unsigned)t, nch, ne, p1, be, ee, bee,
               (unsigned long long)hg, (unsigned long long)hc, (unsigned long long)he);
        if (traj) {
            /* rows "iter\tdeg1\trecovered\tfirst_pos\n", terminated by an empty line (BPT:988,1051,1145) */
            printf("TRAJ_BEGIN\n%sTRAJ_END\n", traj);
            free(traj);
        }
        if (dump) {
            printf("VNADJ");
            for (int j = 0; j < n; j++) for (int i = 0; i < Def_dv; i++) printf(" %d", VNdegree[j][1 + i]);
            printf("\nCNDEG");
            for (int i = 0; i < nk; i++) printf(" %d", CNdegree[i][0]);
            printf("\nCHAN ");
            for (int j = 0; j < n; j++) putchar('0' + LLRsChannel[j]);
            printf("\nERASED ");
            for (int j = 0; j < n; j++) putchar('0' + VNerased[j]);
            printf("\n");
        }
        if (whole_run)   /* counters accumulate exactly as main_terminated does (BPF:2140) */
            plr_computation(ne, p1, be, ee, bee);
    }
    if (whole_run)
        printf("RUN users_err=%d frame_err=%d frame_errP1=%d block_err=%d users_err_exp=%d frame_err_exp=%d block_err_exp=%d\n",
               users_err, frame_err, frame_errP1, block_err, users_err_exp, frame_err_exp, block_err_exp);
    return 0;
}
