/*
 * TEST INFRASTRUCTURE — deterministic driver around the reference's STREAMING mode
 * (main_streaming / decodeBP_SW_circular, BPF:1403-1500, 1934-2054), which the shipped source compiles out
 * (`#define CIRCULAR` / `#undef CIRCULAR`, BPF:33-34).  oracle/Makefile drops the #undef on a pipe and appends
 * this text; nothing of the reference is written to disk.
 *
 * Replays the body of main_streaming for ONE ε with srandom(seed) and prints, per decoded position, the value
 * decodeBP_SW_circular returned and the running counters (BPF:2015-2046).
 *
 * usage: ref_stream_* P seed eps W dump ndoped d0 d1 …      (P = positions to decode)
 *        ref_stream_* run seed W npoints eps_ini eps_delta max_blocks_err max_blocks ndoped d0 d1 …
 *                              a whole run of main_streaming with ONE srandom(seed): the ε points back to back, random()
 *                              carried from point to point, each point stopped by main_streaming's own rule (BPF:2033);
 *                              prints the arguments of results_circular per point
 *        ref_stream_* kat      runs the reference's own two table printers for this mode, test_is_position_doped_streaming
 *                              and test_circular_buffer_wrapping (BPF:1891-1924): its only known-answer material here
 */
#undef main
#include <stdint.h>

int main(int argc, char **argv)
{
    if (argc >= 2 && argv[1][0] == 'k') {
        test_is_position_doped_streaming();
        printf("----\n");
        test_circular_buffer_wrapping();
        return 0;
    }
    if (argc >= 10 && argv[1][0] == 'r') {
        unsigned seed = (unsigned)strtoul(argv[2], 0, 10);
        int W = atoi(argv[3]), npoints = atoi(argv[4]);
        double eps_ini = atof(argv[5]), eps_delta = atof(argv[6]);
        int max_blocks_err = atoi(argv[7]), max_blocks = atoi(argv[8]), num_doped = atoi(argv[9]);
        int doped_positions[32] = {0};
        for (int i = 0; i < num_doped && i < 32; i++) doped_positions[i] = atoi(argv[10 + i]);
        int n, nk, L = Def_L, CNsPos = Def_CNsPos, VNsPos = Def_VNsPos; double r, ShLm;
        initialize_variables(&n, &nk, L, &r, &ShLm);
        srandom(seed);                                              /* once, as main_streaming does (BPF:1942-1945) */
        for (sim = 0; sim < npoints; sim++) {
            int num_erasures = 0, num_blocks_err = 0, num_erasures_exp = 0, num_blocks_err_exp = 0;
            int num_bits_generated = 0, num_blocks_generated = 0, num_bits_generated_exp = 0, num_blocks_generated_exp = 0;
            inizio_sim();                                           /* perm_code := identity (its eps is the compile-time grid's) */
            double epsilon = eps_ini - sim * eps_delta;
            initialize_arrays_circular(n, nk, L, CNsPos);
            int gen_stream_pos = 0, pos;
            for (gen_stream_pos = 0; gen_stream_pos < L / 2; gen_stream_pos++) {
                generate_stream_pos(gen_stream_pos, L, epsilon, VNsPos, CNsPos, num_doped, doped_positions);
                initialize_messages_circular(gen_stream_pos, L, VNsPos, CNsPos);
            }
            for (pos = 0;; pos++) {
                int pd = pos - dv + 1, pe = pos - 2 * dv + 1;
                if (pd >= 0 && !is_position_doped_streaming(pd, num_doped, doped_positions)) { num_bits_generated += VNsPos; num_blocks_generated += 1; }
                if (pe >= 0 && !is_position_doped_streaming(pe, num_doped, doped_positions)) { num_bits_generated_exp += VNsPos; num_blocks_generated_exp += 1; }
                num_erasures += decodeBP_SW_circular(pos, n, L, W, VNsPos, CNsPos, &num_blocks_err, &num_erasures_exp, &num_blocks_err_exp);
                if (num_blocks_err_exp >= max_blocks_err || num_blocks_generated_exp >= max_blocks) break;
                generate_stream_pos(gen_stream_pos, L, epsilon, VNsPos, CNsPos, num_doped, doped_positions);
                initialize_messages_circular(gen_stream_pos, L, VNsPos, CNsPos);
                gen_stream_pos++;
            }
            printf("ROW sim=%d eps=%.17g pos=%d ne=%d be=%d ee=%d bee=%d gb=%d gbl=%d gbe=%d gble=%d\n", sim, epsilon, pos,
                   num_erasures, num_blocks_err, num_erasures_exp, num_blocks_err_exp, num_bits_generated,
                   num_blocks_generated, num_bits_generated_exp, num_blocks_generated_exp);
        }
        return 0;
    }
    if (argc < 7) { fprintf(stderr, "usage: %s P seed eps W dump ndoped [d0 ...]\n", argv[0]); return 2; }
    int P = atoi(argv[1]);
    unsigned seed = (unsigned)strtoul(argv[2], 0, 10);
    double epsilon = atof(argv[3]);
    int W = atoi(argv[4]);
    int dump = atoi(argv[5]);
    int num_doped = atoi(argv[6]);
    int doped_positions[32] = {0};
    for (int i = 0; i < num_doped && i < 32; i++) doped_positions[i] = atoi(argv[7 + i]);

    int n, nk, L = Def_L, CNsPos = Def_CNsPos, VNsPos = Def_VNsPos; double r, ShLm;
    initialize_variables(&n, &nk, L, &r, &ShLm);
    sim = 0; inizio_sim();
    srandom(seed);
    printf("HDR dv=%d dc=%d L=%d CNsPos=%d VNsPos=%d n=%d nk=%d P=%d seed=%u eps=%.17g W=%d ndoped=%d",
           Def_dv, Def_dc, L, CNsPos, VNsPos, n, nk, P, seed, epsilon, W, num_doped);
    for (int i = 0; i < num_doped; i++) printf(" d%d=%d", i, doped_positions[i]);
    printf("\n");

    int num_erasures = 0, num_blocks_err = 0, num_erasures_exp = 0, num_blocks_err_exp = 0;
    int num_bits_generated = 0, num_blocks_generated = 0, num_bits_generated_exp = 0, num_blocks_generated_exp = 0;
    int stream_lag = L / 2;
    initialize_arrays_circular(n, nk, L, CNsPos);
    int gen_stream_pos = 0;
    for (gen_stream_pos = 0; gen_stream_pos < stream_lag; gen_stream_pos++) {
        generate_stream_pos(gen_stream_pos, L, epsilon, VNsPos, CNsPos, num_doped, doped_positions);
        initialize_messages_circular(gen_stream_pos, L, VNsPos, CNsPos);
    }
    for (int pos = 0; pos < P; pos++) {
        int pos_vn_decision = pos - dv + 1, pos_vn_decision_exp = pos - 2 * dv + 1;
        if (pos_vn_decision >= 0 && !is_position_doped_streaming(pos_vn_decision, num_doped, doped_positions)) {
            num_bits_generated += VNsPos; num_blocks_generated += 1;
        }
        if (pos_vn_decision_exp >= 0 && !is_position_doped_streaming(pos_vn_decision_exp, num_doped, doped_positions)) {
            num_bits_generated_exp += VNsPos; num_blocks_generated_exp += 1;
        }
        int nep = decodeBP_SW_circular(pos, n, L, W, VNsPos, CNsPos, &num_blocks_err, &num_erasures_exp, &num_blocks_err_exp);
        num_erasures += nep;
        printf("POS pos=%d nep=%d ne=%d be=%d ee=%d bee=%d gb=%d gbl=%d gbe=%d gble=%d\n", pos, nep, num_erasures,
               num_blocks_err, num_erasures_exp, num_blocks_err_exp, num_bits_generated, num_blocks_generated,
               num_bits_generated_exp, num_blocks_generated_exp);
        if (dump && pos_vn_decision >= 0) {
            printf("ERASED ");
            int s0 = (pos_vn_decision % L) * VNsPos;
            for (int j = s0; j < s0 + VNsPos; j++) putchar('0' + VNerased[j]);
            printf("\n");
        }
        generate_stream_pos(gen_stream_pos, L, epsilon, VNsPos, CNsPos, num_doped, doped_positions);
        initialize_messages_circular(gen_stream_pos, L, VNsPos, CNsPos);
        gen_stream_pos++;
    }
    return 0;
}
