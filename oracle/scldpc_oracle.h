/*
 * TEST INFRASTRUCTURE — CPU oracle for the SC-LDPC BEC Monte-Carlo hot path.
 *
 * Plain-C restatement of the reference's C simulators
 *   BPF = simulators_sc_ldpc/bp_decoding/SC_LDPC_Simulator_BPDecoder_BEC_full_BP_LimIter_OlmosRandomEnsemble.c
 *   BPW = …/SC_LDPC_Simulator_BPDecoder_BEC_SlidingWindow_LimIter_OlmosRandomEnsemble.c
 *   BPT = …/trajectories_SC_LDPC_Simulator_BPDecoder_BEC_full_BP_OlmosRandomEnsemble.c
 * Each function cites the reference lines it follows.  Parity status: PINNED — checked in
 * tests/test_oracle_vs_reference.py against (a) the real reference compiled from
 * /root/reference (oracle/_ref/, via oracle/ref_driver_tail.c) where that tree exists and
 * (b) the golden fixtures committed under tests/golden/ that were generated from it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (fl_scaling_sc_ldpc_amd/) never links, imports or executes it.
 */
#ifndef SCLDPC_ORACLE_H
#define SCLDPC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Ensemble geometry.  Naming trap: cns_pos = Def_M = Def_CNsPos, vns_pos = Def_VNsPos
 * = "N" of BASELINE.json; n = vns_pos*L; nk = (L+dv-1)*cns_pos (terminated chain, BPF:37). */
typedef struct {
    int dv, dc, L, cns_pos, vns_pos;
} orc_params;

static inline int orc_n(const orc_params *p)  { return p->vns_pos * p->L; }
static inline int orc_nk(const orc_params *p) { return (p->L + p->dv - 1) * p->cns_pos; }

/* glibc random()/srandom() TYPE_3 replica (third-party arithmetic on the path: glibc 2.35,
 * stdlib/random_r.c; call sites BPF:370,381,1684,2062).  Verified against libc in the tests. */
typedef struct {
    int32_t r[31];
    int f, b;            /* front / rear indices */
} orc_rng;
void    orc_srandom(orc_rng *g, unsigned seed);
int32_t orc_random(orc_rng *g);

uint64_t orc_fnv1a(const void *p, uint64_t nbytes, uint64_t h);   /* h0 = 14695981039346656037 */

/* inizio_sim's perm_code reset (BPF:308-311). perm_code has cns_pos*dc entries. */
void orc_perm_identity(const orc_params *p, int32_t *perm_code);

/* generate_code (BPF:1656-1761).  perm_code is in/out (state carries across calls, BPF:1682-1688).
 * vn_adj[n*dv]: CN id of edge i of VN j at [j*dv+i].  cn_ptr[nk+1], cn_adj[n*dv]: CN→VN lists in
 * the reference's insertion order (BPF:1714-1715). */
void orc_generate_code(const orc_params *p, orc_rng *g, int32_t *perm_code,
                       int32_t *vn_adj, int32_t *cn_ptr, int32_t *cn_adj);

/* channel_doped (BPF:1547-1574): chan[j]=1 iff erased. */
void orc_channel(const orc_params *p, orc_rng *g, double eps, int ndoped, const int *doped, uint8_t *chan);

/* Result block shared by all decoders. */
typedef struct {
    int32_t num_erasures;        /* return value of decodeBP / decodeBP_SW                  */
    int32_t num_blocks_err;      /* *num_blocks_err                                          */
    int32_t num_erasures_exp;    /* *num_erasures_exp                                        */
    int32_t num_blocks_err_exp;  /* *num_blocks_err_exp                                      */
    int32_t num_erasures_p1;     /* *NumErasuresP1 (window decoders only; BPW:846-847)       */
    int32_t iterations;          /* rows emitted = loop bodies executed (full BP); Σ over windows (SW) */
    int32_t status;              /* 0 ok; -1 = "ARGH" invariant abort of BPF:1035-1039       */
} orc_result;

/* Per-iteration trajectory row, the four columns of BPT:988,1051: iter is the row index. */
typedef struct { int32_t deg1, recovered, first_pos; } orc_row;

/* decodeBP, literal per-edge flooding (BPF:900-1140; BPT adds is_term/rows).  max_it<=0 ⇒ no cap
 * (BPT's while(1)).  is_term=0 ⇒ truncated chain (BPT:922-925,944-948).  erased[n] out (VNerased).
 * rows may be NULL; at most rows_cap rows are stored, res->iterations counts all. */
void orc_decode_bp_literal(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                           const int32_t *cn_adj, const uint8_t *chan, int max_it, int is_term,
                           uint8_t *erased, orc_row *rows, int rows_cap, orc_result *res);

/* Same outputs from the node-level level-synchronous peeling model of SURVEY.md §7.4(A). */
void orc_decode_bp_peel(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                        const int32_t *cn_adj, const uint8_t *chan, int max_it, int is_term,
                        uint8_t *erased, orc_row *rows, int rows_cap, orc_result *res);

/* decodeBP_SW.  square=1: BPW:628-912 (square window, init_it for posW==0).
 * square=0: the classical window of BPF:627-897 (VNs from posW-ms, posW<L+ms, decisions for posW>=ms;
 * cap max_it for every window). */
void orc_decode_sw_literal(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                           const int32_t *cn_adj, const uint8_t *chan, int W, int max_it, int init_it,
                           int square, uint8_t *erased, orc_result *res);

/* Node-level windowed peeling model of SURVEY.md §7.4(B) (square window). */
void orc_decode_sw_peel(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                        const int32_t *cn_adj, const uint8_t *chan, int W, int max_it, int init_it,
                        uint8_t *erased, orc_result *res);

/* Size-2 stopping-set expurgation (BPF:1067-1133 first_only=1; BPW:850-908 first_only=0). */
void orc_expurgate(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                   const int32_t *cn_adj, const uint8_t *erased, int first_only,
                   int32_t *num_blocks_err, int32_t *num_erasures_exp, int32_t *num_blocks_err_exp);

/* One self-contained trial exactly as oracle/ref_driver_tail.c replays the reference:
 * identity perm_code, srandom(seed), generate_code, channel_doped, decode.
 * decoder: 0 = full BP literal, 1 = full BP peel, 2 = SW literal (square), 3 = SW peel (square),
 *          4 = SW literal classical.
 * hashes[3]: FNV-1a of (VNdegree rows as int32, LLRsChannel as int32, VNerased as char) — the same
 * digests the reference driver prints.  Optional outputs may be NULL. */
void orc_trial(const orc_params *p, unsigned seed, double eps, int ndoped, const int *doped,
               int decoder, int W, int max_it, int init_it, int is_term,
               orc_result *res, uint64_t hashes[3], int32_t *n_chan_erased,
               int32_t *vn_adj_out, uint8_t *chan_out, uint8_t *erased_out,
               orc_row *rows, int rows_cap);

/* CPU twin of the device's Philox-keyed throughput sampler (csrc/sampler.hip); same integers out.
 * chan_bits: bit (j&31) of word j>>5 = 1 iff VN j erased. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_sample_philox_ens(const orc_params *p, int ensemble, uint64_t seed, uint64_t trial, double eps,
                           int ndoped, const int *doped, int32_t *vn_adj, uint32_t *chan_bits);
void orc_sample_philox(const orc_params *p, uint64_t seed, uint64_t trial, double eps,
                       int ndoped, const int *doped, int32_t *vn_adj, uint32_t *chan_bits);

/* One trial of simulate_peeling_decoder_ldpc (PD:740-785) on the device's Philox pick stream, O(steps log ncn). */
int64_t orc_random_pick_philox(const int32_t *tr, const uint8_t *mask, int n, int l, int ncn, int total_size,
                               int num_steps, uint64_t seed, uint64_t trial, int64_t *r1_out);

/* Streaming mode, the two helpers the reference's own table printers exercise (BPF:1891-1924; scldpc_stream_oracle.c) */
int orc_stream_is_doped(int pos, int ndoped, const int32_t *doped);
void orc_stream_sw_range(int pos, int L, int W, int ms, int32_t out[8]);

#ifdef __cplusplus
}
#endif
#endif
