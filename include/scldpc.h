/*
 * scldpc.h — C-ABI of libscldpc_hip.so: MI355X (gfx950) Monte-Carlo BEC decoding of random
 * SC-LDPC ensembles.  Drop-in for the hot path of rsokolovskii/fl_scaling_sc_ldpc's
 * simulators_sc_ldpc/{bp_decoding,peeling_decoding}.
 *
 * The reference has no FFI: its hot path sits behind process argv + output files and two Python
 * function signatures.  The entry points below are what a binding for that path
 * would call; each cites the reference routine it replaces.  Abbreviations:
 *   BPF = simulators_sc_ldpc/bp_decoding/SC_LDPC_Simulator_BPDecoder_BEC_full_BP_LimIter_OlmosRandomEnsemble.c
 *   BPW = …/SC_LDPC_Simulator_BPDecoder_BEC_SlidingWindow_LimIter_OlmosRandomEnsemble.c
 *   BPT = …/trajectories_SC_LDPC_Simulator_BPDecoder_BEC_full_BP_OlmosRandomEnsemble.c
 *   PD  = simulators_sc_ldpc/peeling_decoding/peeling_decoding.py
 *
 * Conventions: plain pointers and sizes, caller-owned buffers, `int` status return (0 = OK,
 * negative = error, text via scldpc_last_error(), thread-local), no globals shared between calls.  Pointers
 * named d_* are DEVICE pointers (HIP); `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Device entry points only enqueue work; they never synchronise, allocate or free — so they can be
 * captured into a hipGraph and called concurrently on different streams.
 *
 * Workspace: ensembles whose per-CN state exceeds the LDS (N >= 2500 decoders, random-pick peeling at N >= 2000, the
 * sampler beyond 8192 sockets per position) keep it in device memory the CALLER owns: pass (d_workspace,
 * workspace_bytes) with at least scldpc_workspace_bytes(op, ...) bytes, 256-byte aligned, not shared with a call that
 * may run at the same time.  The query returns 0 when the ensemble needs none (NULL / 0 may then be passed).
 *
 * Data layout (one "trial" = one sampled code + one channel realisation = one reference "frame"):
 *   vn_adj   int32 [ntrials][n][dv]       CN index of edge i of VN j   (VNdegree[j][1+i], BPF:87)
 *   chan     uint32[ntrials][nw]          bit (j&31) of word j>>5 = LLRsChannel[j] (1 = erased, BPF:91);
 *                                         nw = (n+31)/32, padding bits of the last word ignored
 *   counters int32 [ntrials][SCLDPC_NCOUNTERS]   see enum below
 *   rows     int32 [ntrials][rows_cap][3] per-iteration (deg_1_iter, recovered, first_erased/VNsPos) — BPT:988,1051
 *   erased   uint32[ntrials][nw]          VNerased after decoding, bit-packed like chan
 * with n = vns_pos*L VNs, nk = (L+dv-1)*cns_pos CNs (terminated chain, BPF:30,37).
 */
#ifndef SCLDPC_H
#define SCLDPC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCLDPC_ABI_VERSION 2

enum {
    SCLDPC_OK = 0,
    SCLDPC_ERR_BAD_ARG = -1,
    SCLDPC_ERR_TOO_LARGE = -2,      /* ensemble does not fit the kernel's LDS budget */
    SCLDPC_ERR_HIP = -3,            /* a HIP runtime call failed                     */
    SCLDPC_ERR_NO_DEVICE = -4
};

/* Device workspace an entry point needs for `ntrials` trials of ensemble p (0: none), or a negative error:
 *   SCLDPC_WS_SAMPLE      scldpc_sample_philox_device(_adj16)
 *   SCLDPC_WS_FULL_BP     scldpc_full_bp_device(_adj16), scldpc_full_bp_fixpoint_device(_adj16); arg0 != 0: with d_rows
 *   SCLDPC_WS_SW_BP       scldpc_sw_bp_device(_adj16), scldpc_swc_bp_device(_adj16);             arg0 = W
 *   SCLDPC_WS_PEEL_SWEEP  scldpc_peel_sweep_device(_adj16);                                      arg0 != 0: the _adj16 form
 *   SCLDPC_WS_PEEL_PICK   scldpc_peel_pick_device(_adj16);  arg0 = total_size, arg1 != 0: with d_mt_state */
enum { SCLDPC_WS_SAMPLE = 0, SCLDPC_WS_FULL_BP = 1, SCLDPC_WS_SW_BP = 2, SCLDPC_WS_PEEL_SWEEP = 3, SCLDPC_WS_PEEL_PICK = 4 };
struct scldpc_code_params;
int64_t scldpc_workspace_bytes(int32_t op, const struct scldpc_code_params *p, int32_t ntrials, int32_t arg0, int32_t arg1);

/* Ensemble geometry — the compile-time #defines of BPF:22-37 as run-time parameters.
 * Naming trap: cns_pos = Def_M = Def_CNsPos; vns_pos = Def_VNsPos = "N" of the papers = Python's M (PD:18). */
typedef struct scldpc_code_params {
    int32_t dv;        /* Def_dv  */
    int32_t dc;        /* Def_dc  */
    int32_t L;         /* Def_L   */
    int32_t cns_pos;   /* Def_CNsPos */
    int32_t vns_pos;   /* Def_VNsPos; dv*vns_pos must equal dc*cns_pos */
} scldpc_code_params;

/* Per-trial outputs of the decoders. */
enum {
    SCLDPC_C_NUM_ERASURES = 0,       /* return value of decodeBP / decodeBP_SW (BPF:1138, BPW:910) */
    SCLDPC_C_NUM_BLOCKS_ERR = 1,     /* *num_blocks_err      (BPF:1123-1125, BPW:843-844)           */
    SCLDPC_C_NUM_ERASURES_EXP = 2,   /* *num_erasures_exp    (BPF:1126-1132, BPW:903-907)           */
    SCLDPC_C_NUM_BLOCKS_ERR_EXP = 3, /* *num_blocks_err_exp                                          */
    SCLDPC_C_NUM_ERASURES_P1 = 4,    /* *NumErasuresP1       (BPW:846-847; 0 for full BP)            */
    SCLDPC_C_ITERATIONS = 5,         /* flooding iterations executed (Σ over windows for SW)          */
    SCLDPC_C_STATUS = 6,             /* 0; -1 = the reference would have aborted (BPF:1035-1039)     */
    SCLDPC_C_CHANNEL_ERASURES = 7,   /* Σ LLRsChannel                                                 */
    SCLDPC_NCOUNTERS = 8
};

/* Run-level accumulators of plr_computation (BPF:1503-1520), in this order. */
enum {
    SCLDPC_R_USERS_ERR = 0, SCLDPC_R_FRAME_ERR = 1, SCLDPC_R_FRAME_ERR_P1 = 2, SCLDPC_R_BLOCK_ERR = 3,
    SCLDPC_R_USERS_ERR_EXP = 4, SCLDPC_R_FRAME_ERR_EXP = 5, SCLDPC_R_BLOCK_ERR_EXP = 6,
    SCLDPC_R_FRAMES = 7,             /* f: frames consumed, incl. the one that tripped willIstop (BPF:2143-2144) */
    SCLDPC_R_ITERATIONS = 8,         /* Σ iterations over the consumed frames (measurement aid)       */
    SCLDPC_NRUN = 9
};

int         scldpc_abi_version(void);
const char *scldpc_last_error(void);           /* thread-local text of the last failing call */
int         scldpc_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Ensemble + channel sampling
 * ------------------------------------------------------------------------------------------- */

/* Exact replay of the reference's sampling on identical seeds, on the host:
 *   perm_code := identity (inizio_sim, BPF:308-311); srandom(seed) (BPF:2062);
 *   generate_code (BPF:1656-1761); channel_doped (BPF:1547-1574).
 * glibc's random() (TYPE_3) is restated inside the library, so results do not depend on the
 * host libc.  Outputs are HOST buffers: vn_adj[n*dv], chan_bits[nw]. */
int scldpc_sample_glibc_host(const scldpc_code_params *p, uint32_t seed, double eps,
                             int32_t ndoped, const int32_t *doped_positions,
                             int32_t *vn_adj, uint32_t *chan_bits);

/* The same for a whole reference run: ONE srandom(seed), frames drawn back to back, perm_code and
 * the RNG stream carried from frame to frame exactly as main_terminated does (BPF:2117-2131).
 * nframes consecutive frames are written: vn_adj[nframes][n*dv], chan_bits[nframes][nw].
 * `state` is an opaque caller-owned blob of scldpc_glibc_state_bytes(p) bytes; initialise it with
 * scldpc_glibc_state_init (= srandom + inizio_sim), then call _next any number of times. */
int64_t scldpc_glibc_state_bytes(const scldpc_code_params *p);
int     scldpc_glibc_state_init(const scldpc_code_params *p, uint32_t seed, void *state);
int     scldpc_glibc_state_reset_perm(const scldpc_code_params *p, void *state);   /* inizio_sim only */
int     scldpc_sample_glibc_next_host(const scldpc_code_params *p, void *state, double eps,
                                      int32_t ndoped, const int32_t *doped_positions, int32_t nframes,
                                      int32_t *vn_adj, uint32_t *chan_bits);

/* Throughput mode: sample ntrials codes + channels ON THE DEVICE with a counter-based generator
 * (Philox4x32-10 keyed by (seed, trial index)), same ensemble law as generate_code/channel_doped:
 * one uniform permutation of the cns_pos*dc sockets per CN position, i.i.d. Bernoulli(eps) erasures,
 * doped positions forced known.  Trial t of the call uses global index trial0+t, so any sharding of
 * a run over calls / GPUs gives identical codes.  d_vn_adj [ntrials][n][dv], d_chan_bits [ntrials][nw]. */
int scldpc_sample_philox_device(const scldpc_code_params *p, uint64_t seed, uint64_t trial0,
                                int32_t ntrials, double eps, int32_t ndoped, const int32_t *doped_positions,
                                int32_t *d_vn_adj, uint32_t *d_chan_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* The other ensembles of the reference's Python simulators, same keying, global CN ids (int32 rows):
 *   SCLDPC_ENS_TAIL_BITING  sc_ldpc.gen_slots_tail_biting (sc_ldpc.py:41-62): L permutations, edge i of VN position q
 *                           lands in CN position (q+i) mod L;
 *   SCLDPC_ENS_PROTOGRAPH   sc_ldpc_protograph.gen_slots_from_position (:6-20): per VN position dc/dv portions x dv
 *                           uniform permutations of cns_pos; edge i of VN u of a portion → CN (q+i, perm_i[u]).
 * SCLDPC_ENS_OLMOS equals scldpc_sample_philox_device.  At most 8192 sockets per position for the two others. */
enum { SCLDPC_ENS_OLMOS = 0, SCLDPC_ENS_TAIL_BITING = 1, SCLDPC_ENS_PROTOGRAPH = 2 };
int scldpc_sample_philox_ensemble_device(const scldpc_code_params *p, int32_t ensemble, uint64_t seed, uint64_t trial0,
                                         int32_t ntrials, double eps, int32_t ndoped, const int32_t *doped_positions,
                                         int32_t *d_vn_adj, uint32_t *d_chan_bits, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Decoders (device)
 * ------------------------------------------------------------------------------------------- */

/* decodeBP — flooding BP over the BEC to the fixpoint or max_it iterations, plus the size-2
 * stopping-set expurgation (BPF:900-1140).  max_it <= 0 ⇒ unlimited (BPT:825 `while(1)`).
 * is_term = 0 ⇒ truncated chain: CNs at positions >= L never send information (BPT:922-925,944-948).
 * d_rows / d_erased_bits may be NULL.  One workgroup decodes one trial.
 * Precondition of all BP decoders (full, window, streaming): the graph has generate_code's position structure — edge i of
 * VN j lies in CN position j / vns_pos + i (BPF:1712) — which is what every sampler of this library and of the reference's
 * C simulators produces.  (The peeling entry points take arbitrary VN→CN tables in their int32 form.) */
int scldpc_full_bp_device(const scldpc_code_params *p, int32_t ntrials,
                          const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                          int32_t max_it, int32_t is_term,
                          int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                          uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* decodeBP with no iteration cap, when only what it converges to is wanted (the reference's bp_lim_iter with MAX_IT beyond
 * reach, BPF:2080-2083, prints nothing that depends on the iteration count): same counters as scldpc_full_bp_device with
 * max_it = 0 EXCEPT SCLDPC_C_ITERATIONS, which here counts the kernel's barrier rounds, and STATUS, which is always 0.
 * On the BEC the fixpoint does not depend on the order in which CNs resolve VNs, so a thread follows the chain its own
 * release opens instead of waiting for the next flooding iteration (full_bp.hip). */
int scldpc_full_bp_fixpoint_device(const scldpc_code_params *p, int32_t ntrials,
                                   const int32_t *d_vn_adj, const uint32_t *d_chan_bits, int32_t is_term,
                                   int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);
int scldpc_full_bp_fixpoint_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                         const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits, int32_t is_term,
                                         int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* The same pair — generate_code + channel_doped (BPF:1656-1761, 1547-1574), then decodeBP to its fixpoint (BPF:900-1140) —
 * for the (dv = 4, dc = 8) chain with at most 8192 sockets per CN position (N <= 2048; the BASELINE ensemble has 4000), in
 * the form the throughput path of bench.py runs:
 *   scldpc_sample_philox_device_cn16      same keys, same law, same d_vn_adj16 / d_chan_bits as scldpc_sample_philox_device_adj16,
 *                                         plus (d_cn_adj16 may be NULL) the CN -> VN table uint16 [ntrials][nk][dc]: the VNs
 *                                         attached to every CN (global VN index; needs n < 65535), 0xFFFF where a CN at a
 *                                         chain end has fewer than dc (BPF:1703-1716).  Order within a CN unspecified (a set).
 *   scldpc_full_bp_fixpoint_device_cn16   counters of scldpc_full_bp_fixpoint_device from both tables: 4 bits of LDS per CN,
 *                                         seven trials per CU in flight (full_bp_small.hip).
 *   scldpc_full_bp_device_cn16            decodeBP WITH its iterations from the same tables and the same 4 bits per CN: one
 *                                         flooding iteration per barrier round, so max_it (MaxNumIt, BPF:1065; <= 0 = unlimited),
 *                                         the stop tests (BPF:1044-1045) and SCLDPC_C_ITERATIONS / SCLDPC_C_STATUS are the
 *                                         reference's — every counter of scldpc_full_bp_device (no trajectory rows), six
 *                                         trials per CU in flight.  The reference's bp_lim_iter with a binding MAX_IT
 *                                         (the published ..._500it_... / ..._1000it_... tables) runs on this one.
 *   scldpc_sample_philox_device_sock16    the same sampler emitting, instead of the CN -> VN table, the CN -> socket table
 *                                         d_cn_sock16 of scldpc_sw_bp_ring_device below (what scldpc_cn_sockets_device
 *                                         builds in a second pass); sockets are position-local, so any chain length.
 * *_supported: 1 if the ensemble is taken, else 0 (use the _adj16 entry points). */
int scldpc_sample_philox_cn16_supported(const scldpc_code_params *p);
int scldpc_sample_philox_sock16_supported(const scldpc_code_params *p);
int scldpc_sample_philox_device_sock16(const scldpc_code_params *p, uint64_t seed, uint64_t trial0, int32_t ntrials,
                                       double eps, int32_t ndoped, const int32_t *doped_positions, uint16_t *d_vn_adj16,
                                       uint16_t *d_cn_sock16, uint32_t *d_chan_bits, void *stream);
int scldpc_sample_philox_device_cn16(const scldpc_code_params *p, uint64_t seed, uint64_t trial0, int32_t ntrials,
                                     double eps, int32_t ndoped, const int32_t *doped_positions, uint16_t *d_vn_adj16,
                                     uint16_t *d_cn_adj16, uint32_t *d_chan_bits, void *stream);
int scldpc_full_bp_cn16_supported(const scldpc_code_params *p);
int scldpc_full_bp_fixpoint_device_cn16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                        const uint16_t *d_cn_adj16, const uint32_t *d_chan_bits, int32_t is_term,
                                        int32_t *d_counters, uint32_t *d_erased_bits, void *stream);
int scldpc_full_bp_device_cn16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                               const uint16_t *d_cn_adj16, const uint32_t *d_chan_bits, int32_t max_it, int32_t is_term,
                               int32_t *d_counters, uint32_t *d_erased_bits, void *stream);
/* The same two decoders on the CN -> SOCKET table (scldpc_sample_philox_device_sock16 / scldpc_cn_sockets_device:
 * position-local sockets s = dv*t + i, 0xFFFF = none), which lifts the n < 65535 limit of the global VN ids — e.g. the
 * published L = 100, N = 1000 runs (n = 100 000).  Needs at most 65536 CNs per trial and 16-bit sockets. */
int scldpc_full_bp_sock16_supported(const scldpc_code_params *p);
int scldpc_full_bp_fixpoint_device_sock16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                          const uint16_t *d_cn_sock16, const uint32_t *d_chan_bits, int32_t is_term,
                                          int32_t *d_counters, uint32_t *d_erased_bits, void *stream);
int scldpc_full_bp_device_sock16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                 const uint16_t *d_cn_sock16, const uint32_t *d_chan_bits, int32_t max_it, int32_t is_term,
                                 int32_t *d_counters, uint32_t *d_erased_bits, void *stream);
/* decodeBP of the trajectory build (BPT:900-1140) on the same tables: the iterations WITH their rows — per iteration
 * deg_1_iter, the VNs recovered and the position of the first erased VN (BPT:988, 1037-1038, 1051): d_rows int32
 * [ntrials][rows_cap][3]; d_counters[SCLDPC_C_ITERATIONS] says how many rows a trial wrote (rows beyond rows_cap are dropped).
 * is_term = 0: the truncated chain of `bp_traj … IS_TERM=0` (BPT:922-925, 944-948). */
int scldpc_full_bp_traj_device_cn16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                    const uint16_t *d_cn_adj16, const uint32_t *d_chan_bits, int32_t max_it, int32_t is_term,
                                    int32_t *d_counters, int32_t *d_rows, int32_t rows_cap, uint32_t *d_erased_bits, void *stream);
int scldpc_full_bp_traj_device_sock16(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                                      const uint16_t *d_cn_sock16, const uint32_t *d_chan_bits, int32_t max_it, int32_t is_term,
                                      int32_t *d_counters, int32_t *d_rows, int32_t rows_cap, uint32_t *d_erased_bits, void *stream);

/* decodeBP_SW, square window (BPW:628-912): window of W positions, init_it iterations for the
 * first window and max_it for the others (init_it == 0 ⇒ max_it, BPW:2101-2102). */
int scldpc_sw_bp_device(const scldpc_code_params *p, int32_t ntrials,
                        const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                        int32_t W, int32_t max_it, int32_t init_it,
                        int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* decodeBP_SW, square window, with only the window's state on chip (sw_ring.hip): a ring of W + 2dv - 1 CN positions
 * (4-bit counts) and W + dv VN positions (S bits) in LDS instead of one word per CN of the whole chain — seven trials per CU
 * at (L=100, N=2000, W=10), no workspace.  Same counters as scldpc_sw_bp_device.  Takes the 2-byte tables: d_vn_adj16 as
 * above and d_cn_sock16 uint16 [ntrials][nk][dc] = the sockets of every CN (socket s = dv*t + i is edge i of VN t of
 * position CNpos - i; 0xFFFF pads chain-end CNs; order within a CN unspecified), which scldpc_cn_sockets_device builds from
 * any position-structured VN -> CN table (one workgroup per trial and CN position).  *_supported: 1 if (p, W) is taken. */
int scldpc_sw_bp_ring_supported(const scldpc_code_params *p, int32_t W);
int scldpc_cn_sockets_device(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                             uint16_t *d_cn_sock16, void *stream);
int scldpc_sw_bp_ring_device(const scldpc_code_params *p, int32_t ntrials, const uint16_t *d_vn_adj16,
                             const uint16_t *d_cn_sock16, const uint32_t *d_chan_bits, int32_t W, int32_t max_it,
                             int32_t init_it, int32_t *d_counters, uint32_t *d_erased_bits, void *stream);

/* decodeBP_SW, classical window — the variant kept in BPF:627-897 (its call is commented out at BPF:2137-2138):
 * L+dv-1 windows, VNs [posW-ms, posW+W), position posW-ms decided when window posW closes, max_it per window. */
int scldpc_swc_bp_device(const scldpc_code_params *p, int32_t ntrials,
                         const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                         int32_t W, int32_t max_it,
                         int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);
int scldpc_swc_bp_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                               const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                               int32_t W, int32_t max_it,
                               int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* Compact adjacency variants.  d_vn_adj16 is uint16 [ntrials][n][dv]: the CN index LOCAL to its position
 * (0 .. cns_pos-1).  Edge i of a VN at position pos always lands in CN position pos+i (BPF:1712), so the
 * global id is (pos+i)*cns_pos + local.  Half the HBM bytes of the int32 table (one 8-byte row per VN at
 * dv = 4); same results bit for bit.  Requires cns_pos <= 65536. */
int scldpc_sample_philox_device_adj16(const scldpc_code_params *p, uint64_t seed, uint64_t trial0,
                                      int32_t ntrials, double eps, int32_t ndoped, const int32_t *doped_positions,
                                      uint16_t *d_vn_adj16, uint32_t *d_chan_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);
int scldpc_full_bp_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                int32_t max_it, int32_t is_term,
                                int32_t *d_counters, int32_t *d_rows, int32_t rows_cap,
                                uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);
int scldpc_sw_bp_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                              const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                              int32_t W, int32_t max_it, int32_t init_it,
                              int32_t *d_counters, uint32_t *d_erased_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Python peeling path (PD = simulators_sc_ldpc/peeling_decoding/peeling_decoding.py)
 * In PD's vocabulary VN = "user", CN = "slot"; `M` = vns_pos, cns_per_pos = cns_pos, `transmissions`
 * = vn_adj.  The graph is always laid out for the full terminated chain (ncn = (L+dv-1)*cns_pos CN words);
 * total_size = cns_pos*(L+dv-1) for a terminated chain or cns_pos*L for a non-terminated one (PD:609-611).
 * ------------------------------------------------------------------------------------------- */

/* One trial of simulate_sc_ldpc's loop body (PD:650-691): sweep peeling
 * `for t in range(sweep_start, total_size): sic_round(schedule, t)` (PD:656-657, 270-313), then
 * lost = VNs still attached to a CN of [lost_lo, lost_hi) whose CNs are all < total_size (PD:659-666),
 * stopping sets = connected components of `lost` (extract_stopping_sets, PD:1077-1095).
 * d_out int32 [ntrials][8]: [0] #lost, [1] #lost in components of more than 2 VNs (num_lost_exp, PD:680),
 * [2] #distinct positions int(birthday/cns_per_pos) over those components (PD:684-689), [5] peeling rounds,
 * [7] #erased VNs handed in; the other slots are 0.  d_lost_bits (optional) receives the lost set. */
int scldpc_peel_sweep_device(const scldpc_code_params *p, int32_t ntrials,
                             const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                             int32_t total_size, int32_t sweep_start, int32_t lost_lo, int32_t lost_hi,
                             int32_t *d_out, uint32_t *d_lost_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);
int scldpc_peel_sweep_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                   const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                   int32_t total_size, int32_t sweep_start, int32_t lost_lo, int32_t lost_hi,
                                   int32_t *d_out, uint32_t *d_lost_bits, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* One trial of simulate_peeling_decoder_ldpc's loop body (PD:750-785): random-pick peeling with the
 * degree-1-CN trajectory.  num_steps = int(M*num_positions*(e+0.1)) (PD:721).
 * Generator: d_mt_state != NULL ⇒ uint32 [ntrials][625] = CPython `random` MT19937 state (624 words + index,
 * as random.getstate() gives it); every pick draws `_randbelow(#degree-1 CNs)` from it exactly as
 * random.choice does (PD:1026) and the advanced state is written back, so a host loop can chain the trials of
 * one reference run.  d_mt_state == NULL ⇒ Philox4x32-10 keyed by (seed, trial0 + trial), same rejection rule.
 * d_r1 (optional) int32 [ntrials][num_steps+1] = r1[o, :] (PD:758,781).  d_moments (optional) int64
 * [3][num_steps+1], accumulated in place over the trials of the call like scldpc_r1_moments_device — for runs
 * whose trajectories are too large to keep (N = 10000: 1.2 MB per trial).  d_out int32 [ntrials][4]:
 * #erased VNs, #picks, last r1 value, #steps that had a degree-1 CN.  At most 262144 pickable CNs. */
int scldpc_peel_pick_device(const scldpc_code_params *p, int32_t ntrials,
                            const int32_t *d_vn_adj, const uint32_t *d_chan_bits,
                            int32_t total_size, int32_t num_steps,
                            uint32_t *d_mt_state, uint64_t seed, uint64_t trial0,
                            int32_t *d_r1, int64_t *d_moments, int32_t *d_out, void *d_workspace, uint64_t workspace_bytes, void *stream);
int scldpc_peel_pick_device_adj16(const scldpc_code_params *p, int32_t ntrials,
                                  const uint16_t *d_vn_adj16, const uint32_t *d_chan_bits,
                                  int32_t total_size, int32_t num_steps,
                                  uint32_t *d_mt_state, uint64_t seed, uint64_t trial0,
                                  int32_t *d_r1, int64_t *d_moments, int32_t *d_out, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* Integer moments of a batch of trajectories, the device half of main_simulate_variance (PD:1264-1294) /
 * calc_nu_chunk (fl_scaling/est_scaling_params.py:90-94,131-138): d_moments int64 [3][ncols], accumulated in
 * place: [0][s] += #{r1[t][s] != 0}, [1][s] += Σ_t r1[t][s], [2][s] += Σ_t r1[t][s]². */
int scldpc_r1_moments_device(int32_t ntrials, int32_t ncols, const int32_t *d_r1, int64_t *d_moments, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Streaming mode (main_streaming, BPF:1934-2054 — compiled out in the shipped source by `#undef CIRCULAR`,
 * BPF:33-34): a circular buffer of p->L positions, periodic doping (is_position_doped_streaming, BPF:1589-1612),
 * one position decoded by decodeBP_SW_circular (BPF:1403-1500) and one generated (generate_stream_pos,
 * BPF:1927-1932) per step.  nstreams independent streams run side by side (one workgroup each); the unit of work
 * is one decoded position.  d_state: nstreams blobs of scldpc_stream_state_bytes(p, W) bytes each, zero-filled by
 * the caller before a stream's first call and carried from call to call; every call decodes npos further
 * positions per stream.  Codes and channels are Philox-keyed by (seed, stream0 + stream index, position).
 * d_counters int64 [nstreams][10]: num_erasures, num_blocks_err, num_erasures_exp, num_blocks_err_exp,
 * num_bits_generated, num_blocks_generated, num_bits_generated_exp, num_blocks_generated_exp (the arguments of
 * results_circular, BPF:522-562), positions decoded, positions generated.  "positions generated" < 0 marks a stream
 * as unusable: its permutation ranking met 256 of a position's Philox keys in one of its >= 1024 buckets (nothing a real
 * draw does), and rather than rank wrongly both kernels leave such a stream untouched from then on — callers check the
 * column when they read the counters (bp_decoding.run_streaming does).  d_trace (optional) int32
 * [nstreams][npos][10]: position, value returned by decodeBP_SW_circular, then the eight counters after it.
 * Requires W + dv - 1 <= L/2 (the stream is generated L/2 positions ahead of the decoder, BPF:2001). */
int64_t scldpc_stream_state_bytes(const scldpc_code_params *p, int32_t W);
int scldpc_stream_run_device(const scldpc_code_params *p, int32_t nstreams, uint64_t seed, uint64_t stream0,
                             double eps, int32_t W, int32_t ndoped, const int32_t *doped_positions,
                             int32_t npos, void *d_state, int64_t *d_counters, int32_t *d_trace, void *stream);

/* Same-input mode: the streams' inputs are given instead of drawn.  scldpc_stream_glibc_inputs_host replays, for ONE
 * stream, exactly what main_streaming draws after srandom(seed) (BPF:2059-2062, 1808-1813, 1927-1932): inter_out uint16
 * [npos_gen + dv - 1][cns_pos*dc] = CN-local id (perm_code[i] / dc, BPF:1782) of socket i of CN position c, chan_out
 * uint32 [npos_gen][ceil(vns_pos/32)] = erasure bits of VN position g (none at doped positions, BPF:1621-1654).
 * scldpc_stream_run_device_inputs decodes from such arrays on the device (d_inter [nstreams][inputs_npos + dv - 1][S],
 * d_chan_bits [nstreams][inputs_npos][wpp], one slice per stream): same state blobs, counters and trace as
 * scldpc_stream_run_device; positions_done = positions decoded by earlier calls (a stream needs L/2 + positions_done +
 * npos generated positions).  With the glibc replay the device reproduces a reference run position by position. */
int scldpc_stream_glibc_inputs_host(const scldpc_code_params *p, uint32_t seed, double eps, int32_t ndoped,
                                    const int32_t *doped_positions, int32_t npos_gen, uint16_t *inter_out,
                                    uint32_t *chan_out);
int scldpc_stream_run_device_inputs(const scldpc_code_params *p, int32_t nstreams, int32_t W, int32_t ndoped,
                                    const int32_t *doped_positions, int32_t npos, void *d_state, int64_t *d_counters,
                                    int32_t *d_trace, const uint16_t *d_inter, const uint32_t *d_chan_bits,
                                    int32_t inputs_npos, int64_t positions_done, void *stream);
/* The same for runs of any length, in pieces.  scldpc_stream_glibc_next_host continues from a carried state
 * (scldpc_glibc_state_init = srandom(seed); scldpc_glibc_state_reset_perm = inizio_sim at the start of an ε point, BPF:1994;
 * main_streaming seeds ONCE and carries random() from point to point, BPF:1942-1945): it shuffles `ninit` CN positions
 * (initialize_arrays_circular: dv-1 at the start of a point, 0 afterwards), then draws generate_stream_pos(gpos0 + k),
 * k < npos, into inter_out [ninit + npos][S] and chan_out [npos][wpp].  scldpc_stream_run_device_inputs_at is
 * scldpc_stream_run_device_inputs with arrays that begin at generated position inputs_pos0 instead of 0: d_inter
 * [nstreams][inputs_npos + dv - 1][S] holds CN positions inputs_pos0 .. inputs_pos0 + inputs_npos + dv - 2 (rows 0 .. dv-2 are
 * read only by a stream's first call, when inputs_pos0 = 0), d_chan_bits [nstreams][inputs_npos][wpp] VN positions
 * inputs_pos0 .. ; the call needs the generated positions L/2 + positions_done .. L/2 + positions_done + npos - 1 (and
 * 0 .. L/2 - 1 on a stream's first call) to lie inside. */
int scldpc_stream_glibc_next_host(const scldpc_code_params *p, void *state, double eps, int32_t ndoped,
                                  const int32_t *doped_positions, int32_t ninit, int64_t gpos0, int32_t npos,
                                  uint16_t *inter_out, uint32_t *chan_out);
int scldpc_stream_run_device_inputs_at(const scldpc_code_params *p, int32_t nstreams, int32_t W, int32_t ndoped,
                                       const int32_t *doped_positions, int32_t npos, void *d_state, int64_t *d_counters,
                                       int32_t *d_trace, const uint16_t *d_inter, const uint32_t *d_chan_bits,
                                       int64_t inputs_pos0, int32_t inputs_npos, int64_t positions_done, void *stream);

/* plr_computation + willIstop over a batch, IN TRIAL ORDER (BPF:1503-1520, 440-451, 2140-2144):
 * adds the per-trial counters of trials 0..k into d_run[SCLDPC_NRUN] (int64, accumulated in place),
 * where k is the first trial at which frame_err reaches stop_frame_err (all trials if it never does
 * or stop_frame_err <= 0).  A run already stopped (d_run[FRAME_ERR] >= stop_frame_err) consumes nothing. */
int scldpc_accumulate_run_device(int32_t ntrials, const int32_t *d_counters, int64_t stop_frame_err,
                                 int64_t *d_run, void *stream);

/* simulate_sc_ldpc's bookkeeping over a batch, IN TRIAL ORDER, with its stop rule (PD:668-699): adds the rows of
 * scldpc_peel_sweep_device (d_out int32 [ntrials][8]) of trials 0..k into d_run[SCLDPC_NPEELRUN] (int64, in place), k =
 * the first trial at which the number of failed trials (`num_fuckups`, #lost >= 1) reaches max_fuckups (all trials if
 * it never does or max_fuckups <= 0).  A run that has already reached it consumes nothing. */
enum {
    SCLDPC_PR_TRIALS = 0,       /* o + 1 of the reference's loop (PD:635, 700)      */
    SCLDPC_PR_FUCKUPS = 1,      /* num_fuckups           (PD:668-670)                */
    SCLDPC_PR_LOST = 2,         /* total_failed          (PD:671)                    */
    SCLDPC_PR_FUCKUPS_EXP = 3,  /* num_fuckups_truncated (PD:691-693)                */
    SCLDPC_PR_LOST_EXP = 4,     /* total_failed_expurgated (PD:680, 694)             */
    SCLDPC_PR_BLOCKS_EXP = 5,   /* total_blocks_failed_exp (PD:684-689, 695)         */
    SCLDPC_NPEELRUN = 8
};
int scldpc_accumulate_peel_device(int32_t ntrials, const int32_t *d_out, int64_t max_fuckups, int64_t *d_run, void *stream);

/* Soft doping of the Python path (gen_users_sc_ldpc_doping, PD:176-183: `erasure[pos*M : pos*M + int(alpha*M)] = False`):
 * clears the channel bits of VNs vn_lo .. vn_hi-1 of every trial in d_chan_bits [ntrials][ceil(n/32)]. */
int scldpc_clear_channel_range_device(const scldpc_code_params *p, int32_t ntrials, int32_t vn_lo, int32_t vn_hi,
                                      uint32_t *d_chan_bits, void *stream);

/* LDS bytes the full-BP kernel needs for this ensemble (<= 163840 to be launchable), or a negative error. */
int64_t scldpc_full_bp_lds_bytes(const scldpc_code_params *p);

#ifdef __cplusplus
}
#endif
#endif
