/*
 * TEST INFRASTRUCTURE — CPU oracle for the reference's STREAMING mode (compiled out in the shipped source by
 * `#undef CIRCULAR`, BPF:33-34; BPF = simulators_sc_ldpc/bp_decoding/
 * SC_LDPC_Simulator_BPDecoder_BEC_full_BP_LimIter_OlmosRandomEnsemble.c):
 * a circular buffer of L positions, one position generated and one decoded per step (main_streaming BPF:1934-2054).
 *
 * Parity status: PINNED — tests/test_stream_oracle.py checks it against the reference itself compiled with the
 * #undef removed (oracle/_ref/ref_stream_*, oracle/ref_stream_tail.c) and against fixtures made from it.
 *
 * rng_mode 0: glibc random() exactly as the reference draws (fill_interleaver_pos BPF:1763-1787, then the channel
 *             BPF:1621-1654; doped positions draw nothing).
 * rng_mode 1: Philox4x32-10 keyed like the device's streaming kernel (CPU twin of csrc/stream_bp.hip).
 * decoder 0 : literal per-edge messages with the reference's list-position indexing (vn_update / cn_update
 *             BPF:1285-1336, decodeBP_SW_circular BPF:1403-1500).
 * decoder 1 : node-level model: S = what the CNs see, live window CNs with one unknown neighbour
 *             release it if it lies in the VN window.
 * Only tests/ may use this.
 */
#include "scldpc_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct orc_stream {
    orc_params p;                 /* p.L = buffer length */
    int n, nk, W, ndoped, doped[32];
    double eps;
    int rng_mode, decoder;
    orc_rng rng;
    uint64_t seed, sid;           /* philox key / stream id */
    int32_t *perm;                /* perm_code */
    int32_t *vn;                  /* [n][dv+1]   VNdegree */
    int32_t *cn;                  /* [nk][dc+1]  CNdegree */
    int32_t *inter;               /* [L][S]      interleaverCN */
    uint8_t *lji, *lij;           /* [n][dv], [nk][dc]  messages (list-position indexed) */
    uint8_t *chan, *erased;       /* [n] LLRsChannel, VNerased */
    uint8_t *S, *live;            /* node-level model */
    int gen_pos, pos;
    int32_t ne, be, ee, bee, gb, gbl, gbe, gble;
} orc_stream;

/* is_position_doped_streaming (BPF:1589-1612): the doping pattern repeats with period last_doped + 1 */
int orc_stream_is_doped(int pos, int ndoped, const int32_t *doped)
{
    if (ndoped == 0) return 0;
    int left = doped[0], period = doped[ndoped - 1] + 1, m = pos % period;
    if (m < left) return 0;
    for (int i = 0; i < ndoped; i++) if (doped[i] == m) return 1;
    return 0;
}

/* calc_sw_range_circular_vn / _cn (BPF:1169-1218): the window of stream position pos inside the circular buffer.
 * out = { start_vn, end_vn, end_vn_wrap, is_wrap_vn, start_cn, end_cn, end_cn_wrap, is_wrap_cn } (positions) */
void orc_stream_sw_range(int pos, int L, int W, int ms, int32_t out[8])
{
    const int posW = pos % L;
    int sc = posW, ec = posW + W, ecw = 0, wc = 0;
    if (ec > L) { wc = 1; ecw = (pos + W) % L; ec = L; }
    int sv, ev, evw = 0, wv = 0;
    if (pos <= ms) { sv = 0; ev = posW + W; }
    else { sv = (pos - ms) % L; ev = sv + ms + W; if (ev > L) { wv = 1; evw = (pos + W) % L; ev = L; } }
    out[0] = sv; out[1] = ev; out[2] = evw; out[3] = wv; out[4] = sc; out[5] = ec; out[6] = ecw; out[7] = wc;
}

static int is_doped(const orc_stream *s, int pos)
{
    int32_t d[32];
    for (int i = 0; i < s->ndoped; i++) d[i] = s->doped[i];
    return orc_stream_is_doped(pos, s->ndoped, d);
}

typedef struct { uint32_t key; int32_t sock; } skeyed;
static int skeyed_cmp(const void *a, const void *b)
{
    const skeyed *x = (const skeyed *)a, *y = (const skeyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->sock < y->sock ? -1 : (x->sock > y->sock);
}

/* fill_interleaver_pos (BPF:1763-1787); cn_position = absolute CN position (for the Philox key) */
static void fill_interleaver(orc_stream *s, int slot, int cn_position)
{
    const int S = s->p.cns_pos * s->p.dc, dc = s->p.dc;
    if (s->rng_mode == 0) {
        for (int i = 0; i < S; i++) {
            int pick = i + orc_random(&s->rng) % (S - i);
            int32_t t = s->perm[i]; s->perm[i] = s->perm[pick]; s->perm[pick] = t;
        }
        for (int i = 0; i < S; i++) s->inter[(size_t)slot * S + i] = slot * s->p.cns_pos + s->perm[i] / dc;
    } else {
        skeyed *ks = (skeyed *)malloc(sizeof(skeyed) * (size_t)S);
        const uint32_t key[2] = {(uint32_t)s->seed, (uint32_t)(s->seed >> 32)};
        for (int q = 0; q < (S + 3) / 4; q++) {
            uint32_t ctr[4] = {(uint32_t)q, (uint32_t)cn_position, (uint32_t)s->sid, (uint32_t)(s->sid >> 32)}, r[4];
            orc_philox4x32_10(ctr, key, r);
            for (int u = 0; u < 4 && q * 4 + u < S; u++) { ks[q * 4 + u].key = r[u]; ks[q * 4 + u].sock = q * 4 + u; }
        }
        qsort(ks, (size_t)S, sizeof(skeyed), skeyed_cmp);
        for (int rank = 0; rank < S; rank++) s->inter[(size_t)slot * S + ks[rank].sock] = slot * s->p.cns_pos + rank / dc;
        free(ks);
    }
}

static void generate_pos(orc_stream *s, int pos)                        /* generate_stream_pos BPF:1927-1932 */
{
    const int L = s->p.L, dv = s->p.dv, dc = s->p.dc, V = s->p.vns_pos, C = s->p.cns_pos, S = C * dc;
    const int pb = pos % L, pgi = (pos + dv - 1) % L;
    fill_interleaver(s, pgi, pos + dv - 1);                               /* BPF:1830 */
    for (int c = 0; c < C; c++) s->cn[(size_t)(pgi * C + c) * (dc + 1)] = 0;   /* BPF:1832-1837 */
    for (int t = 0; t < V; t++) {                                        /* BPF:1841-1854 */
        int VN = pb * V + t;
        s->vn[(size_t)VN * (dv + 1)] = dv;
        for (int i = 0; i < dv; i++) {
            int CN = s->inter[(size_t)((pb + i) % L) * S + dv * t + i];
            s->vn[(size_t)VN * (dv + 1) + 1 + i] = CN;
            int32_t *row = &s->cn[(size_t)CN * (dc + 1)];
            row[1 + row[0]] = VN; row[0]++;
        }
    }
    /* generate_channel_doped_circular (BPF:1621-1654) */
    if (!is_doped(s, pos)) {
        if (s->rng_mode == 0) {
            for (int j = pb * V; j < (pb + 1) * V; j++)
                s->chan[j] = ((double)orc_random(&s->rng) / 2147483647.0 >= s->eps) ? 0 : 1;
        } else {
            double x = s->eps * 2147483647.0, c = (double)(uint64_t)x;
            if (c < x) c += 1.0;
            const uint32_t thresh = (uint32_t)c, key[2] = {(uint32_t)s->seed, (uint32_t)(s->seed >> 32)};
            for (int q = 0; q < (V + 3) / 4; q++) {
                uint32_t ctr[4] = {(uint32_t)q, 0x80000000u | (uint32_t)pos, (uint32_t)s->sid, (uint32_t)(s->sid >> 32)}, r[4];
                orc_philox4x32_10(ctr, key, r);
                for (int u = 0; u < 4 && q * 4 + u < V; u++) s->chan[pb * V + q * 4 + u] = (r[u] >> 1) < thresh;
            }
        }
    } else {
        for (int j = pb * V; j < (pb + 1) * V; j++) s->chan[j] = 0;
    }
    /* initialize_messages_circular (BPF:1149-1166) */
    for (int j = pb * V; j < (pb + 1) * V; j++) {
        for (int i = 0; i < dv; i++) s->lji[(size_t)j * dv + i] = s->chan[j];
        s->S[j] = s->chan[j];
    }
    for (int c = pb * C; c < (pb + 1) * C; c++) {
        for (int k = 0; k < s->cn[(size_t)c * (dc + 1)]; k++) s->lij[(size_t)c * dc + k] = 1;
        s->live[c] = 0;
    }
}

static int slot_in_vn(const orc_stream *s, int vn, int cnid)
{
    const int dv = s->p.dv;
    for (int m = 0; m < s->vn[(size_t)vn * (dv + 1)]; m++) if (s->vn[(size_t)vn * (dv + 1) + 1 + m] == cnid) return m;
    return 0;
}
static int slot_in_cn(const orc_stream *s, int cnid, int vn)
{
    const int dc = s->p.dc;
    for (int m = 0; m < s->cn[(size_t)cnid * (dc + 1)]; m++) if (s->cn[(size_t)cnid * (dc + 1) + 1 + m] == vn) return m;
    return 0;
}

static void cn_update(orc_stream *s, int c0, int c1)                    /* BPF:1312-1336 */
{
    const int dc = s->p.dc, dv = s->p.dv;
    for (int i = c0; i < c1; i++) {
        const int32_t *row = &s->cn[(size_t)i * (dc + 1)];
        for (int j = 0; j < row[0]; j++) {
            int er = 0;
            for (int k = 0; k < row[0]; k++)
                if (k != j) er += s->lji[(size_t)row[1 + k] * dv + slot_in_vn(s, row[1 + k], i)];
            s->lij[(size_t)i * dc + j] = er > 0;
        }
    }
}
static void vn_update(orc_stream *s, int j0, int j1)                    /* BPF:1285-1310 */
{
    const int dc = s->p.dc, dv = s->p.dv;
    for (int j = j0; j < j1; j++) {
        const int32_t *row = &s->vn[(size_t)j * (dv + 1)];
        for (int i = 0; i < row[0]; i++) {
            int er = 0;
            for (int k = 0; k < row[0]; k++)
                if (k != i) er += s->lij[(size_t)row[1 + k] * dc + slot_in_cn(s, row[1 + k], j)];
            s->lji[(size_t)j * dv + i] = (er < row[0] - 1 || s->chan[j] == 0) ? 0 : 1;
        }
    }
}
static int app_erased(const orc_stream *s, int j)                       /* BPF:1345-1357 */
{
    const int dc = s->p.dc, dv = s->p.dv;
    const int32_t *row = &s->vn[(size_t)j * (dv + 1)];
    int er = s->chan[j];
    for (int i = 0; i < row[0]; i++) er += s->lij[(size_t)row[1 + i] * dc + slot_in_cn(s, row[1 + i], j)];
    return er == row[0] + 1;
}

static int deg_two_ss(const orc_stream *s, int slot)                    /* get_deg_two_ss BPF:1227-1283 */
{
    const int V = s->p.vns_pos, dv = s->p.dv, dc = s->p.dc;
    int cnt = 0;
    for (int a = 0; a < V; a++) {
        int va = slot * V + a;
        if (!s->erased[va]) continue;
        cnt++;
        for (int b = a + 1; b < V; b++) {
            int vb = slot * V + b;
            if (!s->erased[vb]) continue;
            int same = 1, others_ok = 1;
            for (int i = 0; i < s->vn[(size_t)va * (dv + 1)]; i++) {
                int c = s->vn[(size_t)va * (dv + 1) + 1 + i], has_b = 0;
                const int32_t *row = &s->cn[(size_t)c * (dc + 1)];
                for (int k = 0; k < row[0]; k++) {
                    int v = row[1 + k];
                    if (v == vb) has_b = 1;
                    else if (v != va && s->erased[v]) others_ok = 0;
                }
                if (!has_b) { same = 0; break; }
                if (!others_ok) break;
            }
            if (same && others_ok) cnt -= 2;
        }
    }
    return cnt;
}

/* decodeBP_SW_circular (BPF:1403-1500); returns NumErasuresPos */
static int decode_pos(orc_stream *s, int pos)
{
    const int L = s->p.L, dv = s->p.dv, dc = s->p.dc, V = s->p.vns_pos, C = s->p.cns_pos, ms = dv - 1, W = s->W;
    int32_t rg[8];
    orc_stream_sw_range(pos, L, W, ms, rg);                              /* BPF:1169-1218 */
    const int sv = rg[0], ev = rg[1], evw = rg[2], sc = rg[4], ec = rg[5], ecw = rg[6];
    const int j0 = sv * V, j1 = ev * V, j1w = evw * V, c0 = sc * C, c1 = ec * C, c1w = ecw * C;
    int prec = s->n, nep = 0;
    for (;;) {
        nep = 0;
        int term = 0;
        if (s->decoder == 0) {
            cn_update(s, c0, c1); cn_update(s, 0, c1w);
            vn_update(s, j0, j1); vn_update(s, 0, j1w);
            if (pos >= ms)
                for (int j = j0; j < j0 + V; j++) { s->erased[j] = (uint8_t)app_erased(s, j); nep += s->erased[j]; }
            for (int j = j0; j < j1; j++) term += app_erased(s, j);
            for (int j = 0; j < j1w; j++) term += app_erased(s, j);
        } else {
            /* node level: resid of window CNs from S; a window VN is released by a live CN that sees only it */
            for (int part = 0; part < 2; part++) {
                int a = part ? 0 : c0, b = part ? c1w : c1;
                for (int c = a; c < b; c++) {
                    const int32_t *row = &s->cn[(size_t)c * (dc + 1)];
                    int r = 0;
                    for (int k = 0; k < row[0]; k++) r += s->S[row[1 + k]];
                    s->lij[(size_t)c * dc] = (uint8_t)(r > 255 ? 255 : r);   /* reuse: resid of CN c */
                    s->live[c] = 1;
                }
            }
            uint8_t *kill = s->lji;                                      /* reuse as scratch [n] */
            for (int part = 0; part < 2; part++) {
                int a = part ? 0 : j0, b = part ? j1w : j1;
                for (int j = a; j < b; j++) {
                    kill[j] = 0;
                    if (!s->S[j]) continue;
                    const int32_t *row = &s->vn[(size_t)j * (dv + 1)];
                    for (int i = 0; i < row[0]; i++) {
                        int c = row[1 + i];
                        if (s->live[c] && s->lij[(size_t)c * dc] == 1) { kill[j] = 1; break; }
                    }
                }
            }
            for (int part = 0; part < 2; part++) {
                int a = part ? 0 : j0, b = part ? j1w : j1;
                for (int j = a; j < b; j++) { if (kill[j]) s->S[j] = 0; term += s->S[j]; }
            }
            if (pos >= ms)
                for (int j = j0; j < j0 + V; j++) { s->erased[j] = s->S[j]; nep += s->S[j]; }
        }
        if (term == 0) break;
        if (term == prec) break;
        prec = term;
    }
    if (nep > 0) s->be++;
    int ep = pos - 2 * dv + 1;
    if (ep >= 0) {
        int c = deg_two_ss(s, ep % L);
        if (c > 0) { s->ee += c; s->bee++; }
    }
    return nep;
}

orc_stream *orc_stream_new(const orc_params *p, int rng_mode, int decoder, uint64_t seed, uint64_t sid, double eps,
                           int W, int ndoped, const int *doped)
{
    orc_stream *s = (orc_stream *)calloc(1, sizeof *s);
    s->p = *p; s->n = p->vns_pos * p->L; s->nk = p->cns_pos * p->L;      /* circular: no termination (BPF:39) */
    s->W = W; s->eps = eps; s->rng_mode = rng_mode; s->decoder = decoder; s->seed = seed; s->sid = sid;
    s->ndoped = ndoped;
    for (int i = 0; i < ndoped && i < 32; i++) s->doped[i] = doped[i];
    const int S = p->cns_pos * p->dc;
    s->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
    for (int i = 0; i < S; i++) s->perm[i] = i;                          /* inizio_sim BPF:308-311 */
    s->vn = (int32_t *)calloc((size_t)s->n * (p->dv + 1), sizeof(int32_t));
    s->cn = (int32_t *)calloc((size_t)s->nk * (p->dc + 1), sizeof(int32_t));
    s->inter = (int32_t *)calloc((size_t)p->L * S, sizeof(int32_t));
    s->lji = (uint8_t *)calloc((size_t)s->n * p->dv + (size_t)s->n, 1);
    s->lij = (uint8_t *)calloc((size_t)s->nk * p->dc, 1);
    s->chan = (uint8_t *)calloc((size_t)s->n, 1); s->erased = (uint8_t *)calloc((size_t)s->n, 1);
    s->S = (uint8_t *)calloc((size_t)s->n, 1); s->live = (uint8_t *)calloc((size_t)s->nk, 1);
    if (rng_mode == 0) orc_srandom(&s->rng, (unsigned)seed);
    for (int pos = 0; pos < p->dv - 1; pos++) fill_interleaver(s, pos, pos);   /* initialize_arrays_circular BPF:1808-1813 */
    for (s->gen_pos = 0; s->gen_pos < p->L / 2; s->gen_pos++) generate_pos(s, s->gen_pos);   /* BPF:2007-2012 */
    return s;
}

/* one pass of the `for (pos = 0;; pos++)` loop (BPF:2015-2046); out[10] = pos, nep, then the 8 running counters */
void orc_stream_step(orc_stream *s, int32_t *out)
{
    const int dv = s->p.dv, V = s->p.vns_pos, pos = s->pos;
    int pd = pos - dv + 1, pe = pos - 2 * dv + 1;
    if (pd >= 0 && !is_doped(s, pd)) { s->gb += V; s->gbl += 1; }
    if (pe >= 0 && !is_doped(s, pe)) { s->gbe += V; s->gble += 1; }
    int nep = decode_pos(s, pos);
    s->ne += nep;
    generate_pos(s, s->gen_pos); s->gen_pos++;
    s->pos++;
    out[0] = pos; out[1] = nep; out[2] = s->ne; out[3] = s->be; out[4] = s->ee; out[5] = s->bee;
    out[6] = s->gb; out[7] = s->gbl; out[8] = s->gbe; out[9] = s->gble;
}

/* VNerased of the position decided by the last step (pos - dv + 1), V bytes; returns 0 if none was decided yet */
int orc_stream_last_erased(const orc_stream *s, uint8_t *out)
{
    int pd = s->pos - 1 - s->p.dv + 1;
    if (pd < 0) return 0;
    memcpy(out, s->erased + (size_t)(pd % s->p.L) * s->p.vns_pos, (size_t)s->p.vns_pos);
    return 1;
}

void orc_stream_free(orc_stream *s)
{
    free(s->perm); free(s->vn); free(s->cn); free(s->inter); free(s->lji); free(s->lij);
    free(s->chan); free(s->erased); free(s->S); free(s->live); free(s);
}
