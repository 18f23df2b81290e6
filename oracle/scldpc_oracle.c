/*
 * TEST INFRASTRUCTURE — CPU oracle (see scldpc_oracle.h for scope, citations and parity status).
 * Own code, own data layout; follows the reference's algorithms line by line only in behaviour.
 */
#include "scldpc_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* glibc TYPE_3 additive-feedback generator: x[i] = x[i-31] + x[i-3] (mod 2^32), output >> 1.  */
/* Seeding: Park–Miller 16807 by Schrage's method, then 310 outputs discarded (random_r.c).    */
/* ------------------------------------------------------------------------------------------ */
void orc_srandom(orc_rng *g, unsigned seed)
{
    int32_t word = (int32_t)(seed ? seed : 1u);
    g->r[0] = word;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        g->r[i] = word;
    }
    g->f = 3; g->b = 0;
    for (int k = 0; k < 310; k++) (void)orc_random(g);
}

int32_t orc_random(orc_rng *g)
{
    uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
    g->r[g->f] = (int32_t)v;
    if (++g->f >= 31) g->f = 0;
    if (++g->b >= 31) g->b = 0;
    return (int32_t)(v >> 1);
}

uint64_t orc_fnv1a(const void *p, uint64_t nbytes, uint64_t h)
{
    const unsigned char *c = (const unsigned char *)p;
    for (uint64_t i = 0; i < nbytes; i++) { h ^= c[i]; h *= 1099511628211ULL; }
    return h;
}

void orc_perm_identity(const orc_params *p, int32_t *perm_code)
{
    for (int i = 0; i < p->cns_pos * p->dc; i++) perm_code[i] = i;   /* BPF:308-311 */
}

/* ------------------------------------------------------------------------------------------ */
/* generate_code (BPF:1656-1761)                                                               */
/* ------------------------------------------------------------------------------------------ */
void orc_generate_code(const orc_params *p, orc_rng *g, int32_t *perm_code,
                       int32_t *vn_adj, int32_t *cn_ptr, int32_t *cn_adj)
{
    const int dv = p->dv, dc = p->dc, L = p->L, S = p->cns_pos * p->dc;
    const int n = orc_n(p), nk = orc_nk(p), D = L + dv - 1;
    int32_t *sock_cn = (int32_t *)malloc(sizeof(int32_t) * (size_t)D * S);

    for (int pos = 0; pos < D; pos++) {
        /* in-place Fisher–Yates continuing from the previous arrangement (BPF:1682-1688) */
        for (int i = 0; i < S; i++) {
            int pick = i + orc_random(g) % (S - i);
            int32_t t = perm_code[i]; perm_code[i] = perm_code[pick]; perm_code[pick] = t;
        }
        for (int i = 0; i < S; i++)                                   /* BPF:1693 */
            sock_cn[(size_t)pos * S + i] = pos * p->cns_pos + perm_code[i] / dc;
    }
    /* VN (pos,t), edge i → socket dv*t+i of CN position pos+i (BPF:1703-1716) */
    for (int j = 0; j < n; j++) {
        int pos = j / p->vns_pos, t = j % p->vns_pos;
        for (int i = 0; i < dv; i++)
            vn_adj[(size_t)j * dv + i] = sock_cn[(size_t)(pos + i) * S + dv * t + i];
    }
    /* CN lists in visiting order (counting pass + stable fill) */
    memset(cn_ptr, 0, sizeof(int32_t) * ((size_t)nk + 1));
    for (size_t e = 0; e < (size_t)n * dv; e++) cn_ptr[vn_adj[e] + 1]++;
    for (int c = 0; c < nk; c++) cn_ptr[c + 1] += cn_ptr[c];
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)nk);
    memcpy(fill, cn_ptr, sizeof(int32_t) * (size_t)nk);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < dv; i++) cn_adj[fill[vn_adj[(size_t)j * dv + i]]++] = j;
    free(fill);
    free(sock_cn);
}

/* channel_doped (BPF:1547-1574); unif_ch = random()/RAND_MAX (BPF:370) */
void orc_channel(const orc_params *p, orc_rng *g, double eps, int ndoped, const int *doped, uint8_t *chan)
{
    const int n = orc_n(p);
    for (int j = 0; j < n; j++) {
        double u = (double)orc_random(g) / 2147483647.0;
        chan[j] = (u >= eps) ? 0 : 1;
    }
    for (int d = 0; d < ndoped; d++)
        for (int j = doped[d] * p->vns_pos; j < (doped[d] + 1) * p->vns_pos; j++) chan[j] = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Edge bookkeeping for the literal decoders.  A directed message lives on edge e = j*dv+i.    */
/* cn_edge[k] = the edge id behind entry k of the CN-side list (replaces the reference's       */
/* linear neighbour-slot searches BPF:956,997,1017; graphs of this ensemble are simple because */
/* the dv edges of a VN land in dv different positions).                                       */
/* ------------------------------------------------------------------------------------------ */
static int32_t *build_cn_edge(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr)
{
    const int n = orc_n(p), nk = orc_nk(p), dv = p->dv;
    int32_t *cn_edge = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * dv);
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)nk);
    memcpy(fill, cn_ptr, sizeof(int32_t) * (size_t)nk);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < dv; i++) cn_edge[fill[vn_adj[(size_t)j * dv + i]]++] = j * dv + i;
    free(fill);
    return cn_edge;
}

/* CN update for CNs [c0,c1): c2v[e] = 1 iff some OTHER incoming v2c is 1 (BPF:943-968).
 * Returns nothing; optionally maintains the deg-1 statistic of BPF:969-978. */
static void cn_update(const int32_t *cn_ptr, const int32_t *cn_edge, int c0, int c1,
                      const uint8_t *v2c, uint8_t *c2v, uint8_t *resolved, int *deg1)
{
    for (int c = c0; c < c1; c++) {
        int tot = 0, out_resolved = 0;
        for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) tot += v2c[cn_edge[k]];
        for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) {
            int e = cn_edge[k];
            if (tot - v2c[e] > 0) c2v[e] = 1; else { c2v[e] = 0; out_resolved++; }
        }
        if (resolved && !resolved[c] && out_resolved > 0) {
            if (out_resolved == 1) (*deg1)++;
            resolved[c] = 1;
        }
    }
}

/* VN update for VNs [j0,j1): v2c[e] = 0 iff channel known or some OTHER incoming c2v is 0 (BPF:985-1005) */
static void vn_update(int dv, int j0, int j1, const uint8_t *chan, const uint8_t *c2v, uint8_t *v2c)
{
    for (int j = j0; j < j1; j++) {
        int tot = 0;
        for (int i = 0; i < dv; i++) tot += c2v[j * dv + i];
        for (int i = 0; i < dv; i++) {
            int others = tot - c2v[j * dv + i];
            v2c[j * dv + i] = (others < dv - 1 || chan[j] == 0) ? 0 : 1;
        }
    }
}

/* a-posteriori: erased iff channel erased and every incoming c2v erased (BPF:1009-1034) */
static inline int vn_app_erased(int dv, int j, const uint8_t *chan, const uint8_t *c2v)
{
    int tot = chan[j];
    for (int i = 0; i < dv; i++) tot += c2v[j * dv + i];
    return tot == dv + 1;
}

void orc_expurgate(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                   const int32_t *cn_adj, const uint8_t *erased, int first_only,
                   int32_t *num_blocks_err, int32_t *num_erasures_exp, int32_t *num_blocks_err_exp)
{
    const int dv = p->dv, V = p->vns_pos;
    int first_done = 0;
    *num_blocks_err = 0; *num_erasures_exp = 0; *num_blocks_err_exp = 0;
    for (int pos = 0; pos < p->L; pos++) {
        int cnt = 0, cnt_exp = 0;
        for (int a = 0; a < V; a++) {
            int va = pos * V + a;
            if (!erased[va]) continue;
            cnt++; cnt_exp++;
            for (int b = a + 1; b < V; b++) {
                int vb = pos * V + b;
                if (!erased[vb]) continue;
                int same = 1, others_ok = 1;
                for (int i = 0; i < dv; i++) {                         /* BPF:1089-1111 */
                    int c = vn_adj[(size_t)va * dv + i], has_b = 0;
                    for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) {
                        int v = cn_adj[k];
                        if (v == vb) has_b = 1;
                        else if (v != va && erased[v]) others_ok = 0;
                    }
                    if (!has_b) { same = 0; break; }
                    if (!others_ok) break;
                }
                if (same && others_ok) cnt_exp -= 2;                    /* BPF:1112-1117 */
            }
        }
        if (first_only) {
            if (cnt > 0) (*num_blocks_err)++;                           /* BPF:1123-1125 */
            if (cnt_exp > 0 && !first_done) {                           /* BPF:1126-1132 */
                first_done = 1;
                *num_erasures_exp += cnt_exp; (*num_blocks_err_exp)++;
            }
        } else if (cnt_exp > 0) {                                       /* BPW:903-907 */
            *num_erasures_exp += cnt_exp; (*num_blocks_err_exp)++;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* decodeBP, literal (BPF:900-1140 + the BPT deltas)                                           */
/* ------------------------------------------------------------------------------------------ */
void orc_decode_bp_literal(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                           const int32_t *cn_adj, const uint8_t *chan, int max_it, int is_term,
                           uint8_t *erased, orc_row *rows, int rows_cap, orc_result *res)
{
    const int n = orc_n(p), nk = orc_nk(p), dv = p->dv;
    const size_t E = (size_t)n * dv;
    int32_t *cn_edge = build_cn_edge(p, vn_adj, cn_ptr);
    uint8_t *v2c = (uint8_t *)malloc(E), *c2v = (uint8_t *)malloc(E);
    uint8_t *resolved = (uint8_t *)calloc((size_t)nk, 1);
    memset(res, 0, sizeof *res);

    for (int j = 0; j < n; j++) for (int i = 0; i < dv; i++) v2c[j * dv + i] = chan[j];   /* BPF:913-917 */
    memset(c2v, 0, E);
    const int cn_lim = is_term ? nk : p->L * p->cns_pos;                                   /* BPT:944-948 */
    if (!is_term)                                                                          /* BPT:922-925 */
        for (int c = cn_lim; c < nk; c++)
            for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) c2v[cn_edge[k]] = 1;

    int prec = n, ne = 0, iter = 0;
    for (;;) {
        int deg1 = 0;
        cn_update(cn_ptr, cn_edge, 0, cn_lim, v2c, c2v, resolved, &deg1);
        vn_update(dv, 0, n, chan, c2v, v2c);
        ne = 0; int first = n;
        for (int j = 0; j < n; j++) {
            erased[j] = (uint8_t)vn_app_erased(dv, j, chan, c2v);
            if (erased[j]) { ne++; if (j < first) first = j; }
        }
        if (rows && res->iterations < rows_cap) {
            rows[res->iterations].deg1 = deg1;
            rows[res->iterations].recovered = prec - ne;
            rows[res->iterations].first_pos = first / p->vns_pos;
        }
        res->iterations++;
        if (deg1 < prec - ne && iter > 0) { res->status = -1; break; }                     /* BPF:1035-1039 */
        if (ne == 0) break;
        if (ne == prec) break;
        prec = ne;
        iter++;
        if (max_it > 0 && iter >= max_it) break;                                           /* BPF:1065 */
    }
    res->num_erasures = ne;
    orc_expurgate(p, vn_adj, cn_ptr, cn_adj, erased, 1,
                  &res->num_blocks_err, &res->num_erasures_exp, &res->num_blocks_err_exp);
    free(cn_edge); free(v2c); free(c2v); free(resolved);
}

/* ------------------------------------------------------------------------------------------ */
/* decodeBP_SW, literal (square: BPW:628-912; classical: BPF:627-897)                          */
/* ------------------------------------------------------------------------------------------ */
void orc_decode_sw_literal(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                           const int32_t *cn_adj, const uint8_t *chan, int W, int max_it, int init_it,
                           int square, uint8_t *erased, orc_result *res)
{
    const int n = orc_n(p), nk = orc_nk(p), dv = p->dv, ms = dv - 1, V = p->vns_pos, C = p->cns_pos;
    const size_t E = (size_t)n * dv;
    int32_t *cn_edge = build_cn_edge(p, vn_adj, cn_ptr);
    uint8_t *v2c = (uint8_t *)malloc(E), *c2v = (uint8_t *)malloc(E);
    memset(res, 0, sizeof *res);
    /* VNerased keeps whatever generate_code left there = 0 (BPF:1709) until a window decides it */
    memset(erased, 0, (size_t)n);

    for (int j = 0; j < n; j++) for (int i = 0; i < dv; i++) v2c[j * dv + i] = chan[j];   /* BPW:650-654 */
    memset(c2v, 1, E);                                                                     /* BPW:655-659 */

    const int last = square ? p->L : p->L + ms;
    for (int posW = 0; posW < last; posW++) {
        int c0 = posW * C, c1 = c0 + W * C; if (c1 > nk) c1 = nk;
        int j0, j1;
        if (square)            { j0 = posW * V;        j1 = j0 + W * V; }                  /* BPW:691-693 */
        else if (posW <= ms)   { j0 = 0;               j1 = (W + posW) * V; }              /* BPF:673-678 */
        else                   { j0 = (posW - ms) * V; j1 = j0 + (W + ms) * V; }           /* BPF:680-684 */
        if (j1 > n) j1 = n;
        int cap = square ? (posW == 0 ? init_it : max_it) : max_it;                        /* BPW:699-702 */
        int iter = 0, prec_term = n, ne_pos = 0;
        do {
            ne_pos = 0;
            cn_update(cn_ptr, cn_edge, c0, c1, v2c, c2v, NULL, NULL);
            vn_update(dv, j0, j1, chan, c2v, v2c);
            if (square || posW >= ms)                                                      /* BPF:745 */
                for (int j = j0; j < j0 + V; j++) {
                    erased[j] = (uint8_t)vn_app_erased(dv, j, chan, c2v);
                    ne_pos += erased[j];
                }
            int ne_term = 0;
            for (int j = j0; j < j1; j++) ne_term += vn_app_erased(dv, j, chan, c2v);
            res->iterations++;
            if (ne_term == 0) break;
            if (ne_term == prec_term) break;
            prec_term = ne_term;
            iter++;
        } while (iter < cap);
        res->num_erasures += ne_pos;
        if (ne_pos > 0) res->num_blocks_err++;
        if (posW >= ms && posW <= W - 2) res->num_erasures_p1 += ne_pos;                    /* BPW:846-847 */
    }
    int32_t be_unused;
    orc_expurgate(p, vn_adj, cn_ptr, cn_adj, erased, 0, &be_unused,
                  &res->num_erasures_exp, &res->num_blocks_err_exp);
    free(cn_edge); free(v2c); free(c2v);
}

/* ------------------------------------------------------------------------------------------ */
/* Node-level models.  One bit per VN + a residual counter per CN.        */
/* ------------------------------------------------------------------------------------------ */
void orc_decode_bp_peel(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                        const int32_t *cn_adj, const uint8_t *chan, int max_it, int is_term,
                        uint8_t *erased, orc_row *rows, int rows_cap, orc_result *res)
{
    const int n = orc_n(p), nk = orc_nk(p), dv = p->dv;
    const int cn_lim = is_term ? nk : p->L * p->cns_pos;
    int32_t *resid = (int32_t *)calloc((size_t)nk, sizeof(int32_t));
    int64_t *idsum = (int64_t *)calloc((size_t)nk, sizeof(int64_t));
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)nk);
    int32_t *nxt = (int32_t *)malloc(sizeof(int32_t) * (size_t)nk);
    int32_t *gone = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    memset(res, 0, sizeof *res);

    int ne = 0;
    for (int j = 0; j < n; j++) {
        erased[j] = chan[j];
        if (!chan[j]) continue;
        ne++;
        for (int i = 0; i < dv; i++) { int c = vn_adj[(size_t)j * dv + i]; resid[c]++; idsum[c] += j; }
    }
    int ncur = 0, extra0 = 0;
    for (int c = 0; c < cn_lim; c++) {
        if (resid[c] == 1) cur[ncur++] = c;
        else if (resid[c] == 0 && cn_ptr[c + 1] - cn_ptr[c] == 1) extra0++;   /* BPF:969-978 quirk */
    }
    int prec = n, iter = 0, first = 0;
    /* (ne above is the channel count; the first row reports prec - ne_after with prec = n) */
    for (;;) {
        int deg1 = ncur + (iter == 0 ? extra0 : 0);
        int ngone = 0;
        for (int q = 0; q < ncur; q++) {
            int c = cur[q];
            if (resid[c] != 1) continue;            /* cannot happen for a frontier CN; kept as a guard */
            int j = (int)idsum[c];
            if (erased[j]) { erased[j] = 0; gone[ngone++] = j; }
        }
        int nnxt = 0;
        for (int q = 0; q < ngone; q++) {
            int j = gone[q];
            for (int i = 0; i < dv; i++) {
                int c = vn_adj[(size_t)j * dv + i];
                resid[c]--; idsum[c] -= j;
                if (resid[c] == 1 && c < cn_lim) nxt[nnxt++] = c;   /* candidate; re-checked below */
            }
        }
        int keep = 0;
        for (int q = 0; q < nnxt; q++) if (resid[nxt[q]] == 1) nxt[keep++] = nxt[q];
        nnxt = keep;
        ne -= ngone;
        while (first < n && !erased[first]) first++;
        if (rows && res->iterations < rows_cap) {
            rows[res->iterations].deg1 = deg1;
            rows[res->iterations].recovered = prec - ne;
            rows[res->iterations].first_pos = first / p->vns_pos;
        }
        res->iterations++;
        if (deg1 < prec - ne && iter > 0) { res->status = -1; break; }
        if (ne == 0 || ne == prec) break;
        prec = ne; iter++;
        int32_t *t = cur; cur = nxt; nxt = t; ncur = nnxt;
        if (max_it > 0 && iter >= max_it) break;
    }
    res->num_erasures = ne;
    orc_expurgate(p, vn_adj, cn_ptr, cn_adj, erased, 1,
                  &res->num_blocks_err, &res->num_erasures_exp, &res->num_blocks_err_exp);
    free(resid); free(idsum); free(cur); free(nxt); free(gone);
}

void orc_decode_sw_peel(const orc_params *p, const int32_t *vn_adj, const int32_t *cn_ptr,
                        const int32_t *cn_adj, const uint8_t *chan, int W, int max_it, int init_it,
                        uint8_t *erased, orc_result *res)
{
    const int n = orc_n(p), nk = orc_nk(p), dv = p->dv, ms = dv - 1, V = p->vns_pos, C = p->cns_pos;
    uint8_t *S = (uint8_t *)malloc((size_t)n);          /* what the CNs "see" */
    int32_t *resid = (int32_t *)calloc((size_t)nk, sizeof(int32_t));
    uint8_t *live = (uint8_t *)calloc((size_t)nk, 1);
    uint8_t *kill = (uint8_t *)malloc((size_t)n);
    memset(res, 0, sizeof *res);
    memcpy(S, chan, (size_t)n);
    memset(erased, 0, (size_t)n);

    for (int posW = 0; posW < p->L; posW++) {
        int c0 = posW * C, c1 = c0 + W * C; if (c1 > nk) c1 = nk;
        int j0 = posW * V, j1 = j0 + W * V; if (j1 > n) j1 = n;
        int cap = posW == 0 ? init_it : max_it;
        int prec = n, iter = 0, ne_pos = 0;
        do {
            for (int c = c0; c < c1; c++) {
                int r = 0;
                for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) r += S[cn_adj[k]];
                resid[c] = r; live[c] = 1;
            }
            for (int j = j0; j < j1; j++) {
                kill[j] = 0;
                if (!S[j]) continue;
                for (int i = 0; i < dv; i++) {
                    int c = vn_adj[(size_t)j * dv + i];
                    if (live[c] && resid[c] == 1) { kill[j] = 1; break; }
                }
            }
            int term = 0;
            for (int j = j0; j < j1; j++) { if (kill[j]) S[j] = 0; term += S[j]; }
            ne_pos = 0;
            for (int j = j0; j < j0 + V; j++) { erased[j] = S[j]; ne_pos += S[j]; }
            res->iterations++;
            if (term == 0 || term == prec) break;
            prec = term; iter++;
        } while (iter < cap);
        res->num_erasures += ne_pos;
        if (ne_pos > 0) res->num_blocks_err++;
        if (posW >= ms && posW <= W - 2) res->num_erasures_p1 += ne_pos;
    }
    int32_t be_unused;
    orc_expurgate(p, vn_adj, cn_ptr, cn_adj, erased, 0, &be_unused,
                  &res->num_erasures_exp, &res->num_blocks_err_exp);
    free(S); free(resid); free(live); free(kill);
}

/* ------------------------------------------------------------------------------------------ */
void orc_trial(const orc_params *p, unsigned seed, double eps, int ndoped, const int *doped,
               int decoder, int W, int max_it, int init_it, int is_term,
               orc_result *res, uint64_t hashes[3], int32_t *n_chan_erased,
               int32_t *vn_adj_out, uint8_t *chan_out, uint8_t *erased_out,
               orc_row *rows, int rows_cap)
{
    const int n = orc_n(p), nk = orc_nk(p), dv = p->dv;
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)p->cns_pos * p->dc);
    int32_t *vn_adj = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * dv);
    int32_t *cn_ptr = (int32_t *)malloc(sizeof(int32_t) * ((size_t)nk + 1));
    int32_t *cn_adj = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * dv);
    uint8_t *chan = (uint8_t *)malloc((size_t)n), *erased = (uint8_t *)malloc((size_t)n);
    orc_rng g;

    orc_perm_identity(p, perm);
    orc_srandom(&g, seed);
    orc_generate_code(p, &g, perm, vn_adj, cn_ptr, cn_adj);
    orc_channel(p, &g, eps, ndoped, doped, chan);
    switch (decoder) {
    case 0: orc_decode_bp_literal(p, vn_adj, cn_ptr, cn_adj, chan, max_it, is_term, erased, rows, rows_cap, res); break;
    case 1: orc_decode_bp_peel(p, vn_adj, cn_ptr, cn_adj, chan, max_it, is_term, erased, rows, rows_cap, res); break;
    case 2: orc_decode_sw_literal(p, vn_adj, cn_ptr, cn_adj, chan, W, max_it, init_it, 1, erased, res); break;
    case 3: orc_decode_sw_peel(p, vn_adj, cn_ptr, cn_adj, chan, W, max_it, init_it, erased, res); break;
    default: orc_decode_sw_literal(p, vn_adj, cn_ptr, cn_adj, chan, W, max_it, init_it, 0, erased, res); break;
    }
    if (hashes) {
        const uint64_t h0 = 14695981039346656037ULL;
        hashes[0] = orc_fnv1a(vn_adj, sizeof(int32_t) * (uint64_t)n * dv, h0);
        uint64_t h = h0;
        for (int j = 0; j < n; j++) { int32_t v = chan[j]; h = orc_fnv1a(&v, 4, h); }
        hashes[1] = h;
        hashes[2] = orc_fnv1a(erased, (uint64_t)n, h0);
    }
    if (n_chan_erased) { int s = 0; for (int j = 0; j < n; j++) s += chan[j]; *n_chan_erased = s; }
    if (vn_adj_out) memcpy(vn_adj_out, vn_adj, sizeof(int32_t) * (size_t)n * dv);
    if (chan_out) memcpy(chan_out, chan, (size_t)n);
    if (erased_out) memcpy(erased_out, erased, (size_t)n);
    free(perm); free(vn_adj); free(cn_ptr); free(cn_adj); free(chan); free(erased);
}

/* ------------------------------------------------------------------------------------------ */
/* CPU twin of the device's throughput-mode sampler (fl_scaling_sc_ldpc_amd/csrc/sampler.hip). */
/* Not a restatement of reference code: the reference draws from one sequential glibc stream;  */
/* this is the same ensemble law (BPF:1656-1761, 1547-1574) keyed by Philox4x32-10             */
/* (Salmon, Moraes, Dror, Shaw, SC'11; published KAT vectors are checked in the tests).        */
/* ------------------------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct { uint32_t key; int32_t sock; } keyed;
static int keyed_cmp(const void *a, const void *b)
{
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->sock < y->sock ? -1 : (x->sock > y->sock);
}

/* ensemble: 0 = Olmos chain (generate_code's law), 1 = its tail-biting closure (sc_ldpc.py:41-45: L permutations,
 * edge i of VN position q lands in CN position (q+i) mod L), 2 = protograph chain (sc_ldpc_protograph.py:6-20: per VN
 * position dc/dv portions x dv uniform permutations of cns_pos; edge i of VN u of a portion → CN (q+i, perm_i[u])).
 * Protograph keys: the top pbits bits of a socket's 32-bit key are its permutation id s / cns_pos, the rest the
 * Philox word >> pbits, so one ranking of the S sockets ranks every permutation of the position at once. */
void orc_sample_philox_ens(const orc_params *p, int ensemble, uint64_t seed, uint64_t trial, double eps,
                           int ndoped, const int *doped, int32_t *vn_adj, uint32_t *chan_bits)
{
    const int dv = p->dv, dc = p->dc, S = p->cns_pos * dc, n = orc_n(p), L = p->L, C = p->cns_pos;
    const int D = ensemble == 0 ? L + dv - 1 : L;
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int32_t *cn_local = (int32_t *)malloc(sizeof(int32_t) * (size_t)D * S);
    keyed *ks = (keyed *)malloc(sizeof(keyed) * (size_t)S);
    int pbits = 0;
    while ((1 << pbits) < dc) pbits++;
    for (int pos = 0; pos < D; pos++) {
        for (int q = 0; q < (S + 3) / 4; q++) {
            uint32_t ctr[4] = {(uint32_t)q, (uint32_t)pos, (uint32_t)trial, (uint32_t)(trial >> 32)}, r[4];
            orc_philox4x32_10(ctr, key, r);
            for (int u = 0; u < 4 && q * 4 + u < S; u++) {
                const int sck = q * 4 + u;
                ks[sck].key = ensemble == 2 ? ((uint32_t)(sck / C) << (32 - pbits)) | (r[u] >> pbits) : r[u];
                ks[sck].sock = sck;
            }
        }
        qsort(ks, (size_t)S, sizeof(keyed), keyed_cmp);
        for (int rank = 0; rank < S; rank++)
            cn_local[(size_t)pos * S + ks[rank].sock] = ensemble == 2 ? rank - (ks[rank].sock / C) * C : rank / dc;
    }
    for (int j = 0; j < n; j++) {
        int pos = j / p->vns_pos, t = j % p->vns_pos;
        for (int i = 0; i < dv; i++) {
            if (ensemble == 2) {
                const int portion = t / C, u = t % C;
                vn_adj[(size_t)j * dv + i] = (pos + i) * C + cn_local[(size_t)pos * S + (portion * dv + i) * C + u];
            } else {
                const int cp = ensemble == 1 ? (pos + i) % L : pos + i;
                vn_adj[(size_t)j * dv + i] = cp * C + cn_local[(size_t)cp * S + dv * t + i];
            }
        }
    }
    /* erased iff r/RAND_MAX < eps, r = 31-bit draw (BPF:370,1554-1562) ⇔ r < ceil(eps*RAND_MAX) */
    double x = eps * 2147483647.0, c = (double)(uint64_t)x;
    if (c < x) c += 1.0;
    const uint32_t thresh = (uint32_t)c;
    const int nw = (n + 31) / 32;
    memset(chan_bits, 0, sizeof(uint32_t) * (size_t)nw);
    for (int q = 0; q < (n + 3) / 4; q++) {
        uint32_t ctr[4] = {(uint32_t)q, 0x80000000u, (uint32_t)trial, (uint32_t)(trial >> 32)}, r[4];
        orc_philox4x32_10(ctr, key, r);
        for (int u = 0; u < 4 && q * 4 + u < n; u++)
            if ((r[u] >> 1) < thresh) chan_bits[(q * 4 + u) >> 5] |= 1u << ((q * 4 + u) & 31);
    }
    for (int d = 0; d < ndoped; d++)
        for (int j = doped[d] * p->vns_pos; j < (doped[d] + 1) * p->vns_pos; j++)
            chan_bits[j >> 5] &= ~(1u << (j & 31));
    free(cn_local); free(ks);
}

void orc_sample_philox(const orc_params *p, uint64_t seed, uint64_t trial, double eps,
                       int ndoped, const int *doped, int32_t *vn_adj, uint32_t *chan_bits)
{
    orc_sample_philox_ens(p, 0, seed, trial, eps, ndoped, doped, vn_adj, chan_bits);
}

/* ------------------------------------------------------------------------------------------ */
/* Random-pick peeling with the degree-1 trajectory — one trial of simulate_peeling_decoder_ldpc */
/* (PD:740-785, pick_random_deg_1_cn PD:1022-1026) on the device's Philox pick stream            */
/* (csrc/peel_pick.hip, rng_mode 1), in O(steps * log ncn): the reference and the numpy model    */
/* (pd_oracle.random_pick_trial) rescan all CN degrees at every step, which takes minutes at the */
/* notebook's size (N = 10000, 290 000 steps); here the degree-1 CNs sit in a Fenwick tree and   */
/* the x-th of them in ascending order is found by descent.  tests/test_pd_oracle.py checks it   */
/* against the numpy model on the same stream.                                                   */
/* draw i of trial t = word (i & 3) of philox4x32_10(counter = (i >> 2, 0x90000000, t_lo, t_hi), */
/* key = seed); getrandbits(k) = word >> (32 - k); _randbelow(n) redraws while the value >= n.   */
/* ------------------------------------------------------------------------------------------ */
static void fen_add(int32_t *f, int size, int i, int d) { for (i++; i <= size; i += i & -i) f[i] += d; }
static int fen_select(const int32_t *f, int size, int lg, int k)      /* index of the k-th (0-based) unit */
{
    int pos = 0;
    for (int step = 1 << lg; step; step >>= 1)
        if (pos + step <= size && f[pos + step] <= k) { pos += step; k -= f[pos]; }
    return pos;
}

/* tr: int32 [n][l] global CN ids; mask[n]: 1 = erased.  r1_out: int64 [num_steps + 1].  Returns #VNs recovered by picks. */
int64_t orc_random_pick_philox(const int32_t *tr, const uint8_t *mask, int n, int l, int ncn, int total_size,
                               int num_steps, uint64_t seed, uint64_t trial, int64_t *r1_out)
{
    int32_t *deg = (int32_t *)calloc((size_t)ncn, sizeof(int32_t));
    int64_t *idsum = (int64_t *)calloc((size_t)ncn, sizeof(int64_t));
    int32_t *fen = (int32_t *)calloc((size_t)total_size + 1, sizeof(int32_t));
    int lg = 0;
    while ((2 << lg) <= total_size) lg++;
    for (int j = 0; j < n; j++)
        if (mask[j])
            for (int d = 0; d < l; d++) { deg[tr[(size_t)j * l + d]]++; idsum[tr[(size_t)j * l + d]] += j; }
    int64_t n1 = 0, picked = 0;
    for (int c = 0; c < total_size; c++)
        if (deg[c] == 1) { fen_add(fen, total_size, c, 1); n1++; }
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t draw = 0, buf[4] = {0, 0, 0, 0};
    r1_out[0] = n1;
    for (int s = 0; s < num_steps; s++) {
        if (n1 == 0) { r1_out[s + 1] = r1_out[s]; continue; }                    /* PD:765-767: nothing is drawn */
        int k = 0;
        while ((n1 >> k) != 0) k++;                                              /* n1.bit_length() */
        uint32_t x;
        do {
            if ((draw & 3u) == 0) {
                uint32_t ctr[4] = {draw >> 2, 0x90000000u, (uint32_t)trial, (uint32_t)(trial >> 32)};
                orc_philox4x32_10(ctr, key, buf);
            }
            x = buf[draw & 3u] >> (32 - k);
            draw++;
        } while ((int64_t)x >= n1);
        const int m = fen_select(fen, total_size, lg, (int)x);
        const int j = (int)idsum[m];
        picked++;
        for (int d = 0; d < l; d++) {
            const int c = tr[(size_t)j * l + d];
            idsum[c] -= j;
            deg[c]--;
            if (c < total_size) {
                if (deg[c] == 1) { fen_add(fen, total_size, c, 1); n1++; }
                else if (deg[c] == 0) { fen_add(fen, total_size, c, -1); n1--; }
            }
        }
        r1_out[s + 1] = n1;
    }
    free(deg); free(idsum); free(fen);
    return picked;
}
