// Host-side exact replay of the reference's ensemble + channel sampling on identical seeds
// (generate_code BPF:1656-1761, channel_doped BPF:1547-1574, inizio_sim's perm_code reset
// BPF:308-311, srandom BPF:2062).  glibc's random() (TYPE_3 additive feedback, stdlib/random_r.c)
// is restated here so the stream does not depend on the host's libc.
#include "common.h"
#include <cstring>
#include <vector>

namespace {

struct GlibcRandom {
    int32_t r[31];
    int f, b;

    void seed(uint32_t s)
    {
        int32_t word = (int32_t)(s ? s : 1u);
        r[0] = word;
        for (int i = 1; i < 31; i++) {          // Park–Miller via Schrage
            const long hi = word / 127773, lo = word % 127773;
            word = (int32_t)(16807 * lo - 2836 * hi);
            if (word < 0) word += 2147483647;
            r[i] = word;
        }
        f = 3; b = 0;
        for (int k = 0; k < 310; k++) next();
    }
    inline int32_t next()
    {
        const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)v;
        if (++f == 31) f = 0;
        if (++b == 31) b = 0;
        return (int32_t)(v >> 1);
    }
};

// Opaque state blob: the generator followed by perm_code[cns_pos*dc].
struct StateHeader {
    GlibcRandom rng;
    int32_t nsock;
    int32_t pad;
};

inline int32_t *perm_of(void *state) { return reinterpret_cast<int32_t *>(static_cast<char *>(state) + sizeof(StateHeader)); }

void draw_frame(const scldpc_code_params *p, StateHeader *st, int32_t *perm, double eps,
                int ndoped, const int32_t *doped, int32_t *vn_adj, uint32_t *chan_bits,
                std::vector<int32_t> &sock_cn)
{
    const int dv = p->dv, dc = p->dc, S = p->cns_pos * dc, D = p->L + dv - 1;
    const int n = scldpc::n_of(p), nw = scldpc::nw_of(p);
    GlibcRandom &g = st->rng;
    sock_cn.resize((size_t)D * S);
    for (int pos = 0; pos < D; pos++) {
        for (int i = 0; i < S; i++) {                              // BPF:1682-1688
            const int pick = i + g.next() % (S - i);
            const int32_t t = perm[i]; perm[i] = perm[pick]; perm[pick] = t;
        }
        int32_t *row = &sock_cn[(size_t)pos * S];
        for (int i = 0; i < S; i++) row[i] = pos * p->cns_pos + perm[i] / dc;   // BPF:1693
    }
    for (int pos = 0; pos < p->L; pos++)                            // BPF:1703-1716
        for (int t = 0; t < p->vns_pos; t++) {
            int32_t *out = vn_adj + ((size_t)pos * p->vns_pos + t) * dv;
            for (int i = 0; i < dv; i++) out[i] = sock_cn[(size_t)(pos + i) * S + dv * t + i];
        }
    memset(chan_bits, 0, sizeof(uint32_t) * (size_t)nw);
    for (int j = 0; j < n; j++) {                                   // BPF:1552-1563, unif_ch BPF:370
        const double u = (double)g.next() / 2147483647.0;
        if (!(u >= eps)) chan_bits[j >> 5] |= 1u << (j & 31);
    }
    for (int d = 0; d < ndoped; d++)                                // BPF:1566-1573
        for (int j = doped[d] * p->vns_pos; j < (doped[d] + 1) * p->vns_pos; j++)
            chan_bits[j >> 5] &= ~(1u << (j & 31));
}

int check_doped(const scldpc_code_params *p, int ndoped, const int32_t *doped)
{
    if (ndoped < 0 || (ndoped > 0 && !doped))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "bad doped-position list");
    for (int d = 0; d < ndoped; d++)
        if (doped[d] < 0 || doped[d] >= p->L)
            return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "doped position %d outside [0,%d)", doped[d], p->L);
    return SCLDPC_OK;
}

}  // namespace

extern "C" int64_t scldpc_glibc_state_bytes(const scldpc_code_params *p)
{
    if (int rc = scldpc::check_params(p)) return rc;
    return (int64_t)sizeof(StateHeader) + 4ll * p->cns_pos * p->dc;
}

extern "C" int scldpc_glibc_state_reset_perm(const scldpc_code_params *p, void *state)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (!state) return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "null state");
    int32_t *perm = perm_of(state);
    for (int i = 0; i < p->cns_pos * p->dc; i++) perm[i] = i;       // BPF:308-311
    return SCLDPC_OK;
}

extern "C" int scldpc_glibc_state_init(const scldpc_code_params *p, uint32_t seed, void *state)
{
    if (int rc = scldpc_glibc_state_reset_perm(p, state)) return rc;
    StateHeader *st = static_cast<StateHeader *>(state);
    st->rng.seed(seed);
    st->nsock = p->cns_pos * p->dc;
    st->pad = 0;
    return SCLDPC_OK;
}

extern "C" int scldpc_sample_glibc_next_host(const scldpc_code_params *p, void *state, double eps,
                                             int32_t ndoped, const int32_t *doped_positions, int32_t nframes,
                                             int32_t *vn_adj, uint32_t *chan_bits)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (int rc = check_doped(p, ndoped, doped_positions)) return rc;
    if (!state || nframes < 0 || (nframes > 0 && (!vn_adj || !chan_bits)))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_sample_glibc_next_host: null buffer or negative nframes");
    StateHeader *st = static_cast<StateHeader *>(state);
    if (st->nsock != p->cns_pos * p->dc)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "state blob was initialised for a different ensemble");
    const size_t adj_stride = (size_t)scldpc::n_of(p) * p->dv, ch_stride = (size_t)scldpc::nw_of(p);
    std::vector<int32_t> sock_cn;
    for (int f = 0; f < nframes; f++)
        draw_frame(p, st, perm_of(state), eps, ndoped, doped_positions,
                   vn_adj + f * adj_stride, chan_bits + f * ch_stride, sock_cn);
    return SCLDPC_OK;
}

extern "C" int scldpc_sample_glibc_host(const scldpc_code_params *p, uint32_t seed, double eps,
                                        int32_t ndoped, const int32_t *doped_positions,
                                        int32_t *vn_adj, uint32_t *chan_bits)
{
    const int64_t bytes = scldpc_glibc_state_bytes(p);
    if (bytes < 0) return (int)bytes;
    std::vector<char> state((size_t)bytes);
    if (int rc = scldpc_glibc_state_init(p, seed, state.data())) return rc;
    return scldpc_sample_glibc_next_host(p, state.data(), eps, ndoped, doped_positions, 1, vn_adj, chan_bits);
}

// main_streaming's draws (BPF:1934-2054, the CIRCULAR build) for the first npos_gen generated positions of ONE stream
// after `srandom(seed)` and inizio_sim's perm_code reset: initialize_arrays_circular shuffles CN positions 0 .. dv-2
// (BPF:1808-1813), then generate_stream_pos(g) shuffles CN position g + dv - 1 (fill_interleaver_pos, BPF:1763-1787:
// Fisher-Yates on perm_code, whose state carries over) and draws the channel of VN position g unless it is doped
// (generate_channel_doped_circular, BPF:1621-1654; periodic doping, is_position_doped_streaming BPF:1589-1612).
// Decoding draws nothing, so the whole input stream is a function of (seed, eps, doping) alone.
//   inter_out uint16 [npos_gen + dv - 1][S]: CN-local id perm_code[i] / dc of socket i of CN position c
//   chan_out  uint32 [npos_gen][ceil(vns_pos/32)]: bit t of position g = 1 iff VN (g, t) is erased
// The same draws from a CARRIED state (scldpc_glibc_state_init / _reset_perm: random() and perm_code), so that a run of any
// length — and a run over several ε points, which main_streaming draws from ONE srandom (BPF:1942-1945) with inizio_sim's
// perm_code reset and initialize_arrays_circular's dv-1 shuffles at the start of every point (BPF:1994-1998) — is replayed
// in pieces: first `ninit` CN positions are shuffled (dv-1 at the start of a point, 0 afterwards), then
// generate_stream_pos(gpos0 + k), k < npos: one more CN position shuffled, the channel of VN position gpos0 + k drawn.
extern "C" int scldpc_stream_glibc_next_host(const scldpc_code_params *p, void *state, double eps, int32_t ndoped,
                                             const int32_t *doped_positions, int32_t ninit, int64_t gpos0, int32_t npos,
                                             uint16_t *inter_out, uint32_t *chan_out)
{
    if (int rc = scldpc::check_params(p)) return rc;
    if (!state || npos < 0 || ninit < 0 || gpos0 < 0 || ((npos > 0 || ninit > 0) && !inter_out) || (npos > 0 && !chan_out) ||
        ndoped < 0 || (ndoped > 0 && !doped_positions))
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_stream_glibc_next_host: null buffer or negative count");
    const int dc = p->dc, S = p->cns_pos * dc, V = p->vns_pos, wpp = (V + 31) / 32;
    if (p->cns_pos > 65536)
        return scldpc::set_error(SCLDPC_ERR_TOO_LARGE, "scldpc_stream_glibc_next_host: cns_pos > 65536");
    StateHeader *st = static_cast<StateHeader *>(state);
    if (st->nsock != S)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "state blob was initialised for a different ensemble");
    GlibcRandom &g = st->rng;
    int32_t *perm = perm_of(state);
    auto shuffle_into = [&](uint16_t *row) {
        for (int i = 0; i < S; i++) {                                          // BPF:1770-1776
            const int pick = i + g.next() % (S - i);
            const int32_t t = perm[i]; perm[i] = perm[pick]; perm[pick] = t;
        }
        for (int i = 0; i < S; i++) row[i] = (uint16_t)(perm[i] / dc);         // BPF:1782
    };
    auto doped = [&](long long pos) {                                          // BPF:1589-1612
        if (ndoped == 0) return false;
        const int period = doped_positions[ndoped - 1] + 1, m = (int)(pos % period);
        if (m < doped_positions[0]) return false;
        for (int d = 0; d < ndoped; d++) if (m == doped_positions[d]) return true;
        return false;
    };
    for (int c = 0; c < ninit; c++) shuffle_into(inter_out + (size_t)c * S);
    if (npos > 0) memset(chan_out, 0, sizeof(uint32_t) * (size_t)npos * wpp);
    for (int k = 0; k < npos; k++) {
        shuffle_into(inter_out + (size_t)(ninit + k) * S);
        if (doped(gpos0 + k)) continue;                                        // no draws for a doped position
        uint32_t *row = chan_out + (size_t)k * wpp;
        for (int t = 0; t < V; t++) {
            const double u = (double)g.next() / 2147483647.0;                 // unif_ch, BPF:360-371
            if (!(u >= eps)) row[t >> 5] |= 1u << (t & 31);
        }
    }
    return SCLDPC_OK;
}

extern "C" int scldpc_stream_glibc_inputs_host(const scldpc_code_params *p, uint32_t seed, double eps, int32_t ndoped,
                                               const int32_t *doped_positions, int32_t npos_gen, uint16_t *inter_out,
                                               uint32_t *chan_out)
{
    const int64_t bytes = scldpc_glibc_state_bytes(p);
    if (bytes < 0) return (int)bytes;
    if (npos_gen < 0 || !inter_out || !chan_out)
        return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "scldpc_stream_glibc_inputs_host: null buffer or negative count");
    std::vector<char> state((size_t)bytes);
    if (int rc = scldpc_glibc_state_init(p, seed, state.data())) return rc;   // srandom(seed) + inizio_sim's perm_code
    return scldpc_stream_glibc_next_host(p, state.data(), eps, ndoped, doped_positions, p->dv - 1, 0, npos_gen, inter_out,
                                         chan_out);
}
