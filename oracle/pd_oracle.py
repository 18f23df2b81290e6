"""TEST INFRASTRUCTURE — CPU restatement (numpy / plain Python) of the reference's Python peeling path,
simulators_sc_ldpc/peeling_decoding/{peeling_decoding.py (PD), sc_ldpc.py}.  Array formulation of what the
reference does with dicts of sets of namedtuples; each function cites the lines it follows.

Parity status: PINNED — tests/test_pd_oracle.py checks it against tests/golden/pd_*.npz, produced by importing
the real reference (oracle/make_golden_pd.py): inputs (transmissions, erasure mask), the informative entries
of simulate_sc_ldpc's 13-tuple, and the r1 / plr outputs of simulate_peeling_decoder_ldpc.

Third-party arithmetic on the path: numpy's legacy RandomState (MT19937; `permutation`, `rand`) and CPython's
`random.choice` — used through the same library calls the reference makes (numpy 2.2 / CPython 3.10 here and on
the GPU box), plus an explicit restatement of `random.choice`'s `_randbelow` rejection rule for the device twin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import random as _pyrandom

import numpy as np


# ------------------------------------------------------------------------------------------------
# sampling (sc_ldpc.py:22-56, PD:147-195)
# ------------------------------------------------------------------------------------------------
def gen_slots(rs, l, r, L, M):
    """sc_ldpc.gen_slots: D = L+l-1 draws of rs.permutation(l*M); CN of socket = position*num_cns + perm//r
    (sc_ldpc.py:22-38); VN (i,u) edge d → cn_indices[i+d][d][u] (sc_ldpc.py:37, 48-50).  int64 [L*M, l]."""
    num_cns = int(l * M / r)
    D = L + l - 1
    cn = np.stack([i * num_cns + rs.permutation(l * M).reshape(l, M) // r for i in range(D)])   # [D, l, M]
    tr = np.empty((L, M, l), dtype=np.int64)
    for d in range(l):
        tr[:, :, d] = cn[d:d + L, d, :]
    return tr.reshape(L * M, l)


def gen_erasures(rs, e, l, r, L, M, doping_points=()):
    """PD:154 (+ doping PD:166-195): mask of the VNs that become `User`s.  Hard doping (list): erased VNs whose
    chain position int(tr[0]/cns_per_pos) is doped are dropped; soft doping (dict pos→α): the first int(α·M)
    VNs of the position are forced known."""
    mask = rs.rand(L * M) <= e
    if isinstance(doping_points, dict):
        for pos, alpha in doping_points.items():
            mask[pos * M: pos * M + int(alpha * M)] = False
    elif len(doping_points):
        pos = np.arange(L * M) // M          # int(tr[0]/cns_per_pos) == VN position for this ensemble
        mask &= ~np.isin(pos, list(doping_points))
    return mask


def gen_slots_tail_biting(rs, l, r, L, M):
    """sc_ldpc.gen_slots_tail_biting (sc_ldpc.py:41-45, 59-62): only L permutations, VN (i,u) edge d →
    cn_indices[(i+d) % L][d][u] — the chain closes on itself, CN ids < L*num_cns."""
    num_cns = int(l * M / r)
    cn = np.stack([i * num_cns + rs.permutation(l * M).reshape(l, M) // r for i in range(L)])   # [L, l, M]
    tr = np.empty((L, M, l), dtype=np.int64)
    for d in range(l):
        tr[:, :, d] = cn[(np.arange(L) + d) % L, d, :]
    return tr.reshape(L * M, l)


def gen_protograph(rs, e, l, r, L, M, doping_points=()):
    """gen_users_sc_ldpc_protograph(_doping) (PD:198-241) as (transmissions, mask).  Per VN position, in this order:
    M/num_cns portions × l draws of rs.permutation(num_cns) (sc_ldpc_protograph.py:6-20: edge i of VN u of a portion
    → CN seed + i*num_cns + perm_i[u]), then rs.rand(M) <= e.  Hard doping drops the erased VNs of doped positions
    (PD:237); soft doping hits the reference's undefined name `position` (PD:228) and is not restated."""
    if isinstance(doping_points, dict):
        raise NameError("name 'position' is not defined")            # what PD:228 raises
    num_cns = int(l * M / r)
    portions = int(M / num_cns)
    tr = np.empty((L, M, l), dtype=np.int64)
    mask = np.empty((L, M), dtype=bool)
    for pos in range(L):
        rows = [np.stack([i * num_cns + rs.permutation(num_cns) for i in range(l)]).T for _ in range(portions)]
        tr[pos] = pos * num_cns + np.vstack(rows)
        mask[pos] = rs.rand(M) <= e
        if pos in doping_points:
            mask[pos] = False
    return tr.reshape(L * M, l), mask.reshape(L * M)


def gen_slots_uncoupled(rs, l, r, N):
    """ldpc.gen_slots (ldpc.py:45-84): rs.permutation(l*N).reshape(l, N) // r, redrawn until no VN meets a CN twice."""
    while True:
        tr = (rs.permutation(l * N).reshape(l, N) // r).T.astype(np.int64)
        if not any(len(np.unique(row)) != l for row in tr):
            return tr


def sample_trial(rs, e, l, r, L, M, doping_points=(), is_protograph=False, is_tail_biting=False):
    """The draws of one `generate_users()` call (PD:618-627): (transmissions, mask of the VNs that become Users)."""
    if is_protograph:
        if is_tail_biting:
            raise NotImplementedError("protograph + tail-biting: PD:204-205 reduces CN indices mod L, not mod L*num_cns")
        return gen_protograph(rs, e, l, r, L, M, doping_points)
    tr = gen_slots_tail_biting(rs, l, r, L, M) if is_tail_biting else gen_slots(rs, l, r, L, M)
    return tr, gen_erasures(rs, e, l, r, L, M, doping_points)


# ------------------------------------------------------------------------------------------------
# sweep peeling + error statistics (PD:270-313, 591-701, 1077-1095)
# ------------------------------------------------------------------------------------------------
def sic_round_literal(schedule, users, t):
    """sic_round + subtract_interference (PD:270-313) restated on plain data: schedule = {slot: set(uid)},
    users = {uid: {"transmissions": [...], "k": k, "recovered": set()}} (the reference's User tuple, PD:50-51; a user is
    recovered once it has been seen alone in k slots, PD:255-256 — k = 1 for every user of the SC-LDPC path, PD:160).
    Mutates both like the reference; returns the set of uids decoded in this round.  Small cases only: this is the literal
    form that the reference's own known-answer case (test_2_6_csa_sync, PD:1164-1175) pins; peel_closure below is the
    array form every other test uses, and tests/test_pd_oracle.py ties the two together."""
    decoded = set()
    if t not in schedule or len(schedule[t]) != 1:                        # PD:273-274
        return decoded
    rec = lambda us: {u for u in us if len(users[u]["recovered"]) >= users[u]["k"]}      # PD:251-256
    single = next(iter(schedule[t]))
    users[single]["recovered"].add(t)                                     # decode_slice, PD:259-261
    ds = rec(schedule[t])
    decoded |= ds
    while ds:                                                             # PD:281-286
        d = ds.pop()
        revealed = set()
        for slot_idx in users[d]["transmissions"]:                        # subtract_interference, PD:294-313
            if slot_idx not in schedule:
                continue
            slot = schedule[slot_idx]
            slot.remove(d)
            if len(slot) == 1 and slot_idx <= t:                          # "we can not decode transmissions in the future"
                users[next(iter(slot))]["recovered"].add(slot_idx)
                revealed |= slot
            if len(slot) == 0:
                schedule.pop(slot_idx)
        revealed = rec(revealed)
        ds |= revealed
        decoded |= revealed
    return decoded


def sweep_literal(tr, mask, total_size, sweep_start):
    """`for t in range(sweep_start, total_size): sic_round(schedule, t)` (PD:656-657) with the literal round above on the
    erased VNs of (tr, mask), k = 1: the boolean array of VNs still in the schedule — what peel_closure computes."""
    users = {int(j): {"transmissions": [int(c) for c in tr[j]], "k": 1, "recovered": set()} for j in np.flatnonzero(mask)}
    schedule = {}
    for j, u in users.items():                                            # add_to_schedule, PD:328-334
        for c in u["transmissions"]:
            schedule.setdefault(c, set()).add(j)
    for t in range(sweep_start, total_size):
        sic_round_literal(schedule, users, t)
    alive = np.zeros(len(mask), dtype=bool)
    for us in schedule.values():
        for j in us:
            alive[j] = True
    return alive


def peel_closure(tr, mask, total_size, sweep_start):
    """Residual of `for t in range(sweep_start, total_size): sic_round(schedule, t)` (PD:656-657).
    A CN fires when it is swept holding exactly one VN (PD:273-277) or when a removal leaves it with exactly one VN
    and its index <= t (PD:305-308) — i.e. any CN < total_size may fire on a TRANSITION to one VN, but a CN below
    the sweep start that holds one VN from the outset never does; CNs >= total_size never fire.  The closure does
    not depend on the order.  Returns the boolean array of VNs still in the schedule."""
    n, l = tr.shape
    ncn = int(tr.max()) + 1 if n else 0
    alive = mask.copy()
    cnt = np.zeros(ncn, dtype=np.int64)
    idsum = np.zeros(ncn, dtype=np.int64)
    for d in range(l):
        np.add.at(cnt, tr[alive, d], 1)
        np.add.at(idsum, tr[alive, d], np.flatnonzero(alive))
    stack = [c for c in range(min(total_size, ncn)) if cnt[c] == 1 and c >= sweep_start]
    while stack:
        c = stack.pop()
        if cnt[c] != 1:
            continue
        j = int(idsum[c])
        alive[j] = False
        for d in range(l):
            c2 = int(tr[j, d])
            cnt[c2] -= 1
            idsum[c2] -= j
            if cnt[c2] == 1 and c2 < total_size:
                stack.append(c2)
    return alive


def components(tr, members):
    """extract_stopping_sets (PD:1077-1095): connected components of the VNs `members` through ALL the CNs
    they share.  Returns the list of component sizes and, per member, its component id."""
    idx = np.flatnonzero(members)
    parent = {}

    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for j in idx:
        a = find(("c", int(tr[j, 0])))
        for d in range(1, tr.shape[1]):
            b = find(("c", int(tr[j, d])))
            if a != b:
                parent[b] = a
    roots = [find(("c", int(tr[j, 0]))) for j in idx]
    ids = {}
    comp = np.array([ids.setdefault(r, len(ids)) for r in roots], dtype=np.int64)
    sizes = np.bincount(comp, minlength=len(ids)) if len(idx) else np.zeros(0, dtype=np.int64)
    return sizes, comp, idx


def sc_ldpc_trial_stats(tr, mask, l, r, L, M, is_terminated, is_bounded, doping_points=()):
    """One iteration of the `for o in the_range` loop of simulate_sc_ldpc (PD:632-691) from captured inputs.
    L is the caller's L (before the function widens it, PD:604-607).  Returns a dict of the per-trial integers."""
    ignored_head = 0 if is_bounded else 20
    ignored_head_schedule = 0 if is_bounded else 10
    ignored_tail = 0 if is_terminated else 20
    Lw = L + ignored_head + ignored_tail
    cpp = int(l / r * M)
    num_positions = Lw + l - 1 if is_terminated else Lw
    total_size = cpp * num_positions
    nd = len(doping_points)
    if isinstance(doping_points, dict):
        gen = (Lw - ignored_head - ignored_tail) * M - sum(int(a * M) for a in doping_points.values())
    else:
        gen = (Lw - nd - ignored_head - ignored_tail) * M
    blocks = Lw - nd - ignored_head - ignored_tail
    alive = peel_closure(tr, mask, total_size, ignored_head_schedule * cpp)
    lo, hi = cpp * ignored_head, total_size - cpp * ignored_tail
    in_range = ((tr >= lo) & (tr < hi)).any(axis=1)                       # PD:661-664
    lost = alive & in_range & (tr < total_size).all(axis=1)               # PD:665
    sizes, comp, idx = components(tr, lost)
    big = sizes[comp] > 2 if len(idx) else np.zeros(0, dtype=bool)
    birthday_pos = (tr[idx, 0] // cpp)                                    # int(u.birthday / cns_per_pos), PD:160,687
    return dict(num_lost=int(lost.sum()), num_lost_exp=int(sizes[sizes > 2].sum()),
                frame_err=int(lost.any()), frame_err_exp=int((sizes > 2).any()),
                blocks_failed_exp=int(len(np.unique(birthday_pos[big]))), generated=gen, blocks=blocks,
                sset_sizes=np.sort(sizes))


def simulate_sc_ldpc(seed, e, l, r, L, M, is_terminated, is_bounded, num_repeats=1, max_fuckups=2000,
                     doping_points=(), is_protograph=False, is_tail_biting=False):
    """simulate_sc_ldpc (PD:591-701) after `np.random.seed(seed)`: the informative entries of its 13-tuple, in the
    order (FER, FER_exp, PLR, PLR_exp, #FER_exp, #trials, #lost_exp, #generated, #blocks_failed_exp, #blocks, BLER_exp)."""
    rs = np.random.RandomState(seed)
    Lw = L + (0 if is_bounded else 20) + (0 if is_terminated else 20)
    fu = fu_exp = failed = failed_exp = gen = blk_failed = blk = 0
    for o in range(num_repeats):
        tr, mask = sample_trial(rs, e, l, r, Lw, M, doping_points, is_protograph, is_tail_biting)
        s = sc_ldpc_trial_stats(tr, mask, l, r, L, M, is_terminated, is_bounded, doping_points)
        fu += s["frame_err"]; fu_exp += s["frame_err_exp"]; failed += s["num_lost"]; failed_exp += s["num_lost_exp"]
        gen += s["generated"]; blk_failed += s["blocks_failed_exp"]; blk += s["blocks"]
        if fu >= max_fuckups:
            break
    T = o + 1
    return (fu / T, fu_exp / T, failed / gen, failed_exp / gen, fu_exp, T, failed_exp, gen, blk_failed, blk,
            blk_failed / blk)


# ------------------------------------------------------------------------------------------------
# random-pick peeling with the degree-1 trajectory (PD:705-789, 1022-1026)
# ------------------------------------------------------------------------------------------------
def randbelow(rng, n):
    """CPython 3.10 Random._randbelow_with_getrandbits, what random.choice(seq) draws (PD:1026)."""
    k = n.bit_length()
    x = rng.getrandbits(k)
    while x >= n:
        x = rng.getrandbits(k)
    return x


def random_pick_trial(tr, mask, l, r, L, M, e, is_terminated, rng, num_doping_points=0):
    """One trial of simulate_peeling_decoder_ldpc (PD:740-785) from captured inputs; rng = the Python `random`
    stream.  Returns (r1 int64 [num_pd_steps+1], plr)."""
    cpp = int(l / r * M)
    num_positions = L + l - 1 if is_terminated else L
    total_size = cpp * num_positions
    num_pd_steps = int(M * num_positions * (e + 0.1))                    # PD:721
    n = tr.shape[0]
    ncn = int(tr.max()) + 1
    deg = np.zeros(ncn, dtype=np.int64)
    idsum = np.zeros(ncn, dtype=np.int64)
    alive = mask.copy()
    for d in range(l):
        np.add.at(deg, tr[alive, d], 1)
        np.add.at(idsum, tr[alive, d], np.flatnonzero(alive))
    total_generated = (L - num_doping_points) * M                         # PD:747
    total_recovered = total_generated - int(mask.sum())                   # PD:753 (uid of the last user + 1 = #erased)
    rdeg = deg[:total_size].copy()                                        # PD:756-757: only CNs < total_size are pickable
    r1 = np.zeros(num_pd_steps + 1, dtype=np.int64)
    r1[0] = np.count_nonzero(rdeg == 1)
    for s in range(num_pd_steps):
        ones = np.flatnonzero(rdeg == 1)                                  # ascending order (PD:1023)
        if len(ones) == 0:
            r1[s + 1] = r1[s]
            continue
        m = int(ones[randbelow(rng, len(ones))])
        j = int(idsum[m])
        total_recovered += 1
        for d in range(l):
            c = int(tr[j, d])
            idsum[c] -= j
            deg[c] -= 1
            if c < total_size:
                rdeg[c] -= 1
        r1[s + 1] = np.count_nonzero(rdeg == 1)
    return r1, (total_generated - total_recovered) / total_generated


def random_pick_trial_philox_fast(tr, mask, l, r, L, M, e, is_terminated, seed, trial, num_doping_points=0):
    """random_pick_trial(..., PhiloxPickStream(seed, trial)) in O(steps log ncn) (oracle/scldpc_oracle.c,
    orc_random_pick_philox): what a full-size check (N = 10000, 290 000 steps) needs.  Same returns."""
    import ctypes as C
    from oracle import oracle as O
    cpp = int(l / r * M)
    num_positions = L + l - 1 if is_terminated else L
    total_size = cpp * num_positions
    num_pd_steps = int(M * num_positions * (e + 0.1))                    # PD:721
    tr32 = np.ascontiguousarray(tr, dtype=np.int32)
    m8 = np.ascontiguousarray(mask, dtype=np.uint8)
    ncn = int(tr32.max()) + 1
    r1 = np.zeros(num_pd_steps + 1, dtype=np.int64)
    fn = O.lib().orc_random_pick_philox
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
    picked = fn(tr32.ctypes.data, m8.ctypes.data, tr32.shape[0], l, ncn, min(total_size, ncn), num_pd_steps,
                int(seed), int(trial), r1.ctypes.data)
    total_generated = (L - num_doping_points) * M                         # PD:747
    total_recovered = total_generated - int(m8.sum()) + int(picked)       # PD:753, 771
    return r1, (total_generated - total_recovered) / total_generated


def simulate_peeling_decoder_ldpc(seed, e, l, r, L, M, is_terminated, num_repeats=1, doping_points=(),
                                  is_protograph=False):
    """simulate_peeling_decoder_ldpc (PD:705-789) after `np.random.seed(seed); random.seed(seed)`."""
    rs = np.random.RandomState(seed)
    rng = _pyrandom.Random(seed)
    r1s, plrs = [], []
    for _ in range(num_repeats):
        tr, mask = sample_trial(rs, e, l, r, L, M, doping_points, is_protograph)
        r1, plr = random_pick_trial(tr, mask, l, r, L, M, e, is_terminated, rng, len(doping_points))
        r1s.append(r1); plrs.append(plr)
    return np.stack(r1s), np.array(plrs)


def simulate_peeling_decoder_ldpc_uncoupled(seed, e, l, r, M, num_repeats=1):
    """simulate_peeling_decoder_ldpc_uncoupled (PD:793-869) after `np.random.seed(seed); random.seed(seed)`: one CN
    position of int(l/r*M) CNs, int(M*(e+0.1)) steps, plr over M.  Returns (r1, plrs, num_vns)."""
    rs = np.random.RandomState(seed)
    rng = _pyrandom.Random(seed)
    r1s, plrs, nv = [], [], []
    for _ in range(num_repeats):
        tr = gen_slots_uncoupled(rs, l, r, M)
        mask = rs.rand(M) <= e                                            # PD:135
        # L = 1 non-terminated in random_pick_trial's terms: total_size = cpp, steps = int(M*1*(e+0.1)), generated = M
        r1, plr = random_pick_trial(tr, mask, l, r, 1, M, e, False, rng, 0)
        r1s.append(r1); plrs.append(plr); nv.append(int(mask.sum()))
    return np.stack(r1s), np.array(plrs), nv


# ------------------------------------------------------------------------------------------------
# variance reduction (fl_scaling/est_scaling_params.py:42-49, 90-94; driver PD:1264-1294)
# ------------------------------------------------------------------------------------------------
def calc_nu_chunk(r1s, r1s_theory, M):
    """find_level + calc_nu_chunk: crop to the support of the theory curve (theory > 0), d = r1/M − theory/M,
    steps with r1 == 0 masked, ssquares = nansum(d², axis 0), counts = #unmasked."""
    last = int(np.max(np.where(r1s_theory > 0))) + 1                      # est_scaling_params.py:91
    th = r1s_theory[r1s_theory > 0]                                       # :92 (a prefix for a real theory curve)
    x = (r1s[:, :last][:, :r1s_theory.shape[0]] / M)                      # :93, :132-133
    d = x - th / M
    d[x == 0] = np.nan                                                    # :135
    return np.nansum(d ** 2, axis=0), np.sum(~np.isnan(d), axis=0)        # :136-137


# ------------------------------------------------------------------------------------------------
# CPU twin of the device's Philox pick stream (csrc/peel_pick.hip, rng_mode 1): draw i of trial t is word (i & 3)
# of philox4x32_10(counter = (i >> 2, 0x90000000, t_lo, t_hi), key = seed); getrandbits(k) = word >> (32 - k).
# ------------------------------------------------------------------------------------------------
class PhiloxPickStream:
    def __init__(self, seed, trial):
        from oracle import oracle as O
        self._f, self.seed, self.trial, self.i = O.philox4x32_10, seed, trial, 0

    def getrandbits(self, k):
        w = self._f([self.i >> 2, 0x90000000, self.trial & 0xFFFFFFFF, self.trial >> 32],
                    [self.seed & 0xFFFFFFFF, self.seed >> 32])[self.i & 3]
        self.i += 1
        return w >> (32 - k)
