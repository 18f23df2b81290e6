"""A CPU model of the barrier-free peeling protocol of fl_scaling_sc_ldpc_amd/csrc/peel_fixpoint.h, run under random
interleavings of its atomic micro-steps: whatever the schedule, the VNs left erased must be exactly the peeling closure
(= what unlimited flooding BP converges to), no VN may ever be released twice, and the packed CN words must end up
consistent.  The GPU tests cannot choose interleavings; this test can.

Model (one "worker" = one wave with its private queue; every numbered line is one atomic step):
    entry (c, h) popped from the worker's queue            h = the half-word [cnt:4 | fold:12] the entry carries
    1. decode fold → (edge i1, t) → VN j; reject impossible decodes; row = adjacency[j]; reject if row[i1] != c
    2. claim: test-and-clear U[j]; give up if it was clear
    3. for each edge i: fold[row[i]] ^= id(i, t)                      (4 separate atomic XORs)
    4. for each edge i: old = word[row[i]]; cnt[row[i]] -= 1          (4 separate atomic decrements, each returns old)
       if old.cnt == 2: push (row[i], [1 | old.fold]) to the own queue
"""
import numpy as np
import pytest


def make_graph(rs, L, V, dv=4, dc=8):
    """Olmos chain like sc_ldpc.gen_slots: per CN position a permutation of the dv*V sockets, CN = rank // dc."""
    C = dv * V // dc
    D = L + dv - 1
    cn = np.stack([p * C + rs.permutation(dv * V).reshape(dv, V) // dc for p in range(D)])     # [D, dv, V]
    adj = np.empty((L * V, dv), dtype=np.int64)
    for q in range(L):
        for i in range(dv):
            adj[q * V:(q + 1) * V, i] = cn[q + i, i, :]
    return adj, C, D * C


def closure(adj, erased, ncn):
    erased = erased.copy()
    cnt = np.zeros(ncn, dtype=np.int64)
    for j in np.flatnonzero(erased):
        cnt[adj[j]] += 1
    changed = True
    while changed:
        changed = False
        for j in np.flatnonzero(erased):
            if (cnt[adj[j]] == 1).any():
                erased[j] = False
                cnt[adj[j]] -= 1
                changed = True
    return erased


def worker(wid, queue, st, count_first=False, validate=True):
    """Generator: yields after every atomic step."""
    adj, V, C, L, dv = st["adj"], st["V"], st["C"], st["L"], st["dv"]
    while queue:
        c, h = queue.pop(0)
        if (h >> 12) != 1:
            continue
        lid = h & 0xFFF
        i1, t = lid // V, lid % V
        pos = c // C - i1
        if i1 >= dv or pos < 0 or pos >= L:
            continue
        j = pos * V + t
        row = adj[j]                                   # immutable table: no step
        if validate and row[i1] != c:
            continue
        yield                                          # --- claim (atomic test-and-clear)
        if not st["U"][j]:
            continue
        st["U"][j] = False
        st["released"].append(j)
        if not count_first:
            for i in range(dv):
                yield                                  # --- one atomic XOR
                st["fold"][row[i]] ^= i * V + t
        for i in range(dv):
            yield                                      # --- one atomic decrement, returns the old word
            old_cnt, old_fold = st["cnt"][row[i]], st["fold"][row[i]]
            st["cnt"][row[i]] -= 1
            if old_cnt == 2:
                queue.append((int(row[i]), (1 << 12) | int(old_fold ^ ((i * V + t) if count_first else 0))))
        if count_first:
            for i in range(dv):
                yield
                st["fold"][row[i]] ^= i * V + t


@pytest.mark.parametrize("seed", range(40))
def test_any_interleaving_reaches_the_peeling_closure(seed):
    rs = np.random.RandomState(seed)
    L, V = int(rs.randint(4, 9)), int(rs.choice([8, 12, 16]))
    adj, C, ncn = make_graph(rs, L, V)
    eps = float(rs.choice([0.3, 0.42, 0.47, 0.5, 0.6]))
    erased = rs.rand(L * V) <= eps
    want = closure(adj, erased, ncn)
    for trial in range(6):
        st = dict(adj=adj, V=V, C=C, L=L, dv=4, U=erased.copy(), released=[],
                  cnt=np.zeros(ncn, dtype=np.int64), fold=np.zeros(ncn, dtype=np.int64))
        for j in np.flatnonzero(erased):
            pos, t = divmod(int(j), V)
            for i in range(4):
                st["cnt"][adj[j, i]] += 1
                st["fold"][adj[j, i]] ^= i * V + t
        # the opening scan: every CN showing one erased neighbour, dealt out round-robin to the workers
        nworkers = int(rs.randint(1, 7))
        queues = [[] for _ in range(nworkers)]
        for k, c in enumerate(np.flatnonzero(st["cnt"] == 1)):
            queues[k % nworkers].append((int(c), (1 << 12) | int(st["fold"][c])))
        live = [worker(w, queues[w], st) for w in range(nworkers)]
        # adversarial-ish scheduler: random worker, random burst length
        while live:
            g = live[rs.randint(len(live))]
            try:
                for _ in range(int(rs.randint(1, 6))):
                    next(g)
            except StopIteration:
                live.remove(g)
        assert len(st["released"]) == len(set(st["released"])), "a VN was released twice"
        assert (st["U"] == want).all(), (seed, trial, int(st["U"].sum()), int(want.sum()))
        # words consistent with the residual
        cnt = np.zeros(ncn, dtype=np.int64); fold = np.zeros(ncn, dtype=np.int64)
        for j in np.flatnonzero(st["U"]):
            pos, t = divmod(int(j), V)
            for i in range(4):
                cnt[adj[j, i]] += 1; fold[adj[j, i]] ^= i * V + t
        assert (cnt == st["cnt"]).all() and (fold == st["fold"]).all()


def test_the_model_has_teeth():
    """Negative control: decrement first / XOR second (the order the barrier-synchronous kernels use, safe THERE) and no
    adjacency check must go wrong under some interleaving — a VN released that the closure keeps erased, or one it
    releases left behind."""
    bad = 0
    for seed in range(60):
        rs = np.random.RandomState(1000 + seed)
        L, V = 6, 12
        adj, C, ncn = make_graph(rs, L, V)
        erased = rs.rand(L * V) <= 0.47
        want = closure(adj, erased, ncn)
        st = dict(adj=adj, V=V, C=C, L=L, dv=4, U=erased.copy(), released=[],
                  cnt=np.zeros(ncn, dtype=np.int64), fold=np.zeros(ncn, dtype=np.int64))
        for j in np.flatnonzero(erased):
            pos, t = divmod(int(j), V)
            for i in range(4):
                st["cnt"][adj[j, i]] += 1
                st["fold"][adj[j, i]] ^= i * V + t
        queues = [[] for _ in range(5)]
        for k, c in enumerate(np.flatnonzero(st["cnt"] == 1)):
            queues[k % 5].append((int(c), (1 << 12) | int(st["fold"][c])))
        live = [worker(w, queues[w], st, count_first=True, validate=False) for w in range(5)]
        try:
            while live:
                g = live[rs.randint(len(live))]
                try:
                    next(g)
                except StopIteration:
                    live.remove(g)
            bad += not (st["U"] == want).all()
        except IndexError:
            bad += 1                                    # a garbage fold decoded to a VN outside the graph
    assert bad > 0


# ------------------------------------------------------------------------------------------------------------------------
# The count-only protocol of fl_scaling_sc_ldpc_amd/csrc/full_bp_small.hip (4 bits per CN, no fold): a CN whose count is one
# reads its neighbour list and takes the neighbour whose U bit is still set.
#     entry c popped from the worker's queue
#     1. for each of c's neighbours (one LDS read each, at different times): remember the last one whose U bit is set
#     2. claim: test-and-clear U[j]; give up if it was clear (or if no neighbour had its bit set)
#     3. for each edge i: old = cnt[row[i]]; cnt[row[i]] -= 1    (4 separate atomic decrements, each returns old)
#        if old == 2: push row[i] to the own queue
# Safe because the claim comes BEFORE the decrements: a neighbour that is counted but no longer erased has its bit clear.
# ------------------------------------------------------------------------------------------------------------------------
def count_worker(queue, st, claim_first=True):
    adj, nbrs = st["adj"], st["nbrs"]
    while queue:
        c = queue.pop(0)
        j = -1
        for v in nbrs[c]:
            yield                                      # --- one read of the bitmap
            if st["U"][v]:
                j = v
        if j < 0:
            continue
        row = adj[j]
        if claim_first:
            yield                                      # --- claim (atomic test-and-clear)
            if not st["U"][j]:
                continue
            st["U"][j] = False
            st["released"].append(j)
        for i in range(4):
            yield                                      # --- one atomic decrement, returns the old count
            old = st["cnt"][row[i]]
            st["cnt"][row[i]] -= 1
            if old == 2:
                queue.append(int(row[i]))
        if not claim_first:                            # (negative control: the bit is cleared only after the decrements)
            yield
            if st["U"][j]:
                st["U"][j] = False
            st["released"].append(j)


def _count_state(adj, erased, ncn):
    cnt = np.zeros(ncn, dtype=np.int64)
    nbrs = [[] for _ in range(ncn)]
    for j in range(adj.shape[0]):
        for i in range(4):
            nbrs[adj[j, i]].append(j)
    for j in np.flatnonzero(erased):
        cnt[adj[j]] += 1
    return dict(adj=adj, nbrs=nbrs, U=erased.copy(), released=[], cnt=cnt)


@pytest.mark.parametrize("seed", range(40))
def test_count_only_protocol_reaches_the_closure_under_any_interleaving(seed):
    rs = np.random.RandomState(500 + seed)
    L, V = int(rs.randint(4, 9)), int(rs.choice([8, 12, 16]))
    adj, C, ncn = make_graph(rs, L, V)
    eps = float(rs.choice([0.3, 0.42, 0.47, 0.5, 0.6]))
    erased = rs.rand(L * V) <= eps
    want = closure(adj, erased, ncn)
    for trial in range(6):
        st = _count_state(adj, erased, ncn)
        nworkers = int(rs.randint(1, 7))
        queues = [[] for _ in range(nworkers)]
        for k, c in enumerate(np.flatnonzero(st["cnt"] == 1)):
            queues[k % nworkers].append(int(c))
        live = [count_worker(queues[w], st) for w in range(nworkers)]
        while live:
            g = live[rs.randint(len(live))]
            try:
                for _ in range(int(rs.randint(1, 6))):
                    next(g)
            except StopIteration:
                live.remove(g)
        assert len(st["released"]) == len(set(st["released"])), "a VN was released twice"
        assert (st["U"] == want).all(), (seed, trial, int(st["U"].sum()), int(want.sum()))
        cnt = np.zeros(ncn, dtype=np.int64)
        for j in np.flatnonzero(st["U"]):
            cnt[adj[j]] += 1
        assert (cnt == st["cnt"]).all() and (st["cnt"] >= 0).all()


def test_the_count_only_model_has_teeth():
    """Negative control: clearing the U bit only AFTER the decrements lets two CNs release the same VN (its counts go
    negative / it is released twice) under some interleaving."""
    bad = 0
    for seed in range(60):
        rs = np.random.RandomState(2000 + seed)
        adj, C, ncn = make_graph(rs, 6, 12)
        erased = rs.rand(6 * 12) <= 0.47
        want = closure(adj, erased, ncn)
        st = _count_state(adj, erased, ncn)
        queues = [[] for _ in range(5)]
        for k, c in enumerate(np.flatnonzero(st["cnt"] == 1)):
            queues[k % 5].append(int(c))
        live = [count_worker(queues[w], st, claim_first=False) for w in range(5)]
        while live:
            g = live[rs.randint(len(live))]
            try:
                next(g)
            except StopIteration:
                live.remove(g)
        bad += (len(st["released"]) != len(set(st["released"]))) or (st["cnt"] < 0).any() or not (st["U"] == want).all()
    assert bad > 0
