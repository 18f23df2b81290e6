"""The CPU oracle (oracle/scldpc_oracle.c) against the golden vectors produced by the REAL reference
(oracle/make_golden.py → tests/golden/*.npz): sampled graph / channel / residual-pattern digests,
the four counters of decodeBP / decodeBP_SW, and the per-iteration trajectory rows of bp_traj.
Both restatements are checked: the literal per-edge flooding decoder and the node-level peeling model."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

DEC = {"bpf": (0, 1), "bpt": (0, 1), "bpw": (2, 3), "bpfsw": (4,)}


def _check(O, g, decoders, limit=None):
    m = g.meta
    p = O.Params(m["dv"], m["dc"], m["L"], m["CNsPos"], m["VNsPos"])
    T = g.T if limit is None else min(g.T, limit)
    for t in range(T):
        for dec in decoders:
            o = O.trial(p, int(g["seed"][t]), m["eps"], decoder=dec, W=m["W"], max_it=g.max_it,
                        init_it=m["init_it"], is_term=m["is_term"], rows_cap=8192, want_arrays=g.has("vn_adj"))
            tag = f"{g.name} trial {t} decoder {dec}"
            assert o["status"] == 0, tag
            assert (o["nch"], o["hg"], o["hc"]) == (int(g["nch"][t]), int(g["hg"][t]), int(g["hc"][t])), tag
            assert (o["num_erasures"], o["num_blocks_err"], o["num_erasures_exp"], o["num_blocks_err_exp"],
                    o["num_erasures_p1"]) == tuple(int(g[k][t]) for k in ("ne", "be", "ee", "bee", "p1")), tag
            assert o["he"] == int(g["he"][t]), tag
            if g.has("rows"):
                ref = g.rows_of(t)
                got = np.stack([o["rows"]["deg1"], o["rows"]["recovered"], o["rows"]["first_pos"]], axis=1)
                assert o["iterations"] == len(ref) and (got == ref).all(), tag
            if g.has("vn_adj"):
                assert (o["vn_adj"] == g["vn_adj"][t]).all() and (o["chan"] == g["chan"][t]).all(), tag
                assert (o["erased"] == g["erased"][t]).all(), tag


@pytest.mark.parametrize("name", golden_names(prefixes=("tiny_", "ss2_", "mid_")))
def test_oracle_small(oracle, name):
    g = load_golden(name)
    _check(oracle, g, DEC[g.variant])


@pytest.mark.parametrize("name", golden_names(prefixes=("c2_", "c3_", "c4_")))
def test_oracle_config_size(oracle, name):
    """(4,8), L=50, N=1000 — the BASELINE.json ensemble.  The literal decoder on a third of the trials
    (≈0.25 s each), the peeling model on all of them."""
    g = load_golden(name)
    lit, peel = DEC[g.variant][0], DEC[g.variant][-1]
    _check(oracle, g, (peel,))
    _check(oracle, g, (lit,), limit=max(2, g.T // 3))


def test_size2_stopping_sets_are_exercised():
    """The ss2_* fixtures must actually contain expurgation events (ee < ne with bee counted)."""
    events = 0
    for name in golden_names(prefixes=("ss2_",)):
        g = load_golden(name)
        if g.variant == "bpw":          # every position contributes ⇒ ee == ne unless pairs were removed
            events += int((g["ee"] < g["ne"]).sum())
    assert events >= 20


@pytest.mark.parametrize("name", golden_names(whole_run=True))
def test_oracle_whole_run(oracle, name):
    """No re-seeding between frames: perm_code and the random() stream carry over (BPF:2117-2144)."""
    import ctypes as C
    O = oracle
    g = load_golden(name)
    m = g.meta
    p = O.Params(m["dv"], m["dc"], m["L"], m["CNsPos"], m["VNsPos"])
    n, nk, dv = p.n, p.nk, p.dv
    perm = np.empty(p.cns_pos * p.dc, dtype=np.int32)
    vn_adj = np.empty((n, dv), dtype=np.int32)
    cn_ptr = np.empty(nk + 1, dtype=np.int32)
    cn_adj = np.empty(n * dv, dtype=np.int32)
    chan = np.empty(n, dtype=np.uint8)
    rng = O.Rng()
    L = O.lib()
    L.orc_perm_identity(C.byref(p), O._p(perm, C.c_int32))
    L.orc_srandom(C.byref(rng), m["seed0"])
    run = dict(users_err=0, frame_err=0, block_err=0, users_err_exp=0, frame_err_exp=0, block_err_exp=0)
    for t in range(g.T):
        L.orc_generate_code(C.byref(p), C.byref(rng), O._p(perm, C.c_int32), O._p(vn_adj, C.c_int32),
                            O._p(cn_ptr, C.c_int32), O._p(cn_adj, C.c_int32))
        L.orc_channel(C.byref(p), C.byref(rng), m["eps"], 0, None, O._p(chan, C.c_uint8))
        res, erased, rows = O.decode_bp(O.Graph(p, vn_adj, cn_ptr, cn_adj), chan, max_it=g.max_it, literal=False,
                                        is_term=m.get("is_term", 1), rows_cap=512 if g.has("rows") else 0)
        if g.has("rows"):               # bp_traj runs: the rows the reference wrote, frame after frame
            gr = g.rows_of(t)
            assert len(rows) == len(gr) and (rows["deg1"] == gr[:, 0]).all() and (rows["recovered"] == gr[:, 1]).all() \
                and (rows["first_pos"] == gr[:, 2]).all(), (name, t)
        assert int(chan.sum()) == int(g["nch"][t]), (name, t)
        assert (res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"], res["num_blocks_err_exp"]) == \
            tuple(int(g[k][t]) for k in ("ne", "be", "ee", "bee")), (name, t)
        if res["num_erasures"] > 0:
            run["users_err"] += res["num_erasures"]; run["frame_err"] += 1; run["block_err"] += res["num_blocks_err"]
        if res["num_erasures_exp"] > 0:
            run["users_err_exp"] += res["num_erasures_exp"]; run["frame_err_exp"] += 1
            run["block_err_exp"] += res["num_blocks_err_exp"]
    for k, v in run.items():
        assert v == m["run_counters"][k], k


def test_glibc_random_against_libc(oracle):
    """The TYPE_3 restatement against the libc of this image (glibc; both here and on the GPU box)."""
    import ctypes
    import ctypes.util
    libc = ctypes.CDLL(ctypes.util.find_library("c"))
    libc.random.restype = ctypes.c_long
    for seed in (1, 2, 12345, 0, 4294967295):
        libc.srandom(seed)
        ref = np.array([libc.random() for _ in range(1500)])
        assert (ref == oracle.glibc_random_stream(seed, 1500)).all(), seed
    assert oracle.glibc_random_stream(1, 3).tolist() == [1804289383, 846930886, 1681692777]


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10 (Salmon et al., SC'11)."""
    f = oracle.philox4x32_10
    assert f([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert f([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert f([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_literal_and_peel_models_agree_on_random_small_graphs(oracle):
    """Literal flooding ≡ node-level peeling far beyond the fixture seeds."""
    O = oracle
    rng = np.random.RandomState(5)
    for (M, L) in ((3, 6), (5, 10), (8, 7), (25, 12)):
        p = O.Params(4, 8, L, M, 2 * M)
        for _ in range(120):
            seed = int(rng.randint(1, 2**31 - 1))
            eps = float(rng.choice([0.2, 0.35, 0.43, 0.47, 0.5, 0.6, 0.8]))
            is_term = int(rng.randint(0, 2))
            mi = int(rng.choice([0, 0, 1, 2, 5]))
            a = O.trial(p, seed, eps, decoder=0, max_it=mi, is_term=is_term, rows_cap=512)
            b = O.trial(p, seed, eps, decoder=1, max_it=mi, is_term=is_term, rows_cap=512)
            for k in ("num_erasures", "num_blocks_err", "num_erasures_exp", "num_blocks_err_exp", "iterations", "he"):
                assert a[k] == b[k], (M, L, seed, eps, is_term, mi, k)
            assert (a["rows"] == b["rows"]).all()
            W, cap, init = int(rng.randint(1, L + 2)), int(rng.randint(1, 8)), int(rng.randint(0, 12))
            a = O.trial(p, seed, eps, decoder=2, W=W, max_it=cap, init_it=init)
            b = O.trial(p, seed, eps, decoder=3, W=W, max_it=cap, init_it=init)
            for k in ("num_erasures", "num_blocks_err", "num_erasures_exp", "num_blocks_err_exp",
                      "num_erasures_p1", "iterations", "he"):
                assert a[k] == b[k], (M, L, seed, eps, W, cap, init, k)


@pytest.mark.parametrize("ens", ["olmos", "tail_biting", "protograph"])
def test_philox_ensemble_twins_have_the_reference_samplers_structure(oracle, ens):
    """orc_sample_philox_ens (the CPU twin of the device samplers): socket counts, position structure and — for the
    protograph chain — one permutation of the position's CNs per (portion, edge), as sc_ldpc.py / sc_ldpc_protograph.py
    construct them; plus a first-moment check of the permutation law (uniform position of a fixed socket)."""
    L, M, C = 9, 24, 12
    po = oracle.Params(4, 8, L, C, M)
    seen = np.zeros(C, dtype=np.int64)
    for t in range(600):
        adj, ch = oracle.sample_philox(po, 3, t, 0.4, ensemble=ens)
        a = adj.reshape(L, M, 4)
        for i in range(4):
            want = (np.arange(L) + i) % L if ens == "tail_biting" else np.arange(L) + i
            assert ((a[:, :, i] // C) == want[:, None]).all()
        if ens == "tail_biting":
            assert (np.bincount(adj.ravel(), minlength=L * C) == 8).all()
        elif ens == "protograph":
            loc = (a % C).reshape(L, 2, C, 4)
            assert (np.sort(loc, axis=2) == np.arange(C)[None, None, :, None]).all()
        else:
            cnt = np.bincount(adj.ravel(), minlength=(L + 3) * C).reshape(L + 3, C)
            assert (cnt[3:L] == 8).all() and cnt[0].sum() == M and cnt[L + 2].sum() == M
        seen[a[4, 5, 2] % C] += 1
    # uniform over the C CNs of the position: chi-square with C-1 = 11 dof, 99.9 % quantile 31.3
    chi2 = ((seen - 600 / C) ** 2 / (600 / C)).sum()
    assert chi2 < 31.3, (ens, seen)
