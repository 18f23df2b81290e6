"""oracle/pd_oracle.py (numpy restatement of the reference's Python peeling path) against the golden vectors
produced by importing the REAL reference (oracle/make_golden_pd.py → tests/golden/pd_*.npz)."""
import glob
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN_DIR

ER = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_er_*.npz")))
TR = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_tr_*.npz")))
UNC = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_unc_*.npz")))


def _load(path):
    z = np.load(path)
    m = json.loads(str(z["meta"]))
    doping = m["doping"]
    if m.get("doping_soft"):
        doping = {int(k): v for k, v in doping.items()}
    return z, m, doping


@pytest.mark.parametrize("path", ER, ids=[os.path.basename(p)[:-4] for p in ER])
def test_simulate_sc_ldpc_tuple(path):
    from oracle import pd_oracle as P
    z, m, doping = _load(path)
    seeds = z["seed"] if m["M"] <= 200 else z["seed"][:3]
    for k, s in enumerate(seeds):
        got = P.simulate_sc_ldpc(int(s), m["e"], m["l"], m["r"], m["L"], m["M"], m["is_terminated"], m["is_bounded"],
                                 1, 2000, doping, m.get("is_protograph", False), m.get("is_tail_biting", False))
        assert np.allclose(got, z["tuple11"][k], rtol=0, atol=1e-15), (path, s, got, z["tuple11"][k])
    if m["M"] <= 200:
        got = P.simulate_sc_ldpc(int(z["seed"][0]), m["e"], m["l"], m["r"], m["L"], m["M"], m["is_terminated"],
                                 m["is_bounded"], 3, 2000, doping, m.get("is_protograph", False),
                                 m.get("is_tail_biting", False))
        assert np.allclose(got, z["multi3"], rtol=0, atol=1e-15)
    if "transmissions" in z.files:                      # the sampled inputs themselves
        Lw = m["L"] + (0 if m["is_bounded"] else 20) + (0 if m["is_terminated"] else 20)
        for k in range(min(5, len(seeds))):
            rs = np.random.RandomState(int(seeds[k]))
            if m.get("is_protograph") or m.get("is_tail_biting"):
                tr, mask = P.sample_trial(rs, m["e"], m["l"], m["r"], Lw, m["M"], doping if m.get("is_protograph") else (),
                                          m.get("is_protograph", False), m.get("is_tail_biting", False))
            else:
                tr = P.gen_slots(rs, m["l"], m["r"], Lw, m["M"])
                mask = rs.rand(Lw * m["M"]) <= m["e"]
            assert (tr == z["transmissions"][k]).all() and (mask == z["mask"][k].astype(bool)).all()


@pytest.mark.parametrize("path", TR, ids=[os.path.basename(p)[:-4] for p in TR])
def test_random_pick_trajectories(path):
    from oracle import pd_oracle as P
    z, m, doping = _load(path)
    seeds = z["seed"] if m["M"] <= 200 else z["seed"][:1]
    for k, s in enumerate(seeds):
        r1, plr = P.simulate_peeling_decoder_ldpc(int(s), m["e"], m["l"], m["r"], m["L"], m["M"], m["is_terminated"],
                                                  1, doping, m.get("is_protograph", False))
        assert (r1[0] == z["r1"][k]).all() and plr[0] == z["plr"][k], (path, s)
    if m["M"] <= 200:
        r1, plr = P.simulate_peeling_decoder_ldpc(int(z["seed"][0]), m["e"], m["l"], m["r"], m["L"], m["M"],
                                                  m["is_terminated"], 2, doping, m.get("is_protograph", False))
        assert (r1 == z["multi2_r1"]).all() and (plr == z["multi2_plr"]).all()


@pytest.mark.parametrize("path", UNC, ids=[os.path.basename(p)[:-4] for p in UNC])
def test_uncoupled_ensemble(path):
    """ldpc.gen_slots (repeat rejection) + simulate_peeling_decoder_ldpc_uncoupled (PD:793-869)."""
    from oracle import pd_oracle as P
    z = np.load(path)
    m = json.loads(str(z["meta"]))
    for k, s in enumerate(z["seed"]):
        rs = np.random.RandomState(int(s))
        tr = P.gen_slots_uncoupled(rs, m["l"], m["r"], m["M"])
        assert (tr == z["transmissions"][k]).all() and ((rs.rand(m["M"]) <= m["e"]) == z["mask"][k].astype(bool)).all()
        r1, plr, nv = P.simulate_peeling_decoder_ldpc_uncoupled(int(s), m["e"], m["l"], m["r"], m["M"], 1)
        assert (r1[0] == z["r1"][k]).all() and plr[0] == z["plr"][k] and nv[0] == z["num_vns"][k]
    r1, plr, nv = P.simulate_peeling_decoder_ldpc_uncoupled(int(z["seed"][0]), m["e"], m["l"], m["r"], m["M"], 2)
    assert (r1 == z["multi2_r1"]).all() and (plr == z["multi2_plr"]).all() and nv == z["multi2_num_vns"].tolist()


def test_randbelow_is_random_choice():
    from oracle import pd_oracle as P
    a, b = random.Random(7), random.Random(7)
    for n in (1, 2, 3, 5, 17, 255, 256, 1000, 4097, 26500):
        seq = range(n)
        assert [a.choice(seq) for _ in range(50)] == [P.randbelow(b, n) for _ in range(50)]


def test_calc_nu_chunk_matches_definition():
    from oracle import pd_oracle as P
    rng = np.random.RandomState(0)
    th = np.concatenate([np.linspace(50, 1, 40), np.zeros(10)])
    r1 = rng.randint(0, 80, size=(7, 60))
    ss, cnt = P.calc_nu_chunk(r1, th, 100)
    x = r1[:, :40] / 100.0
    d = x - th[:40] / 100.0
    ok = x != 0
    assert np.allclose(ss, np.where(ok, d ** 2, 0).sum(0)) and (cnt == ok.sum(0)).all()


@pytest.mark.parametrize("L,M,e,term", [(10, 20, 0.45, False), (10, 20, 0.5, True), (12, 40, 0.3, True), (8, 64, 0.62, False),
                                        (20, 200, 0.47, False)])
def test_fast_random_pick_twin_equals_the_numpy_model(L, M, e, term):
    """oracle/scldpc_oracle.c orc_random_pick_philox (Fenwick tree, O(steps log n)) — what the full-size C3 check on the
    GPU uses — against pd_oracle.random_pick_trial (the literal model pinned to the imported reference) on the device's
    Philox pick stream, incl. runs that exhaust their degree-1 CNs early and redraws of _randbelow."""
    from oracle import oracle as O
    from oracle import pd_oracle as P
    O.build(with_reference=False)
    for seed, trial in ((5, 0), (5, 1), (99, 123456789012)):
        rs = np.random.RandomState(seed + trial % 1000)
        tr = P.gen_slots(rs, 4, 8, L, M)
        mask = P.gen_erasures(rs, e, 4, 8, L, M)
        a = P.random_pick_trial(tr, mask, 4, 8, L, M, e, term, P.PhiloxPickStream(seed, trial))
        b = P.random_pick_trial_philox_fast(tr, mask, 4, 8, L, M, e, term, seed, trial)
        assert (a[0] == b[0]).all() and a[1] == b[1], (L, M, seed, trial)


VAR = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_var_chunk_*.npz")))


@pytest.mark.parametrize("path", VAR, ids=[os.path.basename(p)[:-4] for p in VAR])
def test_calc_nu_chunk_equals_the_reference(path):
    """fixtures written by the REAL fl_scaling.est_scaling_params.calc_nu_chunk (:90-94, :131-138;
    oracle/make_golden_var.py) pin (i) the oracle's restatement and (ii) the product's host-side reduction from integer
    moments (count, sum r1, sum r1^2 — what the device accumulates), float tolerance 1e-12 relative."""
    from oracle import pd_oracle as P
    from fl_scaling_sc_ldpc_amd import peeling_decoding as PD
    z = np.load(path)
    r1s, th, M = z["r1s"], z["theory"], int(z["M"])
    ss, cnt = P.calc_nu_chunk(r1s, th, M)
    assert (cnt == z["counts"]).all() and np.allclose(ss, z["ssquares"], rtol=1e-13, atol=0)
    mom = np.stack([(r1s != 0).sum(0), r1s.sum(0), (r1s ** 2).sum(0)]).astype(np.int64)
    ss2, cnt2 = PD.nu_chunk_from_moments(mom, th, M)
    assert (cnt2 == z["counts"]).all() and np.allclose(ss2, z["ssquares"], rtol=1e-12, atol=1e-300)


def test_literal_sic_round_reproduces_the_reference_known_answer_case():
    """test_2_6_csa_sync (PD:1164-1175), the one hand-built known-answer case the reference holds for sic_round: three users
    with k = 2 over ten slots; nothing is decoded for t = 0..9, all three at t = 10, and the schedule ends empty
    (tests/golden/kat_reference.json: the reference's own functions run on its own data by oracle/make_golden_kat.py)."""
    from oracle import pd_oracle as P
    kat = json.load(open(os.path.join(GOLDEN_DIR, "kat_reference.json")))["csa_2_6_sync"]
    users = {u["uid"]: {"transmissions": list(u["transmissions"]), "k": u["k"], "recovered": set()} for u in kat["users"]}
    schedule = {}
    for uid, u in users.items():
        for c in u["transmissions"]:
            schedule.setdefault(c, set()).add(uid)
    got = [sorted(P.sic_round_literal(schedule, users, t)) for t in range(11)]
    assert got == kat["decoded_at_t"] and got[10] == [1, 2, 3]
    assert {str(k): sorted(v) for k, v in schedule.items()} == kat["schedule_left"] == {}
    assert {str(uid): sorted(u["recovered"]) for uid, u in users.items()} == kat["recovered"]


@pytest.mark.parametrize("seed", range(12))
def test_array_closure_equals_the_literal_sweep(seed):
    """peel_closure (what every other test and the device twin rest on) == the literal sic_round sweep pinned above, on
    random small chains: terminated and not, bounded and not (sweep start > 0), CNs beyond total_size that never fire."""
    from oracle import pd_oracle as P
    rs = np.random.RandomState(100 + seed)
    l, r, L, M = 4, 8, 6 + seed % 5, 8 + 4 * (seed % 3)
    tr = P.gen_slots(rs, l, r, L, M)
    mask = P.gen_erasures(rs, 0.35 + 0.05 * (seed % 6), l, r, L, M)
    cpp = M * l // r
    ncn = int(tr.max()) + 1
    for total_size, start in ((ncn, 0), (cpp * L, 0), (cpp * L, 2 * cpp), (ncn, cpp)):
        assert (P.peel_closure(tr, mask, total_size, start) == P.sweep_literal(tr, mask, total_size, start)).all()
