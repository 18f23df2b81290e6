"""Parity of the Python-peeling mirror (-m gpu): fl_scaling_sc_ldpc_amd.peeling_decoding, driving the HIP kernels
peel_sweep / peel_pick / r1_moments through the C-ABI, against
  (a) golden vectors produced by importing the REAL reference (tests/golden/pd_*.npz) on identical seeds, and
  (b) the numpy oracle (oracle/pd_oracle.py) on device-sampled inputs."""
import glob
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN_DIR, require_gpu

pytestmark = pytest.mark.gpu
ER = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_er_*.npz")))
TR = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_tr_*.npz")))
UNC = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_unc_*.npz")))


@pytest.fixture(scope="module")
def PD():
    require_gpu()
    from fl_scaling_sc_ldpc_amd import peeling_decoding
    return peeling_decoding


def _load(path):
    z = np.load(path)
    m = json.loads(str(z["meta"]))
    doping = m["doping"]
    if m.get("doping_soft"):
        doping = {int(k): v for k, v in doping.items()}
    return z, m, doping


def _tuple11(t):
    return np.array([t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[10], t[11], t[12]], dtype=np.float64)


@pytest.mark.parametrize("path", ER, ids=[os.path.basename(p)[:-4] for p in ER])
def test_simulate_sc_ldpc_equals_reference_on_identical_seeds(PD, path):
    z, m, doping = _load(path)
    args = (m["e"], m["l"], m["r"], m["L"], m["M"], m["is_terminated"], m.get("is_protograph", False), m["is_bounded"],
            m.get("is_tail_biting", False))
    for k, s in enumerate(z["seed"]):
        np.random.seed(int(s)); random.seed(int(s))
        t = PD.simulate_sc_ldpc(*args, num_repeats=1, max_fuckups=2000, doping_points=doping)
        assert np.array_equal(_tuple11(t), z["tuple11"][k]), (path, s, _tuple11(t), z["tuple11"][k])
        assert len(t) == 13 and t[8].shape == (1,) and t[9].shape == (1,)
    # several trials from one stream, in one device batch
    np.random.seed(int(z["seed"][0])); random.seed(int(z["seed"][0]))
    t = PD.simulate_sc_ldpc(*args, num_repeats=3, max_fuckups=2000, doping_points=doping)
    assert np.array_equal(_tuple11(t), z["multi3"])


def test_simulate_sc_ldpc_stop_rule_leaves_the_stream_where_the_reference_does(PD, oracle):
    from oracle import pd_oracle as P
    # max_fuckups = 2 at an ε where most trials fail: the call must stop after the 2nd failing trial,
    # whatever the device batch size, and the global numpy stream must sit right after that trial's draws.
    for batch in (1, 3, 64):
        np.random.seed(5)
        t = PD.simulate_sc_ldpc(0.5, 4, 8, 10, 20, True, False, True, False, num_repeats=40, max_fuckups=2, batch=batch)
        after = np.random.rand()
        ref = P.simulate_sc_ldpc(5, 0.5, 4, 8, 10, 20, True, True, 40, 2)
        assert np.array_equal(_tuple11(t), np.array(ref))
        rs = np.random.RandomState(5)
        for _ in range(int(ref[5])):
            P.gen_slots(rs, 4, 8, 10, 20); rs.rand(200)
        assert after == rs.rand()


@pytest.mark.parametrize("path", TR, ids=[os.path.basename(p)[:-4] for p in TR])
def test_random_pick_trajectories_equal_reference_on_identical_seeds(PD, path):
    z, m, doping = _load(path)
    args = (m["e"], m["l"], m["r"], m["L"], m["M"], m["is_terminated"], m.get("is_protograph", False))
    seeds = z["seed"] if m["M"] <= 200 else z["seed"][:2]
    for k, s in enumerate(seeds):
        np.random.seed(int(s)); random.seed(int(s))
        none, r1, plrs = PD.simulate_peeling_decoder_ldpc(*args, 1, doping)
        assert none is None and r1.shape == z["r1"][k:k + 1].shape
        assert (r1[0] == z["r1"][k]).all() and plrs[0] == z["plr"][k], (path, s)
    # two trials from the shared numpy + `random` streams; both streams end where the reference leaves them
    s0 = int(z["seed"][0])
    np.random.seed(s0); random.seed(s0)
    _, r1, plrs = PD.simulate_peeling_decoder_ldpc(*args, 2, doping)
    assert (r1 == z["multi2_r1"]).all() and (plrs == z["multi2_plr"]).all()


def test_python_random_stream_is_advanced_exactly(PD):
    from oracle import pd_oracle as P
    np.random.seed(9); random.seed(9)
    _, r1, _ = PD.simulate_peeling_decoder_ldpc(0.45, 4, 8, 10, 20, False, False, 3, [])
    got = [random.random() for _ in range(3)]
    rng = random.Random(9)
    rs = np.random.RandomState(9)
    for _ in range(3):
        tr = P.gen_slots(rs, 4, 8, 10, 20); mask = P.gen_erasures(rs, 0.45, 4, 8, 10, 20)
        P.random_pick_trial(tr, mask, 4, 8, 10, 20, 0.45, False, rng)
    assert got == [rng.random() for _ in range(3)]


@pytest.mark.parametrize("L,M,e,term,bounded", [(12, 40, 0.47, True, True), (12, 40, 0.47, False, False),
                                                (50, 1000, 0.48, True, True), (20, 200, 0.5, False, True)])
def test_peel_sweep_equals_oracle_on_device_sampled_inputs(PD, L, M, e, term, bounded):
    from oracle import pd_oracle as P
    E = PD.E
    g = PD._Geometry(4, 8, L, M, term, bounded, [])
    T = 6 if M >= 1000 else 24
    d_adj, d_ch = E.sample_philox(g.params, 77, 1000, T, e, adj16=True)
    res = E.peel_sweep(g.params, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi, want_lost=True)
    out = res["out"].cpu().numpy()
    A = E.adj16_to_global(g.params, d_adj.cpu().numpy())
    bits = E.unpack_bits(d_ch.cpu().numpy(), g.params.n).astype(bool)
    lost_bits = E.unpack_bits(res["lost"].cpu().numpy(), g.params.n)
    for t in range(T):
        s = P.sc_ldpc_trial_stats(A[t].astype(np.int64), bits[t], 4, 8, L, M, term, bounded)
        assert (out[t, 0], out[t, 1], out[t, 2], out[t, 7]) == (s["num_lost"], s["num_lost_exp"], s["blocks_failed_exp"],
                                                               bits[t].sum()), (L, M, t)
        assert lost_bits[t].sum() == s["num_lost"]
    # the mirror in philox mode returns the same totals
    t13 = PD.simulate_sc_ldpc(e, 4, 8, L, M, term, False, bounded, False, num_repeats=T, rng="philox", seed=77, batch=4)
    assert t13[5] == T and t13[7] == T * g.generated


@pytest.mark.parametrize("L,M,e,term", [(10, 20, 0.45, False), (10, 20, 0.5, True), (20, 200, 0.47, False)])
def test_peel_pick_philox_stream_equals_cpu_twin(PD, oracle, L, M, e, term):
    from oracle import pd_oracle as P
    E = PD.E
    T = 5
    none, r1, plrs = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=31, batch=3)
    p = E.CodeParams(4, 8, L, M // 2, M)
    d_adj, d_ch = E.sample_philox(p, 31, 0, T, e, adj16=True)
    A = E.adj16_to_global(p, d_adj.cpu().numpy()).astype(np.int64)
    bits = E.unpack_bits(d_ch.cpu().numpy(), p.n).astype(bool)
    for t in range(T):
        ref_r1, ref_plr = P.random_pick_trial(A[t], bits[t], 4, 8, L, M, e, term, P.PhiloxPickStream(31, t))
        assert (r1[t] == ref_r1).all() and plrs[t] == ref_plr, (L, M, t)


def test_moments_and_variance_chunk(PD):
    from oracle import pd_oracle as P
    import torch
    E = PD.E
    rng = np.random.RandomState(1)
    r1 = rng.randint(0, 300, size=(37, 500)).astype(np.int32)
    r1[:, 400:] = 0
    mom = E.r1_moments(torch.from_numpy(r1).cuda())
    mom = E.r1_moments(torch.from_numpy(r1[:5]).cuda(), mom).cpu().numpy()          # accumulates in place
    both = np.concatenate([r1, r1[:5]]).astype(np.int64)
    assert (mom[0] == (both != 0).sum(0)).all() and (mom[1] == both.sum(0)).all() and (mom[2] == (both ** 2).sum(0)).all()
    theory = np.concatenate([np.linspace(250, 3, 380), np.zeros(120)])
    ss, cnt = PD.nu_chunk_from_moments(mom, theory, 1000)
    ss_ref, cnt_ref = P.calc_nu_chunk(both, theory, 1000)
    assert (cnt == cnt_ref).all() and np.allclose(ss, ss_ref, rtol=1e-12, atol=0)     # float tolerance: 1e-12 relative


def test_ber_sim_cli_writes_the_reference_rows(PD, tmp_path):
    from oracle import pd_oracle as P
    out = tmp_path / "ber.dat"
    np.random.seed(3)
    PD.main_simulate_sc_ldpc([str(out), "4", "8", "10", "20", "[0.5, 0.45]", "T", "U", "B", "NTB", "6", "2000", "[4]"])
    lines = open(out).read().strip().split("\n")
    assert lines[0] == ("# SC-LDPC (4,8,L=11,M=20) terminated:True, proto:False, bounded:True, tail biting:False. "
                        "num_repeats=6, max_fuckups=2000, doping_points=[4].")
    rs_seed = 3
    # the two ε points share one numpy stream (PD:1348-1350): replay them back to back with the oracle
    rs = np.random.RandomState(rs_seed)
    for row, e in zip(lines[1:], (0.5, 0.45)):
        acc = np.zeros(7)
        for _ in range(6):
            tr = P.gen_slots(rs, 4, 8, 11, 20); mask = P.gen_erasures(rs, e, 4, 8, 11, 20, [4])
            s = P.sc_ldpc_trial_stats(tr, mask, 4, 8, 11, 20, True, True, [4])
            acc += [s["frame_err"], s["frame_err_exp"], s["num_lost"], s["num_lost_exp"], s["generated"],
                    s["blocks_failed_exp"], s["blocks"]]
        exp = (e, acc[0] / 6, acc[1] / 6, acc[2] / acc[4], acc[3] / acc[4], int(acc[1]), 6, int(acc[3]), int(acc[4]),
               int(acc[5]), int(acc[6]), acc[5] / acc[6])
        assert row == " ".join(str(x) for x in exp)
    with pytest.raises(NotImplementedError):            # protograph + tail-biting (PD:204-205)
        PD.simulate_sc_ldpc(0.4, 4, 8, 10, 20, True, True, True, True, 1)


@pytest.mark.parametrize("L,M,e,term", [(50, 2000, 0.47, False), (20, 2000, 0.46, True), (60, 1000, 0.48, False)])
def test_peel_pick_beyond_the_lds_budget(PD, oracle, L, M, e, term):
    """CN words in the global workspace + two-level rank-select + in-kernel moments, against the CPU twin."""
    from oracle import pd_oracle as P
    E = PD.E
    T = 3
    none, r1, plrs = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=8)
    none, mom, plrs2 = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=8,
                                                        want_moments=True, batch=2)
    p = E.CodeParams(4, 8, L, M // 2, M)
    d_adj, d_ch = E.sample_philox(p, 8, 0, T, e, adj16=True)
    A = E.adj16_to_global(p, d_adj.cpu().numpy()).astype(np.int64)
    bits = E.unpack_bits(d_ch.cpu().numpy(), p.n).astype(bool)
    for t in range(T):
        ref_r1, ref_plr = P.random_pick_trial(A[t], bits[t], 4, 8, L, M, e, term, P.PhiloxPickStream(8, t))
        assert (r1[t] == ref_r1).all() and plrs[t] == ref_plr, (L, M, t)
    assert (plrs2 == plrs).all()
    assert (mom[0] == (r1 != 0).sum(0)).all() and (mom[1] == r1.sum(0)).all() and (mom[2] == (r1 ** 2).sum(0)).all()


def test_peel_pick_with_the_degree1_bitmap_in_the_workspace(PD, oracle):
    """More than ~77 000 pickable CNs: the degree-1 bitmap leaves the LDS too (one small single-wave workgroup per
    trial); few erasures keep the CPU twin's run short.  Philox picks and the exact MT19937 stream."""
    from oracle import pd_oracle as P
    E = PD.E
    L, M, e, T = 44, 4000, 0.04, 2
    none, r1, plrs = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, True, False, T, [], rng="philox", seed=12)
    p = E.CodeParams(4, 8, L, M // 2, M)
    d_adj, d_ch = E.sample_philox(p, 12, 0, T, e, adj16=True)
    A = E.adj16_to_global(p, d_adj.cpu().numpy()).astype(np.int64)
    bits = E.unpack_bits(d_ch.cpu().numpy(), p.n).astype(bool)
    for t in range(T):
        ref_r1, ref_plr = P.random_pick_trial(A[t], bits[t], 4, 8, L, M, e, True, P.PhiloxPickStream(12, t))
        assert (r1[t] == ref_r1).all() and plrs[t] == ref_plr
    np.random.seed(33); random.seed(33)
    _, r1, plrs = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, True, False, 1, [])
    ref_r1, ref_plr = P.simulate_peeling_decoder_ldpc(33, e, 4, 8, L, M, True, 1)
    assert (r1 == ref_r1).all() and (plrs == ref_plr).all()


def test_peel_pick_notebook_size_exact_stream(PD):
    """M = 10000 (the notebook's trajectory size, PD:1216) on a short chain, with the reference's own numpy + `random`
    streams: the device consumes the MT19937 state exactly as random.choice would."""
    from oracle import pd_oracle as P
    np.random.seed(21); random.seed(21)
    _, r1, plrs = PD.simulate_peeling_decoder_ldpc(0.47, 4, 8, 6, 10000, False, False, 1, [])
    ref_r1, ref_plr = P.simulate_peeling_decoder_ldpc(21, 0.47, 4, 8, 6, 10000, False, 1)
    assert (r1 == ref_r1).all() and (plrs == ref_plr).all()
    assert random.random() == (lambda g: ([P.random_pick_trial(P.gen_slots(rs, 4, 8, 6, 10000),
                                                                P.gen_erasures(rs, 0.47, 4, 8, 6, 10000), 4, 8, 6, 10000,
                                                                0.47, False, g) for rs in [np.random.RandomState(21)]],
                                          g.random())[1])(random.Random(21))


# ---- tail-biting / protograph / uncoupled ensembles --------------------------------------------------
@pytest.mark.parametrize("path", UNC, ids=[os.path.basename(p)[:-4] for p in UNC])
def test_uncoupled_ensemble_equals_reference_on_identical_seeds(PD, path):
    z = np.load(path)
    m = json.loads(str(z["meta"]))
    for k, s in enumerate(z["seed"]):
        np.random.seed(int(s)); random.seed(int(s))
        none, r1, plrs, nv = PD.simulate_peeling_decoder_ldpc_uncoupled(m["e"], m["l"], m["r"], m["M"], 1)
        assert none is None and (r1[0] == z["r1"][k]).all() and plrs[0] == z["plr"][k] and nv == [int(z["num_vns"][k])]
    s0 = int(z["seed"][0])
    np.random.seed(s0); random.seed(s0)
    _, r1, plrs, nv = PD.simulate_peeling_decoder_ldpc_uncoupled(m["e"], m["l"], m["r"], m["M"], 2)
    assert (r1 == z["multi2_r1"]).all() and (plrs == z["multi2_plr"]).all() and nv == z["multi2_num_vns"].tolist()


@pytest.mark.parametrize("ens", ["tail_biting", "protograph"])
@pytest.mark.parametrize("L,M,e,doped", [(10, 20, 0.45, ()), (12, 40, 0.5, (3, 4)), (50, 1000, 0.48, ()), (7, 1024, 0.3, (0,))])
def test_ensemble_sampler_equals_cpu_twin(PD, oracle, ens, L, M, e, doped):
    """scldpc_sample_philox_ensemble_device against orc_sample_philox_ens, plus the structure the reference's samplers
    guarantee: every CN of a populated position has exactly dc sockets, edge i of position q lands in position
    (q+i) [mod L when tail-biting]; protograph: each (portion, edge) block is a permutation of the position's CNs."""
    E = PD.E
    p = E.CodeParams(4, 8, L, M // 2, M)
    po = oracle.Params(4, 8, L, M // 2, M)
    T = 3
    d_adj, d_ch = E.sample_philox(p, 91, 5, T, e, doped, ensemble=ens)
    A, Cb = d_adj.cpu().numpy(), d_ch.cpu().numpy().view(np.uint32)
    C = M // 2
    for t in range(T):
        ra, rc = oracle.sample_philox(po, 91, 5 + t, e, doped, ensemble=ens)
        assert (A[t] == ra).all() and (Cb[t] == rc).all(), (ens, L, M, t)
        a = A[t].reshape(L, M, 4)
        for i in range(4):
            want = (np.arange(L) + i) % L if ens == "tail_biting" else np.arange(L) + i
            assert ((a[:, :, i] // C) == want[:, None]).all()
        if ens == "tail_biting":
            assert (np.bincount(A[t].ravel(), minlength=L * C) == 8).all()
        else:
            loc = (a % C).reshape(L, 2, C, 4)
            assert (np.sort(loc, axis=2) == np.arange(C)[None, None, :, None]).all()


@pytest.mark.parametrize("ens,term,bounded", [("tail_biting", False, True), ("tail_biting", True, False),
                                              ("protograph", True, True), ("protograph", False, False)])
@pytest.mark.parametrize("L,M,e", [(12, 40, 0.47), (20, 200, 0.5)])
def test_sweep_and_pick_on_the_other_ensembles_equal_oracle(PD, oracle, ens, term, bounded, L, M, e):
    from oracle import pd_oracle as P
    E = PD.E
    g = PD._Geometry(4, 8, L, M, term, bounded, [])
    T = 12
    d_adj, d_ch = E.sample_philox(g.params, 17, 40, T, e, ensemble=ens)
    out = E.peel_sweep(g.params, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi)["out"].cpu().numpy()
    A = d_adj.cpu().numpy().astype(np.int64)
    bits = E.unpack_bits(d_ch.cpu().numpy(), g.params.n).astype(bool)
    for t in range(T):
        s = P.sc_ldpc_trial_stats(A[t], bits[t], 4, 8, L, M, term, bounded)
        assert (out[t, 0], out[t, 1], out[t, 2]) == (s["num_lost"], s["num_lost_exp"], s["blocks_failed_exp"]), (ens, t)
    t13 = PD.simulate_sc_ldpc(e, 4, 8, L, M, term, ens == "protograph", bounded, ens == "tail_biting", num_repeats=T,
                              rng="philox", seed=17, batch=5)
    assert t13[5] == T
    if ens == "protograph":
        none, r1, plrs = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, True, 4, [], rng="philox", seed=23, batch=3)
        p = E.CodeParams(4, 8, L, M // 2, M)
        da, dc = E.sample_philox(p, 23, 0, 4, e, ensemble="protograph")
        A2 = da.cpu().numpy().astype(np.int64)
        b2 = E.unpack_bits(dc.cpu().numpy(), p.n).astype(bool)
        for t in range(4):
            ref_r1, ref_plr = P.random_pick_trial(A2[t], b2[t], 4, 8, L, M, e, term, P.PhiloxPickStream(23, t))
            assert (r1[t] == ref_r1).all() and plrs[t] == ref_plr


def test_protograph_with_tail_biting_is_refused(PD):
    with pytest.raises(NotImplementedError):
        PD.simulate_sc_ldpc(0.45, 4, 8, 10, 20, True, True, True, True, num_repeats=1)
    with pytest.raises(NameError):                      # the reference's own failure for soft doping there (PD:228)
        PD.simulate_sc_ldpc(0.45, 4, 8, 10, 20, True, True, True, False, num_repeats=1, doping_points={3: 0.5})
