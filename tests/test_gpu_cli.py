"""The executables' mirror end to end on the GPU (-m gpu): bp_lim_iter / sw_lim_iter / bp_traj with the reference's
argv, writing the reference's files — checked against the oracle run on the very same inputs."""
import os

import numpy as np
import pytest

from conftest import require_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    require_gpu()
    from fl_scaling_sc_ldpc_amd import bp_decoding
    return bp_decoding


def test_bp_lim_iter_glibc_replay_equals_reference_run(B, oracle, tmp_path):
    """--rng glibc --seed S replays a whole reference run: same frames, same risultati rows as the oracle fed with
    the reference's own stream (srandom(S) once, perm_code reset per ε point, BPF:2057-2161)."""
    import ctypes as C
    O = oracle
    B.bp_lim_iter(["3", "0", "0", "30", "--L", "10", "--N", "10", "--num-points", "2", "--max-frames", "25",
                   "--min-frame-err", "6", "--batch", "8", "--rng", "glibc", "--seed", "77", "--outdir", str(tmp_path),
                   "--quiet"])
    rows = open(tmp_path / "SC_LDPC_4_8_L10_M5_BP_SW0_30it_Random_BLER_3.dat").read().strip().split("\n")
    assert rows[0] == B.RISULTATI_HEADER.strip() and len(rows) == 3
    p = O.Params(4, 8, 10, 5, 10)
    n, nk = p.n, p.nk
    perm = np.empty(40, dtype=np.int32); vn_adj = np.empty((n, 4), dtype=np.int32)
    cn_ptr = np.empty(nk + 1, dtype=np.int32); cn_adj = np.empty(n * 4, dtype=np.int32); chan = np.empty(n, dtype=np.uint8)
    rng = O.Rng()
    O.lib().orc_srandom(C.byref(rng), 77)
    for sim in range(2):
        eps = 0.48 - sim * 0.00125
        O.lib().orc_perm_identity(C.byref(p), O._p(perm, C.c_int32))             # inizio_sim
        acc = dict(ue=0, fe=0, be=0, uee=0, fee=0, bee=0)
        f = 0
        while f < 25:
            O.lib().orc_generate_code(C.byref(p), C.byref(rng), O._p(perm, C.c_int32), O._p(vn_adj, C.c_int32),
                                      O._p(cn_ptr, C.c_int32), O._p(cn_adj, C.c_int32))
            O.lib().orc_channel(C.byref(p), C.byref(rng), eps, 0, None, O._p(chan, C.c_uint8))
            res, _, _ = O.decode_bp(O.Graph(p, vn_adj, cn_ptr, cn_adj), chan, max_it=30, literal=True)
            f += 1
            if res["num_erasures"] > 0:
                acc["ue"] += res["num_erasures"]; acc["fe"] += 1; acc["be"] += res["num_blocks_err"]
            if res["num_erasures_exp"] > 0:
                acc["uee"] += res["num_erasures_exp"]; acc["fee"] += 1; acc["bee"] += res["num_blocks_err_exp"]
            if acc["fe"] >= 6:
                break
        exp = "%f %e %e %e %e %e %e %d %d %d %d %d %d %d %d %d" % (
            eps, acc["ue"] / n / f, acc["fe"] / f, acc["be"] / 10 / f, acc["uee"] / n / f, acc["fee"] / f,
            acc["bee"] / 10 / f, n, 10, f, acc["ue"], acc["fe"], acc["be"], acc["uee"], acc["fee"], acc["bee"])
        assert rows[1 + sim] == exp, (sim, rows[1 + sim], exp)


def test_bp_traj_and_sw_files(B, oracle, tmp_path):
    O = oracle
    B.bp_traj(["0", "0", "0", "1000000", "0", "--L", "12", "--N", "40", "--max-frames", "9", "--min-frame-err", "9",
               "--eps-ini", "0.47", "--batch", "4", "--seed", "5", "--outdir", str(tmp_path), "--quiet"])
    path = tmp_path / "trajectories_0.4700_truncated_SC_LDPC_4_8_L12_M20_BP_Full_1000000it_Random_BLER_0.dat"
    frames = open(path).read().split("\n\n")
    assert frames[-1] == "" and len(frames) - 1 == 9                               # empty line after every frame (BPT:1145)
    from fl_scaling_sc_ldpc_amd import engine as E
    p = E.make_params(4, 8, 12, 40)
    po = O.Params(4, 8, 12, 20, 40)
    d_adj, d_ch = E.sample_philox(p, 5, 0, 9, 0.47)                                 # the driver's trials: (seed, point 0, frame t)
    A, bits = d_adj.cpu().numpy(), E.unpack_bits(d_ch.cpu().numpy(), p.n)
    for t in range(9):
        res, _, rows = O.decode_bp(O.Graph.from_vn_adj(po, A[t]), bits[t], is_term=0, literal=False, rows_cap=512)
        exp = "".join("%d\t%d\t%d\t%d\n" % (i, r["deg1"], r["recovered"], r["first_pos"]) for i, r in enumerate(rows))
        assert frames[t] + "\n" == exp, t
    B.sw_lim_iter(["1", "5", "0", "4", "9", "--L", "12", "--N", "40", "--num-points", "1", "--max-frames", "16",
                   "--batch", "16", "--seed", "5", "--outdir", str(tmp_path), "--quiet"])
    row = open(tmp_path / "SC_LDPC_4_8_L12_M20_BP_SW5_4it_9init_Random_BLER_1.dat").read().strip().split("\n")[1].split()
    d_adj, d_ch = E.sample_philox(p, 5, B.trial_key(1, 0), 16, 0.475)             # replica INDEX = 1, point 0, frames 0..15
    A, bits = d_adj.cpu().numpy(), E.unpack_bits(d_ch.cpu().numpy(), p.n)
    ue = fe = 0
    for t in range(16):
        res, _ = O.decode_sw(O.Graph.from_vn_adj(po, A[t]), bits[t], 5, 4, 9, literal=True)
        ue += res["num_erasures"]; fe += res["num_erasures"] > 0
    assert int(row[9]) == 16 and int(row[10]) == ue and int(row[11]) == fe


def test_bp_lim_iter_fixpoint_schedule_writes_the_same_rows(B, tmp_path):
    """MAX_IT = 10^6 (the reference's "unlimited"): --schedule fixpoint must produce the very file the flooding schedule
    produces — same frames consumed, same counters — on the glibc replay and on the Philox stream."""
    for rng_args in (["--rng", "glibc", "--seed", "5"], ["--rng", "philox", "--seed", "9"]):
        outs = []
        for sched in ("flooding", "fixpoint"):
            d = tmp_path / (sched + rng_args[1])
            B.bp_lim_iter(["1", "0", "0", "1000000", "--L", "16", "--N", "200", "--num-points", "3", "--max-frames", "60",
                           "--min-frame-err", "20", "--batch", "16", "--eps-ini", "0.47", "--outdir", str(d), "--quiet",
                           "--schedule", sched] + rng_args)
            files = sorted(os.listdir(d))
            assert len(files) == 1
            outs.append(open(d / files[0]).read())
        assert outs[0] == outs[1] and len(outs[0].strip().split("\n")) == 4


@pytest.mark.parametrize("name,argv", [
    ("tiny_bpt_M5_L10_e460_term_it3_wholerun", ["4", "0", "0", "3", "1", "--L", "10", "--N", "10"]),
    ("mid_bpt_M50_L20_e460_trunc_it6_wholerun", ["4", "0", "0", "6", "0", "--L", "20", "--N", "100"]),
])
def test_bp_traj_honours_max_it_like_the_reference(B, tmp_path, name, argv):
    """`bp_traj INDEX W NUM_DOPED MAX_IT IS_TERM` with a binding MAX_IT (BPT:1076 `while (iter < MaxNumIt)`, BPT:2116;
    the published files are the 500it / 1000it ones): the written file — name and text — equals what the REAL reference
    wrote on the same srandom seed (fixture: its rows frame after frame, oracle/make_golden.py), in the 4-column layout
    (BPT:988,1051) and, with --cols 3, in the layout of the published L50_M2500 files (NB cell 40:10-19)."""
    from conftest import load_golden
    g = load_golden(name)
    m = g.meta
    for cols in (4, 3):
        d = tmp_path / f"c{cols}"
        B.bp_traj(argv + ["--max-frames", str(g.T), "--min-frame-err", str(g.T), "--batch", "5", "--rng", "glibc",
                          "--seed", str(m["seed0"]), "--cols", str(cols), "--outdir", str(d), "--quiet"])
        kind = "terminated" if m["is_term"] else "truncated"
        path = d / ("trajectories_0.4600_%s_SC_LDPC_4_8_L%d_M%d_BP_Full_%dit_Random_BLER_4.dat"
                    % (kind, m["L"], m["CNsPos"], m["max_it"]))
        exp = []
        for t in range(g.T):
            r = g.rows_of(t)
            assert len(r) <= m["max_it"]
            if cols == 4:
                exp.append("".join("%d\t%d\t%d\t%d\n" % (i, r[i, 0], r[i, 1], r[i, 2]) for i in range(len(r))) + "\n")
            else:
                exp.append("".join("%d\t%d\t%d\n" % (i, r[i, 0], r[i, 1]) for i in range(len(r))) + "\n")
        assert open(path).read() == "".join(exp)
    assert any(len(g.rows_of(t)) == m["max_it"] for t in range(g.T))        # the cap binds in these fixtures
