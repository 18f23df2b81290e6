"""N > 1 path on CPU: two gloo ranks shard the batches of an ε point; the all-gathered counters give
every rank the same run totals and the same cut as a single rank (device work faked, tests/fakes.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q, cases):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fakes import FakeSimulator
    from fl_scaling_sc_ldpc_amd import engine as E
    p = E.make_params(4, 8, 10, 10)
    out = []
    for (batch, stop, max_frames) in cases:
        sim = FakeSimulator(p, batch)
        assert sim.world == world and sim.rank == rank
        pt = sim.run_point(1, 0.47, stop, max_frames)
        out.append([pt.run[k] for k in E.RUN_NAMES])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    sys.path.insert(0, HERE)
    from fakes import fake_counters, numpy_accumulate
    from fl_scaling_sc_ldpc_amd import engine as E
    cases = [(8, 0, 100), (8, 17, 500), (5, 3, 64), (64, 1000, 100), (7, 40, 333)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, cases)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = E.make_params(4, 8, 10, 10)
    for i, (batch, stop, max_frames) in enumerate(cases):
        ref = numpy_accumulate(fake_counters(1, np.arange(max_frames), p.n, p.L), np.zeros(E.NRUN), stop).tolist()
        assert got[0][i] == ref and got[1][i] == ref, (cases[i], got[0][i], ref)


def _program_worker(rank, world, port, outdir, shard, q):
    """run_program (the body of main_terminated) with the device work faked: files written by rank 0."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import fakes
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    made = []

    class Sim(fakes.FakeSimulator):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            made.append(self)
    B.Simulator = Sim
    argv = ["2", "0", "0", "50", "--L", "10", "--N", "10", "--num-points", "5", "--max-frames", "1000",
            "--min-frame-err", "1000", "--shard", shard, "--outdir", outdir, "--quiet", "--seed", "3"]
    opts = B._parser("bp_lim_iter").parse_args(argv)
    B.run_program("bp_lim_iter", opts.INDEX, opts.W, opts.NUM_DOPED, opts.MAX_IT, None, opts)
    q.put((rank, made[0].frames_decoded))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_points_and_frames_sharding_write_the_single_rank_file(tmp_path):
    """The ε grid sharded over the ranks (rank = point mod world, rows merged by rank 0 in grid order) and the frames of
    every point split evenly over the ranks both write the file one rank writes — at the reference's defaults
    (--batch 2048, 1000 frames per point) with non-zero work on every rank."""
    ctx = mp.get_context("spawn")
    port = 33500 + os.getpid() % 2000
    texts, work = {}, {}
    for k, (world, shard) in enumerate([(1, "auto"), (2, "points"), (2, "frames"), (2, "auto")]):
        d = str(tmp_path / f"w{world}{shard}")
        q = ctx.Queue()
        procs = [ctx.Process(target=_program_worker, args=(r, world, port + 7 * k, d, shard, q)) for r in range(world)]
        for pr in procs:
            pr.start()
        work[(world, shard)] = dict(q.get(timeout=120) for _ in range(world))
        for pr in procs:
            pr.join(timeout=120)
            assert pr.exitcode == 0
        files = sorted(os.listdir(d))
        assert files == ["SC_LDPC_4_8_L10_M5_BP_SW0_50it_Random_BLER_2.dat"]
        texts[(world, shard)] = open(os.path.join(d, files[0])).read()
    one = texts[(1, "auto")]
    assert len(one.strip().split("\n")) == 6
    for key, t in texts.items():
        assert t == one, key
    assert work[(1, "auto")][0] == 5000
    assert work[(2, "points")] == {0: 3000, 1: 2000} and work[(2, "auto")] == work[(2, "points")]
    assert work[(2, "frames")] == {0: 2500, 1: 2500}            # 500 + 500 of every point's 1000 frames


def _case_worker(rank, world, port, outdir, case, q):
    """run_program under gloo with the device work faked; puts (rank, exit code or exception text) on q."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import fakes
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    B.Simulator = fakes.FakeSimulator
    prog, argv = case["prog"], list(case["argv"]) + ["--outdir", outdir, "--quiet", "--seed", "3"]
    fakes.FakeSimulator.bad_frames = tuple(case.get("bad", ()))
    opts = B._parser(prog).parse_args(argv)
    extra = getattr(opts, "IS_TERM", None) if prog == "bp_traj" else None
    try:
        rc = B.run_program(prog, opts.INDEX, opts.W, opts.NUM_DOPED, opts.MAX_IT, extra, opts)
    except SystemExit as e:
        rc = e.code
    q.put((rank, rc))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_case(tmp_path, tag, world, case, port):
    ctx = mp.get_context("spawn")
    d = str(tmp_path / tag)
    os.makedirs(d, exist_ok=True)
    q = ctx.Queue()
    procs = [ctx.Process(target=_case_worker, args=(r, world, port, d, case, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    rcs = dict(q.get(timeout=120) for _ in range(world))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    return d, rcs


def test_glibc_replay_is_refused_on_more_than_one_rank(tmp_path):
    """--rng glibc is ONE srandom stream carried from point to point (BPF:2057-2131): a multi-rank job would replay it from
    its start on every rank.  run_program refuses before choosing a shard mode (points mode used to slip through)."""
    case = {"prog": "bp_lim_iter", "argv": ["0", "0", "0", "50", "--L", "10", "--N", "10", "--num-points", "4", "--max-frames", "20",
                                            "--rng", "glibc"]}
    d, rcs = _run_case(tmp_path, "glibc2", 2, case, 34100 + os.getpid() % 1000)
    assert all(isinstance(rc, str) and "single process" in rc for rc in rcs.values()), rcs
    assert os.listdir(d) == []


def test_points_mode_abort_leaves_together_and_keeps_the_rows_before(tmp_path):
    """A frame that breaks decodeBP's invariant (status != 0, BPF:1035-1039) on ONE rank's point: the flag travels with the
    counters, every rank exits -1 after the same all-reduce (none is left waiting in a collective), and the file holds
    the rows of the points before the broken one, as the reference's per-point append would have left them."""
    base = ["2", "0", "0", "50", "--L", "10", "--N", "10", "--num-points", "5", "--max-frames", "200", "--min-frame-err", "1000",
            "--shard", "points"]
    port = 34300 + os.getpid() % 1000
    d_ok, rcs = _run_case(tmp_path, "ok", 2, {"prog": "bp_lim_iter", "argv": base}, port)
    assert rcs == {0: 0, 1: 0}
    d_bad, rcs = _run_case(tmp_path, "bad", 2, {"prog": "bp_lim_iter", "argv": base, "bad": [(3, 17)]}, port + 11)
    assert rcs == {0: -1, 1: -1}, rcs                          # point 3 belongs to rank 1; rank 0 leaves with it
    name = "SC_LDPC_4_8_L10_M5_BP_SW0_50it_Random_BLER_2.dat"
    assert sorted(os.listdir(d_ok)) == [name] and sorted(os.listdir(d_bad)) == [name]      # part files removed
    ok, bad = open(os.path.join(d_ok, name)).read().split("\n"), open(os.path.join(d_bad, name)).read().split("\n")
    assert len(ok) == 7 and bad[:4] == ok[:4] and len(bad) == 5                          # header + points 0, 1, 2
    # one rank: the abort is immediate, as in the reference
    d_one, rcs = _run_case(tmp_path, "bad1", 1, {"prog": "bp_lim_iter", "argv": base, "bad": [(3, 17)]}, port + 22)
    assert rcs == {0: -1}
    assert open(os.path.join(d_one, name)).read().split("\n")[:4] == ok[:4]


def test_bp_traj_ranks_are_the_replicas_of_an_array_job(tmp_path):
    """bp_traj on two ranks = the processes INDEX and INDEX + 1 of the reference's array job (BPT:2095, 2131-2134;
    NB cell 35:21): two files, each byte for byte what the single process with that INDEX writes."""
    argv = ["0", "0", "1000000", "1", "--L", "10", "--N", "10", "--max-frames", "23", "--min-frame-err", "23", "--batch", "8"]
    port = 34600 + os.getpid() % 1000
    d2, rcs = _run_case(tmp_path, "two", 2, {"prog": "bp_traj", "argv": ["6"] + argv}, port)
    assert rcs == {0: 0, 1: 0}
    names = sorted(os.listdir(d2))
    assert names == ["trajectories_0.4600_terminated_SC_LDPC_4_8_L10_M5_BP_Full_1000000it_Random_BLER_%d.dat" % i for i in (6, 7)]
    texts = [open(os.path.join(d2, nm)).read() for nm in names]
    assert texts[0] != texts[1]                                # different replicas draw different frames
    for k, idx in enumerate(("6", "7")):
        d1, rcs = _run_case(tmp_path, "one" + idx, 1, {"prog": "bp_traj", "argv": [idx] + argv}, port + 13 + k)
        assert rcs == {0: 0} and os.listdir(d1) == [names[k]]
        assert open(os.path.join(d1, names[k])).read() == texts[k]
        assert len(texts[k].split("\n\n")) - 1 == 23         # 23 frames, an empty line after each


def test_split_round_gives_every_rank_work_at_the_defaults():
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    R, sizes, offs = B.Simulator.split_round(1000, 2048, 8)
    assert R == 1000 and sizes == [125] * 8 and offs == [125 * r for r in range(8)]
    R, sizes, offs = B.Simulator.split_round(10, 4, 3)
    assert R == 10 and sizes == [4, 3, 3] and offs == [0, 4, 7]
    R, sizes, offs = B.Simulator.split_round(100, 4, 3)
    assert R == 12 and sizes == [4, 4, 4]


def _stream_worker(rank, world, port, outdir, streams):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    from fakes import FakeStreams
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    B.E.Streams = FakeStreams                      # the device work; everything else is the product's driver
    opts = argparse.Namespace(N=20, L=12, dv=4, dc=8, eps_ini=0.49, eps_delta=0.01, num_points=3, streams=streams, chunk=5,
                              seed=1, outdir=outdir, quiet=True, max_blocks_err=40, max_blocks=100000)
    B.run_streaming(0, 4, [3], opts)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_streaming_driver_two_ranks_equal_one_rank(tmp_path):
    """main_streaming's driver: 2 ranks x S streams stop at the same chunk and write the same results_circular rows as
    1 rank x 2S streams (the only exchange is the all-reduce of eight int64 per chunk)."""
    ctx = mp.get_context("spawn")
    port = 31500 + os.getpid() % 2000
    d2, d1 = str(tmp_path / "two"), str(tmp_path / "one")
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, d2, 6)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    one = ctx.Process(target=_stream_worker, args=(0, 1, port + 1, d1, 12))
    one.start(); one.join(timeout=120)
    assert one.exitcode == 0
    f2, f1 = sorted(os.listdir(d2)), sorted(os.listdir(d1))
    assert f1 == f2 and len(f1) == 1
    a, b = open(os.path.join(d2, f2[0])).read(), open(os.path.join(d1, f1[0])).read()
    assert a == b and len(a.splitlines()) == 4          # header + one row per eps point


# ---- the Python peeling mirror: trial ranges per rank, one all-reduce / all-gather (device work faked) --------------------
def _fake_engine(PD):
    """Replace the three device calls of peeling_decoding.py by pure functions of the GLOBAL trial index."""
    import types

    def sample_philox(p, seed, trial0, ntrials, eps, doped=(), device="cpu", out=None, adj16=False, ensemble="olmos"):
        ids = torch.arange(trial0, trial0 + ntrials, dtype=torch.int64)
        return ids, ids

    def peel_sweep(p, d_adj, d_chan, total_size, sweep_start=0, lost_lo=0, lost_hi=None, want_lost=False):
        h = (d_adj * 2654435761 + 12345) & 0xFFFFFFFF
        fail = (h >> 7) % 3 != 0
        lost = torch.where(fail, 1 + h % 97, torch.zeros_like(h))
        out = torch.zeros((d_adj.shape[0], 8), dtype=torch.int32)
        out[:, 0] = lost.to(torch.int32)
        out[:, 1] = torch.where(lost > 2, lost, torch.zeros_like(lost)).to(torch.int32)
        out[:, 2] = (lost > 2).to(torch.int32) * (1 + (h % 5)).to(torch.int32)
        return {"out": out, "lost": None}

    def peel_pick(p, d_adj, d_chan, total_size, num_steps, mt_state=None, seed=0, trial0=0, want_r1=True, moments=None):
        T = d_adj.shape[0]
        steps = torch.arange(num_steps + 1, dtype=torch.int64)[None, :]
        r1 = ((d_adj[:, None] * 7919 + steps * 104729) % 23) * ((steps + d_adj[:, None]) % 5 != 0)
        out = torch.zeros((T, 4), dtype=torch.int32)
        out[:, 0] = (50 + d_adj % 11).to(torch.int32)
        out[:, 1] = (d_adj % 7).to(torch.int32)
        if moments is not None:
            moments[0] += (r1 != 0).sum(0)
            moments[1] += r1.sum(0)
            moments[2] += (r1 * r1).sum(0)
        return {"out": out, "r1": r1.to(torch.int32) if want_r1 else None, "moments": moments}

    def r1_moments(d_r1, moments=None):
        r = d_r1.to(torch.int64)
        if moments is None:
            moments = torch.zeros((3, r.shape[1]), dtype=torch.int64)
        moments[0] += (r != 0).sum(0)
        moments[1] += r.sum(0)
        moments[2] += (r * r).sum(0)
        return moments

    def accumulate_peel(d_out, run, max_fuckups=0):        # literal loop of PD:668-699 over the rows, in trial order
        r = run.numpy()
        for row in d_out.numpy():
            if max_fuckups > 0 and r[1] >= max_fuckups:
                break
            r[0] += 1; r[1] += row[0] >= 1; r[2] += row[0]; r[3] += row[1] > 0; r[4] += row[1]; r[5] += row[2]
        return run

    PD.E = types.SimpleNamespace(sample_philox=sample_philox, peel_sweep=peel_sweep, peel_pick=peel_pick,
                                 r1_moments=r1_moments, CodeParams=PD.E.CodeParams, accumulate_peel=accumulate_peel,
                                 NPEELRUN=PD.E.NPEELRUN, PEELRUN_NAMES=PD.E.PEELRUN_NAMES)


def _pd_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from fl_scaling_sc_ldpc_amd import peeling_decoding as PD
    _fake_engine(PD)
    out = []
    for reps, maxf, batch in ((37, 2000, 4), (50, 9, 8), (5, 2000, 64)):
        t = PD.simulate_sc_ldpc(0.45, 4, 8, 10, 20, True, False, True, False, num_repeats=reps, max_fuckups=maxf,
                                rng="philox", seed=3, batch=batch, device="cpu")
        out.append([float(x) for x in t[:8]] + [float(x) for x in t[10:]])
    _, r1, plrs = PD.simulate_peeling_decoder_ldpc(0.45, 4, 8, 10, 20, False, False, 11, [], rng="philox", seed=3, batch=3,
                                                   device="cpu")
    _, mom, plrs2 = PD.simulate_peeling_decoder_ldpc(0.45, 4, 8, 10, 20, False, False, 11, [], rng="philox", seed=3, batch=3,
                                                     device="cpu", want_moments=True)
    _, mom_k, _ = PD.simulate_peeling_decoder_ldpc(0.45, 4, 8, 10, 20, False, False, 11, [], rng="philox", seed=3, batch=3,
                                                   device="cpu", want_moments=True, moments_from="kernel")
    assert (mom_k == mom).all()                            # rows + reduction pass == accumulation inside the kernel
    q.put((rank, out, r1.tolist(), plrs.tolist(), mom.tolist(), plrs2.tolist()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_peeling_mirror_two_ranks_equal_one_rank():
    """simulate_sc_ldpc (ordered stop rule at max_fuckups, PD:698) and simulate_peeling_decoder_ldpc (r1 rows, plrs, moment
    vectors) in Philox mode: two ranks return, on BOTH ranks, exactly what one rank returns."""
    ctx = mp.get_context("spawn")
    port = 35500 + os.getpid() % 2000
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        procs = [ctx.Process(target=_pd_worker, args=(r, world, port + world, q)) for r in range(world)]
        for pr in procs:
            pr.start()
        got = [q.get(timeout=120) for _ in range(world)]
        for pr in procs:
            pr.join(timeout=60)
            assert pr.exitcode == 0
        res[world] = {g[0]: g[1:] for g in got}
    one = res[1][0]
    assert res[2][0] == one and res[2][1] == one
    assert one[0][1][5] < 50            # the stop rule cut the second case short
