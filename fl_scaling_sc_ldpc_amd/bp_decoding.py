"""Host-side mirror of the reference's three BP executables (simulators_sc_ldpc/bp_decoding):

    bp_lim_iter INDEX W NUM_DOPED MAX_IT              (BPF main_terminated, BPF:2057-2161)
    sw_lim_iter INDEX W NUM_DOPED MAX_IT INIT_IT      (BPW:2099-2158)
    bp_traj     INDEX W NUM_DOPED MAX_IT IS_TERM      (BPT:2095-2178)

Same positional argv, same ε grid / stop rule / output files (names and row formats of `risultati`,
BPF:458-519, and of the trajectory dump, BPT:988,1051,1145), but the ensemble size and grid — compile
time #defines in the reference (BPF:22-67) — are options with the reference's values as defaults, the
frames of an ε point are decoded in device batches.  Under torch.distributed (one process per GPU) the work
is sharded the way the reference is run on a cluster — one process per subset of ε points (NB cell 35:21-25,
loop BPF:2111-2114): point `sim` belongs to rank `sim % world`, no collective on the data path, and after
every wave of `world` points one all-reduce of the 9 run counters per point lets rank 0 append the rows in
grid order.  With fewer points than ranks (or --shard frames) the frames of a point are split evenly over
the ranks instead; the only exchange is then an all-gather of the per-trial counter rows (32 B per trial),
so that the ordered stop rule `frame_err >= 1000` cuts at the same frame on every rank.

`bp_traj` is a one-point program that the reference runs as an array of processes, one per INDEX (BPT:2095, file name
BPT:2131-2134; NB cell 35:21): under torch.distributed rank r IS process INDEX + r — it decodes that replica's frames
alone and writes that replica's file, so an N-rank job leaves the N files that N single runs leave.

Two sampling modes:
  * rng="philox" (default): codes and channels drawn on the device, counter-based, trial t of point s of
    replica INDEX keyed by (seed, INDEX·2^52 + s·2^40 + t) — any batch size / GPU count gives the same files, and the
    processes of an array job (same --seed, different INDEX) draw disjoint streams.
  * rng="glibc": the reference's own stream, srandom(seed) once and frames drawn back to back with
    perm_code carried over — reproduces a reference run bit for bit on an identical seed (the
    reference seeds from gettimeofday, BPF:2059-2062; pass --seed to pin it).  Sampling is then a
    sequential host loop (as in the reference); decoding still runs on the device.  Single rank only.

All compute is in libscldpc_hip.so; this file is orchestration and file formats.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

from . import engine as E
from .engine import CodeParams, NCOUNTERS, NRUN, RUN_NAMES  # noqa: F401

POINT_STRIDE = 1 << 40          # philox trial index = replica * REPLICA_STRIDE + point * POINT_STRIDE + frame
REPLICA_STRIDE = 1 << 52        # replica = the executable's INDEX argument (one process of the reference's array jobs)


def trial_key(index, sim, frame=0):
    """Philox trial index of frame `frame` of ε point `sim` in the run of replica `index`."""
    if not (0 <= index < 4096 and 0 <= sim < 4096 and 0 <= frame < POINT_STRIDE):
        raise ValueError("INDEX and the point number must lie in [0, 4096)")
    return index * REPLICA_STRIDE + sim * POINT_STRIDE + frame


class GridSpec:
    """ε grid and stop rule: Def_epsIni − sim·Def_epsDelta for sim < Def_NUM_POINTS (BPF:55-61,301);
    stop at frame_err >= min_frame_err or after max_frames frames (BPF:63-65, 440-451, 2117)."""

    def __init__(self, eps_ini, eps_delta, num_points, min_frame_err, max_frames):
        self.eps_ini, self.eps_delta, self.num_points = eps_ini, eps_delta, num_points
        self.min_frame_err, self.max_frames = min_frame_err, max_frames

    def eps(self, sim):
        return float(self.eps_ini) - sim * float(self.eps_delta)


# the shipped #defines of the three sources
DEFAULTS = {
    "bp_lim_iter": dict(N=1000, L=50, grid=GridSpec(0.48, 0.00125, 26, 1000, 1000)),
    "sw_lim_iter": dict(N=1000, L=50, grid=GridSpec(0.475, 0.00125, 18, 1000, 1000)),
    "bp_traj": dict(N=5000, L=50, grid=GridSpec(0.46, 0.005, 1, 500, 500)),
}


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


class PointResult:
    """Counters of one ε point after the stop rule, as `risultati` needs them (BPF:499-515)."""

    def __init__(self, eps, n, L, run, bad=False):
        self.eps, self.n, self.L = eps, n, L
        self.run = {k: int(v) for k, v in zip(RUN_NAMES, run)}
        self.f = self.run["frames"]
        self.bad = bool(bad)            # a frame broke decodeBP's invariant (BPF:1035-1039): the reference aborts there

    def row(self):
        r, f, n, L = self.run, self.f, self.n, self.L
        vals = (self.eps, r["users_err"] / n / f, r["frame_err"] / f, r["block_err"] / L / f,
                r["users_err_exp"] / n / f, r["frame_err_exp"] / f, r["block_err_exp"] / L / f)
        ints = (n, L, f, r["users_err"], r["frame_err"], r["block_err"], r["users_err_exp"],
                r["frame_err_exp"], r["block_err_exp"])
        return "%f %e %e %e %e %e %e " % vals + " ".join("%d" % v for v in ints) + "\n"


RISULTATI_HEADER = ("p BER FER BLER BER_EXP FER_EXP BLER_EXP n L f users_err frame_err block_err "
                    "users_err_exp frame_err_exp block_err_exp\n")


def result_filename(prog, p, W, max_it, init_it, index):
    """BPF:487 / BPW:488.  (The reference prints Def_M = CNs per position as `M`.)"""
    if prog == "sw_lim_iter":
        return "SC_LDPC_%d_%d_L%d_M%d_BP_SW%d_%dit_%dinit_Random_BLER_%d.dat" % (
            p.dv, p.dc, p.L, p.cns_pos, W, max_it, init_it, index)
    return "SC_LDPC_%d_%d_L%d_M%d_BP_SW%d_%dit_Random_BLER_%d.dat" % (p.dv, p.dc, p.L, p.cns_pos, W, max_it, index)


def traj_filename(p, eps, max_it, is_term, index):
    """BPT:2131-2134."""
    kind = "terminated" if is_term else "truncated"
    return "trajectories_%.4f_%s_SC_LDPC_%d_%d_L%d_M%d_BP_Full_%dit_Random_BLER_%d.dat" % (
        eps, kind, p.dv, p.dc, p.L, p.cns_pos, max_it, index)


def write_risultati(path, sim, point):
    """First point truncates and writes the header, later points append (BPF:489-497)."""
    with open(path, "w" if sim == 0 else "a") as f:
        if sim == 0:
            f.write(RISULTATI_HEADER)
        f.write(point.row())


class Simulator:
    """Batched Monte-Carlo driver around the device decoders."""

    def __init__(self, p, decoder="full", W=0, max_it=0, init_it=0, is_term=True, doped=(), batch=2048,
                 rng="philox", seed=1, device=None, rows_cap=0, schedule="flooding", shard_frames=True, index=0,
                 verbose=False):
        self.p, self.decoder, self.W, self.max_it, self.init_it = p, decoder, W, max_it, init_it
        self.index = index              # replica (the executables' INDEX): part of the Philox key
        self.schedule = schedule        # "fixpoint": unlimited full BP without the iteration count (1.2x faster)
        self.is_term, self.doped, self.batch, self.rng, self.seed = is_term, tuple(doped), batch, rng, seed
        self.rows_cap, self.verbose = rows_cap, verbose
        self.dist, self.rank, self.world = _dist()
        job_world = self.world
        if not shard_frames:                    # ε points / replicas are sharded by the caller: every point runs on one rank
            self.dist, self.rank, self.world = None, 0, 1
        self.device = torch.device(device) if device is not None else E.local_device()
        if rng == "glibc":
            # checked on the JOB's world size: with the points sharded every rank would otherwise replay the same
            # srandom(seed) stream from its start, where the reference carries random() and perm_code from point to point
            if job_world != 1:
                raise ValueError("rng='glibc' replays one sequential reference stream: single rank only")
            self.glibc = E.GlibcRun(p, seed)
        elif rng != "philox":
            raise ValueError("rng must be 'philox' or 'glibc'")
        self._alloc()

    def _alloc(self):
        p, batch = self.p, self.batch
        # compact 2-byte position-local ids for device-sampled codes; the reference's int32 VNdegree for host replays
        adj_dtype = torch.int16 if (self.rng == "philox" and p.cns_pos <= 65536) else torch.int32
        self.d_adj = torch.empty((batch, p.n, p.dv), dtype=adj_dtype, device=self.device)
        self.d_ch = torch.empty((batch, p.nw), dtype=torch.int32, device=self.device)
        self.d_cnt = torch.empty((batch, NCOUNTERS), dtype=torch.int32, device=self.device)
        # full BP on the BASELINE ensemble family: the second-generation pair (sampler_v2 + the 4-bits-per-CN decoder) needs
        # the CN -> VN table next to the VN -> CN one.  gen2: unlimited, no iteration statistics (fixpoint); lvl2: the same
        # decoder walked one flooding iteration per round — iteration caps (the published ..._500it_... tables) and counts
        small = self.rng == "philox" and self.decoder == "full"
        self.sock = small and not E.cn16_supported(p) and E.full_bp_sock16_supported(p)    # n >= 65535: CN -> socket table
        cn16 = small and (E.cn16_supported(p) or self.sock)
        self.gen2 = cn16 and self.rows_cap == 0 and self.schedule == "fixpoint" and (self.max_it <= 0 or self.max_it >= 1000000)
        self.lvl2 = cn16 and not self.gen2
        # square-window decoding with the window's state in LDS reads a CN -> socket table: sampled with the code where the
        # second-generation sampler takes the ensemble (else E.sw_bp builds it in a pass of its own)
        self.ring2 = (self.rng == "philox" and self.decoder == "sw" and adj_dtype == torch.int16
                      and E.sock16_supported(p) and E.sw_ring_supported(p, self.W))
        self.d_cn = (torch.empty((batch, p.nk, p.dc), dtype=torch.int16, device=self.device)
                     if (self.gen2 or self.lvl2 or self.ring2) else None)
        if self.verbose:
            print("[scldpc] kernels: " + self.kernel_choice(), file=sys.stderr, flush=True)

    def kernel_choice(self):
        """Which device kernels this configuration runs (the second-generation ones take dv = 4, dc = 8, N <= 2048)."""
        if self.decoder == "sw":
            return ("sampler_v2 (CN->socket table) + sw_ring (window state in LDS)" if self.ring2 else
                    "sampler (first generation) + " + ("sw_ring + cn_sockets pass" if E.sw_ring_supported(self.p, self.W)
                                                       and self.d_adj.dtype == torch.int16 else "sw_bp (whole chain)"))
        samp = "glibc replay on the host" if self.rng == "glibc" else \
            ("sampler_v3 (CN->socket table)" if self.sock else "sampler_v3 (CN->VN table)") if (self.gen2 or self.lvl2) \
            else "sampler (first generation)"
        if self.gen2:
            return samp + " + full_bp_small fixpoint (4-bit CN counts)"
        if self.lvl2:
            return samp + " + full_bp_small level-synchronous (4-bit CN counts" + (", trajectory rows)" if self.rows_cap else ")")
        return samp + " + full_bp (16-bit CN words" + (", trajectory rows)" if self.rows_cap else ")") + \
            ": the second-generation decoder takes dv = 4, dc = 8, N <= 2048 with device sampling"

    def _accumulate(self, allcnt, run, stop_frame_err):
        return E.accumulate_run(allcnt, run, stop_frame_err)

    def _new_run(self):
        return E.new_run(self.device)

    # -- one device batch ------------------------------------------------------------------------
    def decode_batch(self, nb, want_rows=False):
        adj, ch, cnt = self.d_adj[:nb], self.d_ch[:nb], self.d_cnt[:nb]
        if self.decoder == "sw":
            return E.sw_bp(self.p, adj, ch, self.W, self.max_it, self.init_it, counters=cnt,
                           d_cn_sock=self.d_cn[:nb] if self.ring2 else None)
        if self.gen2 and not want_rows:
            return E.full_bp_fixpoint_cn16(self.p, adj, self.d_cn[:nb], ch, is_term=self.is_term, counters=cnt, sockets=self.sock)
        if self.lvl2:
            return E.full_bp_cn16(self.p, adj, self.d_cn[:nb], ch, max_it=self.max_it, is_term=self.is_term, counters=cnt,
                                  sockets=self.sock, rows_cap=self.rows_cap if want_rows else 0)
        if self.schedule == "fixpoint" and not want_rows and (self.max_it <= 0 or self.max_it >= 1000000):
            return E.full_bp_fixpoint(self.p, adj, ch, is_term=self.is_term, counters=cnt)    # no iteration counts
        return E.full_bp(self.p, adj, ch, max_it=self.max_it, is_term=self.is_term,
                         rows_cap=self.rows_cap if want_rows else 0, counters=cnt)

    def fill_batch(self, sim, eps, frame0, nb):
        if self.rng == "philox" and (self.gen2 or self.lvl2) and self.sock:
            E.sample_philox_sock16(self.p, self.seed, trial_key(self.index, sim, frame0), nb, eps, self.doped,
                                   out=(self.d_adj[:nb], self.d_cn[:nb], self.d_ch[:nb]))
        elif self.rng == "philox" and (self.gen2 or self.lvl2):
            E.sample_philox_cn16(self.p, self.seed, trial_key(self.index, sim, frame0), nb, eps, self.doped,
                                 out=(self.d_adj[:nb], self.d_cn[:nb], self.d_ch[:nb]))
        elif self.rng == "philox" and self.ring2:
            E.sample_philox_sock16(self.p, self.seed, trial_key(self.index, sim, frame0), nb, eps, self.doped,
                                   out=(self.d_adj[:nb], self.d_cn[:nb], self.d_ch[:nb]))
        elif self.rng == "philox":
            E.sample_philox(self.p, self.seed, trial_key(self.index, sim, frame0), nb, eps, self.doped,
                            out=(self.d_adj[:nb], self.d_ch[:nb]))
        else:
            adj, ch = self.glibc.next_frames(nb, eps, self.doped)
            self.d_adj[:nb].copy_(torch.from_numpy(adj))
            self.d_ch[:nb].copy_(torch.from_numpy(ch.view(np.int32)))

    # -- one ε point -------------------------------------------------------------------------------
    @staticmethod
    def split_round(frames_left, batch, world):
        """Frames of one round and their even split over the ranks (contiguous ranges in frame order): a round
        takes min(frames_left, world*batch) frames, rank r gets ⌈·⌉ or ⌊·⌋ of them — at the reference's defaults
        (1000 frames per point, BPF:65) every rank of an 8-GPU job decodes 125 frames, none idles."""
        R = min(frames_left, world * batch)
        base, rem = divmod(R, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
        offs = [sum(sizes[:r]) for r in range(world)]
        return R, sizes, offs

    def run_point(self, sim, eps, min_frame_err, max_frames, on_batch=None, defer_abort=False):
        """Frames 0,1,2,… of the point until frame_err >= min_frame_err or max_frames frames, in frame
        order (BPF:2117-2144).  Every round takes the next min(frames left, world·batch) frames and splits them
        evenly over the ranks; the per-trial counter rows are all-gathered and accumulated in frame order on
        every rank, so all ranks cut at the same frame.  The host looks at the run counters only in rounds in
        which the stop rule could trip (frames so far + this round >= min_frame_err); other rounds stay
        asynchronous.  on_batch(frame0, frames_used, result) sees this rank's batches.
        A frame that breaks decodeBP's invariant aborts the process as in the reference (BPF:1035-1039) — every rank of a
        frame-sharded job sees it in the gathered rows and leaves together; with defer_abort the point comes back with
        .bad set instead, for callers whose other ranks are busy elsewhere and must be told first."""
        p, B, W = self.p, self.batch, self.world
        i_frames, i_ferr, i_status = RUN_NAMES.index("frames"), RUN_NAMES.index("frame_err"), \
            E.COUNTER_NAMES.index("status")
        if self.rng == "glibc":
            self.glibc.new_point()
        run = self._new_run()
        bad = torch.zeros((), dtype=torch.bool, device=run.device)
        frame0, consumed = 0, 0
        while frame0 < max_frames:
            R, sizes, offs = self.split_round(max_frames - frame0, B, W)
            nb, off = sizes[self.rank], offs[self.rank]
            res = None
            snap = self.glibc.snapshot() if self.rng == "glibc" else None
            if nb:
                self.fill_batch(sim, eps, frame0 + off, nb)
                res = self.decode_batch(nb, want_rows=on_batch is not None and self.rows_cap > 0)
            if W > 1:
                m = max(sizes)
                if nb < m:
                    self.d_cnt[nb:m].zero_()
                gathered = [torch.empty_like(self.d_cnt[:m]) for _ in range(W)]
                self.dist.all_gather(gathered, self.d_cnt[:m].contiguous())
                allcnt = torch.cat([g[:s] for g, s in zip(gathered, sizes)], dim=0).contiguous()
            else:
                allcnt = self.d_cnt[:nb]
            self._accumulate(allcnt, run, min_frame_err)
            can_trip = min_frame_err > 0 and frame0 + R >= min_frame_err
            if can_trip:
                r = run.cpu().numpy()
                used_round = int(r[i_frames]) - consumed
                consumed = int(r[i_frames])
                stopped = r[i_ferr] >= min_frame_err
            else:
                used_round, stopped = R, False
                consumed += R
            bad |= (allcnt[:used_round, i_status] != 0).any()
            if on_batch is not None and nb:
                on_batch(frame0 + off, max(0, min(nb, used_round - off)), res)
            if stopped and snap is not None and used_round < nb:
                # the reference stops drawing at the tripping frame: rewind the stream to just after it
                self.glibc.restore(snap)
                self.glibc.next_frames(used_round, eps, self.doped)
            frame0 += R
            if stopped:
                break
        if bool(bad) and not defer_abort:
            abort_invariant()
        return PointResult(eps, p.n, p.L, run.cpu().numpy(), bad=bool(bad))


def abort_invariant():
    """The reference aborts the process at a frame that recovers more VNs than it had degree-1 CNs (BPF:1035-1039)."""
    print("ARGH! RECOVERED MORE VNs THAN deg-1 CNs! Aborting!", flush=True)
    raise SystemExit(-1)


def _write_traj_rows(fh, rows, counters, nb, cols=4):
    """Per frame: one line per iteration `iter\\tdeg1\\trecovered\\tfirst_pos`, then an empty line
    (BPT:988,1051,1145).  cols=3 drops the last column: the layout of the published `…L50_M2500…` files
    (an older build of the program; read by NB cell 40:10-19 with a 3-column unpack)."""
    rows = rows.cpu().numpy()
    its = counters[:, E.COUNTER_NAMES.index("iterations")].cpu().numpy()
    out = []
    for t in range(nb):
        k = int(its[t])
        if k > rows.shape[1]:
            raise RuntimeError(f"trajectory of {k} iterations exceeds rows_cap={rows.shape[1]}; raise --rows-cap")
        r = rows[t, :k]
        if cols == 3:
            out.append("".join("%d\t%d\t%d\n" % (i, r[i, 0], r[i, 1]) for i in range(k)))
        else:
            out.append("".join("%d\t%d\t%d\t%d\n" % (i, r[i, 0], r[i, 1], r[i, 2]) for i in range(k)))
        out.append("\n")
    fh.write("".join(out))


def run_program(prog, index, W, num_doped, max_it, extra, opts):
    """The body of main_terminated for one of the three executables."""
    d = DEFAULTS[prog]
    N = opts.N if opts.N else d["N"]
    L = opts.L if opts.L else d["L"]
    g = d["grid"]
    grid = GridSpec(opts.eps_ini if opts.eps_ini is not None else g.eps_ini,
                    opts.eps_delta if opts.eps_delta is not None else g.eps_delta,
                    opts.num_points if opts.num_points else g.num_points,
                    opts.min_frame_err if opts.min_frame_err is not None else g.min_frame_err,
                    opts.max_frames if opts.max_frames else g.max_frames)
    p = E.make_params(opts.dv, opts.dc, L, N)
    # argv quirk kept: main_terminated reads the doped positions starting at argv[4], which is also
    # MAX_IT (BPF:2083-2091) — so with NUM_DOPED > 0 the first doped position equals MAX_IT.
    tail = [max_it] + list(opts.doped_argv)
    doped = [int(x) for x in tail[:num_doped]]
    if len(doped) < num_doped:
        raise SystemExit("NUM_DOPED=%d but only %d position arguments" % (num_doped, len(doped)))
    init_it, is_term = 0, True
    decoder = "full"
    if prog == "sw_lim_iter":
        decoder = "sw"
        init_it = extra if extra else max_it                       # BPW:2101-2102
    elif prog == "bp_traj":
        is_term = bool(extra)
    dist, rank, world = _dist()
    if opts.rng == "glibc" and world > 1:
        # one srandom(seed) stream carried from frame to frame and from point to point (BPF:2057-2131): there is nothing to
        # shard, and ranks replaying it from its start would write correlated points that look like a reference replay
        raise SystemExit("--rng glibc replays the reference's one sequential random() stream: run it as a single process "
                         "(%d ranks here); use --rng philox for multi-GPU jobs" % world)
    # Sharding (module docstring): bp_traj — one replica (INDEX + rank) per rank; else ε points over the ranks when there
    # are at least as many points as ranks (the reference's own cluster model), else the frames of every point.
    shard = getattr(opts, "shard", "auto")
    if prog == "bp_traj":
        shard = "replicas"
    elif shard == "auto":
        shard = "points" if (world > 1 and grid.num_points >= world) else "frames"
    by_points = shard == "points" and world > 1
    replica = index + rank if shard == "replicas" else index
    # the decoders' loop is do { … } while (iter < MaxNumIt) (BPF:1065, BPT:1076): at least one iteration runs
    cap = max(1, max_it)
    sim_obj = Simulator(p, decoder=decoder, W=W, max_it=cap, init_it=init_it,
                        is_term=is_term, doped=doped, batch=opts.batch, rng=opts.rng, seed=opts.seed,
                        rows_cap=opts.rows_cap if prog == "bp_traj" else 0, schedule=getattr(opts, "schedule", "flooding"),
                        shard_frames=shard == "frames", device=getattr(opts, "device", None), index=replica,
                        verbose=rank == 0 and not opts.quiet)
    outdir = opts.outdir
    os.makedirs(outdir, exist_ok=True)
    t0 = time.time()

    def report(eps, point):
        if rank == 0 and not opts.quiet:
            r = point.run
            print("%f %e %e %e   (f=%d, %.1fs)" % (eps, r["users_err"] / p.n / point.f, r["frame_err"] / point.f,
                                                   r["block_err"] / p.L / point.f, point.f, time.time() - t0),
                  flush=True)

    if by_points:
        # Rank r runs points r, r + world, r + 2·world, … back to back — no barrier between points, so a rank whose points
        # stop early (frame_err >= 1000 long before max_frames) is not held up by the others.  Every finished point is
        # appended to the rank's own part file at once (what a killed job leaves behind, like the reference's per-point
        # fopen("a"), BPF:494-497); ONE all-reduce of the table [points][NRUN counters + abort flag] at the end hands the
        # rows to rank 0, which writes the file in grid order and removes the parts.  A point that breaks decodeBP's
        # invariant (BPF:1035-1039) only raises the flag: every rank reaches the all-reduce and all leave together.
        path = os.path.join(outdir, result_filename(prog, p, W, max_it, init_it, index))
        part = path + ".rank%d.part" % rank
        table = torch.zeros((grid.num_points, NRUN + 1), dtype=torch.int64, device=sim_obj.device)
        for sim in range(rank, grid.num_points, world):
            point = sim_obj.run_point(sim, grid.eps(sim), grid.min_frame_err, grid.max_frames, defer_abort=True)
            table[sim, :NRUN] = torch.tensor([point.run[k] for k in RUN_NAMES], dtype=torch.int64, device=sim_obj.device)
            table[sim, NRUN] = int(point.bad)
            with open(part, "a" if sim != rank else "w") as f:
                f.write("%d %s" % (sim, point.row()))
            if point.bad:
                break                                   # the reference's process is gone at this point
        dist.all_reduce(table)
        rows = table.cpu().numpy()
        first_bad = next((s for s in range(grid.num_points) if rows[s, NRUN]), None)
        if rank == 0:
            for sim in range(grid.num_points if first_bad is None else first_bad):
                pt = PointResult(grid.eps(sim), p.n, p.L, rows[sim, :NRUN])
                if pt.f == 0:
                    break                               # a rank stopped at an abort before reaching this point
                write_risultati(path, sim, pt)
                report(pt.eps, pt)
        dist.barrier()
        if os.path.exists(part):
            os.remove(part)
        if first_bad is not None:
            abort_invariant()
        return 0

    for sim in range(grid.num_points):
        eps = grid.eps(sim)
        if prog == "bp_traj":
            # one file per ε point and per replica; this rank's replica writes its frames in order
            path = os.path.join(outdir, traj_filename(p, eps, max_it, is_term, replica))
            with open(path, "w") as fh:
                def on_batch(frame0, used, res):
                    _write_traj_rows(fh, res["rows"], res["counters"], used, cols=getattr(opts, "cols", 4))
                point = sim_obj.run_point(sim, eps, grid.min_frame_err, grid.max_frames, on_batch=on_batch,
                                          defer_abort=world > 1)
        else:
            point = sim_obj.run_point(sim, eps, grid.min_frame_err, grid.max_frames)
            if rank == 0:
                write_risultati(os.path.join(outdir, result_filename(prog, p, W, max_it, init_it, index)), sim, point)
        if prog == "bp_traj" and world > 1:
            # the replicas run independently; an abort in one of them (BPF:1035-1039) ends the job for all, together
            flag = torch.tensor([int(point.bad)], dtype=torch.int64, device=sim_obj.device)
            dist.all_reduce(flag)
            if int(flag.item()):
                abort_invariant()
        report(eps, point)
    return 0


def _parser(prog):
    ap = argparse.ArgumentParser(prog=prog, description=__doc__.split("\n\n")[0])
    ap.add_argument("INDEX", type=int)
    ap.add_argument("W", type=int)
    ap.add_argument("NUM_DOPED", type=int)
    ap.add_argument("MAX_IT", type=int)
    if prog == "sw_lim_iter":
        ap.add_argument("INIT_IT", type=int)
    elif prog == "bp_traj":
        ap.add_argument("IS_TERM", type=int)
    ap.add_argument("doped_argv", nargs="*", type=int, help="further doped positions (see the argv quirk)")
    ap.add_argument("--dv", type=int, default=4)
    ap.add_argument("--dc", type=int, default=8)
    ap.add_argument("--L", type=int, default=0, help="Def_L (default: the source's value)")
    ap.add_argument("--N", type=int, default=0, help="VNs per position = Def_VNsPos = 2*Def_M")
    ap.add_argument("--eps-ini", type=float, default=None)
    ap.add_argument("--eps-delta", type=float, default=None)
    ap.add_argument("--num-points", type=int, default=0)
    ap.add_argument("--min-frame-err", type=int, default=None)
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--batch", type=int, default=2048, help="frames per device batch and rank")
    ap.add_argument("--rng", choices=("philox", "glibc"), default="philox")
    ap.add_argument("--seed", type=int, default=None, help="default: time-based like the reference (BPF:2059-2062)")
    ap.add_argument("--rows-cap", type=int, default=4096, help="bp_traj: max iterations kept per frame")
    ap.add_argument("--schedule", choices=("flooding", "fixpoint"), default="flooding",
                    help="bp_lim_iter with MAX_IT >= 10^6: 'fixpoint' decodes to the same residual without walking "
                         "the flooding iterations (same files; no iteration statistics)")
    ap.add_argument("--shard", choices=("auto", "points", "frames"), default="auto",
                    help="multi-GPU: ε points over the ranks (the reference's cluster model; default when there are at "
                         "least as many points as ranks) or the frames of every point")
    if prog == "bp_traj":
        ap.add_argument("--cols", type=int, choices=(3, 4), default=4,
                        help="4: iter, deg1, recovered, first erased position (BPT:988,1051); 3: without the last "
                             "(the published L50_M2500 files, NB cell 40)")
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--quiet", action="store_true")
    return ap


def main(prog, argv=None):
    opts = _parser(prog).parse_args(argv)
    if opts.seed is None:
        opts.seed = int((time.time() % 1) * 1e6)                      # te.tv_usec (BPF:2061)
    joined = E.init_distributed()                                     # one process per GPU under torch.distributed.run
    extra = getattr(opts, "INIT_IT", None) if prog == "sw_lim_iter" else getattr(opts, "IS_TERM", None)
    rc = run_program(prog, opts.INDEX, opts.W, opts.NUM_DOPED, opts.MAX_IT, extra, opts)
    if joined:
        import torch.distributed as dist
        dist.destroy_process_group()
    return rc


def bp_lim_iter(argv=None):
    return main("bp_lim_iter", argv)


def sw_lim_iter(argv=None):
    return main("sw_lim_iter", argv)


def bp_traj(argv=None):
    return main("bp_traj", argv)


# ------------------------------------------------------------------------------------------------
# streaming mode: the CIRCULAR build of the reference (`sw INDEX W NUM_DOPED DOPED_POSITIONS…`, BPF:1934-2054)
# ------------------------------------------------------------------------------------------------
STREAM_HEADER = ("p BER BLER BER_EXP BLER_EXP bit_err bit_gen block_err block_gen bit_err_exp bit_gen_exp "
                 "block_err_exp block_gen_exp\n")


def stream_filename(p, num_doped, W, index):
    """results_circular, BPF:535."""
    return "SC_LDPC_%d_%d_L%d_M%d_DOP%d_BP_Stream_SW%d_Random_BLER_%d.dat" % (p.dv, p.dc, p.L, p.cns_pos, num_doped, W, index)


def stream_row(eps, c):
    """results_circular's row (BPF:547-560) from the eight counters (order of engine.STREAM_COUNTERS)."""
    ne, be, ee, bee, gb, gbl, gbe, gble = (int(x) for x in c[:8])
    return "%f %e %e %e %e %d %d %d %d %d %d %d %d\n" % (eps, ne / gb, be / gbl, ee / gbe, bee / gble,
                                                          ne, gb, be, gbl, ee, gbe, bee, gble)


def run_streaming(index, W, doped, opts):
    """main_streaming: per ε point, decode positions until num_blocks_err_exp >= max_blocks_err or
    num_blocks_generated_exp >= max_blocks (Def_MaxNumberBlocksError / Def_MaxNumberBlocksSim, BPF:41-42, 2033).
    The reference runs ONE stream; here `--streams` independent streams (× ranks) advance in lock step, `--chunk`
    positions per launch, and their counters are summed — the stop rule is applied to the sums after every chunk."""
    g0 = DEFAULTS["bp_lim_iter"]["grid"]
    N = opts.N if opts.N else 1000
    L = opts.L if opts.L else 50
    p = E.make_params(opts.dv, opts.dc, L, N)
    grid = GridSpec(opts.eps_ini if opts.eps_ini is not None else g0.eps_ini,
                    opts.eps_delta if opts.eps_delta is not None else g0.eps_delta,
                    opts.num_points if opts.num_points else g0.num_points, 0, 0)
    dist, rank, world = _dist()
    device = E.local_device()
    os.makedirs(opts.outdir, exist_ok=True)
    path = os.path.join(opts.outdir, stream_filename(p, len(doped), W, index))
    if getattr(opts, "rng", "philox") == "glibc":
        # The reference's own experiment, row for row: ONE stream, ONE srandom(seed), the points back to back with random()
        # carried over, every point stopped at the very position at which main_streaming stops (BPF:2033).
        if world > 1:
            raise SystemExit("--rng glibc replays the reference's one sequential random() stream: run it as a single process")
        run = E.GlibcStreamRun(p, opts.seed, W, doped, device=device)

        def tripped(c):
            return c[3] >= opts.max_blocks_err or c[7] >= opts.max_blocks

        for sim in range(grid.num_points):
            eps = grid.eps(sim)
            run.new_point(eps)
            while True:
                rows = run.run(opts.chunk, stop=tripped)
                if tripped(rows[-1, 2:]):
                    break
            tot = rows[-1, 2:]
            with open(path, "w" if sim == 0 else "a") as f:
                if sim == 0:
                    f.write(STREAM_HEADER)
                f.write(stream_row(eps, tot))
            if not opts.quiet:
                print("%f %e %e %e %e" % (eps, tot[0] / tot[4], tot[1] / tot[5], tot[2] / tot[6], tot[3] / tot[7]), flush=True)
        return 0
    for sim in range(grid.num_points):
        eps = grid.eps(sim)
        st = E.Streams(p, opts.streams, opts.seed, eps, W, doped,
                       stream0=(sim * world + rank) * opts.streams, device=device)
        while True:
            cnt, _ = st.run(opts.chunk)
            # (column 9 = positions generated; negative = the stream was marked unusable, include/scldpc.h)
            tot = torch.cat([cnt[:, :8].sum(dim=0), (cnt[:, 9] < 0).sum().reshape(1)])
            if dist is not None:
                dist.all_reduce(tot)                    # the only exchange: nine int64 per chunk
            tot = tot.cpu().numpy()
            if tot[8] > 0:                              # every rank sees the same sum and leaves together
                raise SystemExit("sw: %d stream(s) marked unusable by the generation kernel (a ranking bucket overflowed)" % tot[8])
            tot = tot[:8]
            if tot[3] >= opts.max_blocks_err or tot[7] >= opts.max_blocks:
                break
        if rank == 0:
            with open(path, "w" if sim == 0 else "a") as f:
                if sim == 0:
                    f.write(STREAM_HEADER)
                f.write(stream_row(eps, tot))
            if not opts.quiet:
                print("%f %e %e %e %e" % (eps, tot[0] / tot[4], tot[1] / tot[5], tot[2] / tot[6], tot[3] / tot[7]), flush=True)
    return 0


def streaming(argv=None):
    ap = argparse.ArgumentParser(prog="sw", description="doped SC-LDPC streaming window decoder (CIRCULAR build)")
    ap.add_argument("INDEX", type=int)
    ap.add_argument("W", type=int)
    ap.add_argument("NUM_DOPED", type=int)
    ap.add_argument("DOPED", nargs="*", type=int)
    ap.add_argument("--dv", type=int, default=4)
    ap.add_argument("--dc", type=int, default=8)
    ap.add_argument("--L", type=int, default=0, help="circular buffer length Def_L")
    ap.add_argument("--N", type=int, default=0)
    ap.add_argument("--eps-ini", type=float, default=None)
    ap.add_argument("--eps-delta", type=float, default=None)
    ap.add_argument("--num-points", type=int, default=0)
    ap.add_argument("--max-blocks-err", type=int, default=1000, help="Def_MaxNumberBlocksError (BPF:41)")
    ap.add_argument("--max-blocks", type=int, default=1000000, help="Def_MaxNumberBlocksSim (BPF:42)")
    ap.add_argument("--streams", type=int, default=512, help="independent streams per rank (--rng philox)")
    ap.add_argument("--rng", choices=("philox", "glibc"), default="philox",
                    help="glibc: the reference's own experiment — one stream drawn from srandom(--seed) exactly as "
                         "main_streaming draws it (BPF:1942-1945), reproduced row for row; single process")
    ap.add_argument("--chunk", type=int, default=64, help="positions per stream and launch")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--quiet", action="store_true")
    opts = ap.parse_args(argv)
    if opts.seed is None:
        opts.seed = int((time.time() % 1) * 1e6)
    if len(opts.DOPED) < opts.NUM_DOPED:
        raise SystemExit("NUM_DOPED=%d but only %d positions given" % (opts.NUM_DOPED, len(opts.DOPED)))
    joined = E.init_distributed()
    rc = run_streaming(opts.INDEX, opts.W, opts.DOPED[:opts.NUM_DOPED], opts)
    if joined:
        import torch.distributed as dist
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "sw":
        sys.exit(streaming(sys.argv[2:]))
    if len(sys.argv) < 2 or sys.argv[1] not in DEFAULTS:
        raise SystemExit("usage: python -m fl_scaling_sc_ldpc_amd.bp_decoding {bp_lim_iter|sw_lim_iter|bp_traj|sw} ARGS…")
    sys.exit(main(sys.argv[1], sys.argv[2:]))
