#!/usr/bin/env python3
"""Benchmarks of the hot path on N GPUs of one node.  Default = the headline of BASELINE.json:

    Monte-Carlo codeword trials/sec, (4,8) SC-LDPC, L=50, N=1000, eps=0.48, full BP to the fixpoint   (config C2)

A *step* = one pass of the whole hot path over one batch of B trials per GPU, everything on the device:
    sample (fresh random code + channel per trial, Philox-keyed)  ->  decodeBP  ->  plr_computation
i.e. what one frame of the reference's main_terminated loop does (BPF:2117-2144), B times.  Nothing is cached between
steps: every step draws new trial indices.  Trials are independent, so N GPUs take disjoint trial ranges with no
data-path collective ("weak" scaling: B per GPU is fixed); the only exchange is the final sum of the run counters (one
RCCL all-reduce of 9 int64, outside the timed region except for the closing barrier).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config C2|C3|C4|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     HBM roofline of the step's memory-bound kernel, the decoder: `achieved` = its share of the algorithmic bytes
               (B_alg = 16*E + n/8 per trial = both adjacency directions written once by the sampler (8E) and read once by
               the decoder (8E) as int32, plus the channel bits; the decoder's share is 8E + n/8) per launch / its mean
               duration measured here with HIP events on its stream; `peak` 8 TB/s; `traffic` = the HBM bytes it really
               moves per launch (rocprofv3 PMC passes kept under profiles/, 128 B per L2 read request as calibrated by
               tools/calib, + WRITE_SIZE), `traffic_raw` = FETCH_SIZE + WRITE_SIZE as the counters print them.
               `step` = BASELINE.md §3.4's whole-step figure B_alg x value / peak; `kernels` = the same accounting for
               the sampler (share 8E + n/8, written).  `--flooding` adds the literal-flooding figure 8*E*(1 + sum I).
  cpu_baseline the REAL reference (oracle/_ref/ref_*: the reference's own C source compiled by oracle/Makefile, kind
               "reference") timed on this box's host cores on a bounded sample of the same workload, one single-threaded
               process per core as the reference is run on clusters (NB cell 35:21); kind "port" where only the oracle's
               restatement can run (config C3: the reference path is Python).  Rank 0, N=1 only.  A reported baseline.
Other configs (`--config`): the same line for BASELINE.json's configs 3-5 with SURVEY.md §8d's formula for each.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DV, DC = 4, 8
HBM_PEAK_GBS = 8000.0
SEED = 20261004


# ------------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1): the reference binaries of oracle/_ref, one single-threaded process per core
# ------------------------------------------------------------------------------------------------------------------
def _run_procs(cmds):
    t0 = time.time()
    procs = [subprocess.Popen(c, stdout=subprocess.DEVNULL) for c in cmds]
    ok = all(p.wait() == 0 for p in procs)
    return ok, time.time() - t0


def _cores():
    return max(1, min(os.cpu_count() or 1, 16))


def cpu_baseline_c2(eps, per_core=16):
    """generate_code + channel_doped + decodeBP per trial (BPF:2117-2144), ref_bpf_M500_L50."""
    cores, ref = _cores(), os.path.join(ROOT, "oracle", "_ref", "ref_bpf_M500_L50")
    if os.path.exists(ref):
        # per-process wall times give the spread over cores
        t0 = time.time()
        procs = [(subprocess.Popen([ref, str(per_core), str(900001 + 1000 * k), repr(eps), "1000000", "0", "0", "1", "0"],
                                   stdout=subprocess.DEVNULL), k) for k in range(cores)]
        ends = {}
        while len(ends) < cores:
            for p, k in procs:
                if k not in ends and p.poll() is not None:
                    ends[k] = time.time() - t0
            time.sleep(0.02)
        if all(p.returncode == 0 for p, _ in procs):
            dt = max(ends.values())
            rates = [per_core / ends[k] for k in range(cores)]
            mean = sum(rates) / cores
            sd = (sum((r - mean) ** 2 for r in rates) / max(1, cores - 1)) ** 0.5
            return {"value": cores * per_core / dt, "unit": "trials/s", "cores": cores, "kind": "reference",
                    "per_core": mean, "per_core_ci95": 1.96 * sd / cores ** 0.5,
                    "sample": f"{cores} single-threaded processes x {per_core} trials (srandom seeds 900001+1000k), "
                              f"oracle/_ref/ref_bpf_M500_L50, {dt:.1f} s wall"}
    code = ("import sys,time; sys.path.insert(0,%r); from oracle import oracle as O; p=O.Params(4,8,50,500,1000); "
            "[O.trial(p, int(sys.argv[1])+t, %r, decoder=0) for t in range(%d)]" % (ROOT, eps, per_core))
    ok, dt = _run_procs([[sys.executable, "-c", code, str(900001 + 1000 * k)] for k in range(cores)])
    return {"value": cores * per_core / dt, "unit": "trials/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x {per_core} trials, oracle literal flooding decoder, {dt:.1f} s wall"}


def cpu_baseline_c4(eps, W, it, per_core=2):
    cores, ref = _cores(), os.path.join(ROOT, "oracle", "_ref", "ref_bpw_M1000_L100")
    if not os.path.exists(ref):
        return None
    ok, dt = _run_procs([[ref, str(per_core), str(700001 + 1000 * k), repr(eps), str(it), str(it), str(W), "1", "0"]
                         for k in range(cores)])
    return {"value": cores * per_core / dt, "unit": "trials/s", "cores": cores, "kind": "reference",
            "sample": f"{cores} single-threaded processes x {per_core} frames, oracle/_ref/ref_bpw_M1000_L100 "
                      f"(decodeBP_SW, BPW:628-912), {dt:.1f} s wall"} if ok else None


def cpu_baseline_c5(eps, W, doped, positions=12):
    cores, ref = _cores(), os.path.join(ROOT, "oracle", "_ref", "ref_stream_M2500_L50")
    if not os.path.exists(ref):
        return None
    ok, dt = _run_procs([[ref, str(positions), str(500001 + 1000 * k), repr(eps), str(W), "0", str(len(doped))] +
                         [str(d) for d in doped] for k in range(cores)])
    return {"value": cores * positions / dt, "unit": "positions/s", "cores": cores, "kind": "reference",
            "sample": f"{cores} single-threaded processes x {positions} decoded positions (+25 generated ahead), "
                      f"oracle/_ref/ref_stream_M2500_L50 (main_streaming, BPF:1934-2054), {dt:.1f} s wall"} if ok else None


def cpu_baseline_c3(eps, per_core=4):
    """The reference path is Python (PD:705-789, ~70 s per trial at this size) and does not travel: the oracle's own
    restatement (numpy sampler on the reference's draws + the O(steps log n) random-pick twin) is timed instead."""
    cores = _cores()
    code = ("import sys; sys.path.insert(0,%r); import numpy as np; from oracle import pd_oracle as P\n"
            "rs=np.random.RandomState(int(sys.argv[1]))\n"
            "for t in range(%d):\n"
            "    tr=P.gen_slots(rs,4,8,50,10000); m=P.gen_erasures(rs,%r,4,8,50,10000)\n"
            "    P.random_pick_trial_philox_fast(tr,m,4,8,50,10000,%r,False,int(sys.argv[1]),t)\n" % (ROOT, per_core, eps, eps))
    ok, dt = _run_procs([[sys.executable, "-c", code, str(300001 + k)] for k in range(cores)])
    return {"value": cores * per_core / dt, "unit": "trials/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x {per_core} trials, oracle/pd_oracle.py sampler + orc_random_pick_philox "
                      f"(the reference's simulate_peeling_decoder_ldpc takes ~70 s per trial at this size), {dt:.1f} s wall"} if ok else None


# ------------------------------------------------------------------------------------------------------------------
def measured_traffic(kernel, batch, config):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/traffic.json, written by
    tools/summarize_prof.py from separate rocprofv3 --pmc passes), if it was taken on this config and batch."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(config, {})
        if t.get("batch") == batch and kernel in t.get("kernels", {}):
            return t["kernels"][kernel]
    except Exception:
        pass
    return None


def _traffic_fields(tr):
    return {"traffic": tr["hbm_bytes"] if tr else None, "traffic_raw": tr["hbm_bytes_raw"] if tr else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0,
                    help="timed steps; 0 = the config's default (C2: 100 = 3 s of GPU time, long enough for an outside observer of the GPU's "
                         "activity and clocks; C3: 10, C4: 50, C5: 50)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="trials (streams for C5) per GPU per step; 0 = the config's default")
    ap.add_argument("--config", default="C2", choices=("C2", "C3", "C4", "C5"))
    ap.add_argument("--eps", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flooding", action="store_true",
                    help="C2: decode one flooding iteration per barrier round (scldpc_full_bp_device_cn16; with --gen1 the "
                         "16-bit-CN-word kernel): reports the iteration count and the literal-flooding figure")
    ap.add_argument("--traj", action="store_true",
                    help="C2: the trajectory build's decoder (bp_traj): one flooding iteration per round AND the per-iteration rows "
                         "(deg_1_iter, recovered, first erased position) written for every trial")
    ap.add_argument("--overlap", action="store_true",
                    help="(default) two streams, two buffers: the sampler of step k+1 fills the tail of the decoder of step k")
    ap.add_argument("--no-overlap", action="store_true", help="one stream: sample, then decode, then accumulate")
    ap.add_argument("--adj32", action="store_true", help="C2: int32 global-id adjacency (first-generation kernels)")
    ap.add_argument("--gen1", action="store_true",
                    help="C2: first-generation kernels: sampler that ranks every key + fixpoint decoder on 16-bit CN words "
                         "(default: sampler_v2 + the 4-bits-per-CN decoder, which also reads the CN -> VN table)")
    a = ap.parse_args()
    if a.steps <= 0:
        a.steps = {"C2": 100, "C3": 10, "C4": 50, "C5": 50}[a.config]

    # `python bench.py --gpus N` launched plainly (no launcher, no WORLD_SIZE): start the N ranks ourselves, as fresh child
    # processes, BEFORE anything here touches the GPU (never an exec of this process), relay rank 0's one JSON line and the
    # job's exit code.  Under a launcher (WORLD_SIZE set) a --gpus that disagrees with it is an error, not a note.
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...) "
                         f"or run `python bench.py --gpus {a.gpus}` without a launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # One rank per GPU over RCCL.  Rehearsal on a box with fewer GPUs than ranks (tests/test_gpu_bench_ranks.py: two ranks
    # on the one GPU of a test box): SCLDPC_BENCH_BACKEND=gloo, ranks wrap around the devices; never used for a reported line.
    backend = os.environ.get("SCLDPC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > 1 and local >= ndev:
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible); RCCL needs one device per rank")
    local_dev = local % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    global REHEARSAL, RCCL_RANKS
    if backend != "nccl" or world > ndev:
        REHEARSAL = f"backend={backend}, {world} ranks on {ndev} device(s)"
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # one real all-reduce over the job's backend before anything is timed: every rank contributes 1, so the sum is the
        # number of ranks that are really connected (RCCL over xGMI when backend == "nccl"); reported as `rccl_ranks`
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        if int(ones.item()) != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py: all-reduce saw {int(ones.item())} ranks, WORLD_SIZE={world}")
        RCCL_RANKS = int(ones.item()) if backend == "nccl" else 0
    else:
        RCCL_RANKS = 1

    from fl_scaling_sc_ldpc_amd import engine as E

    def fence():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def finish(dt_local):
        t_max = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        if dist:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item())

    {"C2": run_c2, "C3": run_c3, "C4": run_c4, "C5": run_c5}[a.config](a, E, dev, rank, world, dist, fence, finish)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


REHEARSAL = None
RCCL_RANKS = 1


def spawn_ranks(n, argv):
    """Start `n` ranks of this script under torch.distributed.run as a CHILD process (rendezvous on 127.0.0.1, a free
    port), pass its stdout/stderr through and return its exit code.  The parent has not touched the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def emit(out):
    out["rccl_ranks"] = RCCL_RANKS                  # ranks counted by a real all-reduce over RCCL (0: rehearsal backend)
    if REHEARSAL:
        out["config"]["rehearsal"] = REHEARSAL      # not a reportable line: ranks shared devices / no RCCL
    print(json.dumps(out), flush=True)


def _events(n, k):
    return [tuple(torch.cuda.Event(enable_timing=True) for _ in range(k)) for _ in range(n)]


# ------------------------------------------------------------------------------------------------------------------
# C2 (headline): (4,8) L=50 N=1000 eps=0.48 full BP, unlimited iterations
# ------------------------------------------------------------------------------------------------------------------
def run_c2(a, E, dev, rank, world, dist, fence, finish):
    L_CHAIN, N_POS = 50, 1000
    EPS = 0.48 if a.eps is None else a.eps
    p = E.make_params(DV, DC, L_CHAIN, N_POS)
    B = a.batch or 65536                                # 36.6 rounds of the 1792 resident decoder workgroups: the tail of long trials weighs less (+1.5 % over 32768)
    TRAJ_ROWS = 640                                     # iterations kept per trial in --traj mode (the longest run at this size: ~520)
    if a.traj:
        a.flooding = True                               # the rows come from the iteration-exact decoder
        B = a.batch or 16384                            # 16384 x 640 x 12 B of rows per step
    workload = f"({DV},{DC}) SC-LDPC L={L_CHAIN} N={N_POS} eps={EPS} full BP unlimited iterations"
    gen2 = not (a.gen1 or a.adj32)
    if gen2 and not E.cn16_supported(p):
        raise SystemExit("bench.py: the second-generation kernels do not take this ensemble")
    # Two streams by default: two (tables, channel, counters) buffers, the sampler of step k+1 beside the decoder of step k.
    # The decoder's trials differ 10x in length (a successful decode walks the whole chain), so its launch ends in a
    # tail of long trials on a mostly empty chip: the next step's sampler fills it (+3-4 %).
    nbuf = 1 if a.no_overlap else 2
    d_adj = [torch.empty((B, p.n, p.dv), dtype=torch.int32 if a.adj32 else torch.int16, device=dev) for _ in range(nbuf)]
    d_cn = [torch.empty((B, p.nk, p.dc), dtype=torch.int16, device=dev) if gen2 else None for _ in range(nbuf)]
    d_ch = [torch.empty((B, p.nw), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    d_cnt = [torch.empty((B, E.NCOUNTERS), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    run = E.new_run(dev)
    s_dec = torch.cuda.current_stream(dev)
    s_samp = s_dec if nbuf == 1 else torch.cuda.Stream(dev)
    sampled = [torch.cuda.Event() for _ in range(nbuf)]
    decoded = [torch.cuda.Event() for _ in range(nbuf)]
    ev = _events(a.steps, 4)

    def step(k, timed_idx=None):
        trial0 = (k * world + rank) * B                 # global trial index: step-major, then rank — disjoint ranges
        b = k % nbuf
        e = ev[timed_idx] if timed_idx is not None else None
        with torch.cuda.stream(s_samp):
            s_samp.wait_event(decoded[b])               # buffer b is free once its previous decode has finished
            if e:
                e[0].record(s_samp)
            if gen2:
                E.sample_philox_cn16(p, SEED, trial0, B, EPS, out=(d_adj[b], d_cn[b], d_ch[b]))
            else:
                E.sample_philox(p, SEED, trial0, B, EPS, out=(d_adj[b], d_ch[b]))
            if e:
                e[1].record(s_samp)
            sampled[b].record(s_samp)
        with torch.cuda.stream(s_dec):
            s_dec.wait_event(sampled[b])
            if e:
                e[2].record(s_dec)
            if a.traj and gen2:
                E.full_bp_cn16(p, d_adj[b], d_cn[b], d_ch[b], counters=d_cnt[b], rows_cap=TRAJ_ROWS)
            elif a.flooding and gen2:
                E.full_bp_cn16(p, d_adj[b], d_cn[b], d_ch[b], counters=d_cnt[b])
            elif a.flooding:
                E.full_bp(p, d_adj[b], d_ch[b], counters=d_cnt[b])
            elif gen2:
                E.full_bp_fixpoint_cn16(p, d_adj[b], d_cn[b], d_ch[b], counters=d_cnt[b])
            else:
                E.full_bp_fixpoint(p, d_adj[b], d_ch[b], counters=d_cnt[b])
            if e:
                e[3].record(s_dec)
            E.accumulate_run(d_cnt[b], run, 0)
            decoded[b].record(s_dec)

    for k in range(a.warmup):
        step(k)
    fence()
    run.zero_()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k, k)
    fence()
    dt = finish(time.perf_counter() - t0)
    if dist:
        dist.all_reduce(run, op=dist.ReduceOp.SUM)      # the optional RCCL counter reduce
    r = dict(zip(E.RUN_NAMES, run.cpu().tolist()))
    total_trials = a.steps * B * world
    assert r["frames"] == total_trials, (r["frames"], total_trials)
    if rank != 0:
        return
    ms_sample = sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps
    ms_bp = sum(e[2].elapsed_time(e[3]) for e in ev) / a.steps
    E_edges = p.n * p.dv
    b_alg = 16 * E_edges + p.n // 8
    share = 8 * E_edges + p.n // 8                      # each kernel's half: one direction of both tables + the channel bits
    # what one kernel really writes (sampler) or has to read (decoder) per trial: the tables as they lie in HBM
    table_bytes = (d_adj[0][0].numel() * d_adj[0].element_size() + (d_cn[0][0].numel() * 2 if gen2 else 0) + 4 * p.nw)
    value = total_trials / dt
    dec_name = "full_bp_small_kernel" if gen2 else "full_bp_kernel" if a.flooding else "full_bp_fixpoint_kernel"
    samp_name = "sample_philox_v2_kernel" if gen2 else "sample_philox_kernel"
    tr_dec, tr_samp = measured_traffic(dec_name, B, "C2"), measured_traffic(samp_name, B, "C2")
    if a.flooding:
        tr_dec = None                                   # the PMC passes under profiles/ are of the fixpoint variant
    ach_dec = share * B / (ms_bp * 1e-3) / 1e9
    ach_samp = share * B / (ms_sample * 1e-3) / 1e9
    step_ach = b_alg * value / world / 1e9
    roof = {"bound": "hbm", "kernel": dec_name, "achieved": ach_dec, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach_dec / HBM_PEAK_GBS, "traffic": tr_dec["hbm_bytes"] if tr_dec else None,
            "traffic_raw": tr_dec["hbm_bytes_raw"] if tr_dec else None,
            "alg_bytes_per_trial": share, "table_bytes_per_trial": table_bytes, "trials_per_launch": B, "ms_per_launch": ms_bp,
            "note": "decoder share of B_alg = 8E + n/8 (reads both tables once + the channel bits); `traffic` counts 128 B "
                    "per L2 read request (tools/calib: one request per random 8- or 16-byte gather)",
            "step": {"achieved": step_ach, "frac": step_ach / HBM_PEAK_GBS, "alg_bytes_per_trial": b_alg,
                     "definition": "B_alg x trials/s per GPU / 8 TB/s (BASELINE.md 3.4)"},
            "kernels": {samp_name: {"achieved": ach_samp, "frac": ach_samp / HBM_PEAK_GBS, "ms_per_launch": ms_sample,
                                    "alg_bytes_per_trial": share,
                                    "traffic": tr_samp["hbm_bytes"] if tr_samp else None,
                                    "traffic_raw": tr_samp["hbm_bytes_raw"] if tr_samp else None,
                                    "bound": "LDS pipe (random atomics / scatter) + VALU (Philox), not HBM"},
                        dec_name: {"achieved": ach_dec, "frac": ach_dec / HBM_PEAK_GBS, "ms_per_launch": ms_bp,
                                   "alg_bytes_per_trial": share,
                                   "traffic": tr_dec["hbm_bytes"] if tr_dec else None,
                                   "traffic_raw": tr_dec["hbm_bytes_raw"] if tr_dec else None,
                                   "read_requests_per_s": (tr_dec["rdreq"] / (ms_bp * 1e-3)) if tr_dec and tr_dec.get("rdreq") else None,
                                   "bound": "HBM at line granularity (one 128-B line per 8- or 16-byte row gather)" if gen2 else
                                            "latency of ~230 dependent levels"}}}
    if a.flooding:
        # literal flooding moves 8*E*(1 + I_t) bytes for a trial of I_t iterations (secondary figure)
        sum_it = r["iterations"]
        lit = 8.0 * E_edges * (total_trials + sum_it) / dt / world / 1e9
        roof["literal_flooding"] = {"bytes": "8*E*(1 + sum I)", "sum_iterations": sum_it, "equivalent_GBs": lit,
                                    "x_peak": lit / HBM_PEAK_GBS}
    out = {
        "metric": "MC codeword trials/sec, (4,8) SC-LDPC L=50 N=1000 eps=0.48 full BP",
        "value": value, "unit": "trials/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": workload, "trials_per_gpu_per_step": B,
                   "step": "device sample (code+channel) -> decodeBP -> plr_computation",
                   "decoder": ("bp_traj: flooding, one barrier round per iteration, per-iteration rows written (BPT:988,1051)" if a.traj else
                               "flooding, one barrier round per iteration (iteration counts and caps as the reference's)" if a.flooding else
                               "fixpoint of unlimited flooding by chain-following peeling (every output of decodeBP "
                               "except the iteration count; equality with the flooding kernel on every trial is a test)"),
                   "kernels": "sampler_v2 + full_bp_small (4-bit CN counts, CN->VN table)" if gen2 else "first generation",
                   "rng": "philox4x32-10 keyed by (seed, trial)",
                   "adjacency": "int32 global ids" if a.adj32 else "uint16 VN->CN (position-local) + uint16 CN->VN" if gen2
                                else "uint16 position-local ids", "parallelism": f"trial-sharded x{world}",
                   "streams": "sampler(k+1) || decoder(k), double-buffered" if nbuf == 2 else "single stream"},
        "roofline": roof,
        "kernels_ms": {"sample_philox": ms_sample, "full_bp": ms_bp},
        "decode_only_trials_per_s_per_gpu": B / (ms_bp * 1e-3),
        "results": {"FER": r["frame_err"] / r["frames"], "BLER": r["block_err"] / p.L / r["frames"],
                    "BER": r["users_err"] / p.n / r["frames"], "FER_exp": r["frame_err_exp"] / r["frames"],
                    ("mean_iterations" if a.flooding else "mean_barrier_rounds"): r["iterations"] / r["frames"]},
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c2(EPS)
    emit(out)


# ------------------------------------------------------------------------------------------------------------------
# C3: (4,8) L=50 N=10000 random-pick peeling with degree-1-CN trajectory moments (PD:705-789)
# ------------------------------------------------------------------------------------------------------------------
def run_c3(a, E, dev, rank, world, dist, fence, finish):
    L_CHAIN, N_POS = 50, 10000
    EPS = 0.48 if a.eps is None else a.eps
    p = E.make_params(DV, DC, L_CHAIN, N_POS)
    B = a.batch or 16384                                # one wave per trial, 32 waves per CU x 256 CUs = 8192 in flight: two
                                                        # rounds, the second fills the CUs the early finishers leave
    steps_pd = int(N_POS * L_CHAIN * (EPS + 0.1))       # PD:721 (non-terminated)
    total_size = p.cns_pos * L_CHAIN
    # two (table, channel) buffers of 66 GB: the sampler of step k+1 runs on a second stream beside the picks of step k, whose
    # launch ends in a tail of long trials on a mostly empty chip
    nbuf = 1 if a.no_overlap else 2
    d_adj = [torch.empty((B, p.n, p.dv), dtype=torch.int16, device=dev) for _ in range(nbuf)]
    d_ch = [torch.empty((B, p.nw), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    mom = torch.zeros((3, steps_pd + 1), dtype=torch.int64, device=dev)
    s_dec = torch.cuda.current_stream(dev)
    s_samp = s_dec if nbuf == 1 else torch.cuda.Stream(dev)
    sampled = [torch.cuda.Event() for _ in range(nbuf)]
    decoded = [torch.cuda.Event() for _ in range(nbuf)]
    ev = _events(a.steps, 4)

    def step(k, e=None):
        trial0 = (k * world + rank) * B
        b = k % nbuf
        with torch.cuda.stream(s_samp):
            s_samp.wait_event(decoded[b])
            if e:
                e[0].record(s_samp)
            E.sample_philox(p, SEED, trial0, B, EPS, out=(d_adj[b], d_ch[b]))
            if e:
                e[1].record(s_samp)
            sampled[b].record(s_samp)
        with torch.cuda.stream(s_dec):
            s_dec.wait_event(sampled[b])
            if e:
                e[2].record(s_dec)
            # the trajectories as rows (4 B per step and trial: 9.5 GB per batch) and one reduction pass: 7 % faster than three
            # global atomics per step inside the chain of picks (tools/ab_c3.py)
            r = E.peel_pick(p, d_adj[b], d_ch[b], total_size, steps_pd, seed=SEED, trial0=trial0, want_r1=True)
            E.r1_moments(r["r1"], mom)
            if e:
                e[3].record(s_dec)
            decoded[b].record(s_dec)

    for k in range(a.warmup):
        step(k)
    fence()
    mom.zero_()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k, ev[k])
    fence()
    dt = finish(time.perf_counter() - t0)
    if dist:
        dist.all_reduce(mom)                            # the moment vectors: 3 x 290 001 int64 = 7 MB per reduce
    total = a.steps * B * world
    m = mom.cpu().numpy()
    if rank != 0:
        return
    ms_s = sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps
    ms_p = sum(e[2].elapsed_time(e[3]) for e in ev) / a.steps
    b_alg = 16 * p.n * p.dv + p.n // 8
    value = total / dt
    ach = b_alg * B / (ms_p * 1e-3) / 1e9
    out = {"metric": "random-pick peeling trials/sec, (4,8) SC-LDPC L=50 N=10000, degree-1 CN trajectory moments",
           "value": value, "unit": "trials/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic",
           "config": {"workload": f"({DV},{DC}) SC-LDPC L={L_CHAIN} N={N_POS} eps={EPS} random-pick peeling, "
                                  f"{steps_pd} steps per trial, non-terminated, moments of the degree-1 trajectories",
                      "trials_per_gpu_per_step": B, "step": "device sample -> peel_pick (r1 rows) -> r1_moments",
                      "parallelism": f"trial-sharded x{world}",
                      "streams": "sampler(k+1) || picks(k), double-buffered" if nbuf == 2 else "single stream"},
           "roofline": {"bound": "hbm", "kernel": "peel_pick_multi_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, **_traffic_fields(measured_traffic("peel_pick_multi_kernel", B, "C3")),
                        "alg_bytes_per_trial": b_alg, "moments_bytes_per_batch": 24 * (steps_pd + 1), "ms_per_launch": ms_p,
                        "note": "a chain of 290 000 dependent picks per trial, two trials per wave: latency-bound (SURVEY.md §8d says so); the "
                                "fraction of the HBM roofline is reported, not expected to be high"},
           "kernels_ms": {"sample_philox_big": ms_s, "peel_pick": ms_p},
           "results": {"mean_r1_at_step_0": float(m[1][0]) / max(1.0, float(m[0][0])), "trials_in_moments": int(m[0][0])}}
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c3(EPS)
    emit(out)


# ------------------------------------------------------------------------------------------------------------------
# C4: (4,8) L=100 N=2000 square sliding-window BP, W=10, 20 iterations per window (BPW:628-912)
# ------------------------------------------------------------------------------------------------------------------
def run_c4(a, E, dev, rank, world, dist, fence, finish):
    L_CHAIN, N_POS, W, IT = 100, 2000, 10, 20
    EPS = 0.47 if a.eps is None else a.eps
    p = E.make_params(DV, DC, L_CHAIN, N_POS)
    B = a.batch or 8192
    gen2 = not a.gen1
    if gen2 and not E.sock16_supported(p):
        raise SystemExit("bench.py: the second-generation sampler does not take this ensemble")
    # two sets of buffers, two streams: the sampler of step k+1 beside the decoder of step k (as C2)
    nbuf = 1 if a.no_overlap else 2
    d_adj = [torch.empty((B, p.n, p.dv), dtype=torch.int16, device=dev) for _ in range(nbuf)]
    d_ch = [torch.empty((B, p.nw), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    d_cnt = [torch.empty((B, E.NCOUNTERS), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    d_cs = [torch.empty((B, p.nk, p.dc), dtype=torch.int16, device=dev) if gen2 else None for _ in range(nbuf)]
    run = E.new_run(dev)
    s_dec = torch.cuda.current_stream(dev)
    s_samp = s_dec if nbuf == 1 else torch.cuda.Stream(dev)
    sampled = [torch.cuda.Event() for _ in range(nbuf)]
    decoded = [torch.cuda.Event() for _ in range(nbuf)]
    ev = _events(a.steps, 4)

    def step(k, e=None):
        trial0 = (k * world + rank) * B
        b = k % nbuf
        with torch.cuda.stream(s_samp):
            s_samp.wait_event(decoded[b])
            if e:
                e[0].record(s_samp)
            if gen2:                                    # the sampler emits the CN -> socket table with the code
                E.sample_philox_sock16(p, SEED, trial0, B, EPS, out=(d_adj[b], d_cs[b], d_ch[b]))
            else:                                       # first generation: sampler.hip, then a cn_sockets pass inside sw_bp
                E.sample_philox(p, SEED, trial0, B, EPS, out=(d_adj[b], d_ch[b]))
            if e:
                e[1].record(s_samp)
            sampled[b].record(s_samp)
        with torch.cuda.stream(s_dec):
            s_dec.wait_event(sampled[b])
            if e:
                e[2].record(s_dec)
            E.sw_bp(p, d_adj[b], d_ch[b], W, IT, 0, counters=d_cnt[b], d_cn_sock=d_cs[b])
            if e:
                e[3].record(s_dec)
            E.accumulate_run(d_cnt[b], run, 0)
            decoded[b].record(s_dec)

    for k in range(a.warmup):
        step(k)
    fence()
    run.zero_()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k, ev[k])
    fence()
    dt = finish(time.perf_counter() - t0)
    if dist:
        dist.all_reduce(run)
    r = dict(zip(E.RUN_NAMES, run.cpu().tolist()))
    total = a.steps * B * world
    assert r["frames"] == total
    if rank != 0:
        return
    ms_s = sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps
    ms_w = sum(e[2].elapsed_time(e[3]) for e in ev) / a.steps
    e_w = W * N_POS * DV                                # edges of a full window (the last W-1 windows are shorter)
    lit = 8.0 * e_w * r["iterations"] / world / a.steps / (ms_w * 1e-3) / 1e9
    b_alg = 16 * p.n * p.dv + p.n // 8
    share = 8 * p.n * p.dv + p.n // 8
    ach = share * B / (ms_w * 1e-3) / 1e9
    value = total / dt
    out = {"metric": "window-decoding trials/sec, (4,8) SC-LDPC L=100 N=2000 square window W=10, 20 iterations per window",
           "value": value, "unit": "trials/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic",
           "config": {"workload": f"({DV},{DC}) SC-LDPC L={L_CHAIN} N={N_POS} eps={EPS} decodeBP_SW W={W} I_max={IT} I_init={IT}",
                      "trials_per_gpu_per_step": B, "step": ("device sample (code + CN->socket table) -> decodeBP_SW (window state in LDS) -> plr_computation" if gen2 else
                               "device sample -> CN->socket table -> decodeBP_SW (window state in LDS) -> plr_computation"),
                      "parallelism": f"trial-sharded x{world} (the eps grid shards by point in bp_decoding.py)",
                      "streams": "sampler(k+1) || decoder(k), double-buffered" if nbuf == 2 else "single stream"},
           "roofline": {"bound": "hbm", "kernel": "sw_ring_kernel" if gen2 else "sw_ring_kernel (+ cn_sockets_kernel)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, **_traffic_fields(measured_traffic("sw_ring_kernel", B, "C4")),
                        "alg_bytes_per_trial": share, "ms_per_launch": ms_w,
                        "literal_flooding": {"bytes": "8*E_w*sum I, E_w = W*N*dv", "sum_iterations": r["iterations"],
                                             "equivalent_GBs": lit, "x_peak": lit / HBM_PEAK_GBS},
                        "step": {"achieved": b_alg * value / world / 1e9, "frac": b_alg * value / world / 1e9 / HBM_PEAK_GBS}},
           "kernels_ms": {"sample_philox": ms_s, "sw_bp": ms_w},
           "results": {"FER": r["frame_err"] / r["frames"], "BLER": r["block_err"] / p.L / r["frames"],
                       "BER": r["users_err"] / p.n / r["frames"], "mean_window_iterations": r["iterations"] / r["frames"]}}
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c4(EPS, W, IT)
    emit(out)


# ------------------------------------------------------------------------------------------------------------------
# C5: doped (4,8) streaming ensemble N=5000, circular buffer L=50, W=20, doping {10,11,12} (main_streaming, BPF:1934-2054)
# ------------------------------------------------------------------------------------------------------------------
def run_c5(a, E, dev, rank, world, dist, fence, finish):
    L_BUF, N_POS, W, DOPED, CHUNK = 50, 5000, 20, (10, 11, 12), 16
    EPS = 0.485 if a.eps is None else a.eps
    p = E.make_params(DV, DC, L_BUF, N_POS)
    NS = a.batch or 6144                              # three 256-thread decode workgroups per CU in flight: a half of 3072 streams is four rounds of them
    # The streams are independent, so they are run as two halves on two HIP streams: the generation launches of one half
    # (vector / LDS work, 16 waves per workgroup) overlap the decode launches of the other (waits on row gathers, 4 waves per
    # workgroup) — they share a CU's LDS (75 KB / 50 KB per workgroup).  --no-overlap: one set, one stream.
    halves = 1 if a.no_overlap or NS < 2 else 2
    sizes = [NS - NS // 2, NS // 2][:halves] if halves == 2 else [NS]
    hip_streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(halves - 1)]
    sts, off = [], 0
    for h in range(halves):
        sts.append(E.Streams(p, sizes[h], seed=SEED, eps=EPS, W=W, doped=DOPED, stream0=rank * NS + off, device=dev))
        off += sizes[h]
    ev = _events(a.steps, 2)

    def run_all():
        main = torch.cuda.current_stream(dev)
        for h in range(1, halves):
            hip_streams[h].wait_stream(main)
        for h in range(halves):
            with torch.cuda.stream(hip_streams[h]):
                sts[h].run(CHUNK)
        for h in range(1, halves):
            main.wait_stream(hip_streams[h])

    for _ in range(max(1, a.warmup)):
        run_all()
    fence()
    c0 = [st.counters.clone() for st in sts]
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record()
        run_all()
        ev[k][1].record()
    fence()
    dt = finish(time.perf_counter() - t0)
    tot = sum((st.counters - c)[:, :8].sum(dim=0) for st, c in zip(sts, c0))
    if dist:
        dist.all_reduce(tot)                            # the RCCL counter reduce of the streaming driver: eight int64
    if rank != 0:
        return
    c = tot.cpu().numpy()
    positions = a.steps * CHUNK * NS * world
    ms = sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps
    # per decoded position one VN position (N rows of dv ids) and one CN position enter the buffer and are read by the window
    b_alg = 16 * N_POS * DV + N_POS // 8
    ach = b_alg * CHUNK * NS / (ms * 1e-3) / 1e9
    tg, td = measured_traffic("stream_gen_kernel", sizes[0], "C5"), measured_traffic("stream_dec_kernel", sizes[0], "C5")   # per launch of one half
    # (a step of 16 positions is one decode launch between two generate launches; the PMC figures are means per launch)
    tr = {"hbm_bytes": halves * (2 * tg["hbm_bytes"] + td["hbm_bytes"]),
          "hbm_bytes_raw": halves * (2 * tg["hbm_bytes_raw"] + td["hbm_bytes_raw"])} if tg and td else None
    out = {"metric": "decoded positions/sec, doped (4,8) SC-LDPC streaming ensemble N=5000, buffer L=50, W=20",
           "value": positions / dt, "unit": "positions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic",
           "config": {"workload": f"doped ({DV},{DC}) streaming ensemble N={N_POS} L={L_BUF} W={W} doped={list(DOPED)} eps={EPS}",
                      "streams_per_gpu": NS, "positions_per_stream_per_step": CHUNK,
                      "step": "generate_stream_pos (stream_gen_kernel, 1024 threads) + decodeBP_SW_circular (stream_dec_kernel, 256 threads) per position",
                      "parallelism": f"stream-sharded x{world}",
                      "streams": "two halves of the streams on two HIP streams: generation of one beside decoding of the other"
                                 if halves == 2 else "single stream"},
           "roofline": {"bound": "hbm", "kernel": "stream_gen_kernel + stream_dec_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, **_traffic_fields(tr),
                        "alg_bytes_per_position": b_alg, "ms_per_launch": ms,
                        "note": "per step: the streams' next 16 positions are generated (ranking 20000 sockets per position), then "
                                "decoded; `traffic` sums both kernels' launches of a step"},
           "results": {"BLER": float(c[1]) / max(1.0, float(c[5])), "BLER_exp": float(c[3]) / max(1.0, float(c[7])),
                       "BER": float(c[0]) / max(1.0, float(c[4])), "blocks": int(c[5])}}
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c5(EPS, W, DOPED)
    emit(out)


if __name__ == "__main__":
    main()
