#!/usr/bin/env python3
"""Headline benchmark: Monte-Carlo codeword trials/sec, (4,8) SC-LDPC, L=50, N=1000, ε=0.48, full BP to
the fixpoint (BASELINE.json), on N GPUs of one node.

A *step* = one pass of the whole hot path over one batch of B trials per GPU, everything on the device:
    sample (fresh random code + channel per trial, Philox-keyed)  →  decodeBP  →  plr_computation
i.e. exactly what one frame of the reference's main_terminated loop does (BPF:2117-2144), B times.
Nothing is cached between steps: every step draws new trial indices.  Trials are independent, so N GPUs
take disjoint trial ranges with no data-path collective ("weak" scaling: B per GPU is fixed); the only
exchange is the final sum of the run counters (one RCCL all-reduce of 9 int64, outside the timed region
except for the closing barrier).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N … bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     — dominant kernel (full_bp): algorithmic bytes per launch ÷ its mean duration measured here
                 with HIP events on the launch stream, against the 8 TB/s HBM peak.  Algorithmic bytes per
                 trial = 16·E + n/8 (both adjacency directions written once and read once as
                 int32, plus the channel bits) = 3 206 250 B at this ensemble.  `traffic` = measured HBM bytes
                 per launch from the committed rocprofv3 PMC passes (profiles/), corrected as the MI355X
                 guide prescribes (FETCH_SIZE×2), or null when no profile matches this workload.
  cpu_baseline — the REAL reference decoder (oracle/_ref/ref_bpf_M500_L50 = the reference's own C source
                 compiled by oracle/Makefile; kind "reference") timed on this box's host cores on a bounded
                 sample of the same workload, one single-threaded process per core as the reference is run
                 on clusters (NB cell 35:21).  Falls back to the oracle's literal restatement (kind "port")
                 where the reference binary is absent.  Rank 0, N=1 only.  A reported baseline, not a target.
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

DV, DC, L_CHAIN, N_POS, EPS = 4, 8, 50, 1000, 0.48
HBM_PEAK_GBS = 8000.0


def cpu_baseline(budget_trials_per_core=4):
    """Time the reference's decodeBP path (generate_code + channel_doped + decodeBP per trial) on host cores."""
    cores = max(1, min(os.cpu_count() or 1, 16))
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_bpf_M500_L50")
    T = budget_trials_per_core
    if os.path.exists(ref):
        t0 = time.time()
        procs = [subprocess.Popen([ref, str(T), str(900001 + 1000 * k), repr(EPS), "1000000", "0", "0", "1", "0"],
                                  stdout=subprocess.DEVNULL) for k in range(cores)]
        ok = all(p.wait() == 0 for p in procs)
        dt = time.time() - t0
        if ok:
            return {"value": cores * T / dt, "unit": "trials/s", "cores": cores, "kind": "reference",
                    "sample": f"{cores} single-threaded processes x {T} trials (srandom seeds 900001+1000k), "
                              f"oracle/_ref/ref_bpf_M500_L50, {dt:.1f} s wall"}
    # fallback: the oracle's literal per-edge flooding restatement, one process per core
    code = ("import sys,time; sys.path.insert(0,%r); from oracle import oracle as O; p=O.Params(4,8,50,500,1000); "
            "[O.trial(p, int(sys.argv[1])+t, %r, decoder=0) for t in range(%d)]" % (ROOT, EPS, 4 * T))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(900001 + 1000 * k)]) for k in range(cores)]
    for p in procs:
        p.wait()
    dt = time.time() - t0
    return {"value": cores * 4 * T / dt, "unit": "trials/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x {4 * T} trials, oracle literal flooding decoder, {dt:.1f} s wall"}


def measured_traffic(batch):
    """HBM bytes per full_bp launch from the committed PMC summary, if it was taken on this workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        if t.get("workload") == workload_name() and t.get("batch") == batch:
            return t["full_bp_hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def workload_name():
    return f"({DV},{DC}) SC-LDPC L={L_CHAIN} N={N_POS} eps={EPS} full BP unlimited iterations"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32768, help="trials per GPU per step (27 GB of tables per buffer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flooding", action="store_true",
                    help="decode with the level-synchronous kernel (one barrier round per flooding iteration, reports the "
                         "iteration count) instead of the fixpoint kernel (same outputs, no iteration count)")
    ap.add_argument("--overlap", action="store_true",
                    help="(default) two streams, two buffers: the sampler of step k+1 fills the tail of the decoder of step k")
    ap.add_argument("--no-overlap", action="store_true", help="one stream: sample, then decode, then accumulate")
    ap.add_argument("--adj32", action="store_true", help="int32 global-id adjacency instead of the compact uint16 one")
    ap.add_argument("--gen1", action="store_true",
                    help="first-generation kernels: sampler that ranks every key + fixpoint decoder on 16-bit CN words "
                         "(default: sampler_v2 + the 4-bits-per-CN decoder, which also reads the CN -> socket table)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    if a.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from fl_scaling_sc_ldpc_amd import engine as E
    p = E.make_params(DV, DC, L_CHAIN, N_POS)
    B = a.batch
    # Two streams by default: two (tables, channel, counters) buffers, the sampler of step k+1 beside the decoder of step k.
    # The decoder's trials differ 10x in length (a successful decode walks the whole chain), so its launch ends in a
    # tail of long trials on a mostly empty chip: the next step's sampler fills it (+3-4 %).
    a.overlap = not a.no_overlap
    nbuf = 1 if a.no_overlap else 2
    gen2 = not (a.gen1 or a.flooding or a.adj32)
    if gen2 and not E.cn16_supported(p):
        raise SystemExit("bench.py: the second-generation kernels do not take this ensemble")
    d_adj = [torch.empty((B, p.n, p.dv), dtype=torch.int32 if a.adj32 else torch.int16, device=dev) for _ in range(nbuf)]
    d_cn = [torch.empty((B, p.nk, p.dc), dtype=torch.int16, device=dev) if gen2 else None for _ in range(nbuf)]
    d_ch = [torch.empty((B, p.nw), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    d_cnt = [torch.empty((B, E.NCOUNTERS), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    run = E.new_run(dev)
    seed = 20261004
    s_dec = torch.cuda.current_stream(dev)
    s_samp = s_dec if a.no_overlap else torch.cuda.Stream(dev)
    sampled = [torch.cuda.Event() for _ in range(nbuf)]
    decoded = [torch.cuda.Event() for _ in range(nbuf)]

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def step(k, timed_idx=None):
        # global trial index: step-major, then rank — disjoint ranges on every rank
        trial0 = (k * world + rank) * B
        b = k % nbuf
        e = ev[timed_idx] if timed_idx is not None else None
        with torch.cuda.stream(s_samp):
            s_samp.wait_event(decoded[b])                   # buffer b is free once its previous decode has finished
            if e:
                e[0].record(s_samp)
            if gen2:
                E.sample_philox_cn16(p, seed, trial0, B, EPS, out=(d_adj[b], d_cn[b], d_ch[b]))
            else:
                E.sample_philox(p, seed, trial0, B, EPS, out=(d_adj[b], d_ch[b]))
            if e:
                e[1].record(s_samp)
            sampled[b].record(s_samp)
        with torch.cuda.stream(s_dec):
            s_dec.wait_event(sampled[b])
            if e:
                e[2].record(s_dec)
            if a.flooding:
                E.full_bp(p, d_adj[b], d_ch[b], counters=d_cnt[b])
            elif gen2:
                E.full_bp_fixpoint_cn16(p, d_adj[b], d_cn[b], d_ch[b], counters=d_cnt[b])
            else:
                E.full_bp_fixpoint(p, d_adj[b], d_ch[b], counters=d_cnt[b])
            if e:
                e[3].record(s_dec)
            E.accumulate_run(d_cnt[b], run, 0)
            decoded[b].record(s_dec)

    def fence():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(a.warmup):
        step(k)
    run.zero_()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k, k)
    fence()
    dt = time.perf_counter() - t0

    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(run, op=dist.ReduceOp.SUM)          # the optional RCCL counter reduce
    dt = float(t_max.item())
    r = dict(zip(E.RUN_NAMES, run.cpu().tolist()))
    total_trials = a.steps * B * world
    assert r["frames"] == total_trials, (r["frames"], total_trials)

    if rank == 0:
        ms_sample = sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps
        ms_bp = sum(e[2].elapsed_time(e[3]) for e in ev) / a.steps
        E_edges = p.n * p.dv
        b_alg = 16 * E_edges + p.n // 8
        achieved = b_alg * B / (ms_bp * 1e-3) / 1e9
        traffic = measured_traffic(B)
        out = {
            "metric": "MC codeword trials/sec, (4,8) SC-LDPC L=50 N=1000 eps=0.48 full BP",
            "value": total_trials / dt, "unit": "trials/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload_name(), "trials_per_gpu_per_step": B,
                       "step": "device sample (code+channel) -> decodeBP -> plr_computation",
                       "decoder": ("flooding, one barrier round per iteration" if a.flooding else
                                   "fixpoint of unlimited flooding by chain-following peeling (every output of decodeBP "
                                   "except the iteration count; equality with the flooding kernel is a test)"),
                       "rng": "philox4x32-10 keyed by (seed, trial)",
                       "adjacency": "int32 global ids" if a.adj32 else "uint16 position-local ids", "parallelism": f"trial-sharded x{world}",
                       "streams": "sampler(k+1) || decoder(k), double-buffered" if nbuf == 2 else "single stream"},
            "roofline": {"bound": "hbm", "kernel": "full_bp_kernel" if a.flooding else "full_bp_fixpoint_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_trial": b_alg, "trials_per_launch": B, "ms_per_launch": ms_bp},
            "kernels_ms": {"sample_philox": ms_sample, "full_bp": ms_bp},
            "decode_only_trials_per_s_per_gpu": B / (ms_bp * 1e-3),
            "results": {"FER": r["frame_err"] / r["frames"], "BLER": r["block_err"] / p.L / r["frames"],
                        "BER": r["users_err"] / p.n / r["frames"], "FER_exp": r["frame_err_exp"] / r["frames"],
                        ("mean_iterations" if a.flooding else "mean_barrier_rounds"): r["iterations"] / r["frames"]},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
