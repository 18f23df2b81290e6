#!/usr/bin/env python3
"""Time the BASELINE.json configurations (other than the bench's) on one GPU: functional runs with modest trial
counts, to record throughput beside bench.py's headline number."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fl_scaling_sc_ldpc_amd import engine as E
from fl_scaling_sc_ldpc_amd import peeling_decoding as PD


def timed(label, fn, trials):
    fn()                                    # warm-up (module load, workspace growth)
    torch.cuda.synchronize(); t0 = time.time()
    out = fn()
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{label}: {trials} trials in {dt * 1e3:.1f} ms  = {trials / dt:.0f} trials/s", flush=True)
    return out


# C2: (4,8) L=50 N=1000 full BP — bench.py.  Here: limited iterations and the trajectory mode.
p = E.make_params(4, 8, 50, 1000)
T = 8192
d_adj, d_ch = E.sample_philox(p, 1, 0, T, 0.48, adj16=True)
timed("C2  full BP, unlimited (decode only)", lambda: E.full_bp(p, d_adj, d_ch), T)
timed("C2  full BP, unlimited, fixpoint kernel (decode only)", lambda: E.full_bp_fixpoint(p, d_adj, d_ch), T)
timed("C2  full BP, max 100 iterations", lambda: E.full_bp(p, d_adj, d_ch, max_it=100), T)
timed("C2  full BP with trajectory rows (bp_traj mode)", lambda: E.full_bp(p, d_adj, d_ch, rows_cap=1024), T)
timed("C2  square window W=20, 6/60 iterations", lambda: E.sw_bp(p, d_adj, d_ch, 20, 6, 60), T)
g = PD._Geometry(4, 8, 50, 1000, True, True, [])
timed("C1  sweep peeling + stopping sets (simulate_sc_ldpc)", lambda: E.peel_sweep(g.params, d_adj, d_ch, g.total_size), T)
steps = int(1000 * 50 * 0.58)
Tp = T          # one wave per trial: the kernel wants thousands of trials in flight
timed("C1  random-pick peeling, N=1000, 29000 steps (simulate_peeling_decoder_ldpc)",
      lambda: E.peel_pick(p, d_adj[:Tp], d_ch[:Tp], 500 * 50, steps, seed=3, want_r1=True), Tp)
del d_adj, d_ch

# C4: (4,8) L=100 N=2000 square window W=10, 20 iterations per window
p4 = E.make_params(4, 8, 100, 2000)
T4 = 1024
a4, c4 = timed("C4  sampling L=100 N=2000", lambda: E.sample_philox(p4, 1, 0, T4, 0.47, adj16=True), T4)
timed("C4  square window W=10, 20 it/window (CN words in the workspace)", lambda: E.sw_bp(p4, a4, c4, 10, 20, 0), T4)
del a4, c4

# C3: (4,8) L=50 N=10000 random-pick peeling, non-terminated, moments only
p3 = E.make_params(4, 8, 50, 10000)
T3 = 4096
a3, c3 = timed("C3  sampling L=50 N=10000 (big-ensemble sampler)", lambda: E.sample_philox(p3, 1, 0, T3, 0.48, adj16=True), T3)
steps3 = int(10000 * 50 * 0.58)
mom = torch.zeros((3, steps3 + 1), dtype=torch.int64, device="cuda")
timed("C3  random-pick peeling N=10000, 290000 steps, in-kernel moments",
      lambda: E.peel_pick(p3, a3, c3, 5000 * 50, steps3, seed=3, want_r1=False, moments=mom), T3)
timed("C3  full BP N=10000 (workspace)", lambda: E.full_bp(p3, a3[:512], c3[:512]), 512)

# C5: doped (4,8) streaming ensemble N=5000, buffer L=50, W=20, doping {10,11,12}
p5 = E.make_params(4, 8, 50, 5000)
ns = 512
st = E.Streams(p5, ns, seed=1, eps=0.485, W=20, doped=(10, 11, 12))
st.run(4)
torch.cuda.synchronize(); t0 = time.time()
cnt, _ = st.run(32)
torch.cuda.synchronize(); dt = time.time() - t0
c = cnt[:, :8].sum(0).cpu().numpy()
print(f"C5  streaming N=5000 L=50 W=20 doped(10,11,12): {ns} streams x 32 positions in {dt * 1e3:.1f} ms = {ns * 32 / dt:.0f} positions/s;"
      f" BLER {c[1] / max(1, c[5]):.4f}", flush=True)
p5b = E.make_params(4, 8, 50, 1000)
st = E.Streams(p5b, 1024, seed=1, eps=0.48, W=20, doped=(10, 11, 12))
st.run(8)
torch.cuda.synchronize(); t0 = time.time()
cnt, _ = st.run(128)
torch.cuda.synchronize(); dt = time.time() - t0
c = cnt[:, :8].sum(0).cpu().numpy()
print(f"C5' streaming N=1000 L=50 W=20 doped(10,11,12): 1024 streams x 128 positions in {dt * 1e3:.1f} ms = {1024 * 128 / dt:.0f} positions/s;"
      f" BLER {c[1] / max(1, c[5]):.4f}", flush=True)
