#!/usr/bin/env python3
"""A/B for BASELINE config 3 (random-pick peeling, N = 10000, 290 000 steps): where do the degree-1 trajectory's moments
come from — three global atomics per step inside the pick chain, or r1 rows written per step and reduced afterwards?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L, N, EPS = 50, 10000, 0.48
p = E.make_params(4, 8, L, N)
steps = int(N * L * (EPS + 0.1))
ts = p.cns_pos * L
d_adj, d_ch = E.sample_philox(p, 1, 0, B, EPS, adj16=True)
torch.cuda.synchronize()


def timed(name, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:60s} {e0.elapsed_time(e1):9.1f} ms / {B} trials", flush=True)
    return out


mom = torch.zeros((3, steps + 1), dtype=torch.int64, device="cuda")
timed("warm-up (in-kernel moments)", lambda: E.peel_pick(p, d_adj, d_ch, ts, steps, seed=1, trial0=0, want_r1=False, moments=mom))
mom.zero_()
timed("in-kernel moments (3 global atomics per step)", lambda: E.peel_pick(p, d_adj, d_ch, ts, steps, seed=1, trial0=0, want_r1=False, moments=mom))
timed("no trajectory output at all", lambda: E.peel_pick(p, d_adj, d_ch, ts, steps, seed=1, trial0=0, want_r1=False))
r = timed("r1 rows (one 4-byte store per step)", lambda: E.peel_pick(p, d_adj, d_ch, ts, steps, seed=1, trial0=0, want_r1=True))
m2 = timed("   + r1_moments over the rows", lambda: E.r1_moments(r["r1"]))
assert torch.equal(m2, mom), "moments differ"
print("moments agree")
