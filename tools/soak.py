#!/usr/bin/env python3
"""Soak of the bench's decoder pair on the GPU box: N trials of the BASELINE ensemble through sampler_v2, then BOTH forms of
the 4-bit decoder — the barrier-free fixpoint (what `python bench.py` times; its two-atomic claim protocol is the kind of code
whose failures would be rare events) and the level-synchronous one (iteration-exact, itself equal to the 16-bit-word flooding
kernel on 10^5 trials in tests/test_gpu_fullsize.py) — and a comparison of every counter of every trial except the
iteration / barrier-round count, plus the residual pattern of one batch in sixteen.

    python tools/soak.py [N=50000000] [eps=0.48]        -> one line per 2^20 trials, a summary line at the end
"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
EPS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.48
B = 32768
KEEP = [0, 1, 2, 3, 4, 6, 7]
p = E.make_params(4, 8, 50, 1000)
a = torch.empty((B, p.n, 4), dtype=torch.int16, device="cuda")
cn = torch.empty((B, p.nk, 8), dtype=torch.int16, device="cuda")
ch = torch.empty((B, p.nw), dtype=torch.int32, device="cuda")
c1 = torch.empty((B, E.NCOUNTERS), dtype=torch.int32, device="cuda")
c2 = torch.empty((B, E.NCOUNTERS), dtype=torch.int32, device="cuda")
bad = torch.zeros((), dtype=torch.int64, device="cuda")
fails = torch.zeros((), dtype=torch.int64, device="cuda")
its = torch.zeros((), dtype=torch.int64, device="cuda")
t0 = time.time()
done = 0
for k, b0 in enumerate(range(0, N, B)):
    full = k % 16 == 0
    E.sample_philox_cn16(p, 0x50AC, (1 << 40) + b0, B, EPS, out=(a, cn, ch))
    f = E.full_bp_fixpoint_cn16(p, a, cn, ch, counters=c1, want_erased=full)
    l = E.full_bp_cn16(p, a, cn, ch, counters=c2, want_erased=full)
    bad += (c1[:, KEEP] != c2[:, KEEP]).any(dim=1).sum()
    if full:
        bad += (f["erased"] != l["erased"]).any(dim=1).sum()
    fails += (c2[:, 0] > 0).sum()
    its += c2[:, 5].sum()
    done += B
    if done % (1 << 20) == 0:
        print(f"{done:>11d} trials  mismatching {int(bad.item())}  FER {fails.item() / done:.5f}  "
              f"{done / (time.time() - t0) / 1e3:.0f} k trials/s (sampler + both decoders)", flush=True)
torch.cuda.synchronize()
print(f"SOAK eps={EPS} trials={done} mismatching={int(bad.item())} FER={fails.item() / done:.6f} "
      f"mean_iterations={its.item() / done:.3f} seconds={time.time() - t0:.1f}")
sys.exit(1 if int(bad.item()) else 0)
