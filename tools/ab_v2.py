#!/usr/bin/env python3
"""A/B timing in ONE process on one box: first-generation sampler + 16-bit-word fixpoint decoder against the
second generation (sampler_v2.hip, full_bp_small.hip) on the BASELINE ensemble.  HIP events on the launch stream."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p = E.make_params(4, 8, 50, 1000)
dev = "cuda:0"
a1 = torch.empty((B, p.n, 4), dtype=torch.int16, device=dev)
cn = torch.empty((B, p.nk, 8), dtype=torch.int16, device=dev)
ch = torch.empty((B, p.nw), dtype=torch.int32, device=dev)
cnt = torch.empty((B, 8), dtype=torch.int32, device=dev)


def timeit(fn, name):
    fn(0)
    torch.cuda.synchronize()
    ts = []
    for k in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(k + 1)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print(f"{name:58s} {ms:8.3f} ms / {B} trials = {B / ms / 1e3:8.3f} M trials/s", flush=True)
    return ms


s1 = timeit(lambda k: E.sample_philox(p, 1, k * B, B, 0.48, out=(a1, ch)), "sampler v1 (adj16)")
s2 = timeit(lambda k: E.sample_philox_cn16(p, 1, k * B, B, 0.48, out=(a1, None, ch)), "sampler v2, vn table only")
s3 = timeit(lambda k: E.sample_philox_cn16(p, 1, k * B, B, 0.48, out=(a1, cn, ch)), "sampler v2, vn + cn tables")
d1 = timeit(lambda k: E.full_bp_fixpoint(p, a1, ch, counters=cnt), "decoder: fixpoint, 16-bit CN words (2 trials/CU)")
c1 = cnt.clone()
d2 = timeit(lambda k: E.full_bp_fixpoint_cn16(p, a1, cn, ch, counters=cnt), "decoder: fixpoint, 4-bit counts + CN->VN table (7 trials/CU)")
assert torch.equal(c1[:, [0, 1, 2, 3, 4, 6, 7]], cnt[:, [0, 1, 2, 3, 4, 6, 7]]), "decoders disagree"
print(f"step v1 {s1 + d1:.2f} ms -> {B / (s1 + d1) / 1e3:.3f} M trials/s;  "
      f"step v2 {s3 + d2:.2f} ms -> {B / (s3 + d2) / 1e3:.3f} M trials/s")
