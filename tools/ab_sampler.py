#!/usr/bin/env python3
"""A/B in ONE process (diagnostics): sampler generations on the BASELINE ensemble — second generation (SCLDPC_DEBUG_SAMPLER_V2=1)
against the third (default where one Philox call per thread covers a position).  Checks the tables against each other
(VN -> CN bit for bit, CN -> VN as sets) and times both, with and without the CN table; then the whole step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 50
p = E.make_params(4, 8, L, N)
dev = "cuda:0"
a1 = torch.empty((B, p.n, 4), dtype=torch.int16, device=dev)
cn = torch.empty((B, p.nk, 8), dtype=torch.int16, device=dev)
ch = torch.empty((B, p.nw), dtype=torch.int32, device=dev)
cnt = torch.empty((B, 8), dtype=torch.int32, device=dev)


def gen(v2):
    os.environ["SCLDPC_SAMPLER_GEN"] = "2" if v2 else "3"


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


# ---- equality on a few trials (doped positions included)
T = 4
for doped in ((), (3, 4)):
    gen(True)
    r2 = E.sample_philox_cn16(p, 7, 123, T, 0.48, doped)
    s2 = E.sample_philox_sock16(p, 7, 123, T, 0.48, doped)
    gen(False)
    r3 = E.sample_philox_cn16(p, 7, 123, T, 0.48, doped)
    s3 = E.sample_philox_sock16(p, 7, 123, T, 0.48, doped)
    n3 = E.sample_philox_cn16(p, 7, 123, T, 0.48, doped, want_cn=False)
    torch.cuda.synchronize()
    assert torch.equal(r2[0], r3[0]) and torch.equal(r2[2], r3[2]), "VN table / channel differ"
    assert torch.equal(s2[0], s3[0]) and torch.equal(n3[0], r3[0]) and torch.equal(n3[2], r3[2])
    for x, y in ((r2[1], r3[1]), (s2[1], s3[1])):
        xs, ys = np.sort(x.cpu().numpy().view(np.uint16), axis=2), np.sort(y.cpu().numpy().view(np.uint16), axis=2)
        assert (xs == ys).all(), "CN table differs as sets"
print("tables equal (second vs third generation)", flush=True)

gen(True)
s2n = timeit(lambda k: E.sample_philox_cn16(p, 1, k * B, B, 0.48, out=(a1, None, ch)), "sampler v2, vn table only")
s2c = timeit(lambda k: E.sample_philox_cn16(p, 1, k * B, B, 0.48, out=(a1, cn, ch)), "sampler v2, vn + cn tables")
gen(False)
s3n = timeit(lambda k: E.sample_philox_cn16(p, 1, k * B, B, 0.48, out=(a1, None, ch)), "sampler v3, vn table only")
s3c = timeit(lambda k: E.sample_philox_cn16(p, 1, k * B, B, 0.48, out=(a1, cn, ch)), "sampler v3, vn + cn tables")
if E.cn16_supported(p):
    d = timeit(lambda k: E.full_bp_fixpoint_cn16(p, a1, cn, ch, counters=cnt), "decoder: fixpoint, 4-bit counts + CN->VN table")
    print(f"step v2 {s2c + d:.2f} ms -> {B / (s2c + d) / 1e3:.3f} M trials/s;  step v3 {s3c + d:.2f} ms -> {B / (s3c + d) / 1e3:.3f} M trials/s")
