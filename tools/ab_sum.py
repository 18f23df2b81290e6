#!/usr/bin/env python3
"""A/B on the GPU box: the count-and-sum fixpoint decoder (full_bp_sum.hip, no CN -> VN table, one gather per release)
against the 4-bit-count decoder (full_bp_small.hip, two gathers per release) on ensembles with N <= 512.
    python tools/ab_sum.py [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for L, N, eps in [(50, 500, 0.48), (50, 512, 0.47), (100, 250, 0.48), (50, 100, 0.47)]:
    p = E.make_params(4, 8, L, N)
    a, cn, ch = E.sample_philox_cn16(p, 5, 0, T, eps)
    cnt = torch.empty((T, E.NCOUNTERS), dtype=torch.int32, device=a.device)
    ms_small = timed(lambda: E.full_bp_fixpoint_cn16(p, a, cn, ch, counters=cnt))
    ref = cnt.clone()
    line = f"L={L} N={N} eps={eps} T={T}: full_bp_small {ms_small:8.3f} ms"
    for per_cu in ("5", "4"):
        os.environ["SCLDPC_DEBUG_SUM_PER_CU"] = per_cu
        ms = timed(lambda: E.full_bp_fixpoint_vn16(p, a, ch, counters=cnt))
        keep = [0, 1, 2, 3, 4, 6, 7]
        assert torch.equal(cnt[:, keep], ref[:, keep])
        line += f" | full_bp_sum at {per_cu} per CU {ms:8.3f} ms"
    ms_s_cn = timed(lambda: E.sample_philox_cn16(p, 5, 0, T, eps, out=(a, cn, ch)))
    ms_s = timed(lambda: E.sample_philox_cn16(p, 5, 0, T, eps, out=(a, None, ch)))
    print(line + f" | sampler with / without the CN table {ms_s_cn:.3f} / {ms_s:.3f} ms", flush=True)
