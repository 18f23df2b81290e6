#!/usr/bin/env python3
"""Soak on the GPU box: the kernels that keep their CN words in the workspace (peel_sweep, the whole-chain window kernel, the
first-generation flooding kernel) with the words built through cn_build.hip's LDS ring against their own build by global atomics —
every counter and every erasure / lost pattern of every trial, several sizes and channel parameters.

    python tools/soak_prebuild.py > profiles/r03_soak_prebuild.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402
from fl_scaling_sc_ldpc_amd import peeling_decoding as PD  # noqa: E402


def both(env, fn):
    out = {}
    for mode in ("1", "0"):
        os.environ[env] = mode
        out[mode] = fn()
        torch.cuda.synchronize()
    os.environ.pop(env)
    return out


total = 0
for seed, (L, M, e, term, bounded) in enumerate([(50, 5000, 0.48, True, True), (50, 5000, 0.44, False, True), (30, 10000, 0.49, True, False),
                                                  (100, 3000, 0.47, False, False), (64, 4100, 0.30, True, True)]):
    g = PD._Geometry(4, 8, L, M, term, bounded, [])
    T = 1024
    d_adj, d_ch = E.sample_philox(g.params, 100 + seed, 0, T, e, adj16=True)
    r = both("SCLDPC_DEBUG_SWEEP_PREBUILD", lambda: E.peel_sweep(g.params, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi, want_lost=True))
    cols = [0, 1, 2, 7]
    assert torch.equal(r["1"]["out"][:, cols], r["0"]["out"][:, cols]) and torch.equal(r["1"]["lost"], r["0"]["lost"]), (L, M)
    for cl, W, it in ((False, 10, 20), (True, 12, 1000000), (False, 3, 5)):
        r = both("SCLDPC_DEBUG_SW_PREBUILD", lambda: E.sw_bp(g.params, d_adj, d_ch, W, it, 0, want_erased=True, classical=cl, ring=False))
        assert torch.equal(r["1"]["counters"], r["0"]["counters"]) and torch.equal(r["1"]["erased"], r["0"]["erased"]), (L, M, cl, W)
    for traj in (False, True):
        r = both("SCLDPC_DEBUG_FULLBP_PREBUILD", lambda: E.full_bp(g.params, d_adj, d_ch, max_it=0, is_term=term, want_erased=True,
                                                                   rows_cap=64 if traj else 0))
        assert torch.equal(r["1"]["counters"], r["0"]["counters"]) and torch.equal(r["1"]["erased"], r["0"]["erased"]), (L, M, traj)
        if traj:
            assert torch.equal(r["1"]["rows"], r["0"]["rows"])
    total += T
    print(f"L={L} N={M} eps={e} terminated={term}: {T} trials: sweep, square / classical windows, flooding with and without rows — identical", flush=True)
    del d_adj, d_ch
    torch.cuda.empty_cache()
print(f"prebuild soak: {total} trials x 6 kernels/modes, no mismatch")
