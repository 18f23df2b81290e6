#!/usr/bin/env python3
"""On the GPU box: scldpc_peel_sweep_device_adj16 at the notebook's size (L = 50, M = 10000: CN words in the workspace),
first build through cn_build.hip's LDS ring (default) against one global atomic per edge (SCLDPC_DEBUG_SWEEP_PREBUILD=0).

    python tools/ab_sweep_prebuild.py > profiles/r03_ab_sweep_prebuild.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402
from fl_scaling_sc_ldpc_amd import peeling_decoding as PD  # noqa: E402

T = 2048
for L, M, e in ((50, 10000, 0.48), (50, 5000, 0.48), (100, 4000, 0.47)):
    g = PD._Geometry(4, 8, L, M, True, True, [])
    d_adj, d_ch = E.sample_philox(g.params, 3, 0, T, e, adj16=True)
    outs = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["SCLDPC_DEBUG_SWEEP_PREBUILD"] = mode
        E.peel_sweep(g.params, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            o = E.peel_sweep(g.params, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi)["out"]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        outs[mode] = o[:, [0, 1, 2, 7]].cpu()
        print(f"L={L} M={M} eps={e}: {T} trials, build {'through the LDS ring' if mode == '1' else 'by global atomics   '}: "
              f"{ms:8.1f} ms per launch = {T / ms:7.2f} k trials/s", flush=True)
    assert (outs["1"] == outs["0"]).all()
    del d_adj, d_ch
    torch.cuda.empty_cache()
