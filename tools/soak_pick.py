#!/usr/bin/env python3
"""Soak on the GPU box: random-pick peeling with one, two and four trials per wave (peel_pick_kernel / peel_pick_multi_kernel)
must give identical trajectories and loss rates — same Philox draws, same ascending-order picks.
    python tools/soak_pick.py >> profiles/r03_soak_pick.txt"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import peeling_decoding as PD
res = {}
for mode in ("1", "2", "4"):
    os.environ["SCLDPC_DEBUG_PICK_TPW"] = mode
    res[mode] = [PD.simulate_peeling_decoder_ldpc(e, 4, 8, L, M, term, False, T, [], rng="philox", seed=s)
                 for (e, L, M, term, T, s) in [(0.46, 44, 4000, False, 301, 5), (0.2, 60, 3000, True, 257, 6), (0.48, 50, 10000, False, 33, 7)]]
for mode in ("2", "4"):
    for a, b in zip(res["1"], res[mode]):
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all(), mode
print("pick soak: 301 + 257 + 33 trials (N = 4000, 3000, 10000), one / two / four trials per wave: identical trajectories and loss rates")
