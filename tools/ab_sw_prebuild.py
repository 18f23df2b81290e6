#!/usr/bin/env python3
"""On the GPU box: the whole-chain window kernel (scldpc_sw_bp_device_adj16 / scldpc_swc_bp_device_adj16) at BASELINE config
4's size, CN words in the workspace: built through cn_build.hip's LDS ring (default) against one global atomic per edge
(SCLDPC_DEBUG_SW_PREBUILD=0).

    python tools/ab_sw_prebuild.py > profiles/r03_ab_sw_prebuild.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

T = 2048
for L, N, W, it, cl in ((100, 2000, 10, 20, False), (100, 2000, 10, 20, True)):
    p = E.make_params(4, 8, L, N)
    a, ch = E.sample_philox(p, 3, 0, T, 0.47, adj16=True)
    for mode in ("1", "0", "1", "0"):
        os.environ["SCLDPC_DEBUG_SW_PREBUILD"] = mode
        E.sw_bp(p, a, ch, W, it, 0, classical=cl, ring=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): E.sw_bp(p, a, ch, W, it, 0, classical=cl, ring=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"L={L} N={N} W={W} {'classical' if cl else 'square   '} window, whole-chain kernel, {T} trials, build "
              f"{'through the LDS ring' if mode == '1' else 'by global atomics   '}: {ms:7.1f} ms = {T / ms:6.2f} k trials/s", flush=True)
