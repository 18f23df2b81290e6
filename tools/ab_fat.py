#!/usr/bin/env python3
"""A/B on the GPU box (VERDICT r02 #2a): what would the fixpoint decoder gain from FAT CN rows — per CN its dc neighbours, each
with its whole VN row (64 bytes) — i.e. ONE gather per release instead of two?  The fat table is built here by a gather pass
over the sampler's two tables (far too dear to ship: the question is the decoder's side only).
    python tools/ab_fat.py [T]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = E.lib()
vp, i32 = C.c_void_p, C.c_int32
L.scldpc_ab_fatten_device.argtypes = [C.POINTER(E.CodeParams), i32, vp, vp, vp, vp]
L.scldpc_ab_full_bp_fixpoint_device_fat.argtypes = [C.POINTER(E.CodeParams), i32, vp, vp, vp, vp, i32, vp, vp, vp]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (Lc, N, eps) in [(50, 1000, 0.48), (50, 1000, 0.46)]:
    p = E.make_params(4, 8, Lc, N)
    a, cn, ch = E.sample_philox_cn16(p, 5, 0, T, eps)
    fat = torch.empty((T, p.nk, 8, 2), dtype=torch.int32, device=a.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    E.check(L.scldpc_ab_fatten_device(C.byref(p), T, a.data_ptr(), cn.data_ptr(), fat.data_ptr(), st))
    cnt = torch.empty((T, E.NCOUNTERS), dtype=torch.int32, device=a.device)
    ms_small = timed(lambda: E.full_bp_fixpoint_cn16(p, a, cn, ch, counters=cnt))
    ref = cnt.clone()
    ms_fat = timed(lambda: E.check(L.scldpc_ab_full_bp_fixpoint_device_fat(C.byref(p), T, a.data_ptr(), cn.data_ptr(), fat.data_ptr(),
                                                                             ch.data_ptr(), 1, cnt.data_ptr(), None, st)))
    keep = [0, 1, 2, 3, 4, 6, 7]
    assert torch.equal(cnt[:, keep], ref[:, keep]), "counters differ"
    print(f"L={Lc} N={N} eps={eps} T={T}: two gathers per release (shipping) {ms_small:8.3f} ms | one 64-byte gather per release "
          f"(fat rows) {ms_fat:8.3f} ms | fat table {fat.numel() * 4 / T / 1e6:.2f} MB per trial (counters equal on all trials)", flush=True)
