#!/usr/bin/env python3
"""A/B on the GPU box: the bench's C2 step with the sampler and the decoder on DISJOINT sets of CUs (HIP streams created with
hipExtStreamCreateWithCUMask), sampler of step k+1 beside the decoder of step k, against the default (both kernels take the
whole chip one after the other, the sampler filling the decoder's tail).  DESIGN.md §5 estimated +8 % for a balanced split;
this measures it.

    python tools/ab_cumask.py [steps=10]
"""
import ctypes as C
import glob
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, SEED, EPS = 32768, 20261004, 0.48
p = E.make_params(4, 8, 50, 1000)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip = C.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*"))[0])
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (C.c_uint32 * ((NCU + 31) // 32))()
    for i in bits:
        words[i >> 5] |= 1 << (i & 31)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def spread(frac):
    """CU indices of an evenly spread fraction of the chip (whatever the enumeration over the XCDs, every XCD gets its share)"""
    a, acc = [], 0.0
    for i in range(NCU):
        acc += frac
        if acc >= 1.0:
            acc -= 1.0
            a.append(i)
    return a


def per_xcd(k, hyp):
    """k of every XCD's 32 CUs, under either hypothesis about how the mask enumerates them: "interleaved" (CU i lies in XCD
    i % 8) or "major" (XCD i // 32).  Workgroups go round-robin over the XCDs, so an uneven split stalls on the poorest XCD."""
    return [i for i in range(NCU) if ((i // 8) if hyp == "interleaved" else (i % 32)) < k]


def run(frac):
    if frac is None:
        s_samp, s_dec = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
    else:
        mine = set(frac if isinstance(frac, list) else spread(frac))
        s_samp, s_dec = masked_stream(sorted(mine)), masked_stream([i for i in range(NCU) if i not in mine])
    adj = [torch.empty((B, p.n, 4), dtype=torch.int16, device=dev) for _ in range(2)]
    cn = [torch.empty((B, p.nk, 8), dtype=torch.int16, device=dev) for _ in range(2)]
    ch = [torch.empty((B, p.nw), dtype=torch.int32, device=dev) for _ in range(2)]
    cnt = [torch.empty((B, E.NCOUNTERS), dtype=torch.int32, device=dev) for _ in range(2)]
    run_c = E.new_run(dev)
    sampled = [torch.cuda.Event() for _ in range(2)]
    decoded = [torch.cuda.Event() for _ in range(2)]

    def step(k):
        b = k % 2
        with torch.cuda.stream(s_samp):
            s_samp.wait_event(decoded[b])
            E.sample_philox_cn16(p, SEED, k * B, B, EPS, out=(adj[b], cn[b], ch[b]))
            sampled[b].record(s_samp)
        with torch.cuda.stream(s_dec):
            s_dec.wait_event(sampled[b])
            E.full_bp_fixpoint_cn16(p, adj[b], cn[b], ch[b], counters=cnt[b])
            E.accumulate_run(cnt[b], run_c, 0)
            decoded[b].record(s_dec)

    for k in range(3):
        step(k)
    torch.cuda.synchronize()
    run_c.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(STEPS):
        step(3 + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = dict(zip(E.RUN_NAMES, run_c.cpu().tolist()))
    assert r["frames"] == STEPS * B
    return STEPS * B / dt, r["frame_err"] / r["frames"]


print(f"{NCU} CUs, {B} trials per step, {STEPS} steps")
cases = [("whole chip, two streams", None), ("evenly spread 50 %", 0.5), ("evenly spread 45 %", 0.45)]
for hyp in ("interleaved", "major"):
    for k in (12, 14, 15, 16, 18):
        cases.append((f"{k} of 32 CUs per XCD to the sampler ({hyp} enumeration)", per_xcd(k, hyp)))
cases.append(("whole chip, two streams", None))
for name, frac in cases:
    v, fer = run(frac)
    print(f"{name}: {v / 1e3:.0f} k trials/s  (FER {fer:.4f})", flush=True)
