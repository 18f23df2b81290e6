#!/usr/bin/env python3
"""A/B in ONE process on one box (diagnostics): the headline's two kernels at other occupancies and as persistent launches.
  1. sampler_v2 and full_bp_small stand-alone at k workgroups per CU (extra dynamic LDS: SCLDPC_DEBUG_LDS_PAD_*);
  2. both as PERSISTENT launches (SCLDPC_DEBUG_GRID_*: gs sampler + gd decoder workgroups, each looping over its trials)
     started together on two streams — the co-residency experiment: 1 sampler (56 KB, 16 waves) + 4 decoders (23 KB, 4 waves)
     fit a CU's LDS and wave slots; the question is what the pair delivers against sampler-then-decoder.
HIP events on the launch streams; every variant is checked against the default launch's counters."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = E.make_params(4, 8, 50, 1000)
dev = "cuda:0"
bufs = [(torch.empty((B, p.n, 4), dtype=torch.int16, device=dev), torch.empty((B, p.nk, 8), dtype=torch.int16, device=dev),
         torch.empty((B, p.nw), dtype=torch.int32, device=dev)) for _ in range(2)]
cnt = torch.empty((B, 8), dtype=torch.int32, device=dev)
NCU = 256


def env(**kw):
    for k in ("SCLDPC_DEBUG_LDS_PAD_SAMPLER", "SCLDPC_DEBUG_LDS_PAD_DECODER", "SCLDPC_DEBUG_GRID_SAMPLER", "SCLDPC_DEBUG_GRID_DECODER"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ["SCLDPC_DEBUG_" + k] = str(v)


def timeit(fn, name, unit=B):
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
    print(f"{name:78s} {ms:8.3f} ms / {unit} trials = {unit / ms / 1e3:7.3f} M trials/s", flush=True)
    return ms


def sample(k, b=0):
    E.sample_philox_cn16(p, 1, k * B, B, 0.48, out=bufs[b])


def decode(b=0):
    E.full_bp_fixpoint_cn16(p, bufs[b][0], bufs[b][1], bufs[b][2], counters=cnt)


env()
s0 = timeit(lambda k: sample(1), "sampler, default (2 WG/CU)")
d0 = timeit(lambda k: decode(), "decoder, default (7 WG/CU)")
KEEP = [0, 1, 2, 3, 4, 6, 7]                     # all but the barrier-round count (not an output of the reference)
ref = cnt[:, KEEP].clone()
print(f"serial step: {s0 + d0:.2f} ms -> {B / (s0 + d0) / 1e3:.3f} M trials/s", flush=True)

# ---- 1. stand-alone at k workgroups per CU
env(LDS_PAD_SAMPLER=30000)
timeit(lambda k: sample(1), "sampler, 1 WG/CU (LDS padded)")
for kcu in (6, 5, 4, 3, 2):
    pad = 160 * 1024 // kcu - 23600 - 64
    env(LDS_PAD_DECODER=pad)
    cnt.zero_()
    timeit(lambda k: decode(), f"decoder, {kcu} WG/CU (LDS padded by {pad})")
    assert torch.equal(cnt[:, KEEP], ref)

# ---- 2. persistent launches, stand-alone and together
for gs in (1, 2):
    env(GRID_SAMPLER=gs * NCU)
    timeit(lambda k: sample(1), f"sampler, persistent {gs} WG/CU")
for gd in (4, 5, 6, 7):
    env(GRID_DECODER=gd * NCU)
    cnt.zero_()
    timeit(lambda k: decode(), f"decoder, persistent {gd} WG/CU")
    assert torch.equal(cnt[:, KEEP], ref)

s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
env()
sample(1, 0); sample(2, 1)
torch.cuda.synchronize()


def both(k, order):
    """decoder of buffer 0 beside the sampler of buffer 1"""
    cur = torch.cuda.current_stream()
    for s in (s_a, s_b):
        s.wait_stream(cur)
    for which in order:
        if which == "d":
            with torch.cuda.stream(s_a):
                decode(0)
        else:
            with torch.cuda.stream(s_b):
                sample(2, 1)
    cur.wait_stream(s_a)
    cur.wait_stream(s_b)


for gs, gd in ((1, 4), (1, 3), (1, 5), (2, 2), (2, 1)):
    for order in ("ds", "sd"):
        env(GRID_SAMPLER=gs * NCU, GRID_DECODER=gd * NCU)
        cnt.zero_()
        ms = timeit(lambda k: both(k, order), f"persistent pair: {gs} sampler + {gd} decoders per CU, launch order {order}")
        assert torch.equal(cnt[:, KEEP], ref)
        print(f"{'':40s}-> vs serial {s0 + d0:.2f} ms: x{(s0 + d0) / ms:.3f}", flush=True)
env()
cnt.zero_()
timeit(lambda k: both(k, "ds"), "default launches on two streams (as bench.py overlaps them)")
