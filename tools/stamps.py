#!/usr/bin/env python3
"""Diagnostic: where do the sampler and the full-BP kernel spend their cycles?  Uses the stamped build
(make -C fl_scaling_sc_ldpc_amd/csrc stamps → libscldpc_hip_stamps.so): lane 0 of wave 0 of every workgroup
sums s_memtime deltas per phase.  Shares, not absolute times, are what to read (stamps cost cycles)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SCLDPC_LIB_PATH"] = os.path.join(ROOT, "fl_scaling_sc_ldpc_amd", "libscldpc_hip_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fl_scaling_sc_ldpc_amd import engine as E

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
adj16 = "--adj32" not in sys.argv
p = E.make_params(4, 8, 50, 1000)
L = E.lib()
L.scldpc_debug_set_stamps.argtypes = [C.c_void_p]
buf = torch.zeros((T, 16), dtype=torch.int64, device="cuda")
assert L.scldpc_debug_set_stamps(buf.data_ptr()) == 0


def show(title, names):
    torch.cuda.synchronize()
    s = buf.cpu().numpy().astype(np.float64)
    tot = s.sum(axis=1)
    print(f"{title}: mean {tot.mean():.0f} cycles/workgroup  (min {tot.min():.0f}, max {tot.max():.0f})")
    for k, nm in enumerate(names):
        print(f"   {nm:28s} {s[:, k].mean():12.0f}  {100 * s[:, k].sum() / tot.sum():5.1f} %")
    buf.zero_()


d_adj, d_ch = E.sample_philox(p, 1, 0, T, 0.48, adj16=adj16)
show("sample_philox_kernel", ["clear", "keys+histogram", "scan", "group", "rank", "emit", "channel"])
out = E.full_bp(p, d_adj, d_ch)
show("full_bp_kernel", ["clear+channel", "build", "release", "wave reductions", "barrier wait", "bookkeeping",
                        "final+expurgation"])
out2 = E.full_bp_fixpoint(p, d_adj, d_ch)
show("full_bp_fixpoint_kernel", ["channel+build", "peeling (rounds + barrier-free phase)", "final+expurgation"])
os.environ["SCLDPC_SAMPLER_GEN"] = "2"
a2, cn2, ch2 = E.sample_philox_cn16(p, 1, 0, T, 0.48)
show("sample_philox_v2_kernel", ["emit(prev)", "keys+histogram", "scan", "classify", "rank+stage+clear", "emit", "channel"])
os.environ["SCLDPC_SAMPLER_GEN"] = "3"
a2, cn2, ch2 = E.sample_philox_cn16(p, 1, 0, T, 0.48)
show("sample_philox_v3_kernel", ["(loop top)", "A: worklist of p-1 + clear", "A: keys + histogram", "barrier wait A", "B: scan + copy-out",
                                 "barrier wait B", "C: classify", "barrier wait C", "channel"])
out3 = E.full_bp_fixpoint_cn16(p, a2, cn2, ch2)
show("full_bp_small_kernel", ["channel+build", "peeling", "final+expurgation"])
if "--pick" in sys.argv:                        # BASELINE config 3: where does a pick's time go?  (8192 single-wave trials = all wave slots)
    pp = E.make_params(4, 8, 50, 10000)
    NT = 8192
    buf = torch.zeros((NT, 16), dtype=torch.int64, device="cuda")
    assert L.scldpc_debug_set_stamps(buf.data_ptr()) == 0
    da, dc = E.sample_philox(pp, 1, 0, NT, 0.48, adj16=True)
    os.environ["SCLDPC_DEBUG_PICK_TPW"] = "1"   # the stamps sit in the one-trial-per-wave kernel
    buf.zero_()
    E.peel_pick(pp, da, dc, pp.cns_pos * 50, 290000, seed=1, trial0=0, want_r1=True)
    show("peel_pick_kernel (290 000 picks per trial)", ["loop end: r1 store", "the draw (Philox once per four)", "rank-select incl. the bitmap words' trip",
                                                     "the CN word's trip", "the VN row's trip", "returning atomics' trip + bitmap updates"])
    del da, dc
    buf = torch.zeros((T, 16), dtype=torch.int64, device="cuda")
    assert L.scldpc_debug_set_stamps(buf.data_ptr()) == 0
out4 = E.full_bp_cn16(p, a2, cn2, ch2)
show("full_bp_small_kernel<LEVEL>", ["channel+build", "tail", "final+expurgation", "-", "this wave's releases (queue, two gathers, atomics, append)",
                                   "reductions", "barrier wait", "bookkeeping"])
if "--stream" in sys.argv:                      # BASELINE config 5: where does a decoded position's time go?
    ps = E.make_params(4, 8, 50, 5000)
    NS = 2048
    buf = torch.zeros((NS, 16), dtype=torch.int64, device="cuda")
    assert L.scldpc_debug_set_stamps(buf.data_ptr()) == 0
    st = E.Streams(ps, NS, seed=1, eps=0.485, W=20, doped=(10, 11, 12))
    st.run(16)
    torch.cuda.synchronize()
    buf.zero_()
    st.run(16)
    show("stream_gen_kernel + stream_dec_kernel (16 positions; sums over both kernels of a stream)", ["(loop top)", "window frontier", "window rounds", "decision + expurgation",
                                             "generate: straddlers' ranks + inverse row", "generate: copy-out + channel", "generate: wiring",
                                             "ranking: draw + count", "ranking: scan", "ranking: classify + stage"])
it = out["counters"][:, 5].float().mean().item()
print("mean iterations", it)
