#!/usr/bin/env python3
"""Soak on the GPU box: the streaming kernels' three generation paths — fused ranking in LDS (shipping), the 16-bit-counter
fallback with CN rows built in a pass of their own (SCLDPC_DEBUG_STREAM_WIDE=1) and the kernel's other LDS layout
(SCLDPC_DEBUG_STREAM_LEGACY=1) — must leave identical counters on every stream: same Philox keys, same permutations, same
decisions.  (The CPU twin and the reference fixtures pin a few streams position by position in tests/; this runs many.)
    python tools/soak_stream.py > profiles/r03_soak_stream.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from fl_scaling_sc_ldpc_amd import engine as E
N, NS, NPOS, eps, doped = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), eval(sys.argv[5])
p = E.make_params(4, 8, 50, N)
st = E.Streams(p, NS, seed=77, eps=eps, W=20, doped=doped, stream0=1 << 33)
for _ in range(NPOS // 64):
    c, _ = st.run(64)
torch.cuda.synchronize()
torch.save(c.cpu(), sys.argv[6])
''' % ROOT


def run(mode, N, NS, NPOS, eps, doped, out):
    env = dict(os.environ)
    if mode:
        env[mode] = "1"
    subprocess.run([sys.executable, "-c", CHILD, str(N), str(NS), str(NPOS), str(eps), repr(doped), out], check=True, env=env)


def main():
    import torch
    for N, NS, NPOS, eps, doped in [(1000, 4096, 1024, 0.48, (10, 11, 12)), (5000, 2048, 512, 0.485, (10, 11, 12)), (200, 4096, 2048, 0.47, ())]:
        res = {}
        for mode in ("", "SCLDPC_DEBUG_STREAM_WIDE", "SCLDPC_DEBUG_STREAM_LEGACY"):
            out = f"/tmp/soak_{N}_{mode or 'fused'}.pt"
            run(mode, N, NS, NPOS, eps, doped, out)
            res[mode or "fused"] = torch.load(out)
        ref = res["fused"]
        same = all(torch.equal(ref, v) for v in res.values())
        tot = ref[:, :8].sum(dim=0).tolist()
        print(f"N={N} eps={eps} doped={list(doped)}: {NS} streams x {NPOS} positions, three generation paths: "
              f"{'identical counters on every stream' if same else 'MISMATCH'}; totals {tot}", flush=True)
        assert same


if __name__ == "__main__":
    main()
