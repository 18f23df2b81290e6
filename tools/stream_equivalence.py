#!/usr/bin/env python3
"""On the GPU box: does the block error rate of MANY SHORT streams equal that of FEW LONG ones?

The reference's streaming experiment (main_streaming, BPF:1934-2054) is ONE stream run until it has seen enough block errors;
the multi-stream driver (`sw --streams S`) advances S independent streams in lock step and sums their counters.  With doping
that decouples the chain (dv - 1 consecutive known positions per period) a stream is a sequence of independent, identically
distributed segments — the first one, which starts from the known left end, included — so both estimators see the same
process.  Without doping a stream that has failed once keeps failing (the window has lost its known left end), and the
reference's figure is errors / (time to the first failure + errors): a property of ONE stream that no sum over streams
reproduces.  This tool measures both cases; tests/test_gpu_stream.py asserts the first and documents the second.

    python tools/stream_equivalence.py [N] > profiles/r03_stream_equivalence.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fl_scaling_sc_ldpc_amd import engine as E  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L_BUF, W = 50, 20


def samples(p, nstreams, npos_total, chunk, seed, eps, doped, stream0):
    """Block errors and blocks (after expurgation: counters 3, 7) of every (stream, chunk of `chunk` positions)."""
    st = E.Streams(p, nstreams, seed=seed, eps=eps, W=W, doped=doped, stream0=stream0)
    prev = torch.zeros_like(st.counters)
    out = []
    for _ in range(npos_total // chunk):
        c, _ = st.run(chunk)
        d = (c - prev).cpu().numpy()
        prev = c.clone()
        out.append(d[:, [3, 7]])
    return np.concatenate(out, axis=0)             # [(chunks * streams), 2]


def summary(s):
    err, blk = s[:, 0].astype(np.float64), s[:, 1].astype(np.float64)
    bler = err.sum() / blk.sum()
    # standard error of the ratio from the spread of the samples (errors cluster inside a segment: no binomial shortcut)
    se = np.std(err - bler * blk, ddof=1) * np.sqrt(len(err)) / blk.sum()
    return bler, se, int(err.sum()), int(blk.sum())


def main():
    p = E.make_params(4, 8, L_BUF, N)
    print(f"(4,8) streaming ensemble N={N}, buffer L={L_BUF}, W={W}; BLER after expurgation; samples = 2000-position pieces")
    for doped, eps_list in (((10, 11, 12), (0.46, 0.47, 0.475, 0.48)), ((), (0.44, 0.45, 0.46))):
        for eps in eps_list:
            a = samples(p, 512, 2000, 2000, 11, eps, doped, 0)
            b = samples(p, 4, 256000, 2000, 11, eps, doped, 1 << 20)
            (ba, sa, ea, na), (bb, sb, eb, nb) = summary(a), summary(b)
            z = (ba - bb) / max(1e-30, np.hypot(sa, sb))
            print(f"doped={list(doped)} eps={eps}: 512 x 2000 positions BLER {ba:.5f} +- {sa:.5f} ({ea} / {na}) | "
                  f"4 x 256000 positions BLER {bb:.5f} +- {sb:.5f} ({eb} / {nb}) | z = {z:+.2f}", flush=True)
            if not doped:
                # the long streams, piece by piece: once a piece has failed, do the later ones recover?
                e = b[:, 0].reshape(-1, 4).T                   # [stream][piece]
                first = [int(np.argmax(x > 0)) if (x > 0).any() else -1 for x in e]
                after = [float((x[f:] > 0).mean()) if f >= 0 else float("nan") for x, f in zip(e, first)]
                print(f"    undoped long streams: first failing piece {first}, share of later pieces with errors {after}", flush=True)


if __name__ == "__main__":
    main()
