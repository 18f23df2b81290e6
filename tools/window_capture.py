#!/usr/bin/env python3
"""CPU (numpy): how wide is the zone in which full BP works at a time?  — the upper bound of what a position-windowed decoder
(the VN rows and CN rows of the positions around the two decoding waves staged in LDS, gathers only outside)
could save on BASELINE config 2, (4,8) L = 50 N = 1000 eps = 0.48.

Flooding peeling of random trials of the Olmos ensemble (BPF:1656-1761; any uniform permutations do: the statistic is the
ensemble's, not a key stream's).  At the start of every iteration pL / pR = the leftmost / rightmost VN position that still holds
an erased VN; a window of W positions per wave stages VN positions [pL, pL+W) and (pR-W, pR] and the CN positions their edges
reach, [pL, pL+W+3) and (pR-W, pR+3].  A release = one degree-1 CN resolving one VN = the decoder's two gathers (the CN's row of
dc VN ids, 16 B; the VN's row of dv CN ids, 8 B).  Printed: the share of those gathers that fall inside the staged positions, the
LDS the stage needs next to the 23 KB of counts, bits and queues, and the 128-byte lines a trial would then request: the build's
pass over all VN rows (3.1 k lines, as today) + every position staged once, coalesced (6.4 k) + the gathers that still leave the
window at today's measured rate (profiles/r03_pmc.json: 24.8 k lines per trial = 3.1 k of the build + 21.7 k for the 2 x 13.5 k
gathers of the releases, 0.81 line per gather).

    python tools/window_capture.py [trials] > profiles/r03_window_capture.txt"""
import sys

import numpy as np

DV, DC, L, N, EPS = 4, 8, 50, 1000, 0.48
C = N * DV // DC
D = L + DV - 1


def sample(rs):
    vn = np.empty((L * N, DV), dtype=np.int64)
    cn = np.full((D * C, DC), -1, dtype=np.int64)
    for p in range(D):
        perm = rs.permutation(DV * N)                       # rank of socket s
        c = p * C + perm // DC
        s = np.arange(DV * N)
        t, i = s // DV, s % DV
        q = p - i
        ok = (q >= 0) & (q < L)
        vn[q[ok] * N + t[ok], i[ok]] = c[ok]
    # CN -> VN table by a stable sort of the edges
    e_c = vn.reshape(-1)
    e_v = np.repeat(np.arange(L * N), DV)
    o = np.argsort(e_c, kind="stable")
    e_c, e_v = e_c[o], e_v[o]
    first = np.searchsorted(e_c, np.arange(D * C))
    slot = np.arange(len(e_c)) - first[e_c]
    cn[e_c, slot] = e_v
    return vn, cn


def run(trials, seed=1):
    rs = np.random.RandomState(seed)
    WS = list(range(1, 13))
    hit_v = np.zeros(len(WS)); hit_c = np.zeros(len(WS))
    rel = 0; iters = 0; failed = 0; width = []
    for _ in range(trials):
        vn, cn = sample(rs)
        er = rs.random_sample(L * N) < EPS
        cnt = np.bincount(vn[er].reshape(-1), minlength=D * C)
        while True:
            ones = np.nonzero(cnt == 1)[0]
            if len(ones) == 0:
                break
            nb = cn[ones]                                   # [k, dc]
            alive = (nb >= 0) & er[np.clip(nb, 0, None)]
            j = nb[np.arange(len(ones)), alive.argmax(1)]
            j, idx = np.unique(j, return_index=True)        # two CNs may name the same VN: one release
            cpos = ones[idx] // C
            pos_alive = np.nonzero(er.reshape(L, N).any(1))[0]
            pL, pR = pos_alive[0], pos_alive[-1]
            vpos = j // N
            width.append(vpos.max() - vpos.min() + 1)
            for k, W in enumerate(WS):
                hit_v[k] += ((vpos < pL + W) | (vpos > pR - W)).sum()
                hit_c[k] += ((cpos < pL + W + 3) | (cpos > pR - W)).sum()
            rel += len(j); iters += 1
            er[j] = False
            np.subtract.at(cnt, vn[j].reshape(-1), 1)
        failed += er.any()
    return WS, hit_v / rel, hit_c / rel, rel / trials, iters / trials, failed, np.array(width)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    WS, hv, hc, rel, iters, failed, width = run(trials)
    print(f"({DV},{DC}) L={L} N={N} eps={EPS}: {trials} trials, {rel:.0f} releases and {iters:.0f} flooding iterations per trial, "
          f"{failed} trials left with erasures (FER 0.83 at this point: profiles/r03_bench_C2.json)")
    print(f"span of VN positions released in ONE iteration: mean {width.mean():.1f}, median {np.median(width):.0f}, "
          f"90 % {np.percentile(width, 90):.0f}, max {width.max()}")
    print("W = staged VN positions per wave | VN-row gathers inside | CN-row gathers inside | LDS for the stage | lines requested per trial")
    row = 8 * N / 1024                                       # KB of VN rows per position = KB of CN rows per position
    build = L * N * 8 / 128
    per_gather = (24.8e3 - build) / (2 * rel)
    for W, a, b in zip(WS, hv, hc):
        lds = 2 * (W + W + 3) * row
        staged_lines = (L * N * 8 + D * C * 16) / 128        # every position staged once, coalesced
        lines = build + staged_lines + per_gather * rel * ((1 - a) + (1 - b))
        fit = int(160 // (lds + 23))
        print(f"  W = {W:2d} | {100 * a:5.1f} % | {100 * b:5.1f} % | {lds:6.0f} KB = {fit} trial(s) per CU (today 7) | {lines / 1e3:5.1f} k"
              f"  (today 24.8 k)")

if __name__ == "__main__":
    main()
