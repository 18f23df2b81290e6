#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerate tests/golden/*.npz from the REAL reference.

Runs the reference driver binaries of oracle/_ref/ (the reference's own C simulators compiled from
/root/reference by oracle/Makefile + oracle/ref_driver_tail.c) on fixed seeds and stores their
outputs as plain-data fixtures: per-trial counters, FNV-1a digests of the sampled graph / channel /
residual erasure pattern, per-iteration trajectory rows, and — for tiny ensembles — the arrays
themselves.  Only runs in the container that holds /root/reference; the fixtures are committed.

    python oracle/make_golden.py [--only NAME_SUBSTRING] [--jobs 8]

glibc 2.35 / gcc 11.4 produced the committed files (recorded in each file's `meta`).
"""
import argparse
import concurrent.futures as cf
import json
import os
import platform
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(O.HERE), "tests", "golden")

# name, variant, M, L, T, seed0, eps, kwargs
EPS_TINY = (0.30, 0.42, 0.48, 0.55)
SETS = []
for e in EPS_TINY:
    tag = f"{int(round(e * 1000)):03d}"
    SETS += [
        (f"tiny_bpf_M5_L10_e{tag}", "bpf", 5, 10, 48, 100, e, dict(dump=True)),
        (f"tiny_bpf_M5_L10_e{tag}_it3", "bpf", 5, 10, 32, 300, e, dict(dump=True, max_it=3)),
        (f"tiny_bpt_M5_L10_e{tag}_term", "bpt", 5, 10, 32, 500, e, dict(dump=True, is_term=1)),
        (f"tiny_bpt_M5_L10_e{tag}_trunc", "bpt", 5, 10, 32, 700, e, dict(dump=True, is_term=0)),
        # bp_traj with a binding iteration cap (the published files are 500it / 1000it; BPT:1076, 2116)
        (f"tiny_bpt_M5_L10_e{tag}_term_it3", "bpt", 5, 10, 32, 1500, e, dict(dump=True, is_term=1, max_it=3)),
        (f"tiny_bpw_M5_L10_e{tag}_W4_it3_init7", "bpw", 5, 10, 32, 900, e, dict(dump=True, W=4, max_it=3, init_it=7)),
        (f"tiny_bpw_M5_L10_e{tag}_W3_it2", "bpw", 5, 10, 32, 1100, e, dict(dump=True, W=3, max_it=2, init_it=0)),
        (f"tiny_bpfsw_M5_L10_e{tag}_W4_it3", "bpfsw", 5, 10, 32, 1300, e, dict(dump=True, W=4, max_it=3)),
        # Def_M=3: size-2 stopping sets are frequent
        (f"ss2_bpf_M3_L10_e{tag}", "bpf", 3, 10, 400, 2000, e, dict(dump=False)),
        (f"ss2_bpw_M3_L10_e{tag}_W4_it5_init9", "bpw", 3, 10, 400, 3000, e, dict(dump=False, W=4, max_it=5, init_it=9)),
    ]
for e in (0.44, 0.47, 0.49):
    tag = f"{int(round(e * 1000)):03d}"
    SETS += [
        (f"mid_bpf_M50_L20_e{tag}", "bpf", 50, 20, 48, 1, e, {}),
        (f"mid_bpt_M50_L20_e{tag}_trunc", "bpt", 50, 20, 24, 200, e, dict(is_term=0)),
        (f"mid_bpt_M50_L20_e{tag}_term", "bpt", 50, 20, 24, 400, e, dict(is_term=1)),
        (f"mid_bpw_M50_L20_e{tag}_W6_it4_init12", "bpw", 50, 20, 32, 600, e, dict(W=6, max_it=4, init_it=12)),
        (f"mid_bpfsw_M50_L20_e{tag}_W6_it5", "bpfsw", 50, 20, 24, 800, e, dict(W=6, max_it=5)),
    ]
# BASELINE.json config size: (4,8), L=50, N=1000 (Def_M=500).  ≈2–4 s per frame in the reference.
SETS += [
    ("c2_bpf_M500_L50_e480_a", "bpf", 500, 50, 32, 1, 0.48, {}),
    ("c2_bpf_M500_L50_e480_b", "bpf", 500, 50, 32, 33, 0.48, {}),
    ("c2_bpf_M500_L50_e450", "bpf", 500, 50, 16, 1000, 0.45, {}),
    ("c2_bpf_M500_L50_e470", "bpf", 500, 50, 12, 2000, 0.47, {}),
    ("c2_bpf_M500_L50_e490", "bpf", 500, 50, 12, 3000, 0.49, {}),
    ("c2_bpf_M500_L50_e480_it100", "bpf", 500, 50, 12, 4000, 0.48, dict(max_it=100)),
    ("c2_bpt_M500_L50_e480_term", "bpt", 500, 50, 8, 5000, 0.48, dict(is_term=1)),
    ("c2_bpt_M500_L50_e460_trunc", "bpt", 500, 50, 8, 6000, 0.46, dict(is_term=0)),
    ("c2_bpt_M500_L50_e480_term_it100", "bpt", 500, 50, 8, 6500, 0.48, dict(is_term=1, max_it=100)),
    ("mid_bpt_M50_L20_e470_trunc_it5", "bpt", 50, 20, 24, 6600, 0.47, dict(is_term=0, max_it=5)),
    ("mid_bpt_M50_L20_e490_term_it1", "bpt", 50, 20, 24, 6700, 0.49, dict(is_term=1, max_it=1)),
    ("c2_bpw_M500_L50_e465_W20_it6_init60", "bpw", 500, 50, 8, 7000, 0.465, dict(W=20, max_it=6, init_it=60)),
    ("c2_bpw_M500_L50_e470_W10_it20", "bpw", 500, 50, 6, 8000, 0.47, dict(W=10, max_it=20, init_it=0)),
    # bp_traj's shipped size (Def_M = 2500, BPT:25) and BASELINE config 4 (L=100, N=2000): beyond the LDS-resident kernels
    ("c3_bpt_M2500_L50_e460_trunc", "bpt", 2500, 50, 2, 9000, 0.46, dict(is_term=0)),
    ("c3_bpt_M2500_L50_e470_term", "bpt", 2500, 50, 2, 9100, 0.47, dict(is_term=1)),
    ("c4_bpw_M1000_L100_e470_W10_it20", "bpw", 1000, 100, 3, 9200, 0.47, dict(W=10, max_it=20, init_it=0)),
    # whole-run replay: no re-seeding between frames, perm_code and RNG stream carry over (BPF:2117-2144)
    ("c2_bpf_M500_L50_e480_wholerun", "bpf", 500, 50, 6, 7, 0.48, dict(whole_run=True)),
    ("tiny_bpf_M5_L10_e480_wholerun", "bpf", 5, 10, 200, 7, 0.48, dict(whole_run=True)),
    # bp_traj runs (one srandom, frames back to back) with a binding MAX_IT: what `bp_traj INDEX 0 0 MAX_IT IS_TERM` writes
    ("tiny_bpt_M5_L10_e460_term_it3_wholerun", "bpt", 5, 10, 20, 77, 0.46, dict(whole_run=True, max_it=3, is_term=1)),
    ("mid_bpt_M50_L20_e460_trunc_it6_wholerun", "bpt", 50, 20, 12, 78, 0.46, dict(whole_run=True, max_it=6, is_term=0)),
]


def make_one(spec):
    name, variant, M, L, T, seed0, eps, kw = spec
    t0 = time.time()
    hdr, trials, run = O.run_ref(variant, M, L, T, seed0, eps, **kw)
    keys = ("seed", "nch", "ne", "p1", "be", "ee", "bee")
    out = {k: np.array([tr[k] for tr in trials], dtype=np.int64) for k in keys}
    for k in ("hg", "hc", "he"):
        out[k] = np.array([tr[k] for tr in trials], dtype=np.uint64)
    if "rows" in trials[0]:
        lens = np.array([len(tr["rows"]) for tr in trials], dtype=np.int64)
        out["rows_off"] = np.concatenate([[0], np.cumsum(lens)])
        cat = np.concatenate([tr["rows"] for tr in trials])
        out["rows"] = np.stack([cat["deg1"], cat["recovered"], cat["first_pos"]], axis=1).astype(np.int32)
    if "vn_adj" in trials[0]:
        out["vn_adj"] = np.stack([tr["vn_adj"] for tr in trials]).astype(np.int32)
        out["cn_deg"] = np.stack([tr["cn_deg"] for tr in trials]).astype(np.int32)
        out["chan"] = np.stack([tr["chan"] for tr in trials]).astype(np.uint8)
        out["erased"] = np.stack([tr["erased"] for tr in trials]).astype(np.uint8)
    meta = dict(hdr)
    meta.update(name=name, variant=variant, Def_M=M, whole_run=bool(kw.get("whole_run", False)),
                libc=" ".join(platform.libc_ver()), generator="oracle/make_golden.py",
                source="real reference via oracle/_ref/ref_%s_M%d_L%d" % (variant, M, L))
    if run:
        meta["run_counters"] = run
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    return name, len(trials), time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    O.build(with_reference=True)
    os.makedirs(GOLDEN, exist_ok=True)
    todo = [s for s in SETS if a.only in s[0]]
    todo.sort(key=lambda s: -s[2] * s[3] * s[4])      # long jobs first
    with cf.ProcessPoolExecutor(a.jobs) as ex:
        for name, nt, dt in ex.map(make_one, todo):
            print(f"{name}: {nt} trials in {dt:.1f}s", flush=True)


if __name__ == "__main__":
    main()
