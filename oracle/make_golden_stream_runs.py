#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — tests/golden/stream_wholeruns.json: whole runs of the REAL reference's streaming program
(main_streaming, BPF:1934-2054; oracle/_ref/ref_stream_* in `run` mode, oracle/ref_stream_tail.c): ONE srandom(seed), the
ε points back to back with random() carried over, every point stopped by main_streaming's own rule (BPF:2033).  Per point
the eight arguments of results_circular — what `sw INDEX W NUM_DOPED … --rng glibc --seed S` must write row for row."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(O.HERE), "tests", "golden", "stream_wholeruns.json")
# Def_M, L, seed, W, npoints, eps_ini, eps_delta, max_blocks_err, max_blocks, doped
RUNS = [
    (5, 20, 5, 6, 4, 0.5, 0.02, 30, 2000, (5, 6)),
    (5, 20, 77, 7, 3, 0.47, 0.015, 25, 300, ()),
    (50, 30, 21, 10, 3, 0.49, 0.01, 12, 400, ()),
    (50, 30, 8, 12, 2, 0.49, 0.005, 40, 260, (10, 11, 12)),
    (500, 50, 3, 20, 2, 0.485, 0.005, 6, 70, (10, 11, 12)),
]


def main():
    O.build(with_reference=True)
    runs = []
    for M, L, seed, W, npts, e0, de, mbe, mb, doped in RUNS:
        cmd = [os.path.join(O.REF_DIR, f"ref_stream_M{M}_L{L}"), "run", str(seed), str(W), str(npts), repr(e0), repr(de),
               str(mbe), str(mb), str(len(doped))] + [str(d) for d in doped]
        txt = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
        rows = []
        for ln in txt.split("\n"):
            if ln.startswith("ROW "):
                kv = dict(tok.split("=") for tok in ln.split()[1:])
                rows.append({"eps": float(kv["eps"]), "pos": int(kv["pos"]),
                             "counters": [int(kv[k]) for k in ("ne", "be", "ee", "bee", "gb", "gbl", "gbe", "gble")]})
        assert len(rows) == npts
        runs.append(dict(Def_M=M, L=L, seed=seed, W=W, num_points=npts, eps_ini=e0, eps_delta=de, max_blocks_err=mbe,
                         max_blocks=mb, doped=list(doped), rows=rows))
        print(M, L, seed, [r["pos"] for r in rows], flush=True)
    json.dump({"generator": "oracle/make_golden_stream_runs.py", "source": "real reference, oracle/_ref/ref_stream_* run mode",
               "counter_order": ["num_erasures", "num_blocks_err", "num_erasures_exp", "num_blocks_err_exp", "num_bits_generated",
                                 "num_blocks_generated", "num_bits_generated_exp", "num_blocks_generated_exp"],
               "runs": runs}, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
