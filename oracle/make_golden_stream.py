#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — tests/golden/stream_*.npz from the REAL reference's streaming mode
(oracle/_ref/ref_stream_*: the reference source with its `#undef CIRCULAR` line dropped on a pipe, wrapped by
oracle/ref_stream_tail.c).  Per decoded position: the value decodeBP_SW_circular returned and the eight running
counters of main_streaming (BPF:2015-2046); for small ensembles also the decided position's VNerased."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(O.HERE), "tests", "golden")
# name, M, L, P, seed, eps, W, doped
SETS = [
    ("tiny_e450_W6", 5, 20, 200, 3, 0.45, 6, ()),
    ("tiny_e470_W7_dop56", 5, 20, 200, 4, 0.47, 7, (5, 6)),
    ("tiny_e500_W4_dop9", 5, 20, 160, 5, 0.50, 4, (9,)),
    ("tiny_e300_W6", 5, 20, 120, 6, 0.30, 6, ()),
    ("mid_e470_W10", 50, 30, 150, 7, 0.47, 10, ()),
    ("mid_e490_W12_dop101112", 50, 30, 150, 8, 0.49, 12, (10, 11, 12)),
    ("c5_e470_W20", 500, 50, 80, 1, 0.47, 20, ()),
    ("c5_e485_W20_dop101112", 500, 50, 100, 2, 0.485, 20, (10, 11, 12)),
    # BASELINE config 5 at its own size, Def_M = 2500 (N = 5000): 64 decoded positions = 89 generated ones, so the circular
    # buffer of L = 50 positions wraps around (about a minute of the reference each)
    ("c5n5000_e480_W20", 2500, 50, 64, 11, 0.48, 20, ()),
    ("c5n5000_e485_W20_dop101112", 2500, 50, 64, 12, 0.485, 20, (10, 11, 12)),
]


def main():
    O.build(with_reference=True)
    only = set(sys.argv[1:])                        # optional: regenerate just the named sets
    for name, M, L, P, seed, eps, W, doped in SETS:
        if only and name not in only:
            continue
        rows = O.run_ref_stream(M, L, P, seed, eps, W, doped, dump=(M <= 50))
        out = {"rows": np.array([[r[f] for f in O.Stream.FIELDS] for r in rows], dtype=np.int32),
               "meta": np.array(json.dumps(dict(name=name, Def_M=M, L=L, P=P, seed=seed, eps=eps, W=W, doped=list(doped),
                                                fields=list(O.Stream.FIELDS), generator="oracle/make_golden_stream.py",
                                                source=f"real reference, streaming build oracle/_ref/ref_stream_M{M}_L{L}")))}
        if M <= 50:
            V = 2 * M
            out["erased"] = np.stack([r.get("erased", np.zeros(V, np.uint8)) for r in rows]).astype(np.uint8)
        np.savez_compressed(os.path.join(GOLDEN, f"stream_{name}.npz"), **out)
        print(name, len(rows), flush=True)


if __name__ == "__main__":
    main()
