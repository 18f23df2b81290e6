#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerate tests/golden/pd_var_*.npz by IMPORTING the real reference: the variance workflow
of the peeling simulator:

  * fl_scaling/est_scaling_params.py: calc_nu_chunk (:90-94) → calc_var_chunk (:131-138) on fixed trajectory arrays and
    synthetic theory curves (the shipped theory file is a git-LFS pointer);
  * simulators_sc_ldpc/peeling_decoding/peeling_decoding.py: main_simulate_variance (PD:1264-1294) end to end on small
    ensembles — sys.argv as simulate_variance.py passes it, `np.random.seed(s); random.seed(s)` first, a synthetic theory
    pickle in a temporary directory — storing the (ssquares, counts) it pickles.

Only runs in the container that holds /root/reference; the fixtures (inputs + outputs, no reference text) are committed.

    MPLBACKEND=Agg python oracle/make_golden_var.py
"""
import json
import os
import pickle
import platform
import random
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/simulators_sc_ldpc/peeling_decoding")
from fl_scaling import est_scaling_params as esp      # noqa: E402  (the real reference)
import peeling_decoding as pd                          # noqa: E402

pd.trange = range


def meta(**kw):
    kw.update(numpy=np.__version__, python=platform.python_version(), generator="oracle/make_golden_var.py")
    return np.array(json.dumps(kw))


def chunk_cases():
    rng = np.random.RandomState(7)
    # (name, r1s, theory, M)
    r1 = rng.randint(0, 300, size=(37, 500)).astype(np.int64)
    r1[:, 400:] = 0
    r1[rng.rand(37, 500) < 0.05] = 0
    th = np.concatenate([np.linspace(250, 3, 380), np.zeros(120)])
    yield "chunk_tail0", r1, th, 1000
    r2 = rng.randint(0, 2000, size=(8, 1201)).astype(np.int64)
    r2[3, 700:] = 0
    th2 = 900.0 * np.exp(-np.arange(1201) / 400.0) + 0.25            # positive everywhere: nothing is cropped
    yield "chunk_full", r2, th2, 10000
    r3 = rng.randint(0, 50, size=(5, 64)).astype(np.int64)
    th3 = np.linspace(40, 1, 90)                                       # theory longer than the trajectories (est…py:93)
    th3[70:] = 0
    yield "chunk_short_rows", r3, th3[:64], 20


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    for name, r1, th, M in chunk_cases():
        ss, cnt = esp.calc_nu_chunk(r1.copy(), th.copy(), M)
        np.savez_compressed(os.path.join(GOLDEN, f"pd_var_{name}.npz"), r1s=r1, theory=th, M=np.array(M),
                            ssquares=np.asarray(ss, dtype=np.float64), counts=np.asarray(cnt, dtype=np.int64),
                            meta=meta(kind="calc_nu_chunk", name=name))
        print("pd_var_" + name, ss.shape, flush=True)
    # (name, l, r, L, M, e, T|N, num_runs, num_runs_batch, seed)
    for name, l, r, L, M, e, tflag, runs, batch, seed in [("main_tiny_N", 4, 8, 10, 20, 0.45, "N", 6, 3, 11),
                                                          ("main_tiny_T", 4, 8, 10, 20, 0.48, "T", 4, 2, 12),
                                                          ("main_mid_N", 4, 8, 20, 200, 0.47, "N", 4, 2, 13)]:
        term = tflag == "T"
        steps = int(M * (L + l - 1 if term else L) * (e + 0.1))
        # synthetic theory curve: smooth, positive on a prefix, zero afterwards (like the mean-evolution output)
        k = np.arange(steps + 1)
        cut = int(0.8 * steps)
        th = np.where(k < cut, 0.05 * M * (1.0 + np.cos(np.pi * k / cut)) + 1.5, 0.0)
        with tempfile.TemporaryDirectory() as d:
            fth, fout = os.path.join(d, "theory.pkl"), os.path.join(d, "out.pkl")
            with open(fth, "wb") as f:
                pickle.dump((th,), f)
            argv = ["simulate_variance.py", fout, str(l), str(r), str(L), str(M), repr(e), tflag, "U", str(runs), str(batch), fth]
            old = sys.argv
            sys.argv = argv
            np.random.seed(seed); random.seed(seed)
            pd.main_simulate_variance()
            after = (float(np.random.rand()), random.random())
            sys.argv = old
            with open(fout, "rb") as f:
                ss, cnt = pickle.load(f)                   # our own file, written a moment ago by the reference
        np.savez_compressed(os.path.join(GOLDEN, f"pd_var_{name}.npz"), theory=th, ssquares=np.asarray(ss, np.float64),
                            counts=np.asarray(cnt, np.int64), after=np.array(after),
                            meta=meta(kind="main_simulate_variance", name=name, argv=argv[2:-1], seed=seed))
        print("pd_var_" + name, ss.shape, int(cnt.sum()), flush=True)


if __name__ == "__main__":
    main()
