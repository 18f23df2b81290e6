#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — per-iteration statistics of the reference's PUBLISHED BP trajectories
(sim_data/trajectories_bp_decoding/: 2 x 200 files x 500 frames, 956 MB of text written by bp_traj, BPT:988,1051,1145),
condensed into tests/golden/published/bp_trajectories_<tag>.npz for tests/test_gpu_published_curves.py:

  n_t[t]            frames that have a row for iteration t (a frame of k iterations has rows 0..k-1)
  sum / sumsq [c,t] sum and sum of squares over those frames of column c: 0 deg_1_iter, 1 recovered, 2 first erased position
                    (the published L50_M2500 files have no third column: an older 3-column build, NB cell 40)
  frames, meta      number of frames; ensemble, eps, MAX_IT, truncated

    python oracle/make_golden_published_traj.py          (needs /root/reference; reads data files only)
"""
import glob
import json
import os
import re

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(os.environ.get("SCLDPC_REFERENCE", "/root/reference"), "sim_data", "trajectories_bp_decoding")


def condense(pattern, tag):
    files = sorted(glob.glob(os.path.join(SRC, pattern)))
    m = re.match(r"trajectories_([0-9.]+)_(terminated|truncated)_SC_LDPC_(\d+)_(\d+)_L(\d+)_M(\d+)_BP_Full_(\d+)it_", os.path.basename(files[0]))
    eps, kind, dv, dc, L, M, max_it = float(m.group(1)), m.group(2), *(int(m.group(i)) for i in range(3, 8))
    n_t = np.zeros(max_it + 1, dtype=np.int64)
    s1 = np.zeros((3, max_it + 1), dtype=np.float64)
    s2 = np.zeros((3, max_it + 1), dtype=np.float64)
    frames, ncols = 0, None
    for path in files:
        df = pd.read_csv(path, sep="\t", header=None, skip_blank_lines=False)
        a = df.to_numpy(dtype=np.float64)
        ncols = a.shape[1]
        rows = a[~np.isnan(a[:, 0])]
        it = rows[:, 0].astype(np.int64)
        frames += int((it == 0).sum())
        n_t += np.bincount(it, minlength=max_it + 1)[:max_it + 1]
        for c in range(ncols - 1):
            v = rows[:, 1 + c]
            s1[c] += np.bincount(it, weights=v, minlength=max_it + 1)[:max_it + 1]
            s2[c] += np.bincount(it, weights=v * v, minlength=max_it + 1)[:max_it + 1]
    meta = dict(eps=eps, is_term=int(kind == "terminated"), dv=dv, dc=dc, L=L, cns_pos=M, vns_pos=M * dc // dv, max_it=max_it,
                columns=ncols, files=len(files), source="sim_data/trajectories_bp_decoding/" + pattern)
    out = os.path.join(ROOT, "tests", "golden", "published", f"bp_trajectories_{tag}.npz")
    np.savez_compressed(out, n_t=n_t, sum=s1, sumsq=s2, frames=np.int64(frames), meta=json.dumps(meta))
    last = int(np.flatnonzero(n_t)[-1])
    print(tag, meta, "frames", frames, "longest", last + 1, "mean deg1 at t=0,50:", s1[0, 0] / n_t[0], s1[0, 50] / max(1, n_t[50]))


if __name__ == "__main__":
    condense("trajectories_0.4550_truncated_SC_LDPC_4_8_L100_M500_BP_Full_1000it_Random_BLER_*.dat", "L100_M500_e4550_trunc_1000it")
    condense("trajectories_0.4600_truncated_SC_LDPC_4_8_L50_M2500_BP_Full_500it_Random_BLER_*.dat", "L50_M2500_e4600_trunc_500it")
