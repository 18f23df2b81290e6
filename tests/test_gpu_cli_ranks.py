"""The executables' mirrors as multi-rank jobs on a real device (-m gpu): launched with python -m torch.distributed.run as
INTEGRATION.md shows, two ranks — rehearsed on a one-GPU box, both ranks on cuda:0 and gloo instead of RCCL
(SCLDPC_DIST_BACKEND).  Every job must write the very file the single-process run writes: trial indices, the sharding of
ε points / frames / streams / trials, the ordered stop rule and the counter exchange do not depend on the number of ranks.
(tests/test_distributed_gloo.py checks the same logic on the CPU with the device work faked.)"""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT, require_gpu

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, module, args):
    env = dict(os.environ, SCLDPC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable]
    if world > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += ["-m", module, *args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])


def _same_files(d1, d2, expect):
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d2)) and len(names) == expect, (names, sorted(os.listdir(d2)))
    for nm in names:
        a, b = open(os.path.join(d1, nm), "rb").read(), open(os.path.join(d2, nm), "rb").read()
        assert a == b and len(a) > 0, nm


@pytest.mark.parametrize("prog,args", [
    # frames of every point split over the ranks, ordered stop at 40 frame errors; MAX_IT = 150 binds (level kernel)
    ("bp_lim_iter", ["5", "0", "0", "150", "--num-points", "2", "--max-frames", "700", "--min-frame-err", "40", "--batch", "128",
                     "--shard", "frames"]),
    # ε points over the ranks (the reference's cluster model), rank 0 appends the rows in grid order
    ("sw_lim_iter", ["2", "8", "0", "12", "0", "--L", "30", "--N", "400", "--num-points", "4", "--max-frames", "300",
                     "--min-frame-err", "30", "--batch", "128", "--shard", "points"]),
    # streaming: independent streams per rank, counters summed after every chunk, stop rule on the sums
    # (--streams is per rank: 2 x 16 streams are the 32 streams of the single process)
    ("sw", ["1", "12", "2", "6", "7", "--L", "30", "--N", "200", "--num-points", "2", "--chunk", "8", "--max-blocks", "4000",
            "--max-blocks-err", "100", "--streams"]),
])
def test_bp_programs_two_ranks_write_the_single_rank_file(tmp_path, prog, args):
    require_gpu()
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir(); d2.mkdir()
    common = ["--seed", "4242", "--quiet"]
    per_world = (["32"], ["16"]) if prog == "sw" else ([], [])
    _run(1, "fl_scaling_sc_ldpc_amd.bp_decoding", [prog, *args, *per_world[0], *common, "--outdir", str(d1)])
    _run(2, "fl_scaling_sc_ldpc_amd.bp_decoding", [prog, *args, *per_world[1], *common, "--outdir", str(d2)])
    _same_files(str(d1), str(d2), 1)


def test_bp_traj_two_ranks_write_the_files_of_index_and_index_plus_one(tmp_path):
    """bp_traj under torch.distributed: rank r is process INDEX + r of the reference's array job (BPT:2095, file name
    BPT:2131-2134; NB cell 35:21) — the two-rank job leaves exactly the two files that two single runs leave."""
    require_gpu()
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir(); d2.mkdir()
    args = ["0", "0", "1000000", "1", "--L", "20", "--N", "200", "--max-frames", "40", "--min-frame-err", "40", "--batch", "16",
            "--eps-ini", "0.47", "--seed", "77", "--quiet"]
    _run(2, "fl_scaling_sc_ldpc_amd.bp_decoding", ["bp_traj", "3", *args, "--outdir", str(d2)])
    for idx in ("3", "4"):
        _run(1, "fl_scaling_sc_ldpc_amd.bp_decoding", ["bp_traj", idx, *args, "--outdir", str(d1)])
    _same_files(str(d1), str(d2), 2)
    a, b = (open(os.path.join(d2, nm)).read() for nm in sorted(os.listdir(d2)))
    assert a != b and a.count("\n\n") == 40 and b.count("\n\n") == 40


def test_ber_sim_two_ranks_write_the_single_rank_table(tmp_path):
    """ber_sim.py's argv (PD:1327-1356) in throughput mode: the trials of every round split over the ranks, result rows
    all-gathered, ordered stop at max_fuckups; rank 0 writes."""
    require_gpu()
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir(); d2.mkdir()
    args = ["4", "8", "20", "200", "[0.45, 0.47]", "T", "U", "B", "NTB", "600", "50", "[]", "--rng", "philox", "--seed", "99",
            "--batch", "128"]
    _run(1, "fl_scaling_sc_ldpc_amd.peeling_decoding", ["ber_sim", str(d1 / "table.dat"), *args])
    _run(2, "fl_scaling_sc_ldpc_amd.peeling_decoding", ["ber_sim", str(d2 / "table.dat"), *args])
    _same_files(str(d1), str(d2), 1)
    rows = open(d1 / "table.dat").read().strip().split("\n")
    assert rows[0].startswith("# SC-LDPC (4,8,L=20,M=200)") and len(rows) == 3
