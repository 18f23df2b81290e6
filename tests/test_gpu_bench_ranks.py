"""bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run, one process per rank),
rehearsed on a one-GPU box: both ranks on cuda:0, gloo instead of RCCL (SCLDPC_BENCH_BACKEND).  What is checked is the
multi-rank logic of the bench — disjoint trial ranges per rank, the closing counter reduce, max-over-ranks timing, one
JSON line from rank 0 — not a throughput: the line carries config.rehearsal and is never a reported number."""
import json
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


def _bench(world, *args):
    env = dict(os.environ, SCLDPC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable]
    if world > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--no-cpu-baseline", *args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    return json.loads(lines[0])


def test_plain_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the driver's command form): the script starts the two ranks itself
    and rank 0's single line says n_gpus 2, with the error counts of one rank over the same trial indices."""
    require_gpu()
    env = dict(os.environ, SCLDPC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--steps", "2",
                        "--warmup", "1", "--batch", "1024"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    two = json.loads(lines[0])
    one = _bench(1, "--steps", "2", "--warmup", "1", "--batch", "2048")
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 0 and "rehearsal" in two["config"]     # gloo rehearsal: no RCCL
    assert one["n_gpus"] == 1 and one["rccl_ranks"] == 1
    for k in ("FER", "BLER", "BER", "FER_exp"):
        assert two["results"][k] == one["results"][k], (k, two["results"], one["results"])
    # and a launcher whose world size disagrees with --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT, capture_output=True, text=True,
                         timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_c2_two_ranks_decode_the_trials_one_rank_decodes():
    """Two ranks x B trials per step cover the trial indices of one rank x 2B: identical error counts."""
    require_gpu()
    two = _bench(2, "--steps", "3", "--warmup", "1", "--batch", "1024")
    one = _bench(1, "--steps", "3", "--warmup", "1", "--batch", "2048")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and "rehearsal" in two["config"]
    assert two["config"]["parallelism"] == "trial-sharded x2" and two["scaling"] == "weak"
    for k in ("FER", "BLER", "BER", "FER_exp"):
        assert two["results"][k] == one["results"][k], (k, two["results"], one["results"])
    assert two["value"] > 0 and abs(two["value"] * two["ms_per_step"] * 1e-3 - 2 * 1024) < 1e-6 * 2048
    assert two["roofline"]["frac"] > 0 and two["roofline"]["step"]["frac"] > 0


@pytest.mark.parametrize("config,args,units", [
    ("C3", ("--steps", "1", "--warmup", "1", "--batch", "64"), 64),
    ("C4", ("--steps", "2", "--warmup", "1", "--batch", "256"), 256),
    ("C5", ("--steps", "2", "--warmup", "1", "--batch", "16"), 16 * 16),
])
def test_other_configs_run_on_two_ranks(config, args, units):
    """The closing reduce inside run_c3 / run_c4 / run_c5 asserts that every rank's units arrived."""
    require_gpu()
    out = _bench(2, "--config", config, *args)
    assert out["n_gpus"] == 2 and out["value"] > 0
    steps = int(args[1])
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 2 * units) < 1e-6 * units, (out["value"], out["ms_per_step"])
    if config == "C3":
        assert out["results"]["trials_in_moments"] == 2 * 64 * steps
