"""bench.py's rank dispatch (host logic, no GPU): `python bench.py --gpus N` launched plainly must start N ranks itself as
a child torch.distributed.run job (the driver's command form) and relay its exit code; a --gpus that disagrees with the
launcher's WORLD_SIZE is an error, never a one-GPU line labelled otherwise."""
import importlib.util
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plain_launch_with_gpus_n_spawns_n_ranks(monkeypatch):
    bench = _bench_module()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                          # the child job's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]      # same arguments for every rank
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_must_equal_world_size_under_a_launcher():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # no line at all rather than a mislabelled one


def test_single_rank_needs_a_device_and_says_so():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
