"""A real C program against libscldpc_hip.so (-m gpu): tests/c_caller/frame_loop.c — the frame loop of main_terminated
(BPF:2117-2144) that INTEGRATION.md §2a shows — is compiled with gcc as plain C99 (no HIP compiler, no torch), linked
against the shared library and run; its risultati row and its view of the struct / enum layout must equal the Python
binding's."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, require_gpu

pytestmark = pytest.mark.gpu


def test_c_frame_loop_equals_python_binding(tmp_path):
    require_gpu()
    import ctypes as C
    import torch
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    from fl_scaling_sc_ldpc_amd import engine as E
    from fl_scaling_sc_ldpc_amd import _lib
    exe = str(tmp_path / "frame_loop")
    libdir = os.path.join(ROOT, "fl_scaling_sc_ldpc_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c_caller", "frame_loop.c"),
                    "-L", libdir, "-lscldpc_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-o", exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    for L, N, eps, max_it, frames, ferr, seed, batch in ((20, 200, 0.47, 1000000, 300, 40, 7, 64),
                                                         (50, 1000, 0.48, 60, 100, 1000, 11, 48),
                                                         (50, 5000, 0.46, 1000000, 24, 5, 3, 16)):   # workspace needed
        r = subprocess.run([exe, str(L), str(N), repr(eps), str(max_it), str(frames), str(ferr), str(seed), str(batch)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        row, layout = r.stdout.strip().split("\n")
        # the same loop through the Python binding
        p = E.make_params(4, 8, L, N)
        run = E.new_run()
        for f in range(0, frames, batch):
            nb = min(batch, frames - f)
            d_adj, d_ch = E.sample_philox(p, seed, f, nb, eps)
            out = E.full_bp(p, d_adj, d_ch, max_it=max_it)
            E.accumulate_run(out["counters"], run, ferr)
            if int(run[1].item()) >= ferr:
                break
        pt = B.PointResult(eps, p.n, p.L, run.cpu().numpy())
        assert row + "\n" == pt.row(), (row, pt.row())
        assert layout.split() == ["layout", str(C.sizeof(_lib.CodeParams)), str(_lib.CodeParams.cns_pos.offset),
                                  str(_lib.CodeParams.vns_pos.offset), str(_lib.NCOUNTERS), str(_lib.NRUN),
                                  str(_lib.COUNTER_NAMES.index("channel_erasures")), str(_lib.RUN_NAMES.index("frames"))]
    assert int(np.sum([1])) == 1


def test_two_streams_with_their_own_workspaces_do_not_interfere():
    """Ensembles whose CN words live in device memory (N = 5000): the workspace is the caller's, so two calls in flight on
    two streams, each with its own buffer, give what they give one after the other (the library used to keep one hidden
    buffer per device, which two streams would have shared)."""
    require_gpu()
    import torch
    from fl_scaling_sc_ldpc_amd import engine as E
    p = E.make_params(4, 8, 50, 5000)
    T = 96
    a1, c1 = E.sample_philox(p, 5, 0, T, 0.47, adj16=True)
    a2, c2 = E.sample_philox(p, 5, T, T, 0.485, adj16=True)
    ref1 = E.full_bp(p, a1, c1)["counters"].clone()
    ref2 = E.sw_bp(p, a2, c2, 10, 20)["counters"].clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            o1 = E.full_bp(p, a1, c1)["counters"]
        with torch.cuda.stream(s2):
            o2 = E.sw_bp(p, a2, c2, 10, 20)["counters"]
        torch.cuda.synchronize()
        assert torch.equal(o1, ref1) and torch.equal(o2, ref2)
    # too small a workspace is refused, not worked around
    import ctypes as C
    need = E.lib().scldpc_workspace_bytes(E.WS_FULL_BP, C.byref(p), T, 0, 0)
    assert need == T * p.nk * 4
    small = torch.empty(need - 256, dtype=torch.uint8, device="cuda")
    cnt = torch.empty((T, E.NCOUNTERS), dtype=torch.int32, device="cuda")
    rc = E.lib().scldpc_full_bp_device_adj16(C.byref(p), T, a1.data_ptr(), c1.data_ptr(), 0, 1, cnt.data_ptr(), None, 0, None,
                                             small.data_ptr(), need - 256, None)
    assert rc == -1 and b"workspace" in E.lib().scldpc_last_error()


def test_entry_points_can_be_captured_into_a_graph():
    """include/scldpc.h: device entry points only enqueue work — no allocation, no synchronisation, no hidden state — so
    a whole step (sample -> decode -> accumulate) can be captured into a hipGraph and replayed: the replays give what
    eager calls give, and a replay after new inputs were written into the same buffers sees them."""
    require_gpu()
    import torch
    from fl_scaling_sc_ldpc_amd import engine as E
    p = E.make_params(4, 8, 50, 1000)
    T = 512
    a16 = torch.empty((T, p.n, 4), dtype=torch.int16, device="cuda")
    cn16 = torch.empty((T, p.nk, 8), dtype=torch.int16, device="cuda")
    ch = torch.empty((T, p.nw), dtype=torch.int32, device="cuda")
    cnt = torch.empty((T, E.NCOUNTERS), dtype=torch.int32, device="cuda")
    run = E.new_run()

    def step():
        E.sample_philox_cn16(p, 99, 0, T, 0.48, out=(a16, cn16, ch))
        E.full_bp_fixpoint_cn16(p, a16, cn16, ch, counters=cnt)
        E.accumulate_run(cnt, run, 0)

    step()                                                  # eager reference (also loads the code objects)
    torch.cuda.synchronize()
    want_cnt, want_run = cnt.clone(), run.clone()
    run.zero_(); cnt.zero_()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        g.capture_begin()
        step()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    run.zero_(); cnt.zero_()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    keep = [0, 1, 2, 3, 4, 6, 7]
    assert torch.equal(cnt[:, keep], want_cnt[:, keep])
    assert torch.equal(run[:8], 2 * want_run[:8])           # two replays accumulated twice
