"""The C-ABI shared library: loads, exports every symbol include/scldpc.h declares, and its host-side
entry points (exact glibc replay of generate_code / channel_doped) reproduce the golden digests.
No device compute here — the decoders are exercised by the -m gpu tests."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_names, load_golden


@pytest.fixture(scope="module")
def L():
    from fl_scaling_sc_ldpc_amd import _lib
    return _lib.lib()


def test_library_exports_every_declared_symbol(L):
    from fl_scaling_sc_ldpc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "scldpc.h")).read()
    declared = set(re.findall(r"\b(scldpc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found in include/scldpc.h"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), f"libscldpc_hip.so does not export {name}"
    assert L.scldpc_abi_version() == 2


def test_header_is_plain_c():
    """include/scldpc.h must compile as C with no HIP/torch headers (the drop-in boundary)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "scldpc.h"\nint main(void){scldpc_code_params p={4,8,50,500,1000};'
                             'return (int)sizeof(p)-20;}\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src,
                        "-o", os.path.join(d, "t")], check=True)
        assert subprocess.run([os.path.join(d, "t")]).returncode == 0


def test_error_paths_without_device(L):
    from fl_scaling_sc_ldpc_amd import _lib
    bad = _lib.CodeParams(4, 8, 50, 500, 999)              # dv*vns_pos != dc*cns_pos
    assert L.scldpc_glibc_state_bytes(C.byref(bad)) == -1
    assert b"must equal" in L.scldpc_last_error()
    ok = _lib.CodeParams(4, 8, 50, 500, 1000)
    assert L.scldpc_full_bp_lds_bytes(C.byref(ok)) <= 160 * 1024
    big = _lib.CodeParams(4, 8, 50, 5000, 10000)           # N = 10000: CN words go to the global workspace, bitmaps still fit
    assert L.scldpc_full_bp_lds_bytes(C.byref(big)) <= 160 * 1024
    huge = _lib.CodeParams(4, 8, 50, 50000, 100000)        # N = 100000: even the VN bitmap exceeds the LDS
    assert L.scldpc_full_bp_lds_bytes(C.byref(huge)) > 160 * 1024
    assert L.scldpc_full_bp_device(C.byref(ok), -1, None, None, 0, 1, None, None, 0, None, None, 0, None) == -1
    assert L.scldpc_full_bp_device(C.byref(ok), 0, None, None, 0, 1, None, None, 0, None, None, 0, None) == 0    # empty batch
    assert L.scldpc_full_bp_device(C.byref(ok), 1, None, None, 0, 1, None, None, 0, None, None, 0, None) == -1   # null buffers
    assert L.scldpc_sample_philox_device(C.byref(ok), 1, 0, 1, 1.5, 0, None, None, None, None, 0, None) == -1
    assert L.scldpc_sw_bp_device(C.byref(ok), 1, None, None, 0, 1, 1, None, None, None, 0, None) == -1
    # caller-owned workspace: what each operation needs, without a device (pure host arithmetic)
    WS = dict(sample=0, full_bp=1, sw_bp=2, peel_sweep=3, peel_pick=4)
    assert L.scldpc_workspace_bytes(WS["full_bp"], C.byref(ok), 4096, 0, 0) == 0            # N = 1000: everything in LDS
    assert L.scldpc_workspace_bytes(WS["sample"], C.byref(ok), 4096, 0, 0) == 0
    assert L.scldpc_workspace_bytes(WS["full_bp"], C.byref(big), 100, 0, 0) == 100 * (53 * 5000) * 4   # one word per CN
    assert L.scldpc_workspace_bytes(WS["sw_bp"], C.byref(big), 100, 10, 0) == 100 * (53 * 5000) * 4
    assert L.scldpc_workspace_bytes(WS["sample"], C.byref(big), 10, 0, 0) > 0
    assert L.scldpc_workspace_bytes(WS["peel_pick"], C.byref(big), 8, 250000, 0) >= 8 * (53 * 5000) * 4 + 8 * 3907 * 8
    assert L.scldpc_workspace_bytes(WS["peel_sweep"], C.byref(ok), 8, 1, 0) == 0
    assert L.scldpc_workspace_bytes(99, C.byref(ok), 8, 0, 0) == -1
    assert L.scldpc_workspace_bytes(WS["full_bp"], C.byref(bad), 8, 0, 0) == -1
    with pytest.raises(_lib.ScldpcError):
        _lib.check(-1)


def test_missing_library_fails_loudly(monkeypatch):
    from fl_scaling_sc_ldpc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libscldpc_hip.so")
    with pytest.raises(_lib.ScldpcError, match="no CPU fallback"):
        _lib.lib()


def _fnv(oracle, arr):
    a = np.ascontiguousarray(arr)
    return int(oracle.lib().orc_fnv1a(a.ctypes.data, a.nbytes, 14695981039346656037))


@pytest.mark.parametrize("name", ["tiny_bpf_M5_L10_e480", "ss2_bpf_M3_L10_e420", "mid_bpf_M50_L20_e470",
                                  "c2_bpf_M500_L50_e480_a", "c2_bpw_M500_L50_e465_W20_it6_init60"])
def test_glibc_host_sampler_reproduces_reference_digests(oracle, name):
    """scldpc_sample_glibc_host == generate_code + channel_doped of the reference on identical seeds."""
    from fl_scaling_sc_ldpc_amd import engine as E
    g = load_golden(name)
    m = g.meta
    p = E.make_params(m["dv"], m["dc"], m["L"], m["VNsPos"])
    T = min(g.T, 12)
    adj, ch = E.sample_glibc_trials(p, g["seed"][:T], m["eps"])
    bits = E.unpack_bits(ch, p.n)
    for t in range(T):
        assert _fnv(oracle, adj[t]) == int(g["hg"][t])
        assert _fnv(oracle, bits[t].astype(np.int32)) == int(g["hc"][t])
        assert int(bits[t].sum()) == int(g["nch"][t])
        if g.has("vn_adj"):
            assert (adj[t] == g["vn_adj"][t]).all() and (bits[t] == g["chan"][t]).all()


@pytest.mark.parametrize("name", golden_names(whole_run=True))
def test_glibc_run_carries_state_like_main_terminated(oracle, name):
    from fl_scaling_sc_ldpc_amd import engine as E
    g = load_golden(name)
    m = g.meta
    p = E.make_params(m["dv"], m["dc"], m["L"], m["VNsPos"])
    run = E.GlibcRun(p, m["seed0"])
    adj_a, ch_a = run.next_frames(2, m["eps"])
    snap = run.snapshot()
    adj_b, ch_b = run.next_frames(g.T - 2, m["eps"])
    adj = np.concatenate([adj_a, adj_b]); ch = np.concatenate([ch_a, ch_b])
    bits = E.unpack_bits(ch, p.n)
    for t in range(g.T):
        assert _fnv(oracle, adj[t]) == int(g["hg"][t]), t
        assert _fnv(oracle, bits[t].astype(np.int32)) == int(g["hc"][t]), t
    run.restore(snap)                                       # rewind used by the ordered stop rule
    adj_c, _ = run.next_frames(1, m["eps"])
    assert (adj_c[0] == adj[2]).all()


def test_doped_positions_are_never_erased():
    from fl_scaling_sc_ldpc_amd import engine as E
    p = E.make_params(4, 8, 12, 40)
    adj, ch = E.sample_glibc_trials(p, [3, 4], 0.9, doped=(2, 7))
    bits = E.unpack_bits(ch, p.n).reshape(2, p.L, p.vns_pos)
    assert bits[:, 2].sum() == 0 and bits[:, 7].sum() == 0 and bits[:, 3].mean() > 0.7
    adj2, ch2 = E.sample_glibc_trials(p, [3, 4], 0.9)
    assert (adj2 == adj).all()                              # doping does not consume random numbers (BPF:1566-1573)
    b2 = E.unpack_bits(ch2, p.n).reshape(2, p.L, p.vns_pos)
    keep = [q for q in range(p.L) if q not in (2, 7)]
    assert (b2[:, keep] == bits[:, keep]).all()


def test_bit_packing_roundtrip():
    from fl_scaling_sc_ldpc_amd import engine as E
    rng = np.random.RandomState(0)
    for n in (1, 31, 32, 33, 100, 1000):
        b = (rng.rand(3, n) < 0.5).astype(np.uint8)
        w = E.pack_bits(b)
        assert w.shape == (3, (n + 31) // 32) and (E.unpack_bits(w, n) == b).all()


def test_second_generation_entry_points_refuse_what_they_do_not_take(L):
    """Which ensembles the specialised kernels take is a host-side decision (no device needed), and a refusal is an error
    code with a text, never a silent slow path."""
    from fl_scaling_sc_ldpc_amd import _lib
    P = _lib.CodeParams
    base, wide, long_, big, odd = P(4, 8, 50, 500, 1000), P(4, 8, 50, 512, 1024), P(4, 8, 100, 500, 1000), P(4, 8, 100, 1000, 2000), P(3, 6, 20, 100, 200)
    assert L.scldpc_sample_philox_cn16_supported(C.byref(base)) == 1 and L.scldpc_full_bp_cn16_supported(C.byref(base)) == 1
    assert L.scldpc_sample_philox_cn16_supported(C.byref(wide)) == 1 and L.scldpc_sample_philox_cn16_supported(C.byref(P(4, 8, 20, 1024, 2048))) == 1
    too_wide = P(4, 8, 20, 1025, 2050)                                                 # 8192 sockets per position: the limit
    assert L.scldpc_sample_philox_cn16_supported(C.byref(too_wide)) == 0 and L.scldpc_sample_philox_sock16_supported(C.byref(too_wide)) == 0
    assert L.scldpc_sample_philox_sock16_supported(C.byref(big)) == 1                 # sockets are position-local: any chain length
    assert L.scldpc_sample_philox_sock16_supported(C.byref(odd)) == 0
    assert L.scldpc_sample_philox_device_sock16(C.byref(too_wide), 1, 0, 4, 0.5, 0, None, None, None, None, None) == -2
    assert L.scldpc_sample_philox_device_sock16(C.byref(big), 1, 0, 4, 0.5, 0, None, None, None, None, None) == -1    # null buffers
    assert L.scldpc_sample_philox_device_sock16(C.byref(big), 1, 0, 0, 0.5, 0, None, None, None, None, None) == 0     # empty batch
    assert L.scldpc_full_bp_cn16_supported(C.byref(long_)) == 0                       # n = 100 000 VNs: ids beyond 16 bits
    assert L.scldpc_sample_philox_cn16_supported(C.byref(big)) == 0                   # n = 200 000 VNs: ids beyond 16 bits
    assert L.scldpc_sample_philox_cn16_supported(C.byref(odd)) == 0 and L.scldpc_full_bp_cn16_supported(C.byref(odd)) == 0
    assert L.scldpc_sample_philox_device_cn16(C.byref(big), 1, 0, 4, 0.5, 0, None, None, None, None, None) == -2
    assert b"65535" in L.scldpc_last_error()
    assert L.scldpc_full_bp_fixpoint_device_cn16(C.byref(long_), 4, None, None, None, 1, None, None, None) == -2
    assert L.scldpc_sample_philox_device_cn16(C.byref(base), 1, 0, 0, 0.5, 0, None, None, None, None, None) == 0      # empty batch
    assert L.scldpc_sample_philox_device_cn16(C.byref(base), 1, 0, 4, 0.5, 0, None, None, None, None, None) == -1     # null buffers
    assert L.scldpc_full_bp_fixpoint_device_cn16(C.byref(base), 4, None, None, None, 1, None, None, None) == -1
    assert L.scldpc_full_bp_device_cn16(C.byref(long_), 4, None, None, None, 0, 1, None, None, None) == -2
    assert L.scldpc_full_bp_device_cn16(C.byref(base), 4, None, None, None, 500, 1, None, None, None) == -1
    assert L.scldpc_full_bp_device_cn16(C.byref(base), 0, None, None, None, 500, 1, None, None, None) == 0
    # window decoder with the window's state in LDS: (4,8) chains with 16-bit sockets, any window that fits
    assert L.scldpc_sw_bp_ring_supported(C.byref(big), 10) == 1 and L.scldpc_sw_bp_ring_supported(C.byref(base), 20) == 1
    assert L.scldpc_sw_bp_ring_supported(C.byref(odd), 5) == 0 and L.scldpc_sw_bp_ring_supported(C.byref(base), 0) == 0
    assert L.scldpc_sw_bp_ring_supported(C.byref(P(4, 8, 400, 5000, 10000)), 300) == 0   # 307 positions of counts: beyond the LDS
    assert L.scldpc_sw_bp_ring_device(C.byref(odd), 4, None, None, None, 5, 3, 0, None, None, None) == -1             # null buffers first
    assert L.scldpc_cn_sockets_device(C.byref(base), 0, None, None, None) == 0
    assert L.scldpc_cn_sockets_device(C.byref(base), 2, None, None, None) == -1
    # same-input streaming: the inputs must cover what the call will generate
    assert L.scldpc_stream_run_device_inputs(C.byref(base), 1, 20, 0, None, 10, None, None, None, None, None, 30, 0, None) == -1
    assert L.scldpc_stream_glibc_inputs_host(C.byref(base), 1, 0.5, 0, None, -1, None, None) == -1
