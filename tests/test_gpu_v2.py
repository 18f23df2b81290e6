"""Second-generation throughput path (-m gpu): sampler_v2.hip (ranking only where a bucket straddles two CNs, CN -> socket
table) and full_bp_small.hip (4 bits of LDS per CN) against the first generation, the CPU twin, and the reference's
fixtures.  Everything through the C-ABI (engine.py is ctypes plumbing)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, require_gpu

pytestmark = pytest.mark.gpu

KEEP = [0, 1, 2, 3, 4, 6, 7]            # every counter except the iteration / barrier-round count


@pytest.fixture(scope="module")
def E():
    require_gpu()
    from fl_scaling_sc_ldpc_amd import engine
    return engine


def _check_cn_table(E, p, adj16, cn16):
    """cn16 [nk, dc]: as a set per CN, exactly the VNs the VN -> CN table attaches to it (0xFFFF where a chain-end CN has
    fewer than dc) — i.e. the host's inversion of the same table, up to the order within a CN."""
    c = np.sort(np.ascontiguousarray(cn16).view(np.uint16), axis=1)
    want = E.cn_adj_from_vn_adj(p, adj16)[0].view(np.uint16)            # ascending VNs, then 0xFFFF
    assert (c == want).all()


@pytest.mark.parametrize("L,N,eps,doped", [(50, 1000, 0.48, ()), (10, 10, 0.48, ()), (20, 100, 0.3, (3, 4)), (7, 66, 0.9, ()),
                                           (50, 1000, 0.48, (24, 25)), (12, 1024, 0.45, (5,)), (5, 128, 0.0, ()),
                                           (5, 128, 1.0, (0,)), (16, 512, 0.5, ()), (9, 600, 0.47, ()),
                                           # more than 4096 sockets per position: two Philox calls per thread
                                           (7, 2048, 0.47, (2,)), (12, 1100, 0.5, ()), (15, 2000, 0.48, ()), (6, 1028, 0.4, ())])
def test_sampler_v2_equals_first_generation_and_twin(E, oracle, L, N, eps, doped, monkeypatch):
    import torch
    if (L + N) % 2:                                          # every other case through the opt-in third-generation kernel
        monkeypatch.setenv("SCLDPC_SAMPLER_GEN", "3")
    p = E.make_params(4, 8, L, N)
    assert E.cn16_supported(p)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    seed, t0, T = 0x123456789ABCDEF, (1 << 33) + 5, 5
    a1, c1 = E.sample_philox(p, seed, t0, T, eps, doped, adj16=True)
    a2, cn2, c2 = E.sample_philox_cn16(p, seed, t0, T, eps, doped)
    a3, _, c3 = E.sample_philox_cn16(p, seed, t0, T, eps, doped, want_cn=False)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(c1, c2) and torch.equal(a1, a3) and torch.equal(c1, c3)
    A, CN = a2.cpu().numpy(), cn2.cpu().numpy()
    ta, tch = oracle.sample_philox(po, seed, t0, eps, doped)            # CPU twin (global ids)
    assert (E.adj16_to_global(p, A[0]) == ta).all()
    for t in range(2 if N >= 500 else T):
        _check_cn_table(E, p, A[t], CN[t])


@pytest.mark.parametrize("L,N,eps,is_term,cap", [(100, 1000, 0.48, True, 0), (100, 1000, 0.47, False, 150), (50, 1000, 0.48, True, 60),
                                                  (130, 800, 0.46, True, 0), (16, 200, 0.47, True, 5), (9, 24, 0.5, False, 0)])
def test_small_decoder_on_the_socket_table_takes_long_chains(E, L, N, eps, is_term, cap):
    """scldpc_full_bp(_fixpoint)_device_sock16: the 4-bits-per-CN decoder reading position-local sockets instead of global VN
    ids, so that n >= 65535 (the published L = 100, N = 1000 runs: n = 100 000) no longer falls back to the 16-bit-word
    kernels.  Every counter and the residual equal full_bp's (and, where both forms apply, the VN-id form's)."""
    import torch
    p = E.make_params(4, 8, L, N)
    assert E.full_bp_sock16_supported(p) and (p.n < 65535) == E.cn16_supported(p)
    T = 64
    a, cs, ch = E.sample_philox_sock16(p, 31, 500, T, eps)
    ref = E.full_bp(p, a, ch, max_it=cap, is_term=is_term, want_erased=True)
    lvl = E.full_bp_cn16(p, a, cs, ch, max_it=cap, is_term=is_term, want_erased=True, sockets=True)
    torch.cuda.synchronize()
    assert torch.equal(ref["counters"], lvl["counters"]) and torch.equal(ref["erased"], lvl["erased"])
    if cap == 0:
        fix = E.full_bp_fixpoint_cn16(p, a, cs, ch, is_term=is_term, want_erased=True, sockets=True)
        assert (fix["counters"].cpu().numpy()[:, KEEP] == ref["counters"].cpu().numpy()[:, KEEP]).all()
        assert torch.equal(ref["erased"], fix["erased"])
    if E.cn16_supported(p):
        a2, cn, ch2 = E.sample_philox_cn16(p, 31, 500, T, eps)
        assert torch.equal(a, a2) and torch.equal(ch, ch2)
        vn = E.full_bp_cn16(p, a2, cn, ch2, max_it=cap, is_term=is_term, want_erased=True)
        assert torch.equal(vn["counters"], lvl["counters"])


@pytest.mark.parametrize("gen", ["2", "3"])
@pytest.mark.parametrize("L,N,which", [(50, 1000, -2), (9, 600, 3), (7, 2048, -2), (12, 1100, 11), (10, 10, -2)])
def test_sampler_v2_exact_fallback_gives_the_same_tables(E, monkeypatch, L, N, which, gen):
    """A bucket count that does not fit its nibble, or more straddlers than the worklist holds, sends a CN position through
    the exact fallback (rank of every key among all S, ties by socket) instead of trapping the process.  It never happens
    on real draws, so the test forces it (SCLDPC_DEBUG_SAMPLER_EXACT_POS: one position, or -2 = every position): the
    VN -> CN table must come out bit for bit, the CN -> VN / CN -> socket tables as the same sets."""
    import torch
    p = E.make_params(4, 8, L, N)
    seed, t0, T = 99, 7, 3
    monkeypatch.setenv("SCLDPC_SAMPLER_GEN", "2")           # the default kernels give the tables to compare with
    a1, cn1, c1 = E.sample_philox_cn16(p, seed, t0, T, 0.48)
    s1 = E.sample_philox_sock16(p, seed, t0, T, 0.48)[1]
    monkeypatch.setenv("SCLDPC_SAMPLER_GEN", gen)           # (the third generation takes N <= 1024; beyond, the second runs)
    monkeypatch.setenv("SCLDPC_DEBUG_SAMPLER_EXACT_POS", str(which))
    a2, cn2, c2 = E.sample_philox_cn16(p, seed, t0, T, 0.48)
    s2 = E.sample_philox_sock16(p, seed, t0, T, 0.48)[1]
    torch.cuda.synchronize()
    monkeypatch.delenv("SCLDPC_DEBUG_SAMPLER_EXACT_POS")
    assert torch.equal(a1, a2) and torch.equal(c1, c2)
    for x, y in ((cn1, cn2), (s1, s2)):
        assert (np.sort(x.cpu().numpy().view(np.uint16), axis=2) == np.sort(y.cpu().numpy().view(np.uint16), axis=2)).all()
    if which == -2:                                          # exact ranks: within a CN the sockets come in key order, not arrival order
        _check_cn_table(E, p, a2[0].cpu().numpy(), cn2[0].cpu().numpy())


@pytest.mark.parametrize("L,N,eps,is_term", [(50, 1000, 0.48, True), (50, 1000, 0.45, True), (50, 1000, 0.49, False),
                                             (16, 200, 0.47, True), (16, 200, 0.30, True), (9, 24, 0.5, False),
                                             (12, 1024, 0.46, True), (30, 400, 0.44, False), (10, 10, 0.48, True),
                                             (6, 16, 0.9, True), (6, 16, 0.05, True)])
def test_small_decoder_equals_flooding_and_fixpoint_kernels(E, L, N, eps, is_term):
    """Same trials through full_bp (level-synchronous flooding), full_bp_fixpoint (16-bit CN words) and
    full_bp_fixpoint_cn16 (4-bit counts + CN -> VN table): identical counters and identical residual patterns."""
    import torch
    p = E.make_params(4, 8, L, N)
    T = 96 if N >= 1000 else 192
    a, cn, ch = E.sample_philox_cn16(p, 77, 1000, T, eps)
    ref = E.full_bp(p, a, ch, is_term=is_term, want_erased=True)
    fix = E.full_bp_fixpoint(p, a, ch, is_term=is_term, want_erased=True)
    sm = E.full_bp_fixpoint_cn16(p, a, cn, ch, is_term=is_term, want_erased=True)
    torch.cuda.synchronize()
    r, f, s = (x["counters"].cpu().numpy() for x in (ref, fix, sm))
    assert (r[:, KEEP] == s[:, KEEP]).all() and (f[:, KEEP] == s[:, KEEP]).all()
    assert torch.equal(ref["erased"], sm["erased"])


@pytest.mark.parametrize("name", golden_names(prefixes=("c2_bpf", "mid_bpf", "tiny_bpf", "ss2_bpf", "mid_bpt", "tiny_bpt"),
                                               uncapped=True))
def test_small_decoder_on_reference_fixtures(E, name):
    """The reference's own graphs and channels (glibc replay on the fixture's seeds), CN -> VN table built on the host:
    the unlimited-iteration fixtures' counters (incl. the size-2 stopping-set expurgation of the Def_M = 3 ensembles) and
    residual patterns."""
    import torch
    g = load_golden(name)
    m = g.meta
    assert not g.max_it
    p = E.make_params(m["dv"], m["dc"], m["L"], m["VNsPos"])
    T = min(g.T, 16 if p.n > 10000 else 64)
    adj, ch = E.sample_glibc_trials(p, g["seed"][:T], m["eps"])
    a16 = E.global_to_adj16(p, adj)
    cn16 = E.cn_adj_from_vn_adj(p, a16)
    d_a, d_ch = E.to_device(a16, ch)
    d_cn = torch.from_numpy(cn16).to(d_a.device)
    out = E.full_bp_fixpoint_cn16(p, d_a, d_cn, d_ch, is_term=bool(m["is_term"]), want_erased=True)
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy()
    assert (c[:, 0] == g["ne"][:T]).all() and (c[:, 1] == g["be"][:T]).all()
    assert (c[:, 2] == g["ee"][:T]).all() and (c[:, 3] == g["bee"][:T]).all() and (c[:, 7] == g["nch"][:T]).all()
    if g.has("erased"):
        assert (E.unpack_bits(out["erased"].cpu().numpy(), p.n) == g["erased"][:T]).all()


# ---- the same decoder walked one flooding iteration per barrier round (scldpc_full_bp_device_cn16) ------------------------
@pytest.mark.parametrize("L,N,eps,is_term,caps", [
    (50, 1000, 0.48, True, (0, 1, 2, 3, 40, 150, 233, 400)), (50, 1000, 0.45, True, (0, 25, 90)),
    (50, 1000, 0.49, False, (0, 5, 120)), (50, 1000, 0.30, True, (0, 2)), (50, 1000, 0.10, True, (0, 1)),
    (16, 200, 0.47, True, (0, 7)), (9, 24, 0.5, False, (0, 2)), (12, 1024, 0.46, True, (0, 30)),
    (30, 400, 0.44, False, (0, 11)), (10, 10, 0.48, True, (0, 1)), (6, 16, 0.9, True, (0, 3)), (6, 16, 0.0, True, (0,)),
    (60, 1000, 0.485, True, (0, 77))])
def test_level_decoder_equals_flooding_kernel_counter_for_counter(E, L, N, eps, is_term, caps):
    """full_bp_cn16 (4-bit counts, CN -> VN table, one flooding iteration per round) against full_bp (16-bit CN words): ALL
    eight counters — the iteration count and the status among them — and the residual pattern, with and without binding
    iteration caps (a capped run stops mid-way: the residual is the reference's a-posteriori erasure set of that iteration)."""
    import torch
    p = E.make_params(4, 8, L, N)
    T = 96 if N >= 1000 else 192
    a, cn, ch = E.sample_philox_cn16(p, 78, 5000, T, eps)
    for max_it in caps:
        ref = E.full_bp(p, a, ch, max_it=max_it, is_term=is_term, want_erased=True)
        lvl = E.full_bp_cn16(p, a, cn, ch, max_it=max_it, is_term=is_term, want_erased=True)
        torch.cuda.synchronize()
        r, v = ref["counters"].cpu().numpy(), lvl["counters"].cpu().numpy()
        assert (r == v).all(), (max_it, np.argwhere(r != v)[:4].tolist(), r[r != v][:4], v[r != v][:4])
        assert torch.equal(ref["erased"], lvl["erased"]), max_it
        if max_it:
            assert (v[:, 5] <= max_it).all()


@pytest.mark.parametrize("name", golden_names(prefixes=("c2_", "mid_", "tiny_", "ss2_"), variants=("bpf", "bpt")))
def test_level_decoder_on_reference_fixtures(E, name):
    """The reference's own graphs and channels (glibc replay on the fixture's seeds), CN -> VN table built on the host:
    counters, residual patterns and — where the fixture holds the trajectory rows — the iteration count, including the
    fixtures generated with a binding MAX_IT (*_it1, _it3, _it5, _it100) and the truncated chains (BPT is_term = 0)."""
    import torch
    g = load_golden(name)
    m = g.meta
    p = E.make_params(m["dv"], m["dc"], m["L"], m["VNsPos"])
    T = min(g.T, 16 if p.n > 10000 else 64)
    adj, ch = E.sample_glibc_trials(p, g["seed"][:T], m["eps"])
    a16 = E.global_to_adj16(p, adj)
    cn16 = E.cn_adj_from_vn_adj(p, a16)
    d_a, d_ch = E.to_device(a16, ch)
    d_cn = torch.from_numpy(cn16).to(d_a.device)
    out = E.full_bp_cn16(p, d_a, d_cn, d_ch, max_it=g.max_it, is_term=bool(m["is_term"]), want_erased=True)
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy()
    for col, key in ((0, "ne"), (1, "be"), (2, "ee"), (3, "bee"), (7, "nch")):
        assert (c[:, col] == g[key][:T]).all(), (name, key)
    assert (c[:, 6] == 0).all() and (c[:, 4] == 0).all()
    er = E.unpack_bits(out["erased"].cpu().numpy(), p.n)
    assert (er.sum(axis=1) == g["ne"][:T]).all()
    if g.has("erased"):
        assert (er == g["erased"][:T]).all()
    if g.has("rows"):
        assert [int(x) for x in c[:, 5]] == [len(g.rows_of(t)) for t in range(T)], name
        # the trajectory build's rows (deg_1_iter, recovered, first erased position; BPT:988, 1051) from the same kernel,
        # through both table forms (VN ids / sockets)
        d_cs = E.cn_sockets(p, d_a)
        for sockets, tab in ((False, d_cn), (True, d_cs)):
            tr = E.full_bp_cn16(p, d_a, tab, d_ch, max_it=g.max_it, is_term=bool(m["is_term"]), rows_cap=2048, sockets=sockets)
            torch.cuda.synchronize()
            ct, rows = tr["counters"].cpu().numpy(), tr["rows"].cpu().numpy()
            assert (ct == c).all()
            for t in range(T):
                ref = g.rows_of(t)
                assert (rows[t, :len(ref)] == ref).all(), (name, sockets, t)


@pytest.mark.parametrize("L,N,eps,is_term,cap", [(50, 1000, 0.48, True, 0), (50, 1000, 0.46, False, 0), (50, 1000, 0.47, True, 40),
                                                  (100, 1000, 0.47, True, 0), (16, 200, 0.49, False, 9), (10, 10, 0.5, True, 0)])
def test_trajectory_rows_of_the_level_kernel_equal_the_flooding_kernels(E, L, N, eps, is_term, cap):
    """bp_traj's rows from full_bp_small (4-bit counts, six trials per CU) against full_bp's (16-bit CN words, two per CU) on
    device-sampled trials: every row of every trial, the counters, truncated chains and binding caps included."""
    import torch
    p = E.make_params(4, 8, L, N)
    T = 48
    sockets = not E.cn16_supported(p)
    a, tab, ch = (E.sample_philox_sock16 if sockets else E.sample_philox_cn16)(p, 5, 77, T, eps)
    ref = E.full_bp(p, a, ch, max_it=cap, is_term=is_term, rows_cap=1024)
    got = E.full_bp_cn16(p, a, tab, ch, max_it=cap, is_term=is_term, rows_cap=1024, sockets=sockets)
    torch.cuda.synchronize()
    c, r0, r1 = ref["counters"].cpu().numpy(), ref["rows"].cpu().numpy(), got["rows"].cpu().numpy()
    assert (c == got["counters"].cpu().numpy()).all()
    for t in range(T):
        k = int(c[t, 5])
        assert (r0[t, :k] == r1[t, :k]).all(), (t, np.argwhere(r0[t, :k] != r1[t, :k])[:3])


def test_level_decoder_with_a_small_queue_takes_the_frontier_from_the_snapshot(E):
    """Low eps: nearly every erased VN is resolved in iteration 0, so the first frontier is several queue-fulls and the
    pushes of one round overflow the queue (scan rounds back to back) — counters still equal the flooding kernel's."""
    import torch
    p = E.make_params(4, 8, 50, 1000)
    for eps in (0.2, 0.35, 0.6, 0.97):
        a, cn, ch = E.sample_philox_cn16(p, 79, 0, 64, eps)
        ref = E.full_bp(p, a, ch, want_erased=True)
        lvl = E.full_bp_cn16(p, a, cn, ch, want_erased=True)
        torch.cuda.synchronize()
        assert torch.equal(ref["counters"], lvl["counters"]) and torch.equal(ref["erased"], lvl["erased"]), eps


# ---- square window with only the window's state on chip (sw_ring.hip) ---------------------------------------------------
@pytest.mark.parametrize("L,N,W,max_it,init_it,eps", [
    (100, 2000, 10, 20, 0, 0.47), (50, 1000, 20, 6, 60, 0.465), (50, 1000, 10, 20, 0, 0.47), (16, 200, 5, 3, 9, 0.45),
    (14, 1200, 5, 4, 10, 0.47), (9, 24, 2, 1, 0, 0.5), (10, 10, 4, 3, 7, 0.48), (10, 10, 12, 1000000, 0, 0.42),
    (30, 400, 1, 2, 3, 0.44), (12, 64, 11, 5, 0, 0.3), (40, 1000, 40, 1000000, 0, 0.47), (20, 100, 3, 50, 0, 0.9)])
def test_ring_window_decoder_equals_whole_chain_kernel(E, L, N, W, max_it, init_it, eps):
    """scldpc_sw_bp_ring_device (ring of W+7 CN / W+4 VN positions in LDS, 4-bit counts, CN -> socket table from
    scldpc_cn_sockets_device) == scldpc_sw_bp_device (one word per CN of the whole chain): every counter incl. the
    iteration total, NumErasuresP1 and the size-2 stopping-set expurgation, and the VNerased pattern — for windows longer
    than the chain, W = 1, binding and non-binding caps, V not a multiple of 32, and BASELINE config 4's size."""
    import torch
    p = E.make_params(4, 8, L, N)
    T = 48 if N >= 1000 else 160
    a, ch = E.sample_philox(p, 55, 300, T, eps, adj16=True)
    old = E.sw_bp(p, a, ch, W, max_it, init_it, want_erased=True, ring=False)
    new = E.sw_bp(p, a, ch, W, max_it, init_it, want_erased=True, ring=True)
    torch.cuda.synchronize()
    assert torch.equal(old["counters"], new["counters"]), (old["counters"][:4], new["counters"][:4])
    assert torch.equal(old["erased"], new["erased"])


@pytest.mark.parametrize("L,N,W,max_it,classical,eps", [(100, 2000, 10, 20, False, 0.47), (60, 3300, 8, 12, False, 0.465),
                                                        (100, 2000, 12, 15, True, 0.47), (60, 3300, 9, 1000000, True, 0.46),
                                                        (3, 16000, 2, 7, False, 0.45), (2, 12000, 2, 5, True, 0.5)])
def test_whole_chain_window_kernel_cn_words_built_through_lds(E, monkeypatch, L, N, W, max_it, classical, eps):
    """scldpc_sw_bp_device_adj16 / scldpc_swc_bp_device_adj16 with their CN words in the workspace (chains beyond the LDS):
    the words built through cn_build.hip's LDS ring (default) against the kernel's own build (one global atomic per edge) —
    every counter incl. the per-position erasure counts behind them, and the VNerased pattern; V a multiple of 32 and not."""
    import torch
    p = E.make_params(4, 8, L, N)
    a, ch = E.sample_philox(p, 56, 100, 24, eps, adj16=True)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCLDPC_DEBUG_SW_PREBUILD", mode)
        out[mode] = E.sw_bp(p, a, ch, W, max_it, 0, want_erased=True, classical=classical, ring=False)
        torch.cuda.synchronize()
    assert torch.equal(out["1"]["counters"], out["0"]["counters"]) and torch.equal(out["1"]["erased"], out["0"]["erased"])
    assert int(out["1"]["counters"][:, 7].sum()) == int(E.unpack_bits(ch.cpu().numpy(), p.n).sum())


def test_cn_socket_table_is_the_inverse_of_the_vn_table(E):
    import torch
    for L, N in ((10, 10), (12, 1000), (100, 2000)):
        p = E.make_params(4, 8, L, N)
        a, _ = E.sample_philox(p, 9, 0, 2, 0.5, adj16=True)
        cs = E.cn_sockets(p, a).cpu().numpy().view(np.uint16).astype(np.int64)
        A = a.cpu().numpy().view(np.uint16).astype(np.int64)
        for t in range(2):
            c = cs[t].reshape(L + 3, p.cns_pos, 8)
            for cpos in (0, 1, 3, L - 1, L, L + 2):
                valid = c[cpos] != 0xFFFF
                i, tt = c[cpos] % 4, c[cpos] // 4
                q = cpos - i
                assert ((q >= 0) & (q < L))[valid].all()
                # every listed socket points back at this CN, and the list holds all of them
                back = A[t].reshape(L, N, 4)[np.clip(q, 0, L - 1), np.clip(tt, 0, N - 1), i]
                assert (back == np.arange(p.cns_pos)[:, None])[valid].all()
                assert valid.sum() == sum(N for k in range(4) if 0 <= cpos - k < L)


@pytest.mark.parametrize("L,N,W,eps,doped", [(100, 2000, 10, 0.47, ()), (50, 1000, 10, 0.47, (20,)), (10, 10, 4, 0.48, ()),
                                             (9, 24, 2, 0.5, (0, 8)), (14, 1200, 5, 0.47, ()), (30, 2048, 6, 0.46, ())])
def test_sampler_emits_the_socket_table_of_the_ring_decoder(E, L, N, W, eps, doped):
    """scldpc_sample_philox_device_sock16: the code and channel of the first-generation sampler bit for bit, and — as a set
    per CN — the table scldpc_cn_sockets_device builds from that code in a second pass; the ring decoder gives the same
    counters and residual pattern from either table.  N = 2000, L = 100 is BASELINE config 4 (200 000 VNs: the table holds
    position-local sockets, not VN indices)."""
    import torch
    p = E.make_params(4, 8, L, N)
    assert E.sock16_supported(p) and E.sw_ring_supported(p, W)
    T = 24 if N >= 1000 else 96
    a1, c1 = E.sample_philox(p, 31, 7000, T, eps, doped, adj16=True)
    a2, cs2, c2 = E.sample_philox_sock16(p, 31, 7000, T, eps, doped)
    cs1 = E.cn_sockets(p, a1)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(c1, c2)
    s1 = np.sort(cs1.cpu().numpy().view(np.uint16), axis=2)
    s2 = np.sort(cs2.cpu().numpy().view(np.uint16), axis=2)
    assert (s1 == s2).all()
    r1 = E.sw_bp(p, a1, c1, W, 20, 0, want_erased=True, ring=True, d_cn_sock=cs1)
    r2 = E.sw_bp(p, a2, c2, W, 20, 0, want_erased=True, ring=True, d_cn_sock=cs2)
    torch.cuda.synchronize()
    assert torch.equal(r1["counters"], r2["counters"]) and torch.equal(r1["erased"], r2["erased"])


@pytest.mark.parametrize("name", golden_names(variants=("bpw",)))
def test_ring_window_decoder_on_reference_fixtures(E, name):
    """The reference's square-window runs (BPW's decodeBP_SW on its own graphs, glibc replay on the fixture's seeds)."""
    import torch
    g = load_golden(name)
    m = g.meta
    p = E.make_params(m["dv"], m["dc"], m["L"], m["VNsPos"])
    T = min(g.T, 16 if p.n > 10000 else 64)
    adj, ch = E.sample_glibc_trials(p, g["seed"][:T], m["eps"])
    d_a, d_ch = E.to_device(E.global_to_adj16(p, adj), ch)
    out = E.sw_bp(p, d_a, d_ch, m["W"], m["max_it"], m["init_it"], want_erased=True, ring=True)
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy()
    for k, col in (("ne", 0), ("be", 1), ("ee", 2), ("bee", 3), ("p1", 4), ("nch", 7)):
        assert (c[:, col] == g[k][:T]).all(), (name, k)
    if g.has("erased"):
        assert (E.unpack_bits(out["erased"].cpu().numpy(), p.n) == g["erased"][:T]).all()


# ---- the reference's two decoder families on ONE graph ("cross-implementation redundancy") ------------------
@pytest.mark.parametrize("L,N,eps,is_term", [(50, 1000, 0.48, True), (50, 1000, 0.47, False), (20, 200, 0.46, True),
                                             (20, 200, 0.49, False), (12, 64, 0.5, True)])
def test_bp_and_peeling_paths_leave_the_same_stopping_set_on_a_shared_graph(E, L, N, eps, is_term):
    """On the BEC unlimited flooding BP (bp_decoding: decodeBP, BPF:900-1140) and peeling (peeling_decoding: the sic_round
    sweep PD:270-313 and the random-pick peeling PD:740-785) all stop at the maximal stopping set of the erased VNs.  The
    reference never runs its two simulators on a shared graph; here one batch of sampled codes and channels goes through the
    four device decoders of both families: identical residual VN sets (flooding, fixpoint, sweep) and, for random-pick
    peeling — whose picks are random but whose closure is not — the same number of VNs left."""
    import torch
    p = E.make_params(4, 8, L, N)
    T = 64
    adj, ch = E.sample_philox(p, 321, 77, T, eps)                        # int32 global ids: the layout all four kernels take
    total_size = p.nk if is_term else L * p.cns_pos                     # CNs beyond a truncated chain never fire (BPT:944-948, PD:609-611)
    bp = E.full_bp(p, adj, ch, is_term=is_term, want_erased=True)
    fx = E.full_bp_fixpoint(p, adj, ch, is_term=is_term, want_erased=True)
    sw = E.peel_sweep(p, adj, ch, total_size, want_lost=True)
    steps = p.n                                                          # more picks than erased VNs: runs to exhaustion
    pk = E.peel_pick(p, adj, ch, total_size, steps, seed=5, trial0=0, want_r1=False)
    torch.cuda.synchronize()
    assert torch.equal(bp["erased"], fx["erased"])
    ne = bp["counters"][:, 0]
    # simulate_sc_ldpc counts as lost only users all of whose transmissions lie below total_size (PD:659-666): on a truncated
    # chain the VNs of the last dv - 1 positions are outside its statistic, decodeBP's VNerased keeps them
    counted = p.n if is_term else (L - (p.dv - 1)) * N
    er = torch.from_numpy(E.unpack_bits(bp["erased"].cpu().numpy(), p.n)[:, :counted].copy())
    lost = torch.from_numpy(E.unpack_bits(sw["lost"].cpu().numpy(), p.n))
    assert torch.equal(er, lost[:, :counted]) and int(lost[:, counted:].sum()) == 0
    assert torch.equal(er.sum(dim=1).to(torch.int32), sw["out"][:, 0].cpu()) and torch.equal(bp["counters"][:, 7], sw["out"][:, 7])
    out = pk["out"]                                                      # [0] #erased, [1] #picked
    assert torch.equal(out[:, 0], bp["counters"][:, 7]) and torch.equal(out[:, 0] - out[:, 1], ne)
    assert int((ne > 0).sum().item()) > 0 or eps < 0.47                  # the batch does contain failures where expected
