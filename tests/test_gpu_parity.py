"""Parity tests proper (-m gpu): the HIP kernels, called through the C-ABI, against
  (a) the golden vectors generated from the REAL reference (tests/golden, bit-exact), and
  (b) the CPU oracle on the same seeded inputs (bit-exact: integer / bit work only)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden_names, load_golden, require_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    require_gpu()
    from fl_scaling_sc_ldpc_amd import engine
    return engine


def _golden_inputs(E, g, T=None):
    m = g.meta
    p = E.make_params(m["dv"], m["dc"], m["L"], m["VNsPos"])
    T = g.T if T is None else min(T, g.T)
    adj, ch = E.sample_glibc_trials(p, g["seed"][:T], m["eps"])
    return p, T, E.to_device(adj, ch)


@pytest.mark.parametrize("name", golden_names(variants=("bpf", "bpt")))
def test_full_bp_matches_reference_golden(E, name):
    import torch
    g = load_golden(name)
    m = g.meta
    p, T, (d_adj, d_ch) = _golden_inputs(E, g)
    out = E.full_bp(p, d_adj, d_ch, max_it=g.max_it, is_term=bool(m["is_term"]),
                    rows_cap=2048 if g.has("rows") else 0, want_erased=True)
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
        rows = out["rows"].cpu().numpy()
        for t in range(T):
            ref = g.rows_of(t)
            assert c[t, 5] == len(ref) and (rows[t, :len(ref)] == ref).all(), (name, t)


@pytest.mark.parametrize("name", golden_names(variants=("bpw",)))
def test_sw_bp_matches_reference_golden(E, name):
    import torch
    g = load_golden(name)
    m = g.meta
    p, T, (d_adj, d_ch) = _golden_inputs(E, g)
    out = E.sw_bp(p, d_adj, d_ch, m["W"], m["max_it"], m["init_it"], want_erased=True)
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy()
    for col, key in ((0, "ne"), (1, "be"), (2, "ee"), (3, "bee"), (4, "p1"), (7, "nch")):
        assert (c[:, col] == g[key][:T]).all(), (name, key)
    assert (c[:, 6] == 0).all()
    if g.has("erased"):
        assert (E.unpack_bits(out["erased"].cpu().numpy(), p.n) == g["erased"][:T]).all()


@pytest.mark.parametrize("name", golden_names(whole_run=True))
def test_whole_run_replay_matches_reference_run_counters(E, name):
    """glibc stream + perm_code carried over frame to frame, decoded on the GPU, accumulated by the
    run kernel == the reference's own users_err/frame_err/… after the same frames (BPF:2117-2144)."""
    import torch
    g = load_golden(name)
    m = g.meta
    p = E.make_params(m["dv"], m["dc"], m["L"], m["VNsPos"])
    run_src = E.GlibcRun(p, m["seed0"])
    adj, ch = run_src.next_frames(g.T, m["eps"])
    d_adj, d_ch = E.to_device(adj, ch)
    out = E.full_bp(p, d_adj, d_ch, max_it=g.max_it, is_term=bool(m.get("is_term", 1)))
    run = E.accumulate_run(out["counters"], E.new_run(), 0)
    torch.cuda.synchronize()
    r = dict(zip(E.RUN_NAMES, run.cpu().tolist()))
    rc = m["run_counters"]
    assert (r["users_err"], r["frame_err"], r["block_err"], r["users_err_exp"], r["frame_err_exp"],
            r["block_err_exp"], r["frames"]) == (rc["users_err"], rc["frame_err"], rc["block_err"],
                                                 rc["users_err_exp"], rc["frame_err_exp"], rc["block_err_exp"], g.T)
    assert (out["counters"].cpu().numpy()[:, 0] == g["ne"]).all()


@pytest.mark.parametrize("L,N,eps,doped", [(10, 10, 0.48, ()), (20, 100, 0.3, (3, 4)), (50, 1000, 0.48, ()),
                                           (50, 1000, 0.48, (24, 25)), (7, 66, 0.9, ()), (12, 2000, 0.45, (5,)),
                                           (5, 128, 0.0, ()), (5, 128, 1.0, (0,))])
def test_philox_sampler_equals_cpu_twin(E, oracle, L, N, eps, doped):
    import torch
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    seed, t0, T = 0x123456789ABCDEF, (1 << 33) + 5, 4
    d_adj, d_ch = E.sample_philox(p, seed, t0, T, eps, doped)
    torch.cuda.synchronize()
    A, Cb = d_adj.cpu().numpy(), d_ch.cpu().numpy().view(np.uint32)
    for t in range(T):
        a, c = oracle.sample_philox(po, seed, t0 + t, eps, doped)
        assert (a == A[t]).all() and (c == Cb[t]).all(), (L, N, t)
    # every CN position holds a perfect matching: CN degrees are dc in the interior
    deg = np.bincount(A[0].reshape(-1), minlength=p.nk).reshape(L + 3, p.cns_pos)
    assert (deg[3:L] == 8).all() and deg.sum() == p.n * 4
    # sharding invariance: the same trial indices drawn in a different call are identical
    d2, c2 = E.sample_philox(p, seed, t0 + 2, 2, eps, doped)
    assert (d2.cpu().numpy() == A[2:4]).all() and (c2.cpu().numpy().view(np.uint32) == Cb[2:4]).all()


def _oracle_counters(oracle, po, adj, chbits, decoder, **kw):
    g = oracle.Graph.from_vn_adj(po, adj)
    if decoder == "full":
        res, erased, rows = oracle.decode_bp(g, chbits, max_it=kw.get("max_it", 0), is_term=kw.get("is_term", 1),
                                             literal=kw.get("literal", False), rows_cap=kw.get("rows_cap", 0))
        return res, erased, rows
    res, erased = oracle.decode_sw(g, chbits, kw["W"], kw["max_it"], kw.get("init_it", 0), literal=kw.get("literal", False))
    return res, erased, None


@pytest.mark.parametrize("L,N", [(50, 1000), (16, 200), (9, 24), (14, 1200)])    # N = 1200: 32-bit CN words in LDS
@pytest.mark.parametrize("eps", [0.05, 0.3, 0.44, 0.48, 0.52, 0.95])
def test_full_bp_equals_oracle_on_philox_inputs(E, oracle, L, N, eps):
    """Random seeds beyond the fixtures, incl. ε far from threshold: tiny ε floods the frontier queue
    (overflow → scan path), large ε stops at once.  Trajectory rows and residual patterns included."""
    import torch
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    T = 6 if N >= 1000 else 24
    d_adj, d_ch = E.sample_philox(p, 99, int(eps * 1000) * 1000, T, eps)
    for is_term, max_it in ((True, 0), (False, 0), (True, 3)):
        out = E.full_bp(p, d_adj, d_ch, max_it=max_it, is_term=is_term, rows_cap=1024, want_erased=True)
        torch.cuda.synchronize()
        A, Cb = d_adj.cpu().numpy(), E.unpack_bits(d_ch.cpu().numpy(), p.n)
        c, rows = out["counters"].cpu().numpy(), out["rows"].cpu().numpy()
        er = E.unpack_bits(out["erased"].cpu().numpy(), p.n)
        for t in range(T):
            res, erased, orows = _oracle_counters(oracle, po, A[t], Cb[t], "full", max_it=max_it,
                                                  is_term=int(is_term), literal=(t == 0 and N < 1000), rows_cap=1024)
            assert c[t, :4].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                         res["num_blocks_err_exp"]], (L, N, eps, is_term, max_it, t)
            assert c[t, 5] == res["iterations"] and c[t, 6] == 0 and c[t, 7] == Cb[t].sum()
            assert (er[t] == erased).all()
            k = res["iterations"]
            got = rows[t, :k]
            assert (got[:, 0] == orows["deg1"]).all() and (got[:, 1] == orows["recovered"]).all() \
                and (got[:, 2] == orows["first_pos"]).all(), (L, N, eps, is_term, max_it, t)


@pytest.mark.parametrize("L,N,W,max_it,init_it", [(50, 1000, 20, 6, 60), (50, 1000, 10, 20, 0), (16, 200, 5, 3, 9), (14, 1200, 5, 4, 10),
                                                  (16, 200, 30, 50, 0), (9, 24, 1, 1, 1), (9, 24, 4, 2, 0)])
@pytest.mark.parametrize("eps", [0.1, 0.42, 0.47, 0.6])
def test_sw_bp_equals_oracle_on_philox_inputs(E, oracle, L, N, W, max_it, init_it, eps):
    import torch
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    T = 5 if N >= 1000 else 20
    d_adj, d_ch = E.sample_philox(p, 7, int(eps * 1000) * 1000 + W, T, eps)
    out = E.sw_bp(p, d_adj, d_ch, W, max_it, init_it, want_erased=True)
    torch.cuda.synchronize()
    A, Cb = d_adj.cpu().numpy(), E.unpack_bits(d_ch.cpu().numpy(), p.n)
    c = out["counters"].cpu().numpy()
    er = E.unpack_bits(out["erased"].cpu().numpy(), p.n)
    for t in range(T):
        res, erased, _ = _oracle_counters(oracle, po, A[t], Cb[t], "sw", W=W, max_it=max_it, init_it=init_it,
                                          literal=(t == 0))
        assert c[t, :5].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                     res["num_blocks_err_exp"], res["num_erasures_p1"]], (L, N, W, eps, t)
        assert c[t, 5] == res["iterations"] and c[t, 6] == 0
        assert (er[t] == erased).all()


def test_generic_dv_path(E, oracle):
    """(3,6) ensemble: the non-int4 adjacency path of all three kernels."""
    import torch
    p = E.make_params(3, 6, 14, 60)
    po = oracle.Params(3, 6, 14, p.cns_pos, p.vns_pos)
    d_adj, d_ch = E.sample_philox(p, 3, 0, 16, 0.4)
    A, Cb = d_adj.cpu().numpy(), d_ch.cpu().numpy().view(np.uint32)
    for t in range(4):
        a, c = oracle.sample_philox(po, 3, t, 0.4)
        assert (a == A[t]).all() and (c == Cb[t]).all()
    out = E.full_bp(p, d_adj, d_ch, rows_cap=256)
    sw = E.sw_bp(p, d_adj, d_ch, 5, 4, 8)
    torch.cuda.synchronize()
    bits = E.unpack_bits(Cb, p.n)
    for t in range(16):
        res, _, orows = _oracle_counters(oracle, po, A[t], bits[t], "full", rows_cap=256)
        assert out["counters"][t, :4].tolist() == [res["num_erasures"], res["num_blocks_err"],
                                                   res["num_erasures_exp"], res["num_blocks_err_exp"]]
        assert (out["rows"][t, :res["iterations"], 0].cpu().numpy() == orows["deg1"]).all()
        res, _, _ = _oracle_counters(oracle, po, A[t], bits[t], "sw", W=5, max_it=4, init_it=8)
        assert sw["counters"][t, :5].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                                  res["num_blocks_err_exp"], res["num_erasures_p1"]]


def test_edge_cases(E, oracle):
    import torch
    p = E.make_params(4, 8, 10, 40)
    # empty batch: nothing launched, nothing touched
    d_adj = torch.empty((0, p.n, 4), dtype=torch.int32, device="cuda")
    d_ch = torch.empty((0, p.nw), dtype=torch.int32, device="cuda")
    assert E.full_bp(p, d_adj, d_ch)["counters"].shape == (0, 8)
    assert E.sw_bp(p, d_adj, d_ch, 3, 2)["counters"].shape == (0, 8)
    # nothing erased / everything erased
    d_adj, d_ch = E.sample_philox(p, 1, 0, 3, 0.0)
    c = E.full_bp(p, d_adj, d_ch)["counters"].cpu().numpy()
    assert (c[:, :5] == 0).all() and (c[:, 5] == 1).all() and (c[:, 7] == 0).all()
    d_adj, d_ch = E.sample_philox(p, 1, 0, 3, 1.0)
    c = E.full_bp(p, d_adj, d_ch)["counters"].cpu().numpy()
    assert (c[:, 7] == p.n).all() and (c[:, 1] == p.L).all()
    # even a fully erased word loses a few VNs: boundary CNs of degree 1 resolve their only neighbour
    po = oracle.Params(4, 8, 10, p.cns_pos, p.vns_pos)
    A = d_adj.cpu().numpy()
    for t in range(3):
        res, _, _ = oracle.decode_bp(oracle.Graph.from_vn_adj(po, A[t]), np.ones(p.n, np.uint8), literal=True)
        assert c[t, :4].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                     res["num_blocks_err_exp"]] and c[t, 5] == res["iterations"]
        assert res["num_erasures"] < p.n
    # ragged n (not a multiple of 32) is covered by (9,24) above; ensembles beyond the LDS budget are refused
    big = E.make_params(4, 8, 50, 100000)
    with pytest.raises(E.ScldpcError, match="2\\^24|160 KiB"):
        E.full_bp(big, torch.empty((1, big.n, 4), dtype=torch.int32, device="cuda"),
                  torch.empty((1, big.nw), dtype=torch.int32, device="cuda"))


def test_accumulate_run_ordered_stop(E):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fakes import fake_counters, numpy_accumulate
    import torch
    for T in (1, 5, 1023, 1024, 1025, 5000):
        cnt = fake_counters(3, np.arange(T), 400, 10)
        d = torch.from_numpy(cnt).cuda()
        for stop in (0, 1, 7, 333, 100000):
            run = E.new_run()
            run[1] = 2                                        # a run already holding 2 frame errors
            E.accumulate_run(d, run, stop)
            ref = np.zeros(E.NRUN, dtype=np.int64); ref[1] = 2
            assert run.cpu().tolist() == numpy_accumulate(cnt, ref, stop).tolist(), (T, stop)


@pytest.mark.parametrize("L,N,eps", [(50, 1000, 0.48), (16, 200, 0.44), (9, 24, 0.5), (12, 2000, 0.46)])
def test_compact_adjacency_is_equivalent(E, L, N, eps):
    """uint16 position-local adjacency: same codes, same decoder outputs as the int32 table, bit for bit."""
    import torch
    p = E.make_params(4, 8, L, N)
    T = 12
    a32, c32 = E.sample_philox(p, 11, 500, T, eps, doped=(2,))
    a16, c16 = E.sample_philox(p, 11, 500, T, eps, doped=(2,), adj16=True)
    torch.cuda.synchronize()
    assert (c32 == c16).all()
    assert (E.adj16_to_global(p, a16.cpu().numpy()) == a32.cpu().numpy()).all()
    assert (E.global_to_adj16(p, a32.cpu().numpy()[0]) == a16.cpu().numpy()[0]).all()
    for kw in (dict(), dict(max_it=4), dict(is_term=False), dict(rows_cap=1024)):
        o32 = E.full_bp(p, a32, c32, want_erased=True, **kw)
        o16 = E.full_bp(p, a16, c16, want_erased=True, **kw)
        assert (o32["counters"] == o16["counters"]).all() and (o32["erased"] == o16["erased"]).all(), kw
        if kw.get("rows_cap"):
            assert (o32["rows"] == o16["rows"]).all()
    if p.nk * 4 < 150_000:
        s32 = E.sw_bp(p, a32, c32, 5, 4, 9, want_erased=True)
        s16 = E.sw_bp(p, a16, c16, 5, 4, 9, want_erased=True)
        assert (s32["counters"] == s16["counters"]).all() and (s32["erased"] == s16["erased"]).all()


@pytest.mark.parametrize("L,N,eps", [(50, 5000, 0.47), (50, 10000, 0.46), (100, 2000, 0.47)])
def test_ensembles_beyond_the_lds_use_the_global_workspace(E, oracle, L, N, eps):
    """bp_traj's shipped size (N=5000), the notebook's peeling size (N=10000) and BASELINE config 4 (L=100, N=2000):
    the CN words move to a global-memory workspace; results stay bit-exact (rows included)."""
    import torch
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    T = 3
    d_adj, d_ch = E.sample_philox(p, 5, 0, T, eps, adj16=(N <= 5000)) if p.cns_pos * 8 <= 8192 else (None, None)
    if d_adj is None:                       # the device sampler ranks up to 8192 sockets per position: sample on the host
        adj, ch = E.sample_glibc_trials(p, [11, 12, 13], eps)
        d_adj, d_ch = E.to_device(adj, ch)
    A = d_adj.cpu().numpy()
    if A.dtype == np.int16:
        A = E.adj16_to_global(p, A)
    bits = E.unpack_bits(d_ch.cpu().numpy(), p.n)
    out = E.full_bp(p, d_adj, d_ch, rows_cap=4096, want_erased=True)
    lim = E.full_bp(p, d_adj, d_ch, max_it=25, is_term=False)
    torch.cuda.synchronize()
    c, rows = out["counters"].cpu().numpy(), out["rows"].cpu().numpy()
    for t in range(T):
        g = oracle.Graph.from_vn_adj(po, A[t])
        res, erased, orows = oracle.decode_bp(g, bits[t], literal=False, rows_cap=4096)
        assert c[t, :4].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                     res["num_blocks_err_exp"]] and c[t, 5] == res["iterations"]
        k = res["iterations"]
        assert (rows[t, :k, 0] == orows["deg1"]).all() and (rows[t, :k, 1] == orows["recovered"]).all() \
            and (rows[t, :k, 2] == orows["first_pos"]).all()
        assert (E.unpack_bits(out["erased"][t].cpu().numpy(), p.n) == erased).all()
        res2, _, _ = oracle.decode_bp(g, bits[t], max_it=25, is_term=0, literal=False)
        assert lim["counters"][t, :4].tolist() == [res2["num_erasures"], res2["num_blocks_err"],
                                                   res2["num_erasures_exp"], res2["num_blocks_err_exp"]]
    sw = E.sw_bp(p, d_adj, d_ch, 10, 20, 0)["counters"].cpu().numpy()
    for t in range(T):
        res, _ = oracle.decode_sw(oracle.Graph.from_vn_adj(po, A[t]), bits[t], 10, 20, 0, literal=False)
        assert sw[t, :5].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                      res["num_blocks_err_exp"], res["num_erasures_p1"]]


@pytest.mark.parametrize("path", ["fused", "nibble", "wide", "nibble+wide"])
@pytest.mark.parametrize("L,N,eps,doped", [(8, 5000, 0.47, ()), (6, 10000, 0.46, (2,)), (5, 16384, 0.5, ()), (5, 2500, 0.4, ())])
def test_big_ensemble_sampler_equals_cpu_twin(E, oracle, monkeypatch, L, N, eps, doped, path):
    """More than 8192 sockets per position (sample_philox_big_kernel): same integers as the twin through every ranking the
    kernel has — fused in LDS (round 3: stage of the position's sockets, straddling buckets ordered per lane, the socket -> CN row
    as the stage's inverse), nibble-wide counters with the straddlers' records in the workspace (SCLDPC_DEBUG_SAMPLER_FUSED=0),
    and the 16-bit-counter fallback of either (forced: SCLDPC_DEBUG_SAMPLER_WIDE)."""
    import torch
    if "nibble" in path:
        monkeypatch.setenv("SCLDPC_DEBUG_SAMPLER_FUSED", "0")
    if "wide" in path:
        monkeypatch.setenv("SCLDPC_DEBUG_SAMPLER_WIDE", "1")
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    T = 3
    for adj16 in (False, True):
        d_adj, d_ch = E.sample_philox(p, 2, 40, T, eps, doped, adj16=adj16)
        torch.cuda.synchronize()
        A = d_adj.cpu().numpy()
        if adj16:
            A = E.adj16_to_global(p, A)
        Cb = d_ch.cpu().numpy().view(np.uint32)
        for t in range(T):
            a, c = oracle.sample_philox(po, 2, 40 + t, eps, doped)
            assert (a == A[t]).all() and (c == Cb[t]).all(), (L, N, adj16, t)


@pytest.mark.parametrize("L,N,eps,is_term", [(20, 5000, 0.47, True), (12, 10000, 0.46, False), (30, 2500, 0.48, True)])
def test_workspace_cn_words_built_through_lds_equal_the_atomic_build(E, monkeypatch, L, N, eps, is_term):
    """Ensembles whose CN words live in the workspace (N >= 2500, e.g. bp_traj's shipped Def_M = 2500): cn_build.hip builds
    the words through an LDS ring of dv CN positions instead of one global atomic per edge — every counter, the residuals
    and the trajectory rows equal the in-kernel build's."""
    import torch
    p = E.make_params(4, 8, L, N)
    a, c = E.sample_philox(p, 9, 0, 24, eps, adj16=True)
    res = {}
    for pre in ("0", "1"):
        monkeypatch.setenv("SCLDPC_DEBUG_FULLBP_PREBUILD", pre)
        res[pre] = [E.full_bp(p, a, c, is_term=is_term, want_erased=True),
                    E.full_bp(p, a, c, is_term=is_term, rows_cap=1024, want_erased=True)]
        torch.cuda.synchronize()
    for x, y in zip(res["0"], res["1"]):
        assert torch.equal(x["counters"], y["counters"]) and torch.equal(x["erased"], y["erased"])
        assert (x["rows"] is None and y["rows"] is None) or torch.equal(x["rows"], y["rows"])


@pytest.mark.parametrize("name", golden_names(variants=("bpfsw",)))
def test_classical_window_matches_reference_golden(E, name):
    """decodeBP_SW of BPF:627-897 (classical window) against the real reference's outputs."""
    import torch
    g = load_golden(name)
    m = g.meta
    p, T, (d_adj, d_ch) = _golden_inputs(E, g)
    out = E.sw_bp(p, d_adj, d_ch, m["W"], m["max_it"], want_erased=True, classical=True)
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy()
    for col, key in ((0, "ne"), (1, "be"), (2, "ee"), (3, "bee"), (4, "p1"), (7, "nch")):
        assert (c[:, col] == g[key][:T]).all(), (name, key)
    if g.has("erased"):
        assert (E.unpack_bits(out["erased"].cpu().numpy(), p.n) == g["erased"][:T]).all()


@pytest.mark.parametrize("L,N,W,max_it,eps", [(50, 1000, 20, 6, 0.465), (16, 200, 5, 3, 0.45), (9, 24, 2, 1, 0.5), (14, 1200, 5, 4, 0.47),
                                              (16, 200, 30, 50, 0.47), (100, 2000, 10, 20, 0.47)])
def test_classical_window_equals_oracle(E, oracle, L, N, W, max_it, eps):
    import torch
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    T = 4 if N >= 1000 else 16
    d_adj, d_ch = E.sample_philox(p, 13, 7, T, eps, adj16=True)
    out = E.sw_bp(p, d_adj, d_ch, W, max_it, want_erased=True, classical=True)
    torch.cuda.synchronize()
    A = E.adj16_to_global(p, d_adj.cpu().numpy())
    bits = E.unpack_bits(d_ch.cpu().numpy(), p.n)
    c = out["counters"].cpu().numpy()
    er = E.unpack_bits(out["erased"].cpu().numpy(), p.n)
    for t in range(T):
        res, erased = oracle.decode_sw(oracle.Graph.from_vn_adj(po, A[t]), bits[t], W, max_it, literal=True, square=False)
        assert c[t, :5].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                     res["num_blocks_err_exp"], res["num_erasures_p1"]], (L, N, W, t)
        assert c[t, 5] == res["iterations"] and (er[t] == erased).all()


@pytest.mark.parametrize("L,N", [(50, 1000), (16, 200), (9, 24), (60, 1024), (14, 1200)])
@pytest.mark.parametrize("eps", [0.02, 0.3, 0.44, 0.47, 0.48, 0.49, 0.52, 0.7, 0.95])
@pytest.mark.parametrize("is_term", [True, False])
def test_fixpoint_kernel_equals_flooding_kernel(E, L, N, eps, is_term):
    """Chain-following peeling (scldpc_full_bp_fixpoint_device) against the level-synchronous kernel: every counter
    except the iteration count, and the residual erasure pattern, trial by trial; both adjacency formats."""
    import torch
    p = E.make_params(4, 8, L, N)
    T = 48 if N >= 1000 else 96
    d_adj, d_ch = E.sample_philox(p, 5, int(eps * 100) * 1000 + L, T, eps, adj16=True)
    ref = E.full_bp(p, d_adj, d_ch, is_term=is_term, want_erased=True)
    adj32 = torch.from_numpy(E.adj16_to_global(p, d_adj.cpu().numpy())).to(d_adj.device)
    for adj in (d_adj, adj32):
        out = E.full_bp_fixpoint(p, adj, d_ch, is_term=is_term, want_erased=True)
        torch.cuda.synchronize()
        c, r = out["counters"].cpu().numpy(), ref["counters"].cpu().numpy()
        cols = [0, 1, 2, 3, 4, 6, 7]
        assert (c[:, cols] == r[:, cols]).all(), (L, N, eps, np.argwhere(c[:, cols] != r[:, cols])[:3].tolist())
        assert (out["erased"].cpu().numpy() == ref["erased"].cpu().numpy()).all()


@pytest.mark.parametrize("name", golden_names(variants=("bpf",), uncapped=True))
def test_fixpoint_kernel_matches_reference_golden(E, name):
    """The real reference's unlimited-iteration outputs (fixtures with max_it = 0 only)."""
    import torch
    g = load_golden(name)
    m = g.meta
    assert not g.max_it
    p, T, (d_adj, d_ch) = _golden_inputs(E, g)
    out = E.full_bp_fixpoint(p, d_adj, d_ch, is_term=bool(m["is_term"]), want_erased=True)
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy()
    for col, key in ((0, "ne"), (1, "be"), (2, "ee"), (3, "bee"), (7, "nch")):
        assert (c[:, col] == g[key][:T]).all(), (name, key)
    if g.has("erased"):
        assert (E.unpack_bits(out["erased"].cpu().numpy(), p.n) == g["erased"][:T]).all()
