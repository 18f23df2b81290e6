"""Streaming mode on the GPU (-m gpu): stream_bp_kernel through the C-ABI against its CPU twin (the streaming
oracle with Philox keys and the node-level decoder — itself pinned to the reference's CIRCULAR build), position by
position, across launches and buffer wrap-arounds."""
import numpy as np
import pytest

from conftest import require_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    require_gpu()
    from fl_scaling_sc_ldpc_amd import engine
    return engine


@pytest.mark.parametrize("L,N,eps,W,doped,chunks", [
    (20, 10, 0.45, 6, (), (40, 35, 60)), (20, 10, 0.47, 7, (5, 6), (100, 50)), (20, 10, 0.5, 4, (9,), (90,)),
    (30, 100, 0.47, 10, (), (70, 30)), (30, 100, 0.49, 12, (10, 11, 12), (100,)),
    (50, 1000, 0.47, 20, (), (30, 30)), (50, 1000, 0.485, 20, (10, 11, 12), (60,)), (50, 5000, 0.47, 20, (24,), (12,))])
def test_streams_equal_cpu_twin(E, oracle, L, N, eps, W, doped, chunks):
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    ns = 3
    st = E.Streams(p, ns, seed=17, eps=eps, W=W, doped=doped, stream0=5)
    twins = [oracle.Stream(po, 17, eps, W, doped, rng_mode=1, decoder=1, sid=5 + s) for s in range(ns)]
    for npos in chunks:                                   # the state carries over from launch to launch
        cnt, tr = st.run(npos, trace=True)
        tr = tr.cpu().numpy(); cnt = cnt.cpu().numpy()
        for s in range(ns):
            for k in range(npos):
                o = twins[s].step()
                assert tr[s, k].tolist() == [o[f] for f in oracle.Stream.FIELDS], (L, N, s, k)
            assert cnt[s, :8].tolist() == [o[f] for f in oracle.Stream.FIELDS[2:]] and cnt[s, 8] == o["pos"] + 1


@pytest.mark.parametrize("env", ["", "SCLDPC_DEBUG_STREAM_WIDE", "SCLDPC_DEBUG_STREAM_LEGACY"])
@pytest.mark.parametrize("dc,N,eps,W", [(6, 300, 0.6, 12), (12, 300, 0.3, 10), (10, 250, 0.36, 8)])
def test_streams_with_other_check_degrees_equal_cpu_twin(E, oracle, monkeypatch, dc, N, eps, W, env):
    """dc != 8 (no power of two: CN = rank / dc by division; CN rows of 12, 20 or 24 bytes read entry by entry): every
    generation path and the decoder against the CPU twin, position by position."""
    if env:
        monkeypatch.setenv(env, "1")
    L = 30
    p = E.make_params(4, dc, L, N)
    po = oracle.Params(4, dc, L, p.cns_pos, p.vns_pos)
    st = E.Streams(p, 2, seed=31, eps=eps, W=W, doped=(7, 8, 9), stream0=3)
    twins = [oracle.Stream(po, 31, eps, W, (7, 8, 9), rng_mode=1, decoder=1, sid=3 + s) for s in range(2)]
    for npos in (45, 40):
        cnt, tr = st.run(npos, trace=True)
        tr = tr.cpu().numpy()
        for s in range(2):
            for k in range(npos):
                o = twins[s].step()
                assert tr[s, k].tolist() == [o[f] for f in oracle.Stream.FIELDS], (dc, s, k)


@pytest.mark.parametrize("env", ["SCLDPC_DEBUG_STREAM_WIDE", "SCLDPC_DEBUG_STREAM_LEGACY"])
@pytest.mark.parametrize("L,N,eps,W,doped", [(30, 100, 0.49, 12, (10, 11, 12)), (50, 1000, 0.485, 20, (10, 11, 12)), (50, 5000, 0.47, 20, ())])
def test_ranking_fallback_with_16_bit_counters_gives_the_same_stream(E, oracle, monkeypatch, L, N, eps, W, doped, env):
    """The generation kernel ranks a CN position's keys with nibble-wide bucket counters, the CN rows staged beside them; a
    position in which sixteen keys meet in one bucket (never on real draws) is ranked again with 16-bit counters and its CN
    rows built in a pass of their own.  Forced here (SCLDPC_DEBUG_STREAM_WIDE); SCLDPC_DEBUG_STREAM_LEGACY runs the kernel's
    other LDS layout (the one of ensembles beyond 32 768 sockets per position: no separate stage).  Either way the stream
    must still equal its CPU twin position by position."""
    monkeypatch.setenv(env, "1")
    p = E.make_params(4, 8, L, N)
    po = oracle.Params(4, 8, L, p.cns_pos, p.vns_pos)
    st = E.Streams(p, 2, seed=23, eps=eps, W=W, doped=doped, stream0=9)
    twins = [oracle.Stream(po, 23, eps, W, doped, rng_mode=1, decoder=1, sid=9 + s) for s in range(2)]
    npos = 40 if N <= 1000 else 10
    cnt, tr = st.run(npos, trace=True)
    tr = tr.cpu().numpy()
    for s in range(2):
        for k in range(npos):
            o = twins[s].step()
            assert tr[s, k].tolist() == [o[f] for f in oracle.Stream.FIELDS], (L, N, s, k)


def test_a_stream_whose_ranking_overflowed_is_marked_unusable_not_trapped(E, monkeypatch, tmp_path):
    """256 keys of a position in one of the fallback ranking's buckets cannot be ranked with its byte-wide arrival slots: no
    real draw does that, but the kernel must neither trap (a GPU fault resets the node for everyone) nor rank wrongly.
    Forced here (SCLDPC_DEBUG_STREAM_WIDE=2): the stream's "positions generated" column turns negative, later calls leave its
    counters alone, and the `sw` driver stops with a message."""
    import torch
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    monkeypatch.setenv("SCLDPC_DEBUG_STREAM_WIDE", "2")
    st = E.Streams(E.make_params(4, 8, 20, 10), 3, 5, 0.47, 6, (5, 6))
    c1 = st.run(8)[0].clone()
    c2 = st.run(8)[0].clone()
    torch.cuda.synchronize()
    assert (c1[:, 9] < 0).all() and torch.equal(c1, c2) and (c1[:, :8] == 0).all()
    with pytest.raises(SystemExit, match="unusable"):
        B.streaming(["2", "6", "2", "5", "6", "--L", "20", "--N", "10", "--eps-ini", "0.47", "--num-points", "1",
                     "--streams", "4", "--chunk", "25", "--seed", "3", "--outdir", str(tmp_path), "--quiet"])


def test_streaming_cli_writes_results_circular_rows(E, tmp_path):
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    B.streaming(["2", "6", "2", "5", "6", "--L", "20", "--N", "10", "--eps-ini", "0.47", "--num-points", "1",
                 "--max-blocks-err", "50", "--max-blocks", "4000", "--streams", "4", "--chunk", "25", "--seed", "3",
                 "--outdir", str(tmp_path), "--quiet"])
    rows = open(tmp_path / "SC_LDPC_4_8_L20_M5_DOP2_BP_Stream_SW6_Random_BLER_2.dat").read().strip().split("\n")
    assert rows[0] == B.STREAM_HEADER.strip() and len(rows) == 2
    f = rows[1].split()
    assert len(f) == 13 and f[0] == "0.470000"
    ne, gb, be, gbl, ee, gbe, bee, gble = (int(x) for x in f[5:])
    assert bee >= 50 or gble >= 4000
    assert float(f[1]) == pytest.approx(ne / gb) and float(f[4]) == pytest.approx(bee / gble)
    st = E.Streams(E.make_params(4, 8, 20, 10), 4, 3, 0.47, 6, (5, 6), stream0=0)     # same streams, same chunks
    tot = None
    while tot is None or (tot[3] < 50 and tot[7] < 4000):
        tot = st.run(25)[0][:, :8].sum(dim=0).cpu().numpy()
    assert [ne, be, ee, bee, gb, gbl, gbe, gble] == tot.tolist()


_RUNS = _json_runs = None


def _whole_runs():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_wholeruns.json")))["runs"]


@pytest.mark.parametrize("k", range(5))
@pytest.mark.parametrize("chunk", [7, 64])
def test_streaming_cli_replays_whole_reference_runs_row_for_row(E, tmp_path, k, chunk):
    """`sw INDEX W NUM_DOPED … --rng glibc --seed S`: the reference's own experiment — ONE stream, one srandom(seed), the ε
    points back to back with random() carried from point to point, each stopped where main_streaming stops (BPF:2033).
    Fixtures: whole runs of the REAL reference (oracle/make_golden_stream_runs.py), N = 10 … 1000, with and without doping.
    The chunk size (positions per launch) must not matter: a point that trips inside a chunk rewinds the host stream to
    where the reference stopped drawing."""
    from fl_scaling_sc_ldpc_amd import bp_decoding as B
    r = _whole_runs()[k]
    d = r["doped"]
    B.streaming(["3", str(r["W"]), str(len(d)), *[str(x) for x in d], "--L", str(r["L"]), "--N", str(2 * r["Def_M"]),
                 "--eps-ini", repr(r["eps_ini"]), "--eps-delta", repr(r["eps_delta"]), "--num-points", str(r["num_points"]),
                 "--max-blocks-err", str(r["max_blocks_err"]), "--max-blocks", str(r["max_blocks"]), "--rng", "glibc",
                 "--seed", str(r["seed"]), "--chunk", str(chunk), "--outdir", str(tmp_path), "--quiet"])
    name = "SC_LDPC_4_8_L%d_M%d_DOP%d_BP_Stream_SW%d_Random_BLER_3.dat" % (r["L"], r["Def_M"], len(d), r["W"])
    rows = open(tmp_path / name).read().strip().split("\n")
    assert rows[0] == B.STREAM_HEADER.strip() and len(rows) == 1 + r["num_points"]
    for row, ref in zip(rows[1:], r["rows"]):
        f = row.split()
        ne, be, ee, bee, gb, gbl, gbe, gble = ref["counters"]
        assert [int(x) for x in f[5:]] == [ne, gb, be, gbl, ee, gbe, bee, gble], (f, ref)       # results_circular's order
        assert f[0] == "%f" % ref["eps"]


def test_streaming_rejects_windows_beyond_the_generated_stream(E):
    with pytest.raises(E.ScldpcError, match="L/2"):
        E.Streams(E.make_params(4, 8, 20, 10), 1, 1, 0.4, 9)


# ---- same-input mode: the device kernel fed with the reference's own stream ----------------------------------------------
import glob as _glob
import os
import json as _json

_STREAM_GOLDEN = sorted(_glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_*.npz")))


@pytest.mark.parametrize("path", _STREAM_GOLDEN, ids=[os.path.basename(p)[:-4] for p in _STREAM_GOLDEN])
def test_stream_kernel_on_the_reference_stream_equals_the_reference(E, path):
    """stream_bp.hip fed with main_streaming's own draws — scldpc_stream_glibc_inputs_host replays srandom(seed),
    fill_interleaver_pos and generate_channel_doped_circular (BPF:1763-1787, 1621-1654) — against the rows the REAL
    reference (its CIRCULAR build, oracle/make_golden_stream.py) printed position by position: the value returned by
    decodeBP_SW_circular and all eight counters of results_circular, across buffer wrap-arounds, with and without doping,
    incl. the N = 1000 fixtures of BASELINE config 5's ensemble family.  Two launches (continuation from the state blob)
    and two stream slots (the second fed with a different seed's inputs must not disturb the first)."""
    import torch
    z = np.load(path)
    m = _json.loads(str(z["meta"]))
    p = E.make_params(4, 8, m["L"], 2 * m["Def_M"])
    P = int(m["P"])
    G = m["L"] // 2 + P
    inter, chan = E.stream_glibc_inputs(p, m["seed"], m["eps"], m["doped"], G)
    inter2, chan2 = E.stream_glibc_inputs(p, m["seed"] + 1, m["eps"], m["doped"], G)
    st = E.InputStreams(p, np.stack([inter, inter2]), np.stack([chan, chan2]), m["W"], m["doped"])
    first = P // 3
    _, tr1 = st.run(first, trace=True)
    cnt, tr2 = st.run(P - first, trace=True)
    torch.cuda.synchronize()
    rows = np.concatenate([tr1[0].cpu().numpy(), tr2[0].cpu().numpy()])
    assert (rows == z["rows"][:P]).all(), np.argwhere(rows != z["rows"][:P])[:3]
    c = cnt[0].cpu().numpy()
    assert c[:8].tolist() == z["rows"][P - 1][2:].tolist() and c[8] == P and c[9] == G


# ---- many short streams against few long ones -----------------------------------------------------------
def _stream_pieces(E, p, nstreams, npieces, piece, seed, eps, doped, stream0, W=20):
    """Block errors and blocks after expurgation (counters 3, 7) of every (stream, piece of `piece` positions)."""
    import torch
    st = E.Streams(p, nstreams, seed=seed, eps=eps, W=W, doped=doped, stream0=stream0)
    prev = torch.zeros_like(st.counters)
    out = []
    for _ in range(npieces):
        c, _ = st.run(piece)
        out.append((c - prev)[:, [3, 7]].cpu().numpy())
        prev = c.clone()
    return np.stack(out, axis=1)                                # [stream][piece][errors, blocks]


def test_many_short_doped_streams_estimate_the_same_bler_as_few_long_ones(E):
    """The reference's streaming experiment is ONE long stream (main_streaming, BPF:1934-2054); `sw --streams S` sums S
    independent streams advanced in lock step.  With doping that decouples the chain (dv - 1 = 3 consecutive known positions
    per period: doped {10, 11, 12}) a stream is a sequence of independent, identically distributed segments, the first —
    which starts from the known left end — included, so 512 streams x 2000 positions and 4 streams x 256 000 positions see
    the same process: their BLERs agree within the spread of 2000-position samples (profiles/r03_stream_equivalence.txt has
    the same at other eps: z = +0.6 at eps = 0.48).  Tolerance: |z| < 4 on >= 200 block errors each."""
    p = E.make_params(4, 8, 50, 1000)
    a = _stream_pieces(E, p, 512, 1, 2000, 11, 0.482, (10, 11, 12), 0).reshape(-1, 2).astype(np.float64)
    b = _stream_pieces(E, p, 4, 128, 2000, 11, 0.482, (10, 11, 12), 1 << 20).reshape(-1, 2).astype(np.float64)

    def est(s):
        bler = s[:, 0].sum() / s[:, 1].sum()
        return bler, np.std(s[:, 0] - bler * s[:, 1], ddof=1) * np.sqrt(len(s)) / s[:, 1].sum()
    (ba, sa), (bb, sb) = est(a), est(b)
    assert a[:, 0].sum() >= 200 and b[:, 0].sum() >= 200, (a[:, 0].sum(), b[:, 0].sum())
    assert abs(ba - bb) < 4.0 * np.hypot(sa, sb), (ba, sa, bb, sb)


def test_an_undoped_stream_that_failed_keeps_failing(E):
    """Without doping the two estimators are NOT the same experiment, and no burn-in repairs it: once a window has lost its
    known left end every later block is in error (measured: share 1.0), so the reference's single-stream figure is
    errors / (time to the first failure + errors) — a property of ONE stream.  The driver therefore documents `--streams 1`
    (with `--rng glibc`: the reference's run row for row) for undoped ensembles and many streams for doped ones
    (INTEGRATION.md §streaming).  Here: 8 undoped streams near threshold, 40 pieces of 500 positions; every stream that
    fails does so for good."""
    p = E.make_params(4, 8, 50, 1000)
    s = _stream_pieces(E, p, 8, 40, 500, 5, 0.455, (), 0)
    bad = s[:, :, 0] > 0                                        # [stream][piece]
    assert bad.any()
    for k in range(bad.shape[0]):
        if bad[k].any():
            first = int(np.argmax(bad[k]))
            assert bad[k, first:].all(), (k, first, bad[k].astype(int))
            assert (s[k, first + 1:, 0] >= 0.9 * s[k, first + 1:, 1]).all()     # (nearly) every later block of the stream is in error
