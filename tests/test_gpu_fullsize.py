"""BASELINE.json sizes on the GPU: (4,8), L=50, N=1000, 10^5 trials of full BP — checked through
size-independent properties, against the statistical pins the reference publishes, and bit-exactly
against the CPU oracle on a strided subset."""
import numpy as np
import pytest

from conftest import require_gpu

pytestmark = pytest.mark.gpu
T_TOTAL, BATCH = 100_000, 5_000


@pytest.fixture(scope="module")
def big_run():
    require_gpu()
    import torch
    from fl_scaling_sc_ldpc_amd import engine as E
    p = E.make_params(4, 8, 50, 1000)
    d_adj = torch.empty((BATCH, p.n, 4), dtype=torch.int32, device="cuda")
    d_ch = torch.empty((BATCH, p.nw), dtype=torch.int32, device="cuda")
    d_a16 = torch.empty((BATCH, p.n, 4), dtype=torch.int16, device="cuda")
    d_cn16 = torch.empty((BATCH, p.nk, 8), dtype=torch.int16, device="cuda")
    d_ch2 = torch.empty((BATCH, p.nw), dtype=torch.int32, device="cuda")
    cnts = []
    keep = {}
    KEEP = [0, 1, 2, 3, 4, 6, 7]                                    # all but the iteration / barrier-round count
    for b in range(T_TOTAL // BATCH):
        E.sample_philox(p, 2024, b * BATCH, BATCH, 0.48, out=(d_adj, d_ch))
        out = E.full_bp(p, d_adj, d_ch, want_erased=(b == 0))
        cnts.append(out["counters"].cpu().numpy())
        # bench.py's path on the very same trials: second-generation sampler (uint16 tables) + 4-bit-count decoder, and
        # the first-generation fixpoint kernel on the uint16 adjacency — every counter of every trial must agree with
        # the level-synchronous flooding kernel on the reference's int32 layout
        E.sample_philox_cn16(p, 2024, b * BATCH, BATCH, 0.48, out=(d_a16, d_cn16, d_ch2))
        assert torch.equal(d_ch, d_ch2)
        small = E.full_bp_fixpoint_cn16(p, d_a16, d_cn16, d_ch2)["counters"].cpu().numpy()
        fix = E.full_bp_fixpoint(p, d_a16, d_ch2)["counters"].cpu().numpy()
        assert (small[:, KEEP] == cnts[-1][:, KEEP]).all(), b
        assert (fix[:, KEEP] == cnts[-1][:, KEEP]).all(), b
        # `bench.py --flooding` / bp_lim_iter with a cap: the 4-bit decoder walked one flooding iteration per round —
        # ALL eight counters of every trial, the iteration count among them
        assert torch.equal(E.full_bp_cn16(p, d_a16, d_cn16, d_ch2)["counters"], out["counters"]), b
        if b == 0:
            assert (E.adj16_to_global(p, d_a16[::500].cpu().numpy()) == d_adj[::500].cpu().numpy()).all()
            # properties on the first batch
            lim = E.full_bp(p, d_adj, d_ch, max_it=50)["counters"].cpu().numpy()
            again = E.full_bp(p, d_adj, out["erased"])["counters"].cpu().numpy()
            sw = E.sw_bp(p, d_adj, d_ch, p.L + 3, 1_000_000)["counters"].cpu().numpy()
            keep = dict(first=cnts[0], lim=lim, again=again, sw=sw,
                        adj=d_adj[::250].cpu().numpy(), ch=d_ch[::250].cpu().numpy())
    return E, p, np.concatenate(cnts), keep


def test_fer_and_plr_match_published_pins(big_run):
    """Peeling ≡ unlimited-iteration BP on the BEC; the reference's published (4,8,L=50,N=1000, ε=0.48)
    terminated table gives FER_exp 0.8338 = 83380/100000 and PLR 0.2102
    (sim_data/error_rates/terminated_fer_plr_sc_ldpc_4_8_50_1000.dat, last row; unknown seeds ⇒ statistical pin)."""
    E, p, c, _ = big_run
    assert len(c) == T_TOTAL and (c[:, 6] == 0).all()
    fer_exp = np.mean(c[:, 2] > 0)
    sigma = np.sqrt(0.8338 * (1 - 0.8338) * 2 / T_TOTAL)        # both sides are 10^5-trial estimates
    assert abs(fer_exp - 0.8338) < 4.5 * sigma, fer_exp
    plr = c[:, 0].sum() / (T_TOTAL * p.n)
    assert abs(plr - 0.2102) < 0.004, plr
    assert abs(c[:, 7].mean() / p.n - 0.48) < 2e-4             # channel law
    # BLER: terminated_fer_plr_bler_sc_ldpc_4_8_50_1000.dat, row eps = 0.48: 26781 failed blocks in 802 failed frames of
    # 1000 (BLER 0.5356 at that run's FER 0.802) = 33.39 failed blocks per failed frame; a 1000-trial estimate, so the
    # comparison is on the per-failed-frame mean with its standard error
    failed = c[:, 0] > 0
    per_frame = c[failed, 1].astype(np.float64)
    se = per_frame.std() / np.sqrt(802.0)
    assert abs(per_frame.mean() - 26781 / 802) < 4.5 * se, (per_frame.mean(), se)
    bler = c[:, 1].sum() / (p.L * T_TOTAL)
    assert abs(bler - 0.5356 * failed.mean() / 0.802) < 4.5 * se / p.L, bler


def test_size_independent_properties(big_run):
    E, p, c, k = big_run
    first, lim, again, sw = k["first"], k["lim"], k["again"], k["sw"]
    # more iterations never leave more erasures; capped runs stop at the cap
    assert (lim[:, 0] >= first[:, 0]).all() and (lim[:, 5] <= 50).all()
    assert ((lim[:, 0] == first[:, 0]) | (lim[:, 5] == 50)).all()
    # the residual is a fixpoint: decoding it again changes nothing; the loop needs two passes to see
    # "no progress" (NumErasuresPrec starts at n, BPF:910,1045), one when nothing is erased (BPF:1044)
    assert (again[:, 0] == first[:, 0]).all() and (again[:, 1] == first[:, 1]).all()
    assert (again[first[:, 0] > 0, 5] == 2).all() and (again[first[:, 0] == 0, 5] == 1).all()
    # a window as long as the chain with no cap is full BP (same residual ⇒ same NumErasures / blocks)
    assert (sw[:, 0] == first[:, 0]).all() and (sw[:, 1] == first[:, 1]).all()
    # counters are internally consistent
    assert ((c[:, 0] > 0) == (c[:, 1] > 0)).all() and (c[:, 2] <= c[:, 0]).all() and (c[:, 3] <= 1).all()
    assert (c[:, 0] <= c[:, 7]).all()


def test_strided_subset_equals_oracle(big_run, oracle):
    E, p, c, k = big_run
    po = oracle.Params(4, 8, 50, 500, 1000)
    bits = E.unpack_bits(k["ch"], p.n)
    for i in range(len(k["adj"])):
        res, _, _ = oracle.decode_bp(oracle.Graph.from_vn_adj(po, k["adj"][i]), bits[i], literal=False)
        t = i * 250
        assert k["first"][t, :4].tolist() == [res["num_erasures"], res["num_blocks_err"], res["num_erasures_exp"],
                                              res["num_blocks_err_exp"]] and k["first"][t, 5] == res["iterations"]
