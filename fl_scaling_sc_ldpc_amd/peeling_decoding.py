"""Host-side mirror of the reference's Python peeling simulators
(simulators_sc_ldpc/peeling_decoding/peeling_decoding.py = PD): same function names, argument meaning, return
tuples, argv contracts and output formats — with the per-trial work (sweep peeling + stopping-set components,
random-pick peeling with the degree-1 trajectory) done by the HIP kernels of libscldpc_hip.so.

    simulate_sc_ldpc(e, l, r, L, M, is_terminated, is_protograph, is_bounded, is_tail_biting,
                     num_repeats=int(1e5), max_fuckups=2000, doping_points=[])          → 13-tuple   (PD:591-701)
    simulate_peeling_decoder_ldpc(e, l_deg, r_deg, L, M, is_terminated, is_protograph,
                                  num_repeats=None, doping_points=[])                   → (None, r1, plrs)  (PD:705-789)
    main_simulate_sc_ldpc()   = ber_sim.py            (PD:1327-1356)
    main_simulate_variance()  = simulate_variance.py  (PD:1264-1294)
    test_sc_ldpc()            = `python3 peeling_decoding.py`  (PD:1212-1245)

Two sampling modes (keyword `rng`, not in the reference's signatures):
  * rng="numpy" (default): the reference's own streams — the GLOBAL numpy RandomState for codes and channels
    (sc_ldpc.gen_slots → np.random.permutation, np.random.rand; PD:153-154) and the GLOBAL Python `random` for the
    picks (random.choice, PD:1026).  After `np.random.seed(s); random.seed(s)` the results equal the reference's
    bit for bit, and both streams are left exactly where the reference leaves them.
  * rng="philox": codes, channels and picks drawn on the device (counter-based, keyed by (seed, trial)); same
    ensemble law, any batch size / GPU count gives the same numbers.
Ensembles: the Olmos semi-structured chain (default), its tail-biting closure (flag TB, sc_ldpc.py:41-62), the
protograph-based chain (flag P, sc_ldpc_protograph.py) and the uncoupled (l,r) ensemble with repeat rejection
(ldpc.py; simulate_peeling_decoder_ldpc_uncoupled).  P together with TB is refused: the reference reduces the CN
indices mod L there (PD:204-205), which is not an ensemble.
"""
import ast
import os
import pickle
import random
import sys

import numpy as np
import torch

from . import engine as E


# ------------------------------------------------------------------------------------------------
# sampling with the reference's own streams (sc_ldpc.py:22-56; PD:147-195)
# ------------------------------------------------------------------------------------------------
def gen_slots(l, r, L, M):
    """sc_ldpc.gen_slots from the global numpy stream: `transmissions` int32 [L*M, l]."""
    num_cns = int(l * M / r)
    D = L + l - 1
    cn = np.stack([i * num_cns + np.random.permutation(l * M).reshape(l, M) // r for i in range(D)])
    tr = np.empty((L, M, l), dtype=np.int32)
    for d in range(l):
        tr[:, :, d] = cn[d:d + L, d, :]
    return tr.reshape(L * M, l)


def gen_erasures(e, L, M, doping_points):
    """np.random.rand(L*M) <= e (PD:154) with hard / soft doping applied (PD:166-195)."""
    mask = np.random.rand(L * M) <= e
    if isinstance(doping_points, dict):
        for pos, alpha in doping_points.items():
            mask[pos * M: pos * M + int(alpha * M)] = False
    elif len(doping_points):
        mask &= ~np.isin(np.arange(L * M) // M, list(doping_points))
    return mask


def gen_slots_tail_biting(l, r, L, M):
    """sc_ldpc.gen_slots_tail_biting from the global numpy stream: L permutations, edge d of VN position i lands in
    CN position (i+d) % L (sc_ldpc.py:41-45)."""
    num_cns = int(l * M / r)
    cn = np.stack([i * num_cns + np.random.permutation(l * M).reshape(l, M) // r for i in range(L)])
    tr = np.empty((L, M, l), dtype=np.int32)
    for d in range(l):
        tr[:, :, d] = cn[(np.arange(L) + d) % L, d, :]
    return tr.reshape(L * M, l)


def gen_protograph(e, l, r, L, M, doping_points):
    """gen_users_sc_ldpc_protograph(_doping) (PD:198-241) from the global numpy stream → (transmissions, mask).
    Position by position: M/num_cns portions × l permutations of num_cns (edge i of VN u of a portion → CN
    pos*num_cns + i*num_cns + perm_i[u], sc_ldpc_protograph.py:6-20), then rand(M) <= e."""
    if isinstance(doping_points, dict):
        raise NameError("name 'position' is not defined")      # the reference's own failure for soft doping (PD:228)
    num_cns = int(l * M / r)
    portions = int(M / num_cns)
    tr = np.empty((L, M, l), dtype=np.int32)
    mask = np.empty((L, M), dtype=bool)
    for pos in range(L):
        rows = [np.stack([i * num_cns + np.random.permutation(num_cns) for i in range(l)]).T for _ in range(portions)]
        tr[pos] = pos * num_cns + np.vstack(rows)
        mask[pos] = np.random.rand(M) <= e
        if pos in doping_points:
            mask[pos] = False                                   # PD:237
    return tr.reshape(L * M, l), mask.reshape(L * M)


def gen_slots_uncoupled(l, r, N):
    """ldpc.gen_slots from the global numpy stream: redraw until no VN meets a CN twice (ldpc.py:67-84)."""
    while True:
        tr = (np.random.permutation(l * N).reshape(l, N) // r).T
        if not any(len(np.unique(row)) != l for row in tr):
            return np.ascontiguousarray(tr, dtype=np.int32)


def _sample_numpy(e, l, r, L, M, doping_points, is_protograph, is_tail_biting):
    """One generate_users() of the reference (PD:618-627, 729-737) → (transmissions int32 [L*M, l], mask)."""
    if is_protograph:
        return gen_protograph(e, l, r, L, M, doping_points)
    tr = gen_slots_tail_biting(l, r, L, M) if is_tail_biting else gen_slots(l, r, L, M)
    return tr, gen_erasures(e, L, M, doping_points)


def _dist():
    """(torch.distributed | None, rank, world) — one process per GPU (see bp_decoding.py)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def _free_bytes(device):
    """Free memory on `device` right now (the row buffers and batch sizes are budgeted against it, so that a smaller
    part, or a GPU shared between ranks, is not driven out of memory)."""
    dev = torch.device(device)
    return torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else int(32e9)


def _device(device):
    """Default device: this rank's GPU (LOCAL_RANK), as torch.distributed.run sets it."""
    return device if device is not None else str(E.local_device())


def _split(total, world):
    """Contiguous, even shares of `total` trials: offsets [world + 1]."""
    base, rem = divmod(total, world)
    offs = [0]
    for r in range(world):
        offs.append(offs[-1] + base + (1 if r < rem else 0))
    return offs


def _check_flags(is_protograph, is_tail_biting=False):
    if is_protograph and is_tail_biting:
        raise NotImplementedError("protograph + tail-biting: the reference reduces CN indices mod L (PD:204-205), "
                                  "which merges all CNs of a residue class — not an ensemble this path implements")


def _doped_positions(doping_points):
    return sorted(doping_points.keys()) if isinstance(doping_points, dict) else list(doping_points)


class _Geometry:
    """The index bookkeeping of simulate_sc_ldpc (PD:604-611, 635-648)."""

    def __init__(self, l, r, L, M, is_terminated, is_bounded, doping_points):
        self.ignored_head = 0 if is_bounded else 20
        self.ignored_head_schedule = 0 if is_bounded else 10
        self.ignored_tail = 0 if is_terminated else 20
        self.L = L + self.ignored_head + self.ignored_tail
        self.cpp = int(l / r * M)
        self.num_positions = self.L + l - 1 if is_terminated else self.L
        self.total_size = self.cpp * self.num_positions
        nd = len(doping_points)
        if isinstance(doping_points, dict):
            self.generated = (self.L - self.ignored_head - self.ignored_tail) * M - \
                sum(int(a * M) for a in doping_points.values())
        else:
            self.generated = (self.L - nd - self.ignored_head - self.ignored_tail) * M
        self.blocks = self.L - nd - self.ignored_head - self.ignored_tail
        self.params = E.CodeParams(l, r, self.L, self.cpp, M)
        self.sweep_start = self.ignored_head_schedule * self.cpp
        self.lost_lo = self.cpp * self.ignored_head
        self.lost_hi = self.total_size - self.cpp * self.ignored_tail


def simulate_sc_ldpc(e, l, r, L, M, is_terminated, is_protograph, is_bounded, is_tail_biting, num_repeats=int(1e5),
                     max_fuckups=2000, doping_points=[], rng="numpy", seed=0, batch=None, device=None):
    """Error-rate Monte-Carlo of the sweep peeling decoder (PD:591-701); returns the reference's 13-tuple.
    rng="philox" under torch.distributed: every round of world*batch trials is split evenly over the ranks, the
    per-trial result rows (32 B each) are all-gathered and every rank runs the same ordered accumulation, so the stop
    rule (max_fuckups, PD:698) cuts at the same trial and every rank returns the same tuple as a single rank would."""
    _check_flags(is_protograph, is_tail_biting)
    device = _device(device)
    dist, rank, world = _dist()
    if rng == "numpy" and world > 1:
        raise ValueError("rng='numpy' replays the reference's one sequential stream: single rank only")
    g = _Geometry(l, r, L, M, is_terminated, is_bounded, doping_points)
    p = g.params
    if batch is None:
        batch = 64 if rng == "numpy" else 2048
    # The reference's per-trial bookkeeping and its stop rule (PD:668-699) run on the device, in trial order
    # (scldpc_accumulate_peel_device): the host reads the six totals only in rounds in which max_fuckups could trip.
    run = torch.zeros(E.NPEELRUN, dtype=torch.int64, device=device)
    i_tr, i_fu = E.PEELRUN_NAMES.index("trials"), E.PEELRUN_NAMES.index("fuckups")
    done = 0
    fuckups_bound = 0                                   # upper bound on num_fuckups while nobody has looked
    soft = isinstance(doping_points, dict)
    while done < num_repeats:
        nb = min(batch * world, num_repeats - done)         # trials of this round (all ranks)
        offs = _split(nb, world)
        lo, mine = offs[rank], offs[rank + 1] - offs[rank]
        states = None
        if rng == "numpy":
            adj = np.empty((nb, p.n, l), dtype=np.int32)
            ch = np.empty((nb, p.nw), dtype=np.uint32)
            states = []
            for t in range(nb):
                adj[t], mask = _sample_numpy(e, l, r, g.L, M, doping_points, is_protograph, is_tail_biting)
                ch[t] = E.pack_bits(mask.astype(np.uint8))
                states.append(np.random.get_state())
            d_adj, d_ch = E.to_device(adj, ch, device)
        elif rng == "philox":
            if soft and is_protograph:
                raise NameError("name 'position' is not defined")   # the reference's own failure for soft doping (PD:228)
            ens = "protograph" if is_protograph else "tail_biting" if is_tail_biting else "olmos"
            d_adj, d_ch = E.sample_philox(p, seed, done + lo, mine, e, [] if soft else _doped_positions(doping_points),
                                          device=device, adj16=(ens == "olmos"), ensemble=ens)
            if soft:                                        # PD:176-183: the first int(alpha*M) VNs of the position are known
                for pos, alpha in doping_points.items():
                    E.clear_channel_range(p, d_ch, pos * M, pos * M + int(alpha * M))
        else:
            raise ValueError("rng must be 'numpy' or 'philox'")
        d_out = E.peel_sweep(p, d_adj, d_ch, g.total_size, g.sweep_start, g.lost_lo, g.lost_hi)["out"]
        if world > 1:
            m = max(offs[r + 1] - offs[r] for r in range(world))
            pad = torch.zeros((m, d_out.shape[1]), dtype=d_out.dtype, device=d_out.device)
            pad[:mine] = d_out
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            d_out = torch.cat([parts[r][:offs[r + 1] - offs[r]] for r in range(world)], dim=0).contiguous()
        E.accumulate_peel(d_out, run, max_fuckups)
        fuckups_bound += nb
        if fuckups_bound < max_fuckups and states is None:
            done += nb                                      # cannot have tripped: stay asynchronous
            continue
        tot = run.cpu().numpy()
        used = int(tot[i_tr]) - done
        fuckups_bound = int(tot[i_fu])
        done += used
        if used < nb or tot[i_fu] >= max_fuckups:
            if states is not None:
                np.random.set_state(states[used - 1])       # leave the stream where the reference stops drawing
            break
    tot = {k: int(v) for k, v in zip(E.PEELRUN_NAMES, run.cpu().numpy())}
    assert tot["trials"] == done
    num_fuckups, num_fuckups_truncated = tot["fuckups"], tot["fuckups_exp"]
    total_failed, total_failed_expurgated, total_blocks_failed_exp = tot["lost"], tot["lost_exp"], tot["blocks_exp"]
    total_generated, total_blocks_generated = done * g.generated, done * g.blocks
    o1 = done
    failures, gens = np.zeros(num_repeats), np.zeros(num_repeats)
    return (num_fuckups / o1, num_fuckups_truncated / o1, total_failed / total_generated,
            total_failed_expurgated / total_generated, num_fuckups_truncated, o1, total_failed_expurgated,
            total_generated, failures, gens, total_blocks_failed_exp, total_blocks_generated,
            total_blocks_failed_exp / total_blocks_generated)


def simulate_peeling_decoder_ldpc(e, l_deg, r_deg, L, M, is_terminated, is_protograph, num_repeats=None,
                                  doping_points=[], rng="numpy", seed=0, batch=None, device=None,
                                  want_moments=False, moments_from="auto"):
    """Random-pick peeling with the degree-1-CN trajectory (PD:705-789): returns (None, r1, plrs) with
    r1 int64 [num_repeats, num_pd_steps+1] and plrs float64 [num_repeats].  want_moments=True (philox mode) returns
    (None, moments int64 [3, num_pd_steps+1], plrs) instead of the full trajectories; moments_from: "rows" (the batch's
    trajectories stay on the device and one pass reduces them), "kernel" (three atomics per step inside the pick chain, no
    row buffer) or "auto" (rows where they fit 24 GB).
    rng="philox" under torch.distributed: rank r runs a contiguous share of the trials (global trial indices, so the
    draws do not depend on the sharding); one all-reduce sums the moment vectors (24 B per step) and fills in plrs —
    and the r1 rows when they are asked for — so every rank returns what a single rank would."""
    if not num_repeats:
        num_repeats = 100
    device = _device(device)
    dist, rank, world = _dist()
    if rng == "numpy" and world > 1:
        raise ValueError("rng='numpy' replays the reference's one sequential stream: single rank only")
    if isinstance(doping_points, dict):
        raise NotImplementedError("simulate_peeling_decoder_ldpc takes hard doping points only (PD:747)")
    cpp = int(l_deg / r_deg * M)
    num_positions = L + l_deg - 1 if is_terminated else L
    total_size = cpp * num_positions
    num_pd_steps = int(M * num_positions * (e + 0.1))                      # PD:721
    nd = len(doping_points)
    p = E.CodeParams(l_deg, r_deg, L, cpp, M)
    total_generated = (L - nd) * M                                         # PD:747
    plrs = np.zeros(num_repeats)
    if rng == "numpy":
        r1 = np.zeros((num_repeats, num_pd_steps + 1), dtype="int")
        for o in range(num_repeats):                                       # one shared `random` stream: sequential
            adj, mask = _sample_numpy(e, l_deg, r_deg, L, M, doping_points, is_protograph, False)
            d_adj, d_ch = E.to_device(adj[None], E.pack_bits(mask.astype(np.uint8))[None], device)
            st = random.getstate()
            mt = torch.from_numpy(np.array(st[1], dtype=np.uint32).view(np.int32).copy()[None]).to(device)
            res = E.peel_pick(p, d_adj, d_ch, total_size, num_pd_steps, mt_state=mt)
            new = mt.cpu().numpy().view(np.uint32)[0]
            random.setstate((st[0], tuple(int(x) for x in new), st[2]))
            r1[o] = res["r1"][0].cpu().numpy()
            n_users, picked = (int(x) for x in res["out"][0, :2].cpu())
            plrs[o] = (n_users - picked) / total_generated               # (generated − recovered)/generated, PD:753,785
            if r1[o, -1] == 1:
                print("The number of degree-1 CNs at the end of the iterations is 1!!!")
        return None, r1, plrs
    if rng != "philox":
        raise ValueError("rng must be 'numpy' or 'philox'")
    r1_all = None if want_moments else np.zeros((num_repeats, num_pd_steps + 1), dtype="int")
    moments = None
    if batch is None:
        # one wave steps one trial: the kernel wants ~8192 trials in flight; stay below ~16 GB of device buffers
        # (tables + CN words + picks' scratch ~ 3 x the int32 adjacency; r1 rows 4 B per step where they are written)
        rows_wanted = (not want_moments) or moments_from in ("auto", "rows")
        per_trial = 3 * L * M * l_deg * 4 + (4 * (num_pd_steps + 1) if rows_wanted else 0)
        free = _free_bytes(device)
        batch = max(64, min(8192, int(min(16e9 + (24e9 if rows_wanted and want_moments else 0), 0.6 * free) // per_trial)))
    offs = _split(num_repeats, world)
    if want_moments:
        moments = torch.zeros((3, num_pd_steps + 1), dtype=torch.int64, device=device)
    for done in range(offs[rank], offs[rank + 1], batch):
        nb = min(batch, offs[rank + 1] - done)
        d_adj, d_ch = E.sample_philox(p, seed, done, nb, e, list(doping_points), device=device,
                                      adj16=not is_protograph, ensemble="protograph" if is_protograph else "olmos")
        # moments: three global atomics per step inside the pick chain cost 7 % (tools/ab_c3.py); where the batch's rows fit
        # (4 B per step and trial) they are written instead and reduced by one pass of r1_moments
        if moments_from not in ("auto", "rows", "kernel"):
            raise ValueError("moments_from must be 'auto', 'rows' or 'kernel'")
        # "auto": rows where they fit a quarter of what is free on THIS device right now (shared / smaller parts included)
        rows_first = want_moments and (moments_from == "rows" or
                                       (moments_from == "auto" and
                                        4 * (num_pd_steps + 1) * nb <= min(24e9, 0.25 * _free_bytes(device))))
        res = E.peel_pick(p, d_adj, d_ch, total_size, num_pd_steps, mt_state=None, seed=seed, trial0=done,
                          want_r1=not want_moments or rows_first, moments=None if rows_first else moments)
        if rows_first:
            E.r1_moments(res["r1"], moments)
            res["r1"] = None
        o = res["out"].cpu().numpy()
        plrs[done:done + nb] = (o[:, 0] - o[:, 1]) / total_generated
        if not want_moments:
            r1_all[done:done + nb] = res["r1"].cpu().numpy()
    if world > 1:                                           # shares are disjoint: a sum puts them together
        t_plr = torch.from_numpy(plrs).to(device)
        dist.all_reduce(t_plr)
        plrs = t_plr.cpu().numpy()
        if want_moments:
            dist.all_reduce(moments)
        else:
            t_r1 = torch.from_numpy(r1_all).to(device)
            dist.all_reduce(t_r1)
            r1_all = t_r1.cpu().numpy()
    return None, (moments.cpu().numpy() if want_moments else r1_all), plrs


def simulate_peeling_decoder_ldpc_uncoupled(e, l_deg, r_deg, M, num_repeats=None, device=None):
    """Random-pick peeling on the uncoupled (l,r) ensemble (PD:793-869): returns (None, r1, plrs, num_vns_lst).
    Graphs and channels come from the global numpy stream (ldpc.gen_slots with its repeat rejection, PD:134-135), the
    picks from the global `random` stream; both are left where the reference leaves them."""
    if not num_repeats:
        num_repeats = 100
    device = _device(device)
    cpp = int(l_deg / r_deg * M)
    num_pd_steps = int(M * (e + 0.1))                                      # PD:804
    # one CN position of cpp CNs; the device kernels only need CN ids < total_size = cpp (global-id adjacency)
    p = E.CodeParams(l_deg, r_deg, 1, cpp, M)
    r1 = np.zeros((num_repeats, num_pd_steps + 1), dtype="int")
    plrs = np.zeros(num_repeats)
    num_vns_lst = []
    for o in range(num_repeats):
        adj = gen_slots_uncoupled(l_deg, r_deg, M)
        mask = np.random.rand(M) <= e
        d_adj, d_ch = E.to_device(adj[None], E.pack_bits(mask.astype(np.uint8))[None], device)
        st = random.getstate()
        mt = torch.from_numpy(np.array(st[1], dtype=np.uint32).view(np.int32).copy()[None]).to(device)
        res = E.peel_pick(p, d_adj, d_ch, cpp, num_pd_steps, mt_state=mt)
        new = mt.cpu().numpy().view(np.uint32)[0]
        random.setstate((st[0], tuple(int(x) for x in new), st[2]))
        r1[o] = res["r1"][0].cpu().numpy()
        n_users, picked = (int(x) for x in res["out"][0, :2].cpu())
        num_vns_lst.append(n_users)
        plrs[o] = (n_users - picked) / M                                    # PD:823-826, 866
        if r1[o, -1] == 1:
            print("The number of degree-1 CNs at the end of the iterations is 1!!!")
    return None, r1, plrs, num_vns_lst


# ------------------------------------------------------------------------------------------------
# variance reduction (fl_scaling/est_scaling_params.py:42-49, 90-94, 131-138)
# ------------------------------------------------------------------------------------------------
def nu_chunk_from_moments(moments, r1s_theory, M):
    """calc_nu_chunk from the integer moments (cnt, Σr1, Σr1²) of a batch: ssquares[s] = Σ_{r1≠0} (r1/M − θ/M)²,
    counts[s] = cnt[s], over the support of the theory curve.  Equal to the reference's nansum up to float rounding
    (relative 1e-12; the reference sums squared float differences, this sums integers first).  θ is split into its
    nearest integer k and a fraction f: Σ(r1−θ)² = Σ(r1−k)² − 2f·Σ(r1−k) + cnt·f² with the first two sums exact in
    integers, so nothing cancels when the trajectories sit on the theory curve (s2 − 2θ·s1 + cnt·θ² loses five digits
    there)."""
    last = int(np.max(np.where(r1s_theory > 0))) + 1
    th = r1s_theory[r1s_theory > 0].astype(np.float64)
    cnt, s1, s2 = (np.asarray(moments[k])[:last][:r1s_theory.shape[0]].astype(np.int64) for k in range(3))
    k = np.rint(th).astype(np.int64)
    f = th - k
    a = s2 - 2 * k * s1 + cnt * k * k                                       # Σ (r1 − k)²   exact
    b = s1 - cnt * k                                                        # Σ (r1 − k)    exact
    return (a.astype(np.float64) - 2.0 * f * b + cnt * f * f) / (float(M) * float(M)), cnt


class _ArraysOnly(pickle.Unpickler):
    """The theory file is a pickle of numpy arrays (NB cell 20); nothing else is allowed to load."""
    _OK = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
           ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
           ("numpy._core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) not in self._OK:
            raise pickle.UnpicklingError(f"refusing to load {module}.{name}: theory files hold numpy arrays only")
        return super().find_class(module, name)


def load_theory(path):
    if path.endswith(".npy"):
        return np.load(path)
    with open(path, "rb") as f:
        return np.asarray(_ArraysOnly(f).load()[0])


# ------------------------------------------------------------------------------------------------
# command-line entry points
# ------------------------------------------------------------------------------------------------
_SAFE_CALLS = {"arange": np.arange, "linspace": np.linspace, "range": range, "list": list}


def _safe_eval(text):
    """The reference eval()s its `es` and `doping_points` arguments (PD:1333,1340).  Same inputs accepted —
    numbers, lists/tuples/dicts, arithmetic, np.arange/np.linspace/range — without executing arbitrary code."""
    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return node.value
        if isinstance(node, (ast.List, ast.Tuple)):
            return [ev(x) for x in node.elts]
        if isinstance(node, ast.Dict):
            return {ev(k): ev(v) for k, v in zip(node.keys, node.values)}
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = ev(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult, ast.Div)):
            a, b = ev(node.left), ev(node.right)
            return {ast.Add: a + b, ast.Sub: a - b, ast.Mult: a * b, ast.Div: a / b}[type(node.op)] \
                if not isinstance(node.op, ast.Div) or b != 0 else a / b
        if isinstance(node, ast.Call) and not node.keywords:
            f = node.func
            name = f.attr if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in ("np", "numpy") \
                else (f.id if isinstance(f, ast.Name) else None)
            if name in _SAFE_CALLS:
                return _SAFE_CALLS[name](*[ev(x) for x in node.args])
        raise ValueError("unsupported expression in argument: " + ast.dump(node)[:80])
    return ev(ast.parse(text, mode="eval"))


def _cli_options(args, kw):
    """The reference's command lines are positional only.  This mirror also takes, anywhere among them,
    `--rng philox|numpy`, `--seed S`, `--batch B`, `--device D` — the keyword arguments of the simulators (throughput mode:
    device sampling, trials sharded over the ranks of a torch.distributed.run job)."""
    kw, pos, it = dict(kw), [], iter(args)
    for x in it:
        if isinstance(x, str) and x in ("--rng", "--seed", "--batch", "--device"):
            v = next(it)
            kw[x[2:]] = v if x in ("--rng", "--device") else int(v)
        else:
            pos.append(x)
    return pos, kw


def _join_job(kw):
    """Under torch.distributed.run only the throughput mode shards: the reference's own streams (rng="numpy", the default)
    are one sequential sequence, and ranks that each believed to be rank 0 would all write the same output file."""
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and kw.get("rng") != "philox":
        raise SystemExit("rng='numpy' replays the reference's one sequential numpy / random stream: run it as a single "
                         "process, or pass --rng philox to shard the trials over the %s ranks of this job"
                         % os.environ["WORLD_SIZE"])
    return E.init_distributed() if kw.get("rng") == "philox" else False


def main_simulate_sc_ldpc(argv=None, **kw):
    """ber_sim.py: OUT l r L M "es" T|N U|P B|N TB|NTB num_repeats max_fuckups "doping" (PD:1327-1356)."""
    a, kw = _cli_options(sys.argv if argv is None else [None] + list(argv), kw)
    joined = _join_job(kw)
    try:
        return _main_simulate_sc_ldpc(a, kw)
    finally:
        if joined:
            import torch.distributed as dist
            dist.destroy_process_group()


class _Tee:
    """Rank 0 writes the table; the other ranks of a job compute the same rows (the counters are all-gathered) and stay silent."""

    def __init__(self, path, active):
        self.f = open(path, "wt") if active else None

    def line(self, *vals):
        if self.f is not None:
            print(*vals, file=self.f)
            print(*vals)
            self.f.flush()
            sys.stdout.flush()

    def close(self):
        if self.f is not None:
            self.f.close()


def _main_simulate_sc_ldpc(a, kw):
    fname, l, r, L, M = a[1], int(a[2]), int(a[3]), int(a[4]), int(a[5])
    es = _safe_eval(a[6])
    is_terminated, is_protograph = a[7] == "T", a[8] == "P"
    is_bounded, is_tail_biting = a[9] == "B", a[10] == "TB"
    num_repeats, max_fuckups = int(a[11]), int(a[12])
    doping_points = _safe_eval(a[13])
    L += len(doping_points)                                                 # PD:1343
    head = (f"# SC-LDPC ({l},{r},L={L},M={M}) terminated:{is_terminated}, proto:{is_protograph}, bounded:{is_bounded}, "
            f"tail biting:{is_tail_biting}. num_repeats={num_repeats}, max_fuckups={max_fuckups}, "
            f"doping_points={doping_points}.")
    out = _Tee(fname, _dist()[1] == 0)
    try:
        out.line(head)
        for e in (es if isinstance(es, (list, np.ndarray, range)) else [es]):
            ber, ber_truncated, plr, plr_exp, fbl, tbl, fbit, tgen, _flrs, _gens, fblocks, tblocks, bler = \
                simulate_sc_ldpc(e, l, r, L, M, is_terminated, is_protograph, is_bounded, is_tail_biting, num_repeats,
                                 max_fuckups, doping_points, **kw)
            out.line(e, ber, ber_truncated, plr, plr_exp, fbl, tbl, fbit, tgen, fblocks, tblocks, bler)
    finally:
        out.close()


def main_simulate_variance(argv=None, **kw):
    """simulate_variance.py: OUT l r L M e T|N U|P num_runs num_runs_batch THEORY (PD:1264-1294): writes
    pickle.dump((ssquares, counts))."""
    a, kw = _cli_options(sys.argv if argv is None else [None] + list(argv), kw)
    joined = _join_job(kw)
    try:
        return _main_simulate_variance(a, kw)
    finally:
        if joined:
            import torch.distributed as dist
            dist.destroy_process_group()


def _main_simulate_variance(a, kw):
    fname, l, r, L, M, e = a[1], int(a[2]), int(a[3]), int(a[4]), int(a[5]), float(a[6])
    is_terminated, is_protograph = a[7] == "T", a[8] == "P"
    num_runs, num_runs_batch, ftheory = int(a[9]), int(a[10]), a[11]
    r1s_theory = load_theory(ftheory)
    ssquares, counts = None, None
    for i in range(int(num_runs / num_runs_batch)):
        if kw.get("rng", "numpy") == "philox":
            _, mom, _ = simulate_peeling_decoder_ldpc(e, l, r, L, M, is_terminated, is_protograph, num_runs_batch,
                                                      want_moments=True, **dict(kw, seed=kw.get("seed", 0) + i))
        else:
            _, r1s, _ = simulate_peeling_decoder_ldpc(e, l, r, L, M, is_terminated, is_protograph, num_runs_batch, **kw)
            mom = E.r1_moments(torch.from_numpy(r1s.astype(np.int32)).to(_device(kw.get("device")))).cpu().numpy()
        ss, cn = nu_chunk_from_moments(mom, r1s_theory, M)
        ssquares = ss if ssquares is None else ssquares + ss
        counts = cn if counts is None else counts + cn
    if _dist()[1] == 0:                                                     # every rank holds the same sums: rank 0 writes
        with open(fname, "wb") as f:
            pickle.dump((ssquares, counts), f)
    return ssquares, counts


def test_sc_ldpc(l=4, r=8, L=50, M=10000, e=0.48, is_terminated=False, num_runs=100, num_runs_batch=100,
                 out_pattern="r1_sc_ldpc_{l}_{r}_{L}_{M}_{etag}_{term}_{i}.pkl", **kw):
    """`python3 peeling_decoding.py` (PD:1212-1245): batches of trajectories pickled as (r1, plrs)."""
    for i in range(int(num_runs / num_runs_batch)):
        _, r1, plrs = simulate_peeling_decoder_ldpc(e, l, r, L, M, is_terminated, False, num_runs_batch, [], **kw)
        name = out_pattern.format(l=l, r=r, L=L, M=M, etag=("%.3f" % e).replace(".", "")[:4],
                                  term="terminated" if is_terminated else "nonterminated", i=i)
        with open(name, "wb") as f:
            pickle.dump((r1, plrs), f)


if __name__ == "__main__":
    prog = sys.argv[1] if len(sys.argv) > 1 else "test_sc_ldpc"
    if prog == "ber_sim":
        main_simulate_sc_ldpc(sys.argv[2:])
    elif prog == "simulate_variance":
        main_simulate_variance(sys.argv[2:])
    else:
        test_sc_ldpc()
