#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerate tests/golden/pd_*.npz by IMPORTING the real reference
simulators_sc_ldpc/peeling_decoding/peeling_decoding.py (only possible in the container that holds
/root/reference; the fixtures are committed, the reference never travels).

For every case: np.random.seed(s); random.seed(s) — then
  * the inputs the decoder consumed (`transmissions` from sc_ldpc.gen_slots, the erasure mask from
    np.random.rand(L*M) <= e, PD:147-163), captured by replaying the same draws;
  * `simulate_sc_ldpc(...)` → its 13-tuple (PD:591-701), one trial per call;
  * `simulate_peeling_decoder_ldpc(...)` → r1 trajectory and plr (PD:705-789).
numpy 2.2.6 / CPython 3.10.12 produced the committed files (recorded in `meta`).

    MPLBACKEND=Agg python oracle/make_golden_pd.py
"""
import json
import os
import platform
import random
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference/simulators_sc_ldpc/peeling_decoding"
sys.path.insert(0, REF)
import peeling_decoding as pd      # noqa: E402  (the real reference)
import sc_ldpc                     # noqa: E402


class _Range:
    """tqdm.trange stand-in: simulate_sc_ldpc calls set_description on it (PD:697)."""
    def __init__(self, n): self.n = n
    def __iter__(self): return iter(range(self.n))
    def set_description(self, *_a, **_k): pass


pd.trange = lambda n: _Range(n)

# (name, l, r, L, M, e, is_terminated, is_bounded, doping, seeds)
ER_CASES = [
    ("tiny_TB", 4, 8, 10, 20, 0.45, True, True, [], range(0, 40)),
    ("tiny_NTB", 4, 8, 10, 20, 0.45, False, True, [], range(40, 70)),
    ("tiny_TNB", 4, 8, 12, 20, 0.47, True, False, [], range(70, 90)),
    ("tiny_NTNB", 4, 8, 12, 20, 0.47, False, False, [], range(90, 110)),
    ("tiny_hard", 4, 8, 14, 20, 0.5, True, True, [6, 7], range(110, 130)),
    ("tiny_soft", 4, 8, 14, 20, 0.5, True, True, {6: 0.5, 9: 0.25}, range(130, 150)),
    ("mid_TB", 4, 8, 20, 200, 0.47, True, True, [], range(200, 216)),
    ("mid_NTNB", 4, 8, 20, 200, 0.465, False, False, [], range(216, 224)),
    ("mid_hard", 4, 8, 22, 200, 0.49, False, True, [10, 11], range(224, 232)),
    ("c1_TB", 4, 8, 50, 1000, 0.48, True, True, [], range(300, 308)),
    ("c1_TB_e46", 4, 8, 50, 1000, 0.46, True, True, [], range(308, 312)),
]
TR_CASES = [
    ("tiny_NT", 4, 8, 10, 20, 0.45, False, [], range(0, 24)),
    ("tiny_T", 4, 8, 10, 20, 0.48, True, [], range(24, 48)),
    ("tiny_hard", 4, 8, 14, 20, 0.5, False, [6, 7], range(48, 60)),
    ("mid_NT", 4, 8, 20, 200, 0.47, False, [], range(100, 108)),
    ("mid_T", 4, 8, 20, 200, 0.45, True, [], range(108, 114)),
    ("c1_NT", 4, 8, 50, 1000, 0.48, False, [], range(200, 203)),
]


# tail-biting (TB) and protograph (P) ensembles, SURVEY §8(f)3: (name, l, r, L, M, e, term, bnd, doping, seeds, proto, tb)
ER2_CASES = [
    ("tiny_tb_NT", 4, 8, 10, 20, 0.45, False, True, [], range(400, 430), False, True),
    ("tiny_tb_T", 4, 8, 10, 20, 0.47, True, True, [], range(430, 450), False, True),
    ("tiny_tb_NTNB", 4, 8, 12, 20, 0.47, False, False, [], range(450, 466), False, True),
    ("mid_tb_NT", 4, 8, 20, 200, 0.48, False, True, [], range(466, 474), False, True),
    ("tiny_proto_T", 4, 8, 10, 20, 0.45, True, True, [], range(500, 530), True, False),
    ("tiny_proto_NTNB", 4, 8, 12, 20, 0.47, False, False, [], range(530, 546), True, False),
    ("tiny_proto_hard", 4, 8, 14, 20, 0.5, True, True, [6, 7], range(546, 566), True, False),
    ("mid_proto_T", 4, 8, 20, 200, 0.47, True, True, [], range(566, 574), True, False),
    ("c1_proto_T", 4, 8, 50, 1000, 0.48, True, True, [], range(574, 578), True, False),
    ("c1_tb_NT", 4, 8, 50, 1000, 0.48, False, True, [], range(578, 582), False, True),
]
TR2_CASES = [   # (name, l, r, L, M, e, term, doping, seeds) — protograph
    ("tiny_proto_NT", 4, 8, 10, 20, 0.45, False, [], range(600, 620)),
    ("tiny_proto_T", 4, 8, 10, 20, 0.48, True, [], range(620, 636)),
    ("tiny_proto_hard", 4, 8, 14, 20, 0.5, False, [6, 7], range(636, 648)),
    ("mid_proto_NT", 4, 8, 20, 200, 0.47, False, [], range(648, 654)),
]
UNC_CASES = [   # (name, l, r, M, e, seeds) — uncoupled ensemble with repeat rejection (ldpc.py:80-84)
    ("unc_3_6_M16", 3, 6, 16, 0.4, range(700, 712)),
    ("unc_3_6_M24", 3, 6, 24, 0.45, range(712, 718)),
]


def capture_inputs2(l, r, L, M, e, seed, proto, tb, doping):
    """The draws of generate_users() for the TB / protograph ensembles, replayed with the reference's own functions."""
    import sc_ldpc_protograph
    np.random.seed(seed); random.seed(seed)
    if not proto:
        tr = sc_ldpc.gen_slots_tail_biting(l, r, L, M) if tb else sc_ldpc.gen_slots(l, r, L, M)
        mask = np.random.rand(L * M) <= e
        return tr.astype(np.int32), mask.astype(np.uint8)
    cpp = int(l / r * M)
    trs, masks = [], []
    for pos in range(L):                                   # PD:200-207
        trs.append(pos * cpp + sc_ldpc_protograph.gen_slots_from_position(l, r, M))
        m = np.random.rand(M) <= e
        if pos in doping:
            m[:] = False                                   # PD:237: users of a doped position are never created
        masks.append(m)
    return np.vstack(trs).astype(np.int32), np.concatenate(masks).astype(np.uint8)


def capture_inputs(l, r, L, M, e, seed):
    np.random.seed(seed); random.seed(seed)
    tr = sc_ldpc.gen_slots(l, r, L, M)                     # PD:153
    mask = np.random.rand(L * M) <= e                      # PD:154
    return tr.astype(np.int32), mask.astype(np.uint8)


def meta(**kw):
    kw.update(numpy=np.__version__, python=platform.python_version(), generator="oracle/make_golden_pd.py",
              source="real reference imported from " + REF)
    return np.array(json.dumps(kw))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    for name, l, r, L, M, e, term, bnd, doping, seeds in ER_CASES:
        rows, trs, masks = [], [], []
        Lgen = L + (0 if bnd else 20) + (0 if term else 20)   # PD:604-607: the function widens L itself
        for s in seeds:
            tr, mask = capture_inputs(l, r, Lgen, M, e, s)
            np.random.seed(s); random.seed(s)
            t = pd.simulate_sc_ldpc(e, l, r, L, M, term, False, bnd, False, num_repeats=1, max_fuckups=2000,
                                    doping_points=doping)
            rows.append([t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[10], t[11], t[12]])
            if M <= 200:
                trs.append(tr); masks.append(mask)
        # one multi-trial call: the numpy stream runs on from trial to trial
        np.random.seed(seeds[0]); random.seed(seeds[0])
        t3 = pd.simulate_sc_ldpc(e, l, r, L, M, term, False, bnd, False, num_repeats=3, max_fuckups=2000,
                                 doping_points=doping)
        out = dict(seed=np.array(list(seeds)), tuple11=np.array(rows, dtype=np.float64),
                   multi3=np.array([t3[0], t3[1], t3[2], t3[3], t3[4], t3[5], t3[6], t3[7], t3[10], t3[11], t3[12]]),
                   meta=meta(kind="simulate_sc_ldpc", name=name, l=l, r=r, L=L, M=M, e=e, is_terminated=term,
                             is_bounded=bnd, doping=(doping if isinstance(doping, list) else {str(k): v for k, v in doping.items()}),
                             doping_soft=isinstance(doping, dict)))
        if trs:
            out["transmissions"] = np.stack(trs); out["mask"] = np.stack(masks)
        np.savez_compressed(os.path.join(GOLDEN, f"pd_er_{name}.npz"), **out)
        print("pd_er_" + name, len(rows), flush=True)
    for name, l, r, L, M, e, term, doping, seeds in TR_CASES:
        r1s, plrs, trs, masks = [], [], [], []
        for s in seeds:
            tr, mask = capture_inputs(l, r, L, M, e, s)
            np.random.seed(s); random.seed(s)
            _, r1, plr = pd.simulate_peeling_decoder_ldpc(e, l, r, L, M, term, False, 1, doping)
            r1s.append(r1[0].astype(np.int32)); plrs.append(plr[0])
            if M <= 200:
                trs.append(tr); masks.append(mask)
        np.random.seed(seeds[0]); random.seed(seeds[0])
        _, r1m, plrm = pd.simulate_peeling_decoder_ldpc(e, l, r, L, M, term, False, 2, doping)
        out = dict(seed=np.array(list(seeds)), r1=np.stack(r1s), plr=np.array(plrs),
                   multi2_r1=r1m.astype(np.int32), multi2_plr=plrm,
                   meta=meta(kind="simulate_peeling_decoder_ldpc", name=name, l=l, r=r, L=L, M=M, e=e,
                             is_terminated=term, doping=doping))
        if trs:
            out["transmissions"] = np.stack(trs); out["mask"] = np.stack(masks)
        np.savez_compressed(os.path.join(GOLDEN, f"pd_tr_{name}.npz"), **out)
        print("pd_tr_" + name, len(r1s), flush=True)


def main2():
    import ldpc
    for name, l, r, L, M, e, term, bnd, doping, seeds, proto, tb in ER2_CASES:
        rows, trs, masks = [], [], []
        Lgen = L + (0 if bnd else 20) + (0 if term else 20)
        for s in seeds:
            tr, mask = capture_inputs2(l, r, Lgen, M, e, s, proto, tb, doping)
            np.random.seed(s); random.seed(s)
            t = pd.simulate_sc_ldpc(e, l, r, L, M, term, proto, bnd, tb, num_repeats=1, max_fuckups=2000,
                                    doping_points=doping)
            rows.append([t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[10], t[11], t[12]])
            if M <= 200:
                trs.append(tr); masks.append(mask)
        np.random.seed(seeds[0]); random.seed(seeds[0])
        t3 = pd.simulate_sc_ldpc(e, l, r, L, M, term, proto, bnd, tb, num_repeats=3, max_fuckups=2000,
                                 doping_points=doping)
        out = dict(seed=np.array(list(seeds)), tuple11=np.array(rows, dtype=np.float64),
                   multi3=np.array([t3[0], t3[1], t3[2], t3[3], t3[4], t3[5], t3[6], t3[7], t3[10], t3[11], t3[12]]),
                   meta=meta(kind="simulate_sc_ldpc", name=name, l=l, r=r, L=L, M=M, e=e, is_terminated=term,
                             is_bounded=bnd, doping=doping, doping_soft=False, is_protograph=proto, is_tail_biting=tb))
        if trs:
            out["transmissions"] = np.stack(trs); out["mask"] = np.stack(masks)
        np.savez_compressed(os.path.join(GOLDEN, f"pd_er_{name}.npz"), **out)
        print("pd_er_" + name, len(rows), flush=True)
    for name, l, r, L, M, e, term, doping, seeds in TR2_CASES:
        r1s, plrs, trs, masks = [], [], [], []
        for s in seeds:
            tr, mask = capture_inputs2(l, r, L, M, e, s, True, False, doping)
            np.random.seed(s); random.seed(s)
            _, r1, plr = pd.simulate_peeling_decoder_ldpc(e, l, r, L, M, term, True, 1, doping)
            r1s.append(r1[0].astype(np.int32)); plrs.append(plr[0])
            trs.append(tr); masks.append(mask)
        np.random.seed(seeds[0]); random.seed(seeds[0])
        _, r1m, plrm = pd.simulate_peeling_decoder_ldpc(e, l, r, L, M, term, True, 2, doping)
        np.savez_compressed(os.path.join(GOLDEN, f"pd_tr_{name}.npz"), seed=np.array(list(seeds)), r1=np.stack(r1s),
                            plr=np.array(plrs), multi2_r1=r1m.astype(np.int32), multi2_plr=plrm,
                            transmissions=np.stack(trs), mask=np.stack(masks),
                            meta=meta(kind="simulate_peeling_decoder_ldpc", name=name, l=l, r=r, L=L, M=M, e=e,
                                      is_terminated=term, doping=doping, is_protograph=True))
        print("pd_tr_" + name, len(r1s), flush=True)
    for name, l, r, M, e, seeds in UNC_CASES:
        r1s, plrs, nvs, trs, masks = [], [], [], [], []
        for s in seeds:
            np.random.seed(s); random.seed(s)
            tr = ldpc.gen_slots(l, r, M)                       # PD:134
            mask = np.random.rand(M) <= e                      # PD:135
            np.random.seed(s); random.seed(s)
            _, r1, plr, nv = pd.simulate_peeling_decoder_ldpc_uncoupled(e, l, r, M, 1)
            r1s.append(r1[0].astype(np.int32)); plrs.append(plr[0]); nvs.append(nv[0])
            trs.append(tr.astype(np.int32)); masks.append(mask.astype(np.uint8))
        np.random.seed(seeds[0]); random.seed(seeds[0])
        _, r1m, plrm, nvm = pd.simulate_peeling_decoder_ldpc_uncoupled(e, l, r, M, 2)
        np.savez_compressed(os.path.join(GOLDEN, f"pd_{name}.npz"), seed=np.array(list(seeds)), r1=np.stack(r1s),
                            plr=np.array(plrs), num_vns=np.array(nvs), multi2_r1=r1m.astype(np.int32), multi2_plr=plrm,
                            multi2_num_vns=np.array(nvm), transmissions=np.stack(trs), mask=np.stack(masks),
                            meta=meta(kind="simulate_peeling_decoder_ldpc_uncoupled", name=name, l=l, r=r, M=M, e=e))
        print("pd_" + name, len(r1s), flush=True)


# BASELINE config 3 at full size — (4,8), L = 50, N = 10000, 290 000 peeling steps (PD:721), the notebook's trajectory
# ensemble (PD:1213-1239) — one trial per file (minutes each in the reference): (name, e, is_terminated, seed)
C3_CASES = [("c3_NT", 0.48, False, 700), ("c3_T", 0.46, True, 701)]


def main3():
    import time
    for name, e, term, s in C3_CASES:
        t0 = time.time()
        np.random.seed(s); random.seed(s)
        _, r1, plr = pd.simulate_peeling_decoder_ldpc(e, 4, 8, 50, 10000, term, False, 1, [])
        after = (float(np.random.rand()), random.random())               # where both streams stand afterwards
        np.savez_compressed(os.path.join(GOLDEN, f"pd_trbig_{name}.npz"), seed=np.array([s]), r1=r1.astype(np.int32),
                            plr=plr, after=np.array(after),
                            meta=meta(kind="simulate_peeling_decoder_ldpc", name=name, l=4, r=8, L=50, M=10000, e=e,
                                      is_terminated=term, doping=[]))
        print("pd_trbig_" + name, r1.shape, "%.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    if "--c3" in sys.argv:
        main3()
        sys.exit(0)
    if "--new-only" not in sys.argv:
        main()
    main2()
