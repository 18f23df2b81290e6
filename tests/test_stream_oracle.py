"""The streaming oracle (oracle/scldpc_stream_oracle.c: literal messages and the node-level model, glibc draws)
against golden vectors from the REAL reference's streaming mode (tests/golden/stream_*.npz), and — where
oracle/_ref exists — against the reference itself on fresh seeds."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "stream_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_streaming_oracle_matches_reference_golden(oracle, path):
    z = np.load(path)
    m = json.loads(str(z["meta"]))
    p = oracle.Params(4, 8, m["L"], m["Def_M"], 2 * m["Def_M"])
    for dec in (0, 1):
        if dec == 0 and m["Def_M"] >= 2500:
            P = 8                       # N = 5000: the literal decoder takes a second per position
        elif dec == 0 and m["Def_M"] >= 500 and m["P"] > 60:
            P = 60                      # the literal decoder is slow at this size; the node model runs it all
        else:
            P = m["P"]
        s = oracle.Stream(p, m["seed"], m["eps"], m["W"], m["doped"], rng_mode=0, decoder=dec)
        for k in range(P):
            o = s.step()
            assert [o[f] for f in oracle.Stream.FIELDS] == z["rows"][k].tolist(), (path, dec, k)
            if "erased" in z.files and o["pos"] >= 3:
                assert (s.last_erased() == z["erased"][k]).all(), (path, dec, k)


def test_streaming_fresh_seeds_against_reference(oracle):
    if not os.path.exists(os.path.join(oracle.REF_DIR, "ref_stream_M5_L20")):
        pytest.skip("oracle/_ref not built (no /root/reference here): golden fixtures are the pin")
    rng = np.random.RandomState(11)
    p = oracle.Params(4, 8, 20, 5, 10)
    for _ in range(12):
        seed, eps, W = int(rng.randint(1, 2**30)), float(rng.choice([0.35, 0.44, 0.48, 0.52])), int(rng.randint(2, 8))
        doped = tuple(sorted(rng.choice(np.arange(3, 12), size=rng.randint(0, 3), replace=False).tolist()))
        ref = oracle.run_ref_stream(5, 20, 90, seed, eps, W, doped)
        for dec in (0, 1):
            s = oracle.Stream(p, seed, eps, W, doped, rng_mode=0, decoder=dec)
            for k in range(90):
                o = s.step()
                assert all(o[f] == ref[k][f] for f in oracle.Stream.FIELDS), (seed, eps, W, doped, dec, k)


def test_stream_helpers_reproduce_the_reference_table_printers(oracle):
    """test_is_position_doped_streaming and test_circular_buffer_wrapping (BPF:1891-1924): the reference's own (print-only)
    known-answer tables for the streaming mode, captured from the compiled reference by oracle/make_golden_kat.py.  The
    oracle's decoder takes its doping test and its window ranges from exactly these two functions."""
    import ctypes as C
    import json
    import os
    from conftest import GOLDEN_DIR
    kat = json.load(open(os.path.join(GOLDEN_DIR, "kat_reference.json")))
    L = oracle.lib()
    d = kat["is_position_doped_streaming"]
    doped = (C.c_int32 * len(d["doped_positions"]))(*d["doped_positions"])
    assert len(d["pos_is_doped"]) == 30
    for pos, want in d["pos_is_doped"]:
        assert L.orc_stream_is_doped(pos, len(d["doped_positions"]), doped) == want, pos
    assert L.orc_stream_is_doped(17, 0, doped) == 0
    w = kat["circular_buffer_wrapping"]
    assert len(w["rows"]) == 80
    out = (C.c_int32 * 8)()
    for pos, kind, start, end, end_wrap, is_wrap in w["rows"]:
        L.orc_stream_sw_range(pos, w["L"], w["W"], w["ms"], out)
        got = list(out[0:4]) if kind == "VN" else list(out[4:8])
        assert got == [start, end, end_wrap, is_wrap], (pos, kind, got)
