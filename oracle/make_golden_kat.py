#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — the known-answer material the reference itself holds for the hot path, run
HERE from the reference where it lies and kept as data in tests/golden/kat_reference.json:

  * test_2_6_csa_sync (PD:1164-1175): a hand-built 3-user / 10-slot schedule swept by sic_round (PD:270-313); the
    reference prints the users decoded at every t and the schedule that is left.  Captured by IMPORTING the reference's
    peeling_decoding.py and calling its own functions on its own data.
  * test_is_position_doped_streaming and test_circular_buffer_wrapping (BPF:1891-1924): the two table printers of the
    streaming mode, run from the reference compiled by oracle/Makefile (oracle/_ref/ref_stream_M5_L20 kat).

    python oracle/make_golden_kat.py            (needs /root/reference and `make -C oracle ref`)
"""
import io
import json
import os
import re
import subprocess
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SCLDPC_REFERENCE", "/root/reference")


def csa_kat():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, os.path.join(REF, "simulators_sc_ldpc", "peeling_decoding"))
    sys.path.insert(0, REF)
    with redirect_stdout(io.StringIO()):
        import peeling_decoding as PD
    users = [PD.User(1, 0, [1, 2, 3, 4, 7, 10], set(), 2), PD.User(2, 0, [1, 2, 3, 5, 6, 9], set(), 2),
             PD.User(3, 0, [2, 3, 4, 5, 6, 8], set(), 2)]                # the data of test_2_6_csa_sync, PD:1165-1167
    schedule = PD.add_to_schedule(PD.empty_schedule(), users)
    decoded = []
    for t in range(11):                                                   # PD:1172-1174
        with redirect_stdout(io.StringIO()):
            d = PD.sic_round(schedule, t)
        decoded.append(sorted(u.uid for u in d))
    return {"users": [{"uid": u.uid, "transmissions": list(u.transmissions), "k": u.k} for u in users],
            "decoded_at_t": decoded, "schedule_left": {str(k): sorted(u.uid for u in v) for k, v in schedule.items()},
            "recovered": {str(u.uid): sorted(u.recovered) for u in users}}


def stream_kat():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_stream_M5_L20")
    out = subprocess.run([exe, "kat"], check=True, capture_output=True, text=True).stdout
    doped_txt, wrap_txt = out.split("----\n")
    doped = [[int(x) for x in ln.split()] for ln in doped_txt.strip().split("\n")]
    rows = []
    for ln in wrap_txt.strip().split("\n"):
        m = re.match(r"pos: (\d+) (VN|CN):\t(-?\d+)\t(-?\d+)\t0\t(-?\d+)\t\((\d)\)", ln)
        if m:
            rows.append([int(m.group(1)), m.group(2)] + [int(m.group(i)) for i in (3, 4, 5, 6)])
    return {"is_position_doped_streaming": {"doped_positions": [5, 7, 9], "pos_is_doped": doped},
            "circular_buffer_wrapping": {"L": 10, "ms": 2, "W": 5, "columns": ["pos", "kind", "start", "end", "end_wrap", "is_wrap"],
                                         "rows": rows}}


if __name__ == "__main__":
    kat = {"source": "reference functions run where they lie: PD:1164-1175 (imported), BPF:1891-1924 (oracle/_ref/ref_stream_M5_L20 kat)",
           "csa_2_6_sync": csa_kat(), **stream_kat()}
    path = os.path.join(ROOT, "tests", "golden", "kat_reference.json")
    json.dump(kat, open(path, "w"), indent=1)
    print("wrote", path, len(kat["circular_buffer_wrapping"]["rows"]), "range rows,",
          len(kat["is_position_doped_streaming"]["pos_is_doped"]), "doping rows; decoded per t:", kat["csa_2_6_sync"]["decoded_at_t"])
