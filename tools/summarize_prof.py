#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs that tools/profile.sh leaves in gpurun_out/prof_<tag>/ into
profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table, our kernels only),
profiles/<tag>_pmc.json (FETCH_SIZE / WRITE_SIZE per kernel, own passes) and profiles/traffic.json
(the number bench.py reports as roofline.traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE)·1024) and on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B,
so the read side is doubled for wide coalesced streams (our adjacency reads are 16 B/lane).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("full_bp_fixpoint_kernel", "full_bp_kernel", "sample_philox_kernel", "sw_bp_kernel", "accumulate_run_kernel", "peel")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384      # bench.py's default batch
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun merges runs: keep the latest
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(stats))]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if short(r["Name"]):
                w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs",
                                           "MaxNs", "StdDev")])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for kind in ("pmc_fetch", "pmc_write"):
        for path in [newest(os.path.join(src, kind, "*", "*_counter_collection.csv"))]:
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                if k:
                    pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta[k] = dict(grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]), lds=int(r["LDS_Block_Size"]),
                                   vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]))
    out = {"tag": tag, "trials_per_launch": batch, "units": "counter values are KiB per dispatch, mean over dispatches",
           "kernels": {}}
    for k, c in pmc.items():
        fetch = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"])) if "FETCH_SIZE" in c else None
        write = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"])) if "WRITE_SIZE" in c else None
        e = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "dispatches": len(c.get("FETCH_SIZE", [])), **meta[k]}
        if fetch is not None and write is not None:
            e["hbm_bytes_per_launch_raw"] = (fetch + write) * 1024
            e["hbm_bytes_per_launch_corrected"] = (2 * fetch + write) * 1024      # gfx950: FETCH_SIZE x 2
            e["hbm_bytes_per_trial_corrected"] = e["hbm_bytes_per_launch_corrected"] / batch
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    dom = "full_bp_fixpoint_kernel" if "full_bp_fixpoint_kernel" in out["kernels"] else "full_bp_kernel"   # the bench's decoder
    if dom in out["kernels"] and "hbm_bytes_per_launch_corrected" in out["kernels"][dom]:
        json.dump({"workload": "(4,8) SC-LDPC L=50 N=1000 eps=0.48 full BP unlimited iterations", "batch": batch,
                   "kernel": dom,
                   "full_bp_hbm_bytes_per_launch": out["kernels"][dom]["hbm_bytes_per_launch_corrected"],
                   "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2)"},
                  open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
