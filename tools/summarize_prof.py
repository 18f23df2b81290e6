#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs that tools/profile.sh leaves in gpurun_out/prof_<tag>/ into
profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table, our kernels only),
profiles/<tag>_pmc.json (FETCH_SIZE / WRITE_SIZE / L2 requests per kernel, separate passes) and profiles/traffic.json
(what bench.py reports as roofline.traffic / traffic_raw).

HBM bytes (MI355X_MICROARCH.md §HBM + our own calibration, tools/calib, profiles/<tag>_calibration.txt):
  * counters are in KiB: raw bytes = (FETCH_SIZE + WRITE_SIZE) * 1024;
  * FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but a request moves 128 B — measured here for a 16-B-per-lane stream (4 GiB read:
    33.5 M requests) AND for random 8-byte / 16-byte row gathers (one request per gather, the same ~48 G requests/s
    ceiling as the stream) — so the corrected read bytes are RDREQ x 128 = 2 x FETCH_SIZE for every pattern our kernels use;
  * WRITE_SIZE is exact for our 8- and 16-byte-per-lane coalesced stores (calibrated: 4 GiB stored reads 4 194 304 KiB).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = {"C2": "(4,8) SC-LDPC L=50 N=1000 eps=0.48 full BP unlimited iterations",
             "C3": "(4,8) SC-LDPC L=50 N=10000 eps=0.48 random-pick peeling, 290000 steps per trial",
             "C4": "(4,8) SC-LDPC L=100 N=2000 eps=0.47 decodeBP_SW W=10 I_max=20",
             "C5": "doped (4,8) streaming ensemble N=5000 L=50 W=20 (batch = streams per launch, 16 positions each)"}
KERNELS = ("sw_ring_kernel", "cn_sockets_kernel", "r1_moments_kernel", "full_bp_small_kernel", "sample_philox_v3_kernel", "sample_philox_v2_kernel", "full_bp_fixpoint_kernel", "full_bp_kernel", "sample_philox_kernel",
           "sample_philox_big_kernel", "sw_bp_kernel", "accumulate_run_kernel", "peel_pick_multi_kernel", "cn_build_kernel", "peel_pick_kernel", "peel_sweep_kernel",
           "stream_bp_kernel", "stream_gen_kernel", "stream_dec_kernel", "accumulate_peel_kernel")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536      # bench.py default batch (C2)
    config = sys.argv[3] if len(sys.argv) > 3 else "C2"           # tools/profile.sh TAG [CONFIG]
    # share of every kernel's dispatches that belong to the warm-up step and are left out of the means (C5: a new stream's first
    # generation launch draws L/2 + ahead positions instead of a step's share; the PMC passes run --steps 2 --warmup 1: 1/3)
    skip = float(sys.argv[4]) if len(sys.argv) > 4 else (1.0 / 3.0 if config == "C5" else 0.0)
    if config != "C2":
        tag = f"{tag}_{config}"
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
    for kind in ("pmc_fetch", "pmc_write", "pmc_rdreq"):
        files = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))
        if not files:
            continue
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = short(r["Kernel_Name"])
            if k:
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = dict(grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]), lds=int(r["LDS_Block_Size"]),
                               vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]))
    def mean(v):
        v = v[int(round(len(v) * skip)):] if v else v
        return sum(v) / len(v) if v else None
    out = {"tag": tag, "trials_per_launch": batch,
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch, mean over dispatches" + (f" (the first {skip:.2f} of them, the warm-up step's, left out)" if skip else ""),
           "kernels": {}}
    traffic = {}
    for k, c in pmc.items():
        fetch, write = mean(c.get("FETCH_SIZE")), mean(c.get("WRITE_SIZE"))
        rdreq, hit, miss = mean(c.get("TCC_EA0_RDREQ_sum")), mean(c.get("TCC_HIT_sum")), mean(c.get("TCC_MISS_sum"))
        e = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "TCC_EA0_RDREQ": rdreq, "TCC_HIT": hit, "TCC_MISS": miss,
             "dispatches": len(c.get("FETCH_SIZE", [])) - int(round(len(c.get("FETCH_SIZE", [])) * skip)), **meta[k]}
        if fetch is not None and write is not None:
            e["hbm_bytes_per_launch_raw"] = (fetch + write) * 1024
            e["hbm_bytes_per_launch_corrected"] = (2 * fetch + write) * 1024      # 128 B per read request (calibrated)
            e["hbm_bytes_per_trial_corrected"] = e["hbm_bytes_per_launch_corrected"] / batch
            if rdreq:
                e["read_requests_per_trial"] = rdreq / batch
                e["l2_hit_rate"] = hit / (hit + miss) if hit is not None and miss else None
            traffic[k] = {"hbm_bytes": e["hbm_bytes_per_launch_corrected"], "hbm_bytes_raw": e["hbm_bytes_per_launch_raw"],
                          "rdreq": rdreq}
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    tpath = os.path.join(dst, "traffic.json")
    allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
    allt[config] = {"workload": WORKLOADS[config], "batch": batch, "kernels": traffic,
                    "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum, separate "
                              f"passes of `python bench.py --config {config} --steps 2 --warmup 1`; read bytes = 128 B x requests "
                              "= 2 x FETCH_SIZE, calibrated by tools/calib -> profiles/r02_calibration.txt)"}
    json.dump(allt, open(tpath, "w"), indent=1)
    print(json.dumps(out, indent=1))
    # ---- SQ counters: instruction mix and pipe occupancy (which pipe bounds which kernel)
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    span = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(src, "pmc_sq*"))):
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:
            continue
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = short(r["Kernel_Name"])
            if k:
                sq[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                span[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if sq:
        SIMDS, CUS, GHZ = 1024, 256, 2.4
        with open(os.path.join(dst, f"{tag}_sq_counters.txt"), "w") as f:
            f.write(f"rocprofv3 --pmc <SQ group> -- python3 bench.py --steps 2 --warmup 1 (three passes), {batch} trials per launch;\n"
                    "means per dispatch.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count in units of 4 cycles (one issue slot of a\n"
                    f"SIMD); pipe shares assume {SIMDS} SIMDs, {CUS} CUs and {GHZ} GHz over the dispatch's own duration.\n\n")
            for k, c in sq.items():
                m = {n: mean(v) for n, v in c.items()}
                ns = mean(span[k])
                quads = ns * GHZ / 4 * SIMDS
                f.write(f"{k}   ({ns / 1e6:.3f} ms per dispatch under the profiler)\n")
                for n in sorted(m):
                    f.write(f"    {n:24s} {m[n]:16.0f}\n")
                w = m.get("SQ_WAVES")
                if w:
                    f.write("    per wave: " + "  ".join(f"{n[9:]}={m[n] / w:.0f}" for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if n in m) + "\n")
                if "SQ_ACTIVE_INST_VALU" in m:
                    f.write(f"    VALU issue slots busy: {m['SQ_ACTIVE_INST_VALU'] / quads:.2f} of the SIMDs' time\n")
                if "SQ_LDS_IDX_ACTIVE" in m:
                    f.write(f"    LDS pipe busy: {m['SQ_LDS_IDX_ACTIVE'] / (ns * GHZ * CUS):.2f} of the CUs' time "
                            f"({m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m['SQ_LDS_IDX_ACTIVE']):.2f} of it bank conflicts)\n")
                if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" not in m:
                    pass
                f.write("\n")
        print(open(os.path.join(dst, f"{tag}_sq_counters.txt")).read())


if __name__ == "__main__":
    main()
