#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/pmc_cmd.sh <tag> <python script + args ...>
# Three rocprofv3 --pmc passes (instruction mix, pipe activity, LDS / waits) over the given python command; per-kernel
# summary printed and kept in gpurun_out/pmc_<tag>/summary.txt.
TAG=$1; shift
OUT=$(pwd)/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 "$@" > $OUT/g$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
span = collections.defaultdict(list)
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        span[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        m = {n: sum(v) / len(v) for n, v in d.items()}
        ns = sum(span[k]) / len(span[k])
        if ns < 2e5: continue
        quads = ns * 2.4 / 4 * 1024
        o.write(f"{k}  {ns/1e6:.3f} ms/dispatch\n")
        w = m.get("SQ_WAVES", 0)
        if w:
            o.write("    per wave: " + "  ".join(f"{n[9:]}={m[n]/w:.0f}" for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if n in m) + f"  waves={w:.0f}\n")
        if "SQ_ACTIVE_INST_VALU" in m: o.write(f"    VALU busy {m['SQ_ACTIVE_INST_VALU']/quads:.2f}")
        if "SQ_LDS_IDX_ACTIVE" in m: o.write(f"  LDS pipe busy {m['SQ_LDS_IDX_ACTIVE']/(ns*2.4*256):.2f} (conflicts {m.get('SQ_LDS_BANK_CONFLICT',0)/max(1,m['SQ_LDS_IDX_ACTIVE']):.2f})")
        if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m: o.write(f"  waves waiting {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} of their life")
        if "SQ_WAVE_CYCLES" in m: o.write(f"  resident waves/SIMD {m['SQ_WAVE_CYCLES']/quads:.1f}")
        o.write("\n")
print(open("$OUT/summary.txt").read())
PY
