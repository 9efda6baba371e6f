#!/usr/bin/env python3
"""Summarises the rocprofv3 passes tools/gpu_round2.sh pmc took over `bench.py --steps 1` (config 2): per-pop
instruction counts, HBM bytes per launch, L2 hit rate, kernel time."""
import csv, glob, json, sys
out = sys.argv[1]
vals = {}
for d in glob.glob(out + "/pmc_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_search" in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                vals[r["Counter_Name"]] = float(r["Counter_Value"])
                vals["_vgpr"], vals["_sgpr"], vals["_lds"] = r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")
wl = json.load(open("fuxi-planner_amd/workloads.json"))["c2"]
pops = 332391044.0  # pops the kernel executes on config 2 (stale duplicates of one batch are committed together)
print("regs vgpr %s sgpr %s lds %s" % (vals.get("_vgpr"), vals.get("_sgpr"), vals.get("_lds")))
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
    if k in vals:
        print("%-18s %.3e  per pop %.2f" % (k, vals[k], vals[k] / pops))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    f, w = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    print("FETCH %.1f GB  WRITE %.1f GB  sum %.1f GB (uncorrected)  per pop %.0f + %.0f B  algorithmic %.1f GB" % (f / 1e9, w / 1e9, (f + w) / 1e9, f / pops, w / pops, wl["algorithmic_bytes"] / 1e9))
if "TCC_HIT_sum" in vals:
    print("L2 hit rate %.3f" % (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])))
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Name"]:
            print("kernel stats:", r["Name"], "calls", r["Calls"], "avg ns", r["AverageNs"])
json.dump(vals, open(out + "/pmc_values.json", "w"), indent=1)
