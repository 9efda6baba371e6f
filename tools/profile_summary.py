#!/usr/bin/env python3
"""Condenses the rocprofv3 output of one workload (tools/gpu_round2.sh profiles) into one JSON record: kernel-stats rows,
per-launch counters of the search kernel, the bench line.  tools/collect_profiles.py turns these into profiles/."""
import csv, glob, json, os, sys
P, w = sys.argv[1], sys.argv[2]
rec = {"workload": w}
for f in glob.glob(P + "/stats/**/*kernel_stats.csv", recursive=True):
    rec["kernel_stats"] = [r for r in csv.DictReader(open(f))]
vals = {}
for d in glob.glob(P + "/pmc_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_search" in r["Kernel_Name"]]
        if not rows:
            continue
        # sum over the search launches of the one profiled step (a retry on the large pool is a second launch)
        for r in rows:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        vals["_launches_" + os.path.basename(os.path.dirname(d))] = len({r["Dispatch_Id"] for r in rows})
        r0 = rows[0]
        rec["registers"] = {"arch_vgpr": r0.get("VGPR_Count"), "accum_vgpr": r0.get("Accum_VGPR_Count"), "sgpr": r0.get("SGPR_Count"),
                            "lds_block_bytes": r0.get("LDS_Block_Size"), "scratch": r0.get("Scratch_Size"), "kernel": r0["Kernel_Name"]}
rec["counters_per_step"] = vals
try:
    rec["bench"] = json.loads(open(P + "/bench.json").read().strip().splitlines()[-1])
except Exception as e:  # noqa
    rec["bench_error"] = str(e)
print(json.dumps(rec))
