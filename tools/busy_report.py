#!/usr/bin/env python3
"""VALUBusy / SALUBusy of the search kernel from a `rocprofv3 --pmc VALUBusy SALUBusy` pass (tools/gpu_round3.sh busy).
Prints one CSV line per counter: workload, kernel, counter, value (mean over the search launches)."""
import csv, glob, sys
P, w = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(P + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
print("workload,kernel,counter,value_percent,launches")
for (k, c), v in sorted(acc.items()):
    print("%s,\"%s\",%s,%.2f,%d" % (w, k, c, sum(v) / len(v), len(v)))
