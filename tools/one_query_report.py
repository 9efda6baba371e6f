#!/usr/bin/env python3
"""Counters of tools/one_query.py runs under rocprofv3 --pmc (directories given on the command line): mean per search
launch, and per pop when the number of pops is given (second-to-last argument `pops=N`)."""
import csv, glob, sys
pops = None
dirs = []
for a in sys.argv[1:]:
    if a.startswith("pops="):
        pops = float(a[5:])
    else:
        dirs.append(a)
acc = {}
for P in dirs:
    for f in glob.glob(P + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "k_search" in r["Kernel_Name"]:
                per.setdefault((r["Dispatch_Id"], r["Counter_Name"]), 0.0)
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, c), v in per.items():
            acc.setdefault(c, []).append(v)
for c, v in sorted(acc.items()):
    m = sum(v) / len(v)
    print("%-26s %16.0f per launch%s   (%d launches)" % (c, m, ("  %10.2f per pop" % (m / pops)) if pops else "", len(v)))
