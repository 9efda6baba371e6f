#!/usr/bin/env python3
"""Counters of the last k_search launch of a rocprofv3 --pmc pass: tools/pmc_sum.py <dir> <name> <log of tools/gpu_prof.py --noprof>.
Prints every counter, and per pop where the log says how many pops the launch made."""
import csv, glob, re, sys
d, name, log = sys.argv[1:4]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "k_search" in r["Kernel_Name"]]
if not rows:
    print("%s: no k_search rows (counters not available on this device?)" % name)
    sys.exit(0)
last = max(int(r["Dispatch_Id"]) for r in rows)
c = {}
for r in rows:
    if int(r["Dispatch_Id"]) == last:
        c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
m = re.findall(r"pops (\d+)", open(log).read())
pops = float(m[-1]) if m else 0.0
r0 = [r for r in rows if int(r["Dispatch_Id"]) == last][0]
ms = (int(r0["End_Timestamp"]) - int(r0["Start_Timestamp"])) / 1e6
print("%s: kernel %.2f ms, pops %d | " % (name, ms, pops) + ", ".join("%s %.4g%s" % (k, v, (" (%.3f / pop)" % (v / pops)) if pops else "") for k, v in sorted(c.items())))
