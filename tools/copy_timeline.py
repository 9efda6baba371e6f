#!/usr/bin/env python3
"""Kernels (>= 1 ms) and memory copies of a rocprofv3 --kernel-trace --memory-copy-trace run, in ms from the first k_search:
tools/copy_timeline.py <dir> [from_ms to_ms]"""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, e, "KERNEL q%s %s" % (r["Queue_Id"], r["Kernel_Name"][:34])))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, e, "COPY %s stream %s" % (r.get("Direction", "?").replace("MEMORY_COPY_", ""), r.get("Stream_Id", "?"))))
ev.sort()
t0 = min(s for s, e, n in ev if "k_search" in n)
lo = float(sys.argv[2]) if len(sys.argv) > 2 else -1e9
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
for s, e, n in ev:
    a = (s - t0) / 1e6
    if lo <= a <= hi:
        print("%9.3f -> %9.3f (%8.3f ms) %s" % (a, (e - t0) / 1e6, (e - s) / 1e6, n))
