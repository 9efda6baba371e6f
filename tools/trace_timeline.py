#!/usr/bin/env python3
"""Start / end of every kernel of a rocprofv3 --kernel-trace run, in ms from the first one: tools/trace_timeline.py <dir> [min_us]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
for s, e, n, q, st in rows:
    if (e - s) / 1e3 >= mn:
        print("%9.2f -> %9.2f ms  (%8.2f ms)  queue %s stream %s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, n))
