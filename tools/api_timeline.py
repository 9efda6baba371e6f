#!/usr/bin/env python3
"""HIP API calls of a rocprofv3 --hip-runtime-trace run that took longer than min_ms, per thread, in ms from the first kernel:
tools/api_timeline.py <dir> [min_ms]"""
import csv, glob, sys
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
ks = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Kernel_Name"]:
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERNEL q%s" % r["Queue_Id"], ""))
t0 = min(k[0] for k in ks)
ev = list(ks)
for f in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (e - s) / 1e6 >= mn or r["Function"] in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel"):
            ev.append((s, e, r["Function"], "thread %s" % r["Thread_Id"]))
ev.sort()
for s, e, n, t in ev:
    if s >= t0 - 5e6:
        print("%9.2f -> %9.2f (%7.2f ms) %-28s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n, t))
