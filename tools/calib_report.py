#!/usr/bin/env python3
"""Reads the two rocprofv3 passes of tools/calib_scatter (tools/gpu_round2.sh calib) and prints, per access shape,
counter bytes / bytes touched -- the calibration factors recorded in profiles/hbm_traffic.json."""
import csv, glob, json, sys
out = sys.argv[1]
n = 256 * 16 * 256 * 64
touched = {"k_read16_scatter": n * 16, "k_read64_scatter": n * 64, "k_write16_scatter": n * 16, "k_write4_scatter": n * 4,
           "k_read16_stream": 8 << 30, "k_write16_stream": 8 << 30}
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/calib_%s/**/*counter_collection.csv" % (out, c), recursive=True)
    if not f:
        print("no counter file for", c); continue
    rows = list(csv.DictReader(open(f[0])))
    for k in touched:
        v = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"].startswith(k) and r["Counter_Name"] == c]
        if v:
            res.setdefault(k, {})[c] = v[-1] * 1024.0  # KB; last of the two repetitions
for k, d in res.items():
    print("%-18s touched %12d B  " % (k, touched[k]) + "  ".join("%s %14.0f B = x%.3f" % (c, b, b / touched[k]) for c, b in d.items()))
json.dump({k: {c: b / touched[k] for c, b in d.items()} for k, d in res.items()}, open(out + "/calib_factors.json", "w"), indent=1)
