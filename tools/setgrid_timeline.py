#!/usr/bin/env python3
"""What one fxjps_set_grid of a small map is made of on the device: run under
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/setgrid_timeline.py run
then   python3 tools/setgrid_timeline.py report <dir>
prints, for the median call of 40, every copy and kernel with its start (us from the call's first operation) and duration,
and the host's wall time per call beside it (the canvas of BASELINE config 1: the 147 x 112 reference map in 256 x 256)."""
import csv, glob, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import fuxi_planner_amd as fx
    z = np.load(os.path.join(ROOT, "tests", "golden", "maps_png.npz"))
    occ = np.zeros((256, 256), dtype=np.uint8)
    occ[:147, :112] = np.unpackbits(z["-16.20-11.40_out.png"])[:147 * 112].reshape(147, 112)
    alt = occ.copy()
    alt[255, 255] = 1
    p = fx.Planner([0])
    p.set_grid_occ(occ)
    p.plan((0, 0), (146, 111), 2)
    ts = []
    for i in range(40):
        time.sleep(0.002)  # (a gap the report splits the calls at)
        t = time.perf_counter()
        p.set_grid_occ(alt if i & 1 else occ)
        ts.append((time.perf_counter() - t) * 1e6)
        p.plan((0, 0), (146, 111), 2)
    print("set_grid_occ host wall time: median %.1f us, min %.1f" % (float(np.median(ts)), min(ts)))
else:
    ops = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
    for f in glob.glob(sys.argv[2] + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?"))))
    ops.sort()
    calls, cur = [], []
    for o in ops:
        if cur and o[0] - cur[-1][1] > 1000000:  # > 1 ms of nothing: the next call
            calls.append(cur)
            cur = []
        cur.append(o)
    calls.append(cur)
    calls = [c for c in calls if any("k_derive_jd" in o[2] or "k_derive_all" in o[2] for o in c)]
    spans = sorted((c[-1][1] - c[0][0], i) for i, c in enumerate(calls))
    c = calls[spans[len(spans) // 2][1]]
    print("%d calls with a map build; the median one (first operation to the end of the search): %.1f us" % (len(calls), (c[-1][1] - c[0][0]) / 1e3))
    for s, e, n in c:
        print("  %8.1f us  + %7.1f us  %s" % ((s - c[0][0]) / 1e3, (e - s) / 1e3, n))
