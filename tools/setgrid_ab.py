import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
occ = synth.synth_grid(1024, 1024, 1, 0.2)
s, g = synth.synth_queries(occ, 1, 10000)
p = fx.Planner([0])
for knob in (None, "1", None, "1"):
    if knob: os.environ["FXJPS_TABLE_SHRINK"] = knob
    else: os.environ.pop("FXJPS_TABLE_SHRINK", None)
    ts = []
    for rep in range(6):
        o2 = occ.copy(); o2[rep, rep] ^= 1
        t = time.perf_counter(); p.set_grid_occ(o2); t1 = time.perf_counter(); p.plan_batch(s, g, 2, 1024); t2 = time.perf_counter()
        ts.append(((t1 - t) * 1e3, (t2 - t1) * 1e3))
    print("pools re-sized and wiped at every set_grid" if knob else "pools kept across grids of one shape", ": set_grid %.2f ms, plan_batch(10 000) %.1f ms (medians of 6)" % (np.median([a for a, b in ts]), np.median([b for a, b in ts])))
