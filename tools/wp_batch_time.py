#!/usr/bin/env python3
"""Waypoint selection over the 10 000 paths of BASELINE config 2, still resident behind plan_batch: the ccst pruning and the st rule
as device kernels, and the st rule on host threads (FXJPS_WAYPOINT_ST_HOST=1).  Wall time of the call (inputs from host arrays,
outputs back in host arrays), median of 7; the first st call also fills the table of the host's atan2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, waypoints

occ = synth.synth_grid(1024, 1024, 1, 0.20)
nq = 10000
s, g = synth.synth_queries(occ, 1, nq)
with fx.Planner([0]) as p:
    p.set_grid_occ(occ)
    p.plan_batch(s, g, 2, 1024)
    pos = np.c_[s + 0.5, np.zeros(nq)]
    goal = np.c_[g + 0.0, np.ones(nq)]
    ms = s + 1

    def med(f, n=7):
        ts = []
        for _ in range(n):
            t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
        return ts[0] * 1e3, float(np.median(ts[1:])) * 1e3
    print("ccst, device kernel:   first %.2f ms, then %.2f ms" % med(lambda: waypoints.select_ccst_batch(p, nq, 0.2, (0.0, 0.0), pos, goal)))
    print("st, device kernel:     first %.2f ms (fills the atan2 table), then %.2f ms" % med(lambda: waypoints.select_st_batch(p, nq, ms, 0.2, (0.0, 0.0), pos, goal)))
    os.environ["FXJPS_WAYPOINT_ST_HOST"] = "1"
    for nt in (1, 16, 0):
        print("st, %3s host threads:  first %.2f ms, then %.2f ms" % ((nt or "all",) + med(lambda: waypoints.select_st_batch(p, nq, ms, 0.2, (0.0, 0.0), pos, goal, nthreads=nt))))
