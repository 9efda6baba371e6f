#!/usr/bin/env python3
"""Latency of one call of the drop-in (the way the ROS node uses it: one query per tick)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, io, contextlib
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
for W in (64, 256, 1024):
    occ = synth.synth_grid(W, W, 3, 0.15)
    free = np.argwhere(occ == 0)
    s, g = tuple(int(v) for v in free[0]), tuple(int(v) for v in free[-1])
    m = occ.astype(np.float64)
    with contextlib.redirect_stdout(io.StringIO()):
        fx.jps1.method(m, s, g, 2)
        t = time.time(); n = 20
        for _ in range(n): r = fx.jps1.method(m, s, g, 2)
        dt = (time.time() - t) / n
    p = fx.default_planner()
    p.set_grid(m)
    t = time.time()
    for _ in range(n): p.plan(s, g, 2)
    dt2 = (time.time() - t) / n
    print("grid %d^2: jps1.method %.2f ms per call (grid upload + maps + search, path of %s points); plan on the resident grid %.2f ms" % (W, dt * 1e3, len(r[0]) if r[0] else 0, dt2 * 1e3))
