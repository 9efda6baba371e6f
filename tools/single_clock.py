#!/usr/bin/env python3
"""The shader clock a single call's kernel runs at (FXJPS_DEBUG lines of fxjps_plan_one): one wavefront on an otherwise idle chip,
calls back to back / with pauses between them.  usage: FXJPS_DEBUG=1 python tools/single_clock.py 2> log"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(root, "tests", "golden", "maps_png.npz"))
import json
recs = json.load(open(os.path.join(root, "tests", "golden", "maps_png.json")))
nm = "-16.20-11.40_out.png"
W, H = [r for r in recs if r["map"] == nm and "canvas" not in r][0]["shape"]
g = np.unpackbits(z[nm])[:W * H].reshape(W, H).astype(np.uint8)
canvas = np.zeros((256, 256), np.uint8)
canvas[:W, :H] = g
with fx.Planner([0]) as p:
    p.set_grid_occ(canvas)
    for pause in (0.0, 0.001, 0.02, 0.2):
        sys.stderr.write("== pause %.3f s between calls\n" % pause)
        for i in range(8):
            t = time.perf_counter()
            p.plan_one((0, 0), (146, 111), 2)
            dt = time.perf_counter() - t
            sys.stderr.write("   call %.1f us\n" % (dt * 1e6))
            time.sleep(pause)
    big = synth.synth_grid(1024, 1024, 1, 0.20)
    p.set_grid_occ(big)
    s, gq = synth.synth_queries(big, 1, 10000)
    sys.stderr.write("== config 2, query 9206 alone\n")
    for i in range(3):
        p.plan_one(tuple(s[9206]), tuple(gq[9206]), 2)
