#!/usr/bin/env python3
"""Where the time of ONE call of the drop-in goes (the node's real call: one query per tick on a map of ~ 150 x 110 cells):
Planner.plan as a whole, the ctypes calls inside it with preallocated arrays, the library's own wall time of the call, the
kernel.  And jps1.method: the same plus grid conversion, upload and map build.   python tools/single_call_breakdown.py"""
import contextlib, ctypes as C, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from fuxi_planner_amd import _lib
z = np.load(os.path.join(ROOT, "tests", "golden", "maps_png.npz"))
recs = json.load(open(os.path.join(ROOT, "tests", "golden", "maps_png.json")))
name = "-16.20-11.40_out.png"
occ = np.zeros((256, 256), dtype=np.uint8)
occ[:147, :112] = np.unpackbits(z[name])[:147 * 112].reshape(147, 112)
p = fx.Planner([0])
L = _lib.load()
p.set_grid_occ(occ)
N = 200


def med(fn, n=N):
    fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e6


for s, g in (((5, 100), (140, 5)), ((0, 0), (146, 111))):
    t_plan = med(lambda: p.plan(s, g, 2))
    starts, goals = np.array([s], np.int32), np.array([g], np.int32)
    off, st, cost, secs = np.zeros(2, np.int64), np.zeros(1, np.int32), np.zeros(1, np.float64), C.c_double(0)
    cells = np.empty((2048, 2), np.int32)

    def raw():
        L.fxjps_plan_batch_csr(p._h, _lib.ptr(starts, C.c_int32), _lib.ptr(goals, C.c_int32), 1, 2, 1024, _lib.ptr(off, C.c_int64), None, 0,
                               _lib.ptr(st, C.c_int32), _lib.ptr(cost, C.c_double), C.byref(secs))
        L.fxjps_last_cells(p._h, _lib.ptr(cells, C.c_int32), int(off[1]))
    t_raw = med(raw)
    tm = p.timing()
    m64 = occ.astype(np.float64)
    with contextlib.redirect_stdout(io.StringIO()):
        t_method = med(lambda: fx.jps1.method(m64, s, g, 2), 50)
    t_conv = med(lambda: fx.as_occ(m64), 50)
    t_setgrid = med(lambda: p.set_grid_occ(occ), 50)
    print("%s -> %s: Planner.plan %.1f us = Python around the calls %.1f + two ctypes calls %.1f (the library's own wall time of the planning call %.1f, "
          "kernel %.1f); jps1.method %.1f us = that + `matrix == 1` %.1f + set_grid (upload, map build, wait) %.1f" % (
              s, g, t_plan, t_plan - t_raw, t_raw, tm["total_ms"] * 1e3, tm["search_kernel_ms"] * 1e3, t_method, t_conv, t_setgrid))
