#!/usr/bin/env python3
"""What ends the committed prefix of an iteration, and what ends a second round (-DFXJPS_HWID build, libfxjps_hwid.so):
tools/r2_stats.py [query ids | all]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FXJPS_LIB"] = os.path.join(ROOT, "fuxi-planner_amd", "libfxjps_hwid.so")
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
p = fx.Planner([0]); L = _lib.load()
L.fxjps_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
occ = synth.synth_grid(1024, 1024, 1, 0.2); p.set_grid_occ(occ)
s, g = synth.synth_queries(occ, 1, 10000)
for arg in (sys.argv[1:] or ["9206"]):
    if arg == "all":
        ss, gg = s, g
    else:
        q = int(arg); ss, gg = s[q:q + 1], g[q:q + 1]
    p.plan_batch(ss, gg, int(os.environ.get("FX_HC", "2")), 1024)
    c = (C.c_uint64 * 64)(); L.fxjps_debug_counters(p._h, c); c = list(c)
    it = max(c[55], 1)
    print("%s: %d iterations, %.2f nodes popped, %.2f committed by the first round; prefix = batch %.1f %%, cut by ONE intruder %.1f %%, by several at once %.1f %%, by a hazard / goal / parent %.1f %%" % (
        arg, c[55], c[60] / it, c[61] / it, 100.0 * c[56] / it, 100.0 * c[57] / it, 100.0 * c[58] / it, 100.0 * c[59] / it))
    r2 = max(c[48], 1)
    print("   second rounds %d (%.1f %% of the iterations): + %.2f nodes each, %.2f left behind; ended by: end of batch %.1f %%, another push of the first round %.1f %%, a push of s %.1f %%, touched node / hazard %.1f %%" % (
        c[48], 100.0 * c[48] / it, c[53] / r2, c[54] / r2, 100.0 * c[49] / r2, 100.0 * c[50] / r2, 100.0 * c[51] / r2, 100.0 * c[52] / r2))
