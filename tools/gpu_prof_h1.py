import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FXJPS_LIB"] = os.path.join(ROOT, "fuxi-planner_amd", "libfxjps_prof.so")
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
p = fx.Planner([0]); L = _lib.load(); L.fxjps_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
occ = synth.synth_grid(1024, 1024, 1, 0.20); p.set_grid_occ(occ)
names = ["pops 2..K", "c:wait+probe_eval", "node select+ec issue", "x:dirs+issue", "x:eval+diag", "x:more+bcast", "c:math+nvalid+write+push", "c:hazard", "pop 1 (+refill)", "looptop"]
for h in (1, 2):
    s, g = synth.synth_queries(occ, 1, 256)
    t = time.time(); off, cells, cost, st = p.plan_batch(s, g, h, 1024); dt = time.time() - t
    c = (C.c_uint64 * 32)(); L.fxjps_debug_counters(p._h, c); c = list(c)
    print("h=%d wall %.3f pops %d pushes %d refills %d slow %d" % (h, dt, c[0], c[1], c[2], c[3]), flush=True)
    tot = sum(c[8:18])
    print("   cycles/pop %.0f : " % (tot / max(c[0], 1)) + ", ".join("%s %.0f" % (names[k], c[8 + k] / max(c[0], 1)) for k in range(10)), flush=True)
