#!/usr/bin/env python3
"""At what clock do the longest queries of a batch run -- alone on the chip, and inside their batch?  FXJPS_QSTAT records
each search's wall time (s_memrealtime, 100 MHz) and its shader-clock cycles (s_memtime): cycles / microsecond is the
clock the wavefront saw.  A query that takes longer inside the batch at the SAME number of cycles ran at a lower clock
(the chip under load); more cycles mean waiting (memory, a shared CU).   python tools/clock_probe.py [workload=c2] [n=6]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FXJPS_QSTAT"] = "1"
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
wname = sys.argv[1] if len(sys.argv) > 1 else "c2"
ntop = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = json.load(open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")))[wname]
p = fx.Planner([0])
L = _lib.load()
L.fxjps_debug_qstat.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
occ = synth.synth_grid(wl["W"], wl["H"], wl["grid_seed"], wl["p"])
p.set_grid_occ(occ)
nq = wl["nq"]
s, g = synth.synth_queries(occ, wl["qseed"], nq)


def stats(ss, gg):
    for rep in range(2):
        p.plan_batch(ss, gg, wl["hchoice"], wl["max_path_len"])
    q = np.zeros((len(ss), 4), dtype=np.uint64)
    assert L.fxjps_debug_qstat(p._h, q.ctypes.data_as(C.POINTER(C.c_uint64)), len(ss)) == 0
    us = (q[:, 1] - q[:, 0]).astype(float) / 100.0
    return us, (q[:, 3] >> np.uint64(24)).astype(float), q[:, 2].astype(float)


us, cyc, pops = stats(s, g)
top = np.argsort(-us)[:ntop]
print("%s: the %d queries that take longest inside their batch (kernel %.1f ms)" % (wname, ntop, p.timing()["search_kernel_ms"]))
for i in top:
    a_us, a_cyc, a_pops = stats(s[i:i + 1], g[i:i + 1])
    print("  q %5d pops %6d | in the batch: %8.1f us, %6.1f M cycles, %4.0f MHz, %5.0f cycles / pop | alone: %8.1f us, %6.1f M cycles, %4.0f MHz, %5.0f cycles / pop | time x %.3f = clock x %.3f * cycles x %.3f" % (
        i, pops[i], us[i], cyc[i] / 1e6, cyc[i] / us[i], cyc[i] / pops[i], a_us[0], a_cyc[0] / 1e6, a_cyc[0] / a_us[0], a_cyc[0] / a_pops[0],
        us[i] / a_us[0], (a_cyc[0] / a_us[0]) / (cyc[i] / us[i]), cyc[i] / a_cyc[0]))
