#!/usr/bin/env python3
"""How long does the query that ends config 2 (9206) take by the company it keeps?  Its own start / end inside each batch
(FXJPS_QSTAT), for batches made of: the query alone; the 16 queries of the head launch; those + n of the others."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FXJPS_QSTAT"] = "1"
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
wl = json.load(open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")))["c2"]
p = fx.Planner([0])
L = _lib.load()
L.fxjps_debug_qstat.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
occ = synth.synth_grid(wl["W"], wl["H"], wl["grid_seed"], wl["p"])
p.set_grid_occ(occ)
s, g = synth.synth_queries(occ, wl["qseed"], wl["nq"])
d = np.abs(s - g)
key = 2 * d.max(1) + d.min(1)
order = np.argsort(-key, kind="stable")
head = order[:16]
print("9206 is number %d of the longest-first order" % int(np.flatnonzero(order == 9206)[0]))
rest = order[16:]
sets = [("alone", np.array([9206])), ("the 16 of the head launch", head), ("head + next 48", order[:64]), ("head + next 240", order[:256]),
        ("head + next 1008", order[:1024]), ("head + next 4080", order[:4096]), ("head + the 2000 SHORTEST", np.r_[head, order[-2000:]]), ("all 10 000", order)]
for name, ids in sets:
    ids = np.asarray(ids)
    for rep in range(2):
        p.plan_batch(s[ids], g[ids], 2, 1024)
    tm = p.timing()
    q = np.zeros((len(ids), 4), dtype=np.uint64)
    assert L.fxjps_debug_qstat(p._h, q.ctypes.data_as(C.POINTER(C.c_uint64)), len(ids)) == 0
    i = int(np.flatnonzero(ids == 9206)[0])
    t0 = q[q[:, 2] > 0, 0].min()
    print("%-28s kernel %.2f ms | 9206: start %.2f ms, runs %.2f ms, %d pops" % (name, tm["search_kernel_ms"], (q[i, 0] - t0) / 1e5, (q[i, 1] - q[i, 0]) / 1e5, q[i, 2]), flush=True)
