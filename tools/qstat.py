#!/usr/bin/env python3
"""Per-query timeline of one batch (FXJPS_QSTAT diagnostics): when each query started and ended, how many pops it
made, on which wavefront -- the per-pop rate as a function of how busy the chip is.
Usage: python tools/qstat.py [workload=c2] [FXJPS_WAVES=...]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FXJPS_QSTAT"] = "1"
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
wname = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = json.load(open(os.path.join(ROOT, "fuxi-planner_amd", "workloads.json")))[wname]
p = fx.Planner([0])
L = _lib.load()
L.fxjps_debug_qstat.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
occ = synth.synth_grid(wl["W"], wl["H"], wl["grid_seed"], wl["p"])
p.set_grid_occ(occ)
nq = wl["nq"]
s, g = synth.synth_queries(occ, wl["qseed"], nq)
for rep in range(2):
    off, cells, cost, st = p.plan_batch(s, g, wl["hchoice"], wl["max_path_len"])
tm = p.timing()
q = np.zeros((nq, 4), dtype=np.uint64)
assert L.fxjps_debug_qstat(p._h, q.ctypes.data_as(C.POINTER(C.c_uint64)), nq) == 0
ran = q[:, 2] > 0
t0 = q[ran, 0].min()
ts, te, pops = (q[ran, 0] - t0) / 100.0, (q[ran, 1] - t0) / 100.0, q[ran, 2].astype(float)  # microseconds
print("%s waves=%s: kernel %.1f ms, %d searched queries, span %.1f ms, pops %.3e" % (wname, os.environ.get("FXJPS_WAVES", "full"), tm["search_kernel_ms"], ran.sum(), te.max() / 1e3, pops.sum()))
rate = (te - ts) / pops
print("us/pop over queries: mean %.3f  p10 %.3f  p50 %.3f  p90 %.3f" % (rate.mean(), *np.percentile(rate, [10, 50, 90])))
# rate by start time window and by how many queries are in flight at the query's midpoint
edges = np.linspace(0, te.max(), 11)
mid = 0.5 * (ts + te)
for a, b in zip(edges[:-1], edges[1:]):
    m = (mid >= a) & (mid < b)
    inflight = ((ts < 0.5 * (a + b)) & (te > 0.5 * (a + b))).sum()
    if m.any():
        print("  t %6.1f-%6.1f ms: in flight %5d, queries with midpoint here %5d, us/pop %.3f" % (a / 1e3, b / 1e3, inflight, m.sum(), (pops[m] * rate[m]).sum() / pops[m].sum()))
k = np.argsort(-pops)[:5]
for i in k:
    print("  longest: pops %d  start %.1f ms  end %.1f ms  us/pop %.3f" % (pops[i], ts[i] / 1e3, te[i] / 1e3, rate[i]))
# the tail: the queries that end last, with their place in the longest-first order (Chebyshev distance, descending)
ids = np.nonzero(ran)[0]
d = np.maximum(abs(s[:, 0] - g[:, 0]), abs(s[:, 1] - g[:, 1]))
rank = np.empty(nq, dtype=np.int64)
rank[np.argsort(-d, kind="stable")] = np.arange(nq)
for i in np.argsort(-te)[:24]:
    qi = ids[i]
    print("  ends last: q %5d  rank %5d  dx %4d dy %4d  pops %6d  start %.1f  end %.1f ms  us/pop %.3f  wave %d" % (
        qi, rank[qi], abs(s[qi, 0] - g[qi, 0]), abs(s[qi, 1] - g[qi, 1]), pops[i], ts[i] / 1e3, te[i] / 1e3, rate[i], int(q[qi, 3]) & 0xFFFFFF))
for t in (40, 50, 60, 70, 80, 90):
    print("  still running at %d ms: %d" % (t, (te > t * 1e3).sum()))
np.save(os.path.join(ROOT, "gpurun_out", "qstat_%s_%s.npy" % (wname, os.environ.get("FXJPS_WAVES", "full"))), q)
