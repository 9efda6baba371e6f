#!/usr/bin/env python3
"""Cell updates on OPEN maps: 64 x 64 windows re-observed on a 4096 x 4096 map of few obstacles, where the rays -- and the walks
back along them (k_jd_walk) -- are as long as the map.  Three ways, same updates: long walks handed on to wavefronts of k_jd_finish
(default), the records streamed when a walk outgrows its bound (round 5: FXJPS_JD_OVF_CAP=0), every record read (round 4:
FXJPS_JD_WALK=0).  The resulting maps are compared with a fresh upload at the end of each run.
usage: python tools/update_sparse.py [density ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(dens, env):
    for k in ("FXJPS_JD_OVF_CAP", "FXJPS_JD_WALK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    import fuxi_planner_amd as fx
    rng = np.random.default_rng(5)
    occ = (rng.random((4096, 4096)) < dens).astype(np.uint8)
    with fx.Planner([0]) as p, fx.Planner([0]) as q:
        p.set_grid_occ(occ)
        ts = []
        for rep in range(40):
            x0, y0 = int(rng.integers(0, 4096 - 64)), int(rng.integers(0, 4096 - 64))
            xs, ys = np.meshgrid(np.arange(x0, x0 + 64), np.arange(y0, y0 + 64), indexing="ij")
            xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
            val = (rng.random(len(xy)) < max(dens, 0.05)).astype(np.uint8)
            occ[xy[:, 0], xy[:, 1]] = val
            t = time.perf_counter()
            p.update_cells(xy, val)
            ts.append(time.perf_counter() - t)
        q.set_grid_occ(occ)
        a, b = p.debug_maps(), q.debug_maps()
        bad = {k: int((a[k] != b[k]).sum()) for k in ("nb8", "bm", "ci", "dbm", "jd") if not np.array_equal(a[k], b[k])}
    ts = np.array(ts[8:]) * 1e3
    return float(np.median(ts)), float(ts.max()), bad


if __name__ == "__main__":
    rc = 0
    for dens in [float(x) for x in sys.argv[1:]] or [0.0, 0.001, 0.01, 0.05, 0.20]:
        row = []
        for name, env in (("handed on", {}), ("streamed on give-up", {"FXJPS_JD_OVF_CAP": "0"}), ("every record read", {"FXJPS_JD_WALK": "0"})):
            med, mx, bad = run(dens, env)
            row.append("%s %.3f ms (max %.3f)%s" % (name, med, mx, " MAPS DIFFER %r" % bad if bad else ""))
            rc |= bool(bad)
        print("density %.3f: %s" % (dens, " | ".join(row)), flush=True)
    sys.exit(rc)
