#!/usr/bin/env python3
"""Randomised differential stress of the streaming side: random maps, random sequences of cell updates (sensor windows,
scattered cells, walls that split and rejoin components, lists with cells outside the grid, deferred updates, large
lists), through fxjps_update_cells / _deferred / fxjps_replan_frame on one handle.  After every rebuild the plans of the
persistent queries are compared with the CPU oracle on the host's copy of the grid (cells, lengths, float64 costs), every
few steps the derived device maps with those of a fresh upload on a second handle, and the waypoints of the batch
(fxjps_waypoint_ccst_batch on the resident paths) with the one-path host function.
Usage: python tools/gpu_stress_updates.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuxi_planner_amd as fx
from fuxi_planner_amd import waypoints
from oracle import oracle
from gpu_stress import make_map


def roots(par):
    r = par.astype(np.int64).ravel().copy()
    idx = np.flatnonzero(r >= 0)
    while True:
        nxt = r[r[idx]]
        if np.array_equal(nxt, r[idx]):
            return r
        r[idx] = nxt


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    p, q = fx.Planner([0]), fx.Planner([0])
    t0 = time.time(); nmaps = nupd = nq_tot = bad = nmapcmp = 0
    while time.time() - t0 < secs:
        kind, cur = make_map(rng)
        W, H = cur.shape
        if W < 3 or H < 3 or (cur == 0).sum() < 2:
            continue
        n = int(rng.integers(20, 300))
        free = np.argwhere(cur == 0)
        s = free[rng.integers(0, len(free), n)].astype(np.int32); g = free[rng.integers(0, len(free), n)].astype(np.int32)
        h = int(rng.integers(1, 3))
        mpl = 4 * max(W, H) + 64
        p.set_grid_occ(cur)
        p.set_queries(s, g, h, mpl)
        nmaps += 1
        for step in range(int(rng.integers(3, 14))):
            kind_u = int(rng.integers(0, 7))
            if kind_u <= 1:
                w = int(rng.integers(1, min(48, W, H) + 1))
                x0, y0 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - w + 1))
                xs, ys = np.meshgrid(np.arange(x0, x0 + w), np.arange(y0, y0 + w), indexing="ij")
                xy = np.stack([xs.ravel(), ys.ravel()], 1); val = (rng.random(len(xy)) < rng.choice([0.05, 0.2, 0.5])).astype(np.uint8)
            elif kind_u == 2:
                k = int(rng.integers(1, 40))
                xy = np.unique(np.stack([rng.integers(-2, W + 2, k), rng.integers(-2, H + 2, k)], 1), axis=0); val = rng.integers(0, 2, len(xy)).astype(np.uint8)
            elif kind_u == 3:
                x = int(rng.integers(0, W)); xy = np.stack([np.full(H, x), np.arange(H)], 1); val = np.full(H, rng.integers(0, 2), dtype=np.uint8)
            elif kind_u == 4:
                y = int(rng.integers(0, H)); xy = np.stack([np.arange(W), np.full(W, y)], 1); val = np.full(W, rng.integers(0, 2), dtype=np.uint8)
            elif kind_u == 5:
                k = min(W * H, int(rng.integers(9000, 30000))); idx = rng.choice(W * H, k, replace=False)
                xy = np.stack([idx // H, idx % H], 1); val = rng.integers(0, 2, k).astype(np.uint8)
            else:
                xy = np.zeros((0, 2), np.int64); val = np.zeros(0, np.uint8)
            ok = (xy[:, 0] >= 0) & (xy[:, 0] < W) & (xy[:, 1] >= 0) & (xy[:, 1] < H)
            cur[xy[ok, 0], xy[ok, 1]] = val[ok]
            nupd += 1
            mode = int(rng.integers(0, 4))
            if mode == 0:
                p.update_cells(xy.astype(np.int32), val, rebuild=False)  # the next step's call rebuilds
                continue
            if mode == 1:
                p.update_cells(xy.astype(np.int32), val)
                off, cells, cost, st = p.plan_batch(s, g, h, mpl)
                p.set_queries(s, g, h, mpl)
            else:
                off, cells, cost, st = p.replan_frame(xy.astype(np.int32), val)
            oc, ol, ocost, _ = oracle.plan_batch(cur, s, g, h, literal=False, max_len=mpl, nthreads=16)
            good = np.array_equal(st, ol) and cost.tobytes() == ocost.tobytes()
            if good:
                for k in range(n):
                    if not np.array_equal(cells[off[k]:off[k + 1]], oc[k, :max(int(ol[k]), 0)]):
                        good = False; break
            if good and step % 3 == 0:   # waypoints of the resident paths
                pos = np.c_[s + rng.normal(0, 1.0, s.shape), rng.uniform(0, 2, n)]; goal = np.c_[g + 0.5, np.ones(n)]
                wp, gout, nk = waypoints.select_ccst_batch(p, n, 1.0, (0.0, 0.0), pos, goal)
                for k in np.flatnonzero(st > 0)[:40]:
                    w1, k1 = waypoints.select_ccst(cells[off[k]:off[k + 1]], cur, 1.0, (0.0, 0.0), pos[k], goal[k])
                    if wp[k].tobytes() != w1.tobytes() or nk[k] != len(k1):
                        good = False; print("WAYPOINT MISMATCH", k, flush=True); break
                # ... and the st rule (device kernel, angles from the table of the host's atan2) against the one-path host function
                ms = s + 1 + rng.integers(-2, 3, s.shape)
                pw = rng.uniform(-5, 50, (n, 3)); pd = rng.choice([0, 2, 3], n).astype(np.int32)
                wps, dims, gs, angs = waypoints.select_st_batch(p, n, ms, 0.5, (1.0, -1.0), pos, goal, None, pw, pd)
                for k in np.flatnonzero(st > 0)[:40]:
                    w1, g1, a1 = waypoints.select_st(cells[off[k]:off[k + 1]], ms[k], 0.5, (1.0, -1.0), pos[k], goal[k], 0, None if pd[k] == 0 else pw[k, :pd[k]])
                    if wps[k, :dims[k]].tobytes() != w1.tobytes() or angs[k] != a1 or gs[k].tobytes() != g1.tobytes():
                        good = False; print("ST WAYPOINT MISMATCH", k, flush=True); break
            if good and step % 4 == 0:   # derived maps against a fresh upload
                q.set_grid_occ(cur)
                a, b = p.debug_maps(), q.debug_maps()
                nmapcmp += 1
                for name in ("nb8", "bm", "ci", "dbm", "jd"):
                    if not np.array_equal(a[name], b[name]):
                        good = False; print("MAP MISMATCH", name, flush=True)
                fr = np.flatnonzero(cur.ravel() == 0)
                ra, rb = roots(a["comp"])[fr], roots(b["comp"])[fr]
                if (ra < 0).any() or np.unique(np.stack([rb, ra]), axis=1).shape[1] != len(np.unique(rb)):
                    good = False; print("LABEL MISMATCH", flush=True)
            nq_tot += n
            if not good:
                bad += 1
                print("MISMATCH kind=%s %dx%d h=%d n=%d seed=%d map#%d step=%d update=%d mode=%d" % (kind, W, H, h, n, seed, nmaps, step, kind_u, mode), flush=True)
    print("update stress: %d maps, %d updates, %d plans compared, %d map comparisons, %d bad, %.0f s" % (nmaps, nupd, nq_tot, nmapcmp, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
