#!/usr/bin/env python3
"""Randomised differential stress: HIP planner vs CPU oracle on many map families, sizes, densities and both
heuristics.  Usage: python tools/gpu_stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from oracle import oracle

def make_map(rng):
    kind = rng.choice(["iid", "iid", "rooms", "blobs", "open", "maze", "thin"])
    W, H = int(rng.integers(2, 700)), int(rng.integers(2, 700))
    if kind == "thin":
        W = int(rng.integers(1, 4)); H = int(rng.integers(50, 3000))
        if rng.random() < 0.5: W, H = H, W
    occ = np.zeros((W, H), dtype=np.uint8)
    if kind in ("iid", "thin"):
        occ = (rng.random((W, H)) < float(rng.choice([0.02, 0.1, 0.2, 0.3, 0.4, 0.45]))).astype(np.uint8)
    elif kind == "rooms":
        k = int(rng.integers(5, 40))
        occ[::k, :] = 1; occ[:, ::k] = 1
        holes = rng.random((W, H)) < 0.15
        occ[holes & (occ == 1)] = 0
    elif kind == "blobs":
        for _ in range(int(rng.integers(5, 80))):
            x, y = int(rng.integers(0, W)), int(rng.integers(0, H)); r = int(rng.integers(1, 25))
            occ[max(0, x - r):x + r, max(0, y - r):y + r] = 1
    elif kind == "maze":
        occ[:] = 1
        occ[1::2, 1::2] = 0
        carve = rng.random((W, H)) < 0.6
        occ[carve & ((np.add.outer(np.arange(W), np.arange(H)) % 2) == 1)] = 0
    return kind, occ

def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    p = fx.Planner([0])
    t0 = time.time(); nmaps = nq_tot = bad = 0
    while time.time() - t0 < secs:
        kind, occ = make_map(rng)
        W, H = occ.shape
        n = int(rng.integers(50, 800))
        r_ = rng.random()
        if r_ < 0.08:  # now and then a batch large enough for the head launch (4 096 queries and more)
            n = int(rng.integers(4096, 6000))
        elif r_ < 0.2:  # ... and single calls (the one-launch path of nq == 1)
            n = 1
        if rng.random() < 0.7 and (occ == 0).sum() >= 2:
            free = np.argwhere(occ == 0)
            s = free[rng.integers(0, len(free), n)].astype(np.int32); g = free[rng.integers(0, len(free), n)].astype(np.int32)
        else:
            s = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
            g = np.stack([rng.integers(-1, W + 1, n), rng.integers(-1, H + 1, n)], 1).astype(np.int32)
        h = int(rng.integers(1, 3))
        p.set_grid_occ(occ)
        off, cells, cost, st = p.plan_batch(s, g, h)
        ml = max(int(st.max()), 1) + 8
        oc, ol, ocost, _ = oracle.plan_batch(occ, s, g, h, literal=False, max_len=ml, nthreads=min(os.cpu_count() or 16, 64))
        ok = np.array_equal(st, ol) and cost.tobytes() == ocost.tobytes()
        if ok:
            for q in range(n):
                if not np.array_equal(cells[off[q]:off[q + 1]], oc[q, :max(int(ol[q]), 0)]):
                    ok = False; break
        nmaps += 1; nq_tot += n
        if not ok:
            bad += 1
            print("MISMATCH kind=%s %dx%d h=%d n=%d seed=%d map#%d" % (kind, W, H, h, n, seed, nmaps), flush=True)
    print("stress: %d maps, %d queries, %d bad, %.0f s" % (nmaps, nq_tot, bad, time.time() - t0), flush=True)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
