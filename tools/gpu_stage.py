#!/usr/bin/env python3
"""Staged bring-up: prints before every GPU call so a hang is attributable."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
from oracle import oracle
stage = sys.argv[1]
def P(*a):
    print(*a, flush=True)
P("stage", stage)
p = fx.Planner([0])
P("planner created")
if stage == "tiny":
    m = np.zeros((5, 5))
    p.set_grid(m); P("grid set")
    nb = p.debug_nbmask(); P("nbmask\n", nb)
    P(p.plan((0, 0), (4, 4)), p.last_cost)
    P(p.plan((0, 0), (4, 2)), p.last_cost)
    P(p.plan((2, 2), (2, 2)), p.last_cost)
    m = np.zeros((7, 7)); m[3, 1:6] = 1
    p.set_grid(m); P(p.plan((1, 3), (5, 3)), p.last_cost)
    P(p.plan((1, 3), (5, 3), 1), p.last_cost)
elif stage == "small":
    rng = np.random.default_rng(5)
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
        W, H = int(rng.integers(3, 60)), int(rng.integers(3, 60))
        occ = (rng.random((W, H)) < float(rng.choice([.05, .2, .35, .5]))).astype(np.uint8)
        n = 100
        s = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        g = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        p.set_grid_occ(occ)
        P("grid", W, H, "planning")
        off, cells, cost, st = p.plan_batch(s, g, 1 + it % 2)
        oc, ol, ocost, _ = oracle.plan_batch(occ, s, g, 1 + it % 2, max_len=4096)
        bad = [q for q in range(n) if st[q] != ol[q] or (ol[q] > 0 and (not np.array_equal(cells[off[q]:off[q+1]], oc[q, :ol[q]]) or cost[q] != ocost[q]))]
        P("  bad", len(bad), bad[:5], p.timing())
        for q in bad[:2]:
            P("   q", q, s[q], g[q], "gpu", st[q], cost[q], cells[off[q]:off[q+1]].tolist()[:12], "ora", ol[q], ocost[q], oc[q, :max(ol[q],0)].tolist()[:12])
elif stage == "c2":
    occ = synth.synth_grid(1024, 1024, 1, 0.20)
    p.set_grid_occ(occ); P("grid set")
    nq = int(sys.argv[2])
    s, g = synth.synth_queries(occ, 1, nq)
    t = time.time(); off, cells, cost, st = p.plan_batch(s, g, 2, 1024); dt = time.time() - t
    P("gpu done", dt, nq / dt, p.timing())
    oc, ol, ocost, _ = oracle.plan_batch(occ, s, g, 2, max_len=1024, nthreads=8)
    bad = [q for q in range(nq) if st[q] != ol[q] or (ol[q] > 0 and (not np.array_equal(cells[off[q]:off[q+1]], oc[q, :ol[q]]) or cost[q] != ocost[q]))]
    P("bad", len(bad), bad[:10], "nopath", int((st == 0).sum()), "err", int((st < 0).sum()))
    for q in bad[:3]:
        P("   q", q, s[q], g[q], "gpu", st[q], cost[q], "ora", ol[q], ocost[q])
