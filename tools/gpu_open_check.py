#!/usr/bin/env python3
"""Differential check on large, nearly empty grids (long rays, thousands of equal keys: open-list splits, exact sorts,
x-sliced refills) against the CPU oracle.  Usage: python tools/gpu_open_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from oracle import oracle

rng = np.random.default_rng(5)
p = fx.Planner([0])
bad_total = 0
for W, H, dens, nq in ((1500, 1500, 0.0005, 1500), (2048, 1024, 0.002, 1500), (900, 2500, 0.01, 1500), (1200, 1200, 0.0, 600)):
    occ = (rng.random((W, H)) < dens).astype(np.uint8)
    free = np.argwhere(occ == 0)
    s = free[rng.integers(0, len(free), nq)].astype(np.int32)
    g = free[rng.integers(0, len(free), nq)].astype(np.int32)
    p.set_grid_occ(occ)
    for h in (1, 2):
        off, cells, cost, st = p.plan_batch(s, g, h)
        ml = max(int(st.max()), 1) + 8
        oc, ol, ocost, _ = oracle.plan_batch(occ, s, g, h, literal=False, max_len=ml, nthreads=64)
        bad = 0
        if not (np.array_equal(st, ol) and cost.tobytes() == ocost.tobytes()):
            bad = int((st != ol).sum()) + int((cost != ocost).sum())
        else:
            for q in range(nq):
                if not np.array_equal(cells[off[q]:off[q + 1]], oc[q, :max(int(ol[q]), 0)]):
                    bad += 1
        tm = p.timing()
        print("grid %dx%d density %.4f hchoice %d: %d queries, bad %d, kernel %.1f ms, pops %d, retried %d" % (W, H, dens, h, nq, bad, tm["search_kernel_ms"], tm["pops"], tm["retried"]), flush=True)
        bad_total += bad
print("TOTAL BAD", bad_total)
