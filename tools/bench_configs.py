#!/usr/bin/env python3
"""Measurements for the BASELINE configurations that are not the bench.py headline (run on a GPU box).

  c3       4096x4096 grid (seed 2), N queries (default 20000; BASELINE asks for 100k)
  c4shard  1024x1024, 125 000 queries = one GPU's share of config 4
  c5       streaming replan: 1024x1024, 10 % of the cells toggled per frame, 1000 persistent queries
  h1       config 2 with hchoice = 1 (octile x10/x14: many equal keys)
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth

def P(*a): print(*a, flush=True)

def c3(p, nq=20000):
    occ = synth.synth_grid(4096, 4096, 2, 0.20); p.set_grid_occ(occ)
    s, g = synth.synth_queries(occ, 2, nq)
    t = time.time(); off, cells, cost, st = p.plan_batch(s, g, 2, 4096); dt = time.time() - t
    tm = p.timing()
    P("c3: 4096^2, %d queries: %.2f s wall, kernel %.1f ms -> %.0f plans/s (kernel), pops/query %.0f, no path %d, errors %d" % (
        nq, dt, tm["search_kernel_ms"], nq / (tm["search_kernel_ms"] * 1e-3), tm["pops"] / nq, int((st == 0).sum()), int((st < 0).sum())))

def c4shard(p, nq=125000):
    occ = synth.synth_grid(1024, 1024, 1, 0.20); p.set_grid_occ(occ)
    s, g = synth.synth_queries(occ, 1, nq)
    for rep in range(2):
        t = time.time(); off, cells, cost, st = p.plan_batch(s, g, 2, 1024); dt = time.time() - t
    P("c4shard: 1024^2, %d queries: %.3f s -> %.0f plans/s (kernel %.1f ms)" % (nq, dt, nq / dt, p.timing()["search_kernel_ms"]))

def h1(p, nq=10000):
    occ = synth.synth_grid(1024, 1024, 1, 0.20); p.set_grid_occ(occ)
    s, g = synth.synth_queries(occ, 1, nq)
    for rep in range(2):
        t = time.time(); off, cells, cost, st = p.plan_batch(s, g, 1, 1024); dt = time.time() - t
    P("h1: config 2 with hchoice=1: %.3f s -> %.0f plans/s (kernel %.1f ms, pops %d)" % (nq, nq / dt, p.timing()["search_kernel_ms"], p.timing()["pops"]) if False else
      "h1: config 2 with hchoice=1: %.3f s -> %.0f plans/s (kernel %.1f ms, pops %d)" % (dt, nq / dt, p.timing()["search_kernel_ms"], p.timing()["pops"]))

def c5(p, frames=60, nq=1000):
    W = H = 1024
    occ = synth.synth_grid(W, H, 1, 0.20)
    s, g = synth.synth_queries(occ, 5, nq)
    keep = np.zeros((W, H), dtype=bool); keep[s[:, 0], s[:, 1]] = True; keep[g[:, 0], g[:, 1]] = True
    p.set_grid_occ(occ)
    p.plan_batch(s, g, 2, 2048)
    k = int(0.05 * W * H)
    t_upd = t_plan = 0.0; reach = 0
    t0 = time.time()
    for f in range(frames):
        rng = np.random.default_rng(1000 + f)
        on = np.flatnonzero((occ == 1).ravel()); off_ = np.flatnonzero(((occ == 0) & ~keep).ravel())
        a = rng.choice(on, k, replace=False); b = rng.choice(off_, k, replace=False)
        idx = np.concatenate([a, b]); val = np.concatenate([np.zeros(k, np.uint8), np.ones(k, np.uint8)])
        xy = np.stack([idx // H, idx % H], 1).astype(np.int32)
        occ.ravel()[a] = 0; occ.ravel()[b] = 1
        t = time.time(); p.update_cells(xy, val); t_upd += time.time() - t
        t = time.time(); off, cells, cost, st = p.plan_batch(s, g, 2, 2048); t_plan += time.time() - t
        reach += int((st > 0).sum())
    dt = time.time() - t0
    P("c5: %d frames, %d cells toggled/frame, %d queries/frame: update+maps %.2f ms/frame, plan %.1f ms/frame, host toggle gen %.1f ms/frame "
      "-> %.1f frames/s device-side (%.0f plans/s), reachable %.1f %%" % (frames, 2 * k, nq, 1e3 * t_upd / frames, 1e3 * t_plan / frames,
      1e3 * (dt - t_upd - t_plan) / frames, frames / (t_upd + t_plan), frames * nq / (t_upd + t_plan), 100.0 * reach / (frames * nq)))

if __name__ == "__main__":
    p = fx.Planner([0])
    which = sys.argv[1:] or ["h1", "c4shard", "c5", "c3"]
    for w in which:
        {"c3": c3, "c4shard": c4shard, "c5": c5, "h1": h1}[w](p)
