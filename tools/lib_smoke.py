#!/usr/bin/env python3
"""Parity smoke of one library build (FXJPS_LIB) on a GPU box: config-2 queries on cell-indexed and on hashed tables, both
heuristics, a handful of small / sparse maps -- cells, lengths and float64 cost bytes against the C oracle.  Exit code 1 on
any mismatch.   tools/lib_smoke.py [queries = 2000]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
from oracle import oracle

def compare(p, occ, s, g, h, tag, max_len):
    off, cells, cost, st = p.plan_batch(s, g, h, max_len)
    oc, ol, ocost, _ = oracle.plan_batch(occ, s, g, h, literal=False, max_len=max_len, nthreads=16)
    bad = 0
    for q in range(len(s)):
        n = int(ol[q])
        if not (int(st[q]) == n and (n <= 0 or (np.array_equal(cells[off[q]:off[q + 1]], oc[q, :n]) and cost[q].tobytes() == ocost[q].tobytes()))):
            bad += 1
            if bad <= 3:
                print("  MISMATCH %s q=%d start=%s goal=%s len gpu %d oracle %d" % (tag, q, s[q], g[q], st[q], n))
    print("%-34s nq %6d bad %d" % (tag, len(s), bad), flush=True)
    return bad

def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    bad = 0
    occ = synth.synth_grid(1024, 1024, 1, 0.20)
    s, g = synth.synth_queries(occ, 1, 10000)
    idx = np.r_[0:nq - 16, [9206, 606, 5866, 1020, 9206, 606, 5866, 1020, 9206, 606, 5866, 1020, 9206, 606, 5866, 1020]]
    for direct in ("1", "0"):
        os.environ["FXJPS_DIRECT"] = direct
        p = fx.Planner([0])
        p.set_grid_occ(occ)
        for h in (2, 1):
            n = nq if h == 2 else nq // 4
            bad += compare(p, occ, s[idx[-n:]], g[idx[-n:]], h, "c2 direct=%s hchoice=%d" % (direct, h), 1024)
        rng = np.random.default_rng(11)
        for W, H, dens in [(37, 53, 0.3), (150, 110, 0.15), (300, 200, 0.0), (512, 512, 0.04)]:
            o2 = (rng.random((W, H)) < dens).astype(np.uint8)
            n = 400
            s2 = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
            g2 = np.stack([rng.integers(-1, W + 1, n), rng.integers(-1, H + 1, n)], 1).astype(np.int32)
            p.set_grid_occ(o2)
            bad += compare(p, o2, s2, g2, 2, "%dx%d p=%.2f direct=%s" % (W, H, dens, direct), 2048)
        del p
    print("TOTAL BAD %d" % bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
