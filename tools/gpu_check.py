#!/usr/bin/env python3
"""Developer check on a GPU box: HIP planner vs the CPU oracle, verbose."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth
from oracle import oracle

def compare(p, occ, s, g, h, tag, max_len=None, verbose=3):
    t = time.time()
    off, cells, cost, st = p.plan_batch(s, g, h, max_len)
    tg = time.time() - t
    ml = max(int(st.max()), 1) + 8 if max_len is None else max_len
    t = time.time()
    oc, ol, ocost, _ = oracle.plan_batch(occ, s, g, h, literal=False, max_len=max(ml, 2048), nthreads=8)
    to = time.time() - t
    bad = 0
    for q in range(len(s)):
        n = int(ol[q])
        ok = int(st[q]) == n and (n <= 0 or (np.array_equal(cells[off[q]:off[q+1]], oc[q, :n]) and cost[q].tobytes() == ocost[q].tobytes()))
        if not ok:
            bad += 1
            if bad <= verbose:
                print("  MISMATCH q=%d start=%s goal=%s h=%d gpu_len=%d ora_len=%d gpu_cost=%r ora_cost=%r" % (q, s[q], g[q], h, st[q], n, cost[q], ocost[q]))
                if n > 0 and st[q] > 0:
                    a = cells[off[q]:off[q+1]]; b = oc[q, :n]
                    k = 0
                    while k < min(len(a), len(b)) and (a[k] == b[k]).all(): k += 1
                    print("    first diff at", k, a[max(0,k-1):k+3].tolist(), b[max(0,k-1):k+3].tolist())
    tm = p.timing()
    print("%-28s nq=%6d bad=%d gpu=%.3fs (kernel %.1f ms, pops %d, retried %d) oracle8=%.3fs" % (tag, len(s), bad, tg, tm["search_kernel_ms"], tm["pops"], tm["retried"], to))
    return bad

def main():
    p = fx.Planner([0])
    bad = 0
    # sqrt
    t = time.time()
    r = p.selftest_sqrt(0, 1 << 22)
    print("sqrt selftest 4M:", np.array_equal(r, np.sqrt(np.arange(1 << 22, dtype=np.float64))), time.time() - t)
    # known answers
    for rec in json.load(open(os.path.join(ROOT, "tests/golden/known_answers.json"))):
        W, H = rec["shape"]
        m = np.array(rec["grid"], dtype=np.float64).reshape(W, H)
        p.set_grid(m)
        try:
            path = p.plan(rec["start"], rec["goal"], rec["hchoice"])
        except IndexError:
            path = "IndexError"
        exp = [] if rec["path"] is None else [tuple(rec["path"][i:i+2]) for i in range(0, len(rec["path"]), 2)]
        okc = (not exp) or rec["start"] == rec["goal"] or float.fromhex(rec["cost_hex"]) == p.last_cost
        ok = path == exp and okc
        if not ok:
            bad += 1
            print("KNOWN MISMATCH", rec["name"], path, exp, p.last_cost)
    print("known answers done, bad so far", bad)
    # random small
    rng = np.random.default_rng(5)
    for it in range(40):
        W, H = int(rng.integers(3, 60)), int(rng.integers(3, 60))
        dens = float(rng.choice([.05, .2, .35, .5]))
        occ = (rng.random((W, H)) < dens).astype(np.uint8)
        n = 200
        s = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        g = np.stack([rng.integers(-1, W + 1, n), rng.integers(-1, H + 1, n)], 1).astype(np.int32)
        p.set_grid_occ(occ)
        bad += compare(p, occ, s, g, 1 + it % 2, "small %dx%d p=%.2f" % (W, H, dens), verbose=2 if bad < 6 else 0)
    # empty and sparse big maps (long rays)
    for W, H, dens in [(300, 200, 0.0), (700, 900, 0.01), (512, 512, 0.05)]:
        occ = (rng.random((W, H)) < dens).astype(np.uint8)
        n = 500
        s = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        g = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        p.set_grid_occ(occ)
        bad += compare(p, occ, s, g, 2, "sparse %dx%d p=%.2f" % (W, H, dens))
    # config 2
    occ = synth.synth_grid(1024, 1024, 1, 0.20)
    p.set_grid_occ(occ)
    for nq in (128, 2000):
        s, g = synth.synth_queries(occ, 1, nq)
        bad += compare(p, occ, s, g, 2, "C2 1024^2 nq=%d" % nq)
    nq = int(os.environ.get("FX_NQ", "10000"))
    s, g = synth.synth_queries(occ, 1, nq)
    for rep in range(2):
        t = time.time()
        off, cells, cost, st = p.plan_batch(s, g, 2, 1024)
        dt = time.time() - t
        tm = p.timing()
        print("C2 timing nq=%d: %.3f s -> %.0f plans/s (kernel %.1f ms, pops %d, pushes %d, retried %d) nopath=%d err=%d" % (
            nq, dt, nq / dt, tm["search_kernel_ms"], tm["pops"], tm["pushes"], tm["retried"], int((st == 0).sum()), int((st < 0).sum())))
    print("TOTAL BAD", bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
