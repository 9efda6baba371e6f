#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the real reference.

Runs only in the build container (needs /root/reference/scripts/jps1.py, which
never travels to the GPU box).  The outputs are data: inputs (grids as packed
bits / generator seeds, queries) and the reference's outputs (jump-point lists,
the cost it prints at jps1.py:207, and its operation counts observed through a
counting proxy around the grid and heapq).

    python tests/golden/make_golden.py [--only NAME] [--big]

Files written:
    known_answers.json   SURVEY.md section 8(c) table, re-captured
    random_small.json    400 random grids 3..40 cells a side, both hchoice
    maps_png.npz/.json   every reference maps/*.png (+ the 256x256 canvas case)
    synth1024.json       128 queries of the config-2 workload (1024^2, 20 %)
    synth4096.json       (--big) 4 queries of the config-3 workload
"""
import argparse
import contextlib
import glob
import hashlib
import heapq
import io
import json
import multiprocessing as mp
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/scripts")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import jps1  # noqa: E402  (the reference itself)
from oracle import oracle  # noqa: E402  (only for the synthetic-input generator)


class _Row:
    __slots__ = ("r", "c")

    def __init__(self, r, c):
        self.r = r
        self.c = c

    def __getitem__(self, j):
        self.c[0] += 1
        return self.r[j]


class CountingGrid:
    """matrix[x][y] proxy that counts element reads (jps1.py:20-36)."""

    def __init__(self, m):
        self.shape = m.shape
        self.c = [0]
        self.rows = [_Row(m[i], self.c) for i in range(m.shape[0])]

    def __getitem__(self, i):
        return self.rows[i]


def run_ref(m, s, g, h, count=True):
    """-> dict(path=[x0,y0,...] | None, printed=str, cost_hex, cells, pushes, pops)"""
    mm = CountingGrid(m) if count else m
    pushes, pops = [0], [0]

    class H:
        @staticmethod
        def heappush(q, e):
            pushes[0] += 1
            return heapq.heappush(q, e)

        @staticmethod
        def heappop(q):
            pops[0] += 1
            return heapq.heappop(q)

    buf = io.StringIO()
    jps1.heapq = H
    try:
        with contextlib.redirect_stdout(buf):
            r = jps1.method(mm, tuple(s), tuple(g), h)
    finally:
        jps1.heapq = heapq
    printed = buf.getvalue().strip()
    rec = {"start": [int(s[0]), int(s[1])], "goal": [int(g[0]), int(g[1])], "hchoice": int(h),
           "printed": printed}
    if isinstance(r[0], int):
        assert r[0] == 0 and printed == ""
        rec["path"] = None
    else:
        rec["path"] = [int(v) for c in r[0] for v in c]
        rec["cost_hex"] = float(printed).hex()
    if count:
        rec.update(cells=mm.c[0], pushes=pushes[0], pops=pops[0])
    return rec


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dump(name, obj):
    p = os.path.join(HERE, name)
    with open(p, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", p, os.path.getsize(p), "bytes")


# ---------------------------------------------------------------- known answers
def gen_known():
    out = []

    def case(name, m, s, g, h):
        rec = run_ref(m, s, g, h)
        rec.update(name=name, shape=list(m.shape), grid=[int(v) for v in m.astype(np.int64).ravel()])
        out.append(rec)

    e5 = np.zeros((5, 5))
    for h in (1, 2):
        case("5x5 empty corner to corner", e5, (0, 0), (4, 4), h)
        case("5x5 empty knight-ish", e5, (0, 0), (4, 2), h)
    case("start == goal", e5, (2, 2), (2, 2), 2)
    w7 = np.zeros((7, 7))
    w7[3, 1:6] = 1
    for h in (1, 2):
        case("7x7 wall", w7, (1, 3), (5, 3), h)
    m = e5.copy()
    m[4, 4] = 1
    case("goal occupied", m, (0, 0), (4, 4), 2)
    m = e5.copy()
    m[0, 0] = 1
    case("start occupied", m, (0, 0), (4, 4), 2)
    case("goal out of bounds", e5, (0, 0), (5, 5), 2)
    m = np.zeros((3, 3))
    m[0, 1] = m[1, 0] = 1
    case("squeeze forbidden", m, (0, 0), (2, 2), 2)
    m = np.zeros((3, 3))
    m[0, 1] = 1
    case("single corner cut allowed", m, (0, 0), (2, 2), 2)
    case("3x8 empty", np.zeros((3, 8)), (0, 0), (2, 7), 2)
    m = np.zeros((3, 3))
    m[1, 1] = 100
    case("value 100 is free", m, (0, 1), (2, 1), 2)
    m = np.zeros((3, 3))
    m[1, 1] = 1
    case("value 1 is an obstacle", m, (0, 1), (2, 1), 2)
    m = np.zeros((3, 3))
    m[1, 1] = 1
    case("start == goal on an obstacle", m, (1, 1), (1, 1), 2)
    dump("known_answers.json", out)


# ---------------------------------------------------------------- random small
def gen_random_small():
    rng = np.random.default_rng(20261003)
    out = []
    for it in range(400):
        W = int(rng.integers(3, 41))
        H = int(rng.integers(3, 41))
        p = float(rng.choice([.05, .2, .35, .5]))
        m = (rng.random((W, H)) < p).astype(np.float64)
        if it % 9 == 0:  # values other than 1 are free (jps1.py:20-29)
            m[m == 0] = float(rng.choice([100, 2, 0.5]))
        s = (int(rng.integers(0, W)), int(rng.integers(0, H)))
        if it % 13 == 0:  # goals just outside the grid
            g = (int(rng.integers(-1, W + 1)), int(rng.choice([-1, H])))
        else:
            g = (int(rng.integers(0, W)), int(rng.integers(0, H)))
        h = 1 + (it & 1)
        rec = run_ref(m, s, g, h)
        rec.update(shape=[W, H], grid_bits=np.packbits((m == 1).astype(np.uint8)).tobytes().hex())
        out.append(rec)
    dump("random_small.json", out)


# ---------------------------------------------------------------- maps/*.png
def load_png(path):
    """Loader convention of the caller, global_planner_st.py:176-182."""
    from PIL import Image
    img = Image.open(path).convert("L").point(lambda v: 0 if v > 200 else 1)
    return np.array(img)[::-1].T.astype(np.uint8)


def gen_maps():
    rng = np.random.default_rng(7)
    grids = {}
    recs = []
    for path in sorted(glob.glob("/root/reference/maps/*.png")):
        name = os.path.basename(path)
        occ = load_png(path)
        grids[name] = np.packbits(occ)
        W, H = occ.shape
        free = np.argwhere(occ == 0)
        qs = [(tuple(free[0]), tuple(free[-1]), 2), (tuple(free[-1]), tuple(free[0]), 1)]
        for k in range(3):
            a, b = rng.integers(0, len(free), 2)
            qs.append((tuple(free[a]), tuple(free[b]), 2 if k else 1))
        for s, g, h in qs:
            rec = run_ref(occ.astype(np.float64), s, g, h)
            rec.update(map=name, shape=[W, H], sha256=sha(occ))
            recs.append(rec)
    # BASELINE config 1: the 147x112 snapshot pasted into a zero 256x256 canvas
    # (the caller's own np.zeros + paste idiom, global_planner_st.py:248-249).
    name = "-16.20-11.40_out.png"
    occ = load_png("/root/reference/maps/" + name)
    canvas = np.zeros((256, 256), dtype=np.uint8)
    canvas[:occ.shape[0], :occ.shape[1]] = occ
    for s, g in [((0, 0), (146, 111)), ((2, 2), (140, 100)), ((5, 100), (140, 5)),
                 ((0, 0), (255, 255)), ((200, 30), (10, 100))]:
        rec = run_ref(canvas.astype(np.float64), s, g, 2)
        rec.update(map=name, canvas=[256, 256], shape=[256, 256], sha256=sha(canvas))
        recs.append(rec)
    np.savez_compressed(os.path.join(HERE, "maps_png.npz"), **grids)
    dump("maps_png.json", recs)


# ---------------------------------------------------------------- synthetic
_G = {}


def _synth_worker(args):
    W, seed, s, g, h, count = args
    if (W, seed) not in _G:
        _G.clear()
        _G[(W, seed)] = oracle.synth_grid(W, W, seed, 0.20).astype(np.float64)
    return run_ref(_G[(W, seed)], s, g, h, count=count)


def gen_synth(W, seed, qseed, nq, name, count, procs=8, h1_every=8):
    occ = oracle.synth_grid(W, W, seed, 0.20)
    s, g = oracle.synth_queries(occ, qseed, nq)
    jobs = [(W, seed, tuple(map(int, s[i])), tuple(map(int, g[i])),
             1 if (h1_every and i % h1_every == h1_every - 1) else 2, count) for i in range(nq)]
    with mp.Pool(procs) as pool:
        recs = pool.map(_synth_worker, jobs, chunksize=1)
    dump(name, {"W": W, "H": W, "grid_seed": seed, "p": 0.20, "qseed": qseed,
                "grid_sha256": sha(occ), "starts_sha256": sha(s), "goals_sha256": sha(g),
                "density": float(occ.mean()), "queries": recs})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    todo = {"known": gen_known, "random_small": gen_random_small, "maps": gen_maps,
            "synth1024": lambda: gen_synth(1024, 1, 1, 128, "synth1024.json", True)}
    if a.big:
        todo["synth4096"] = lambda: gen_synth(4096, 2, 2, 4, "synth4096.json", False, procs=4,
                                              h1_every=0)
    for k, fn in todo.items():
        if a.only is None or a.only == k:
            fn()
