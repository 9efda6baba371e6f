#!/usr/bin/env python3
"""Golden vectors for the waypoint post-processing that follows a plan, produced by EXECUTING
the reference's own lines: scripts/global_planner_st.py:292-327 and scripts/global_planner_ccst.py:487-544 (with
`map_line_col`, ccst:258-283) are read from /root/reference at generation time, dedented and exec'ed on prepared
inputs.  Only inputs and outputs are stored.

    python tests/golden/make_golden_waypoints.py
"""
import contextlib
import io
import json
import math
import os
import sys
import textwrap
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/scripts"


def ref_lines(name, lo, hi):
    with open(os.path.join(REF, name), encoding="utf-8", errors="replace") as f:
        return "".join(f.readlines()[lo - 1:hi])


def planner_stub():
    """An object whose map_line_col is the reference's method (ccst:258-283)."""
    src = "class P(object):\n" + ref_lines("global_planner_ccst.py", 258, 283)
    ns = {"np": np}
    exec(compile(src, "map_line_col", "exec"), ns)
    return ns["P"]()


def run_st(path, map_start, reso, origin, pos, goal, end_occu, prev_wp):
    ns = {"np": np, "math": math, "path1": (list(map(tuple, path)), 0.0), "map_reso": reso, "map_o": np.array(origin),
          "map_start": np.array(map_start), "global_goal": np.array(goal), "px": pos[0], "py": pos[1], "pz": pos[2],
          "end_occu": end_occu, "dis_wp_tre": 2, "ang_wp_tre": math.pi / 4, "wp": None if prev_wp is None else np.array(prev_wp)}
    block = textwrap.dedent(ref_lines("global_planner_st.py", 292, 327))
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(block, "st:292-327", "exec"), ns)
    return {"wp": [float(v) for v in ns["wp"]], "goal_out": [float(v) for v in ns["global_goal"]], "ang_wp": float(ns["ang_wp"])}


def run_ccst(P, path, mapu, reso, origin, pos, goal, end_occu=0):
    ns = {"np": np, "math": math, "path1": (list(map(tuple, path)), 0.0), "map_reso": reso, "map_o": np.array(origin),
          "global_goal": np.array(goal), "px": pos[0], "py": pos[1], "pz": pos[2], "mapu": mapu, "planner": P, "end_occu": end_occu}
    block = textwrap.dedent(ref_lines("global_planner_ccst.py", 487, 544))
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec(compile(block, "ccst:487-544", "exec"), ns)
    return {"wp": [float(v) for v in ns["wp"]], "kept": [[int(c[0]), int(c[1])] for c in ns["path2_c"]],
            "goal_out": [float(v) for v in ns["global_goal"]]}


def main():
    from oracle import oracle
    rng = np.random.default_rng(20261003)
    P = planner_stub()
    out = []
    tries = 0
    while len(out) < 300 and tries < 5000:
        tries += 1
        W, H = int(rng.integers(8, 70)), int(rng.integers(8, 70))
        occ = (rng.random((W, H)) < float(rng.choice([0.0, 0.05, 0.15, 0.3]))).astype(np.uint8)
        free = np.argwhere(occ == 0)
        if len(free) < 2:
            continue
        s = free[rng.integers(0, len(free))]
        g = free[rng.integers(0, len(free))]
        cells, cost, _ = oracle.plan(occ, (int(s[0]), int(s[1])), (int(g[0]), int(g[1])), 2)
        if cells == 0:
            continue
        path = [[int(c[0]), int(c[1])] for c in cells]
        reso = float(rng.choice([0.1, 0.2, 0.25, 0.5]))
        origin = [float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))]
        near = rng.random() < 0.6  # the vehicle is usually at the start cell, sometimes anywhere
        base = (np.array(path[0]) + 1) * reso + np.array(origin)
        pos = [float(base[0] + rng.normal(0, 0.3)), float(base[1] + rng.normal(0, 0.3)), float(rng.choice([0.0, 0.5, 1.0, 1.5]))] if near else \
              [float(rng.uniform(-8, 20)), float(rng.uniform(-8, 20)), float(rng.uniform(0, 2))]
        goal = [float((g[0] + 1) * reso + origin[0]), float((g[1] + 1) * reso + origin[1]), float(rng.choice([1.0, 1.5, 2.0]))]
        rec = {"W": W, "H": H, "occ_bits": np.packbits(occ).tobytes().hex(), "path": path, "reso": reso, "origin": origin,
               "pos": pos, "goal": goal}
        if len(out) % 2 == 0:
            map_start = [path[0][0] + 1 + int(rng.integers(-1, 2)), path[0][1] + 1 + int(rng.integers(-1, 2))]
            end_occu = int(rng.random() < 0.15)
            prev = None if rng.random() < 0.5 else [float(rng.uniform(-5, 20)), float(rng.uniform(-5, 20))] + ([1.0] if rng.random() < 0.5 else [])
            rec.update({"variant": 0, "map_start": map_start, "end_occu": end_occu, "prev_wp": prev})
            rec["out"] = run_st(path, map_start, reso, origin, pos, goal, end_occu, prev)
        else:
            end_occu = int(rng.random() < 0.15)
            rec.update({"variant": 1, "end_occu": end_occu})
            rec["out"] = run_ccst(P, path, occ.astype(np.float64), reso, origin, pos, goal, end_occu)
        out.append(rec)
    p = os.path.join(HERE, "waypoints.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", p, len(out), "cases;", sum(1 for r in out if r["variant"] == 1 and len(r["out"]["kept"]) < len(r["path"])), "ccst cases pruned points")


if __name__ == "__main__":
    main()
