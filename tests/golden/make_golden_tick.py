#!/usr/bin/env python3
"""End-to-end golden vectors of one planner tick (N1 -> hot path -> N2), produced by EXECUTING the
reference: the grid-preparation lines (global_planner_st.py:230-275 / global_planner_ccst.py:415-464, through end_occu), the real
scripts/jps1.py, and the waypoint lines (global_planner_st.py:292-327 / global_planner_ccst.py:487-544 with
map_line_col) are read from /root/reference at generation time and run on prepared inputs.  Only inputs and outputs
are stored.

    python tests/golden/make_golden_tick.py
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
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/scripts")
import jps1  # noqa: E402  (the reference's module)
from make_golden_gridprep import RANGES, ref_block  # noqa: E402
from make_golden_waypoints import planner_stub, ref_lines  # noqa: E402


def tick(P, variant, raw, start, goal, ifa, reso, origin, pos, goal3, prev_wp):
    ns = {"np": np, "math": math, "ifa": ifa, "mapu": raw.astype(np.float64).copy(), "map_goal": np.array(goal),
          "map_start": np.array(start), "map_o": np.array(origin), "map_reso": reso, "map_c": raw.shape[0], "map_r": raw.shape[1],
          "global_goal": np.array(goal3), "px": pos[0], "py": pos[1], "pz": pos[2], "planner": P,
          "dis_wp_tre": 2, "ang_wp_tre": math.pi / 4, "wp": None if prev_wp is None else np.array(prev_wp), "end_occu": 0}
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec(compile(ref_block(variant), RANGES[variant][0], "exec"), ns)                       # N1
        ms, mg = ns["map_start"], ns["map_goal"]
        if ms[0] > ns["map_c"] or ms[1] > ns["map_r"]:
            return None  # the node does not plan on this tick (st:282 / ccst:471)
        path1 = jps1.method(ns["mapu"], tuple(int(v) for v in ms), tuple(int(v) for v in mg), 2)  # the hot path
        ns["path1"] = path1
        rec = {"map_start": [int(v) for v in ms], "map_goal": [int(v) for v in mg], "grid_shape": list(ns["mapu"].shape)}
        if isinstance(path1[0], int):  # (0, t): no path
            rec.update(path=None, wp=[float(v) for v in goal3], end_occu=int(ns["end_occu"]))
            return rec
        if variant == 0:
            block = textwrap.dedent(ref_lines("global_planner_st.py", 292, 327))                 # N2
        else:
            block = textwrap.dedent(ref_lines("global_planner_ccst.py", 487, 544))
        exec(compile(block, "N2", "exec"), ns)
    rec.update(path=[[int(x), int(y)] for x, y in path1[0]], wp=[float(v) for v in ns["wp"]],
               goal_out=[float(v) for v in ns["global_goal"]], end_occu=int(ns["end_occu"]))
    return rec


def main():
    rng = np.random.default_rng(777)
    P = planner_stub()
    out = []
    tries = 0
    while len(out) < 120 and tries < 3000:
        tries += 1
        variant = len(out) & 1
        ifa = [RANGES[variant][3], 1, 2][tries % 3]
        W0, H0 = int(rng.integers(10, 60)), int(rng.integers(10, 60))
        raw = (rng.random((W0, H0)) < float(rng.choice([0.0, 0.04, 0.1, 0.2]))).astype(np.uint8)
        lo = -4 if tries % 4 == 0 else 0
        start = [int(rng.integers(lo, W0 + 4)), int(rng.integers(lo, H0 + 4))]
        goal = [int(rng.integers(lo, W0 + 4)), int(rng.integers(lo, H0 + 4))]
        occ_cells = np.argwhere(raw == 1)
        if tries % 3 == 1 and len(occ_cells):  # a goal on (or next to) an obstacle: end_occu = 1 ticks
            c = occ_cells[int(rng.integers(0, len(occ_cells)))]
            goal = [int(c[0]) + int(rng.integers(0, 2)), int(c[1]) + int(rng.integers(0, 2))]
        reso = float(rng.choice([0.1, 0.2, 0.5]))
        origin = [float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))]
        pos = [float(start[0] * reso + origin[0] + rng.normal(0, 0.2)), float(start[1] * reso + origin[1] + rng.normal(0, 0.2)), float(rng.choice([0.5, 1.0]))]
        goal3 = [float(goal[0] * reso + origin[0]), float(goal[1] * reso + origin[1]), 1.5]
        prev = None if rng.random() < 0.6 else [float(rng.uniform(-3, 10)), float(rng.uniform(-3, 10)), 1.0]
        try:
            rec = tick(P, variant, raw, start, goal, ifa, reso, origin, pos, goal3, prev)
        except (IndexError, ValueError):
            continue  # the reference itself fails on this input
        if rec is None:
            continue
        rec.update(variant=variant, ifa=ifa, raw_shape=[W0, H0], raw_bits=np.packbits(raw).tobytes().hex(), start=start, goal=goal,
                   reso=reso, origin=origin, pos=pos, goal3=goal3, prev_wp=prev)
        out.append(rec)
    p = os.path.join(HERE, "tick.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", p, len(out), "ticks;", sum(r["end_occu"] for r in out), "with end_occu = 1;", sum(r["path"] is None for r in out), "without a path;",
          sum(r["path"] is not None and r["wp"][:2] != r["goal3"][:2] for r in out), "with an intermediate waypoint")


if __name__ == "__main__":
    main()
