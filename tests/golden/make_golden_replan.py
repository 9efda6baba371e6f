#!/usr/bin/env python3
"""Golden vectors for the replan throttle, produced by EVALUATING the reference's own condition:
the `elif (...)` expression of scripts/global_planner_ccst.py:476 is read from /root/reference at generation time
and eval'ed on prepared values of last_jps_pos, last_jps_time, (px, py, pz) and a stand-in clock.

    python tests/golden/make_golden_replan.py
"""
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def condition():
    with open("/root/reference/scripts/global_planner_ccst.py", encoding="utf-8", errors="replace") as f:
        line = f.readlines()[475]
    m = re.match(r"\s*elif\s*(\(.*\))\s*:\s*$", line)
    assert m, line
    return m.group(1)


class Clock(object):
    def __init__(self):
        self.now = 0.0

    def time(self):
        return self.now


def main():
    expr = compile(condition(), "ccst:476", "eval")
    rng = np.random.default_rng(476)
    out = []
    for it in range(200):
        clk = Clock()
        fresh = it % 5 == 0
        last_pos = np.array([0, 0, 0]) if fresh else np.array([float(rng.choice([0.0, 1.0, -2.5, 3.25])), float(rng.normal()), float(rng.uniform(0, 2))])
        last_time = 0 if fresh else float(rng.uniform(100, 200))
        clk.now = float(last_time + rng.choice([0.0, 0.1, 0.29, 0.3, 0.30000001, 0.5, 2.0])) if not fresh else float(rng.uniform(100, 200))
        step = float(rng.choice([0.0, 0.1, 0.29, 0.3, 0.31, 1.0]))
        d = rng.normal(size=3)
        d = d / np.linalg.norm(d) * step
        pos = [float(v) for v in (np.asarray(last_pos, dtype=np.float64) + d)]
        ns = {"np": np, "time": clk, "last_jps_pos": last_pos, "last_jps_time": last_time, "px": pos[0], "py": pos[1], "pz": pos[2]}
        out.append({"last_pos": [float(v) for v in last_pos], "fresh": fresh, "last_time": float(last_time), "now": clk.now, "pos": pos,
                    "due": bool(eval(expr, ns))})
    p = os.path.join(HERE, "replan.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", p, len(out), "cases,", sum(r["due"] for r in out), "due")


if __name__ == "__main__":
    main()
