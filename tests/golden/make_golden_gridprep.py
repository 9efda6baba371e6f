#!/usr/bin/env python3
"""Golden vectors for the callers' grid preparation, produced by EXECUTING the reference's
own lines: scripts/global_planner_st.py:230-275 and scripts/global_planner_ccst.py:415-464 (through `end_occu`) are read from
/root/reference at generation time, dedented and exec'ed on prepared inputs.  Only inputs and outputs are stored.

    python tests/golden/make_golden_gridprep.py
"""
import json
import os
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scripts"
RANGES = {0: ("global_planner_st.py", 230, 275, 1), 1: ("global_planner_ccst.py", 415, 464, 2)}


def ref_block(variant):
    name, lo, hi, _ = RANGES[variant]
    with open(os.path.join(REF, name), encoding="utf-8", errors="replace") as f:
        lines = f.readlines()[lo - 1:hi]
    return textwrap.dedent("".join(lines))


def run_ref(variant, raw, start, goal, ifa):
    ns = {"np": np, "ifa": ifa, "mapu": raw.astype(np.float64).copy(), "map_goal": np.array(goal), "map_start": np.array(start),
          "map_o": [0.0, 0.0], "map_reso": 0.2, "map_c": raw.shape[0], "map_r": raw.shape[1]}
    exec(compile(ref_block(variant), RANGES[variant][0], "exec"), ns)
    g = ns["mapu"]
    assert set(np.unique(g)) <= {0.0, 1.0}
    return {"grid_shape": list(g.shape), "grid_bits": np.packbits((g == 1).astype(np.uint8)).tobytes().hex(),
            "start_out": [int(v) for v in ns["map_start"]], "goal_out": [int(v) for v in ns["map_goal"]],
            "map_d": [int(v) for v in ns["map_d"]], "end_occu": int(ns["end_occu"])}


def run_ref_callback(msg, width, height):
    """global_planner_st.py:16-20 (map_callback body) executed on a stand-in message object."""
    with open(os.path.join(REF, "global_planner_st.py"), encoding="utf-8", errors="replace") as f:
        block = textwrap.dedent("".join(f.readlines()[15:20]))

    class O(object):
        pass
    data, self_ = O(), O()
    data.info = O()
    data.info.height, data.info.width, data.data = height, width, list(int(v) for v in msg)
    exec(compile(block, "map_callback", "exec"), {"np": np, "data": data, "self": self_})
    return self_.map


def gen_msgs():
    rng = np.random.default_rng(99)
    out = []
    for it in range(40):
        w, h = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        msg = rng.choice(np.array([-1, 0, 0, 0, 100, 100, 1, 50, 99, -2], dtype=np.int8), size=w * h)
        m = run_ref_callback(msg, w, h)
        assert m.shape == (w, h)
        out.append({"width": w, "height": h, "data": [int(v) for v in msg], "map": [int(v) for v in m.ravel()]})
    p = os.path.join(HERE, "occupancy_msg.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", p, len(out), "cases")


def main():
    gen_msgs()
    rng = np.random.default_rng(424242)
    out = []
    for it in range(240):
        variant = it & 1
        ifa = [RANGES[variant][3], 1, 2, 3][it % 4]
        W0, H0 = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        raw = (rng.random((W0, H0)) < float(rng.choice([0.0, 0.03, 0.1, 0.3, 0.6]))).astype(np.uint8)
        lo = -6 if it % 3 == 0 else 0
        start = [int(rng.integers(lo, W0 + 6)), int(rng.integers(lo, H0 + 6))]
        goal = [int(rng.integers(lo, W0 + 6)), int(rng.integers(lo, H0 + 6))]
        try:
            rec = run_ref(variant, raw, start, goal, ifa)
        except (IndexError, ValueError):
            continue  # the reference itself fails on this input (goal row and column fully occupied, ...)
        rec.update(variant=variant, ifa=ifa, raw_shape=[W0, H0], raw_bits=np.packbits(raw).tobytes().hex(), start=start, goal=goal)
        out.append(rec)
    p = os.path.join(HERE, "gridprep.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", p, len(out), "cases", os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
