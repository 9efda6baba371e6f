"""Replan throttle against the reference's own condition (tests/golden/make_golden_replan.py)."""
import json
import os

import numpy as np

from fuxi_planner_amd.replan import ReplanThrottle

HERE = os.path.dirname(os.path.abspath(__file__))


def test_throttle_matches_reference_condition():
    with open(os.path.join(HERE, "golden", "replan.json")) as f:
        cases = json.load(f)
    seen = set()
    for rec in cases:
        now = [rec["now"]]
        t = ReplanThrottle(clock=lambda: now[0])
        if not rec["fresh"]:
            now[0] = rec["last_time"]
            t.mark(rec["last_pos"])
            now[0] = rec["now"]
        assert t.due(rec["pos"]) == rec["due"], rec
        seen.add(rec["due"])
    assert seen == {True, False}


def test_mark_then_due():
    now = [10.0]
    t = ReplanThrottle(clock=lambda: now[0])
    assert t.due((1.0, 2.0, 1.0))                 # nothing recorded yet
    t.mark((1.0, 2.0, 1.0))
    assert not t.due((1.1, 2.0, 1.0))             # moved 0.1 m, 0 s later
    assert t.due((1.4, 2.0, 1.0))                 # moved 0.4 m
    now[0] = 10.31
    assert t.due((1.0, 2.0, 1.0))                 # 0.31 s later
    t.mark((0.0, 5.0, 1.0))
    now[0] = 10.32
    assert t.due((0.0, 5.0, 1.0))                 # the reference's "x == 0 means never searched"
    assert isinstance(t.last_pos, np.ndarray)
