"""Waypoint post-processing against vectors produced by executing the reference's own lines
(tests/golden/make_golden_waypoints.py: global_planner_st.py:292-327, global_planner_ccst.py:487-544 + :258-283).
Host functions of the C ABI: no GPU needed."""
import json
import os

import numpy as np
import pytest

import fuxi_planner_amd as fx
from fuxi_planner_amd import waypoints

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    with open(os.path.join(HERE, "golden", "waypoints.json")) as f:
        return json.load(f)


def grid_of(rec):
    bits = np.unpackbits(np.frombuffer(bytes.fromhex(rec["occ_bits"]), dtype=np.uint8))[:rec["W"] * rec["H"]]
    return bits.reshape(rec["W"], rec["H"]).astype(np.uint8)


def test_st_rule_matches_reference_lines():
    n = 0
    for rec in cases():
        if rec["variant"] != 0:
            continue
        wp, goal, ang = waypoints.select_st(rec["path"], rec["map_start"], rec["reso"], rec["origin"], rec["pos"], rec["goal"],
                                            rec["end_occu"], rec["prev_wp"])
        exp = rec["out"]
        assert wp.tolist() == exp["wp"], (n, wp, exp["wp"])           # bit-exact, including the number of components
        assert goal.tolist() == exp["goal_out"]
        assert ang == exp["ang_wp"]
        n += 1
    assert n == 150


def test_ccst_pruning_matches_reference_lines():
    n = pruned = held = 0
    for rec in cases():
        if rec["variant"] != 1:
            continue
        wp, kept, goal = waypoints.select_ccst(rec["path"], grid_of(rec), rec["reso"], rec["origin"], rec["pos"], rec["goal"],
                                               rec["end_occu"], return_goal=True)
        exp = rec["out"]
        assert kept.tolist() == exp["kept"], (n, kept.tolist(), exp["kept"])
        assert wp.tolist() == exp["wp"], (n, wp, exp["wp"])
        assert goal.tolist() == exp["goal_out"]
        pruned += len(exp["kept"]) < len(rec["path"])
        held += rec["end_occu"]
        n += 1
    assert n == 150 and pruned > 100 and held > 10  # end_occu = 1: the vehicle holds position (ccst:541-544)


def test_numpy_oracle_matches_reference_lines():
    """oracle/waypoints.py (the checker of the device kernel behind fxjps_waypoint_ccst_batch) against the same vectors."""
    from oracle import waypoints as ow
    n = 0
    for rec in cases():
        if rec["variant"] != 1:
            continue
        wp, kept, goal = ow.select_ccst(rec["path"], grid_of(rec).astype(np.float64), rec["reso"], rec["origin"], rec["pos"], rec["goal"],
                                        rec["end_occu"])
        exp = rec["out"]
        assert kept.tolist() == exp["kept"] and wp.tolist() == exp["wp"] and goal.tolist() == exp["goal_out"], n
        n += 1
    assert n == 150


def test_edge_cases():
    occ = np.zeros((10, 10), dtype=np.uint8)
    # a single-cell path (start == goal): nothing to prune, the waypoint is the goal
    wp, kept = waypoints.select_ccst([(3, 3)], occ, 0.2, (0.0, 0.0), (0.0, 0.0, 1.0), (5.0, 5.0, 1.5))
    assert wp.tolist() == [5.0, 5.0, 1.5] and kept.tolist() == [[3, 3]]
    wp, goal, ang = waypoints.select_st([(3, 3)], (4, 4), 0.2, (0.0, 0.0), (0.0, 0.0, 1.0), (5.0, 5.0, 1.5))
    assert wp.tolist() == [5.0, 5.0, 1.5] and ang == 0.0
    # vertical segment (equal x): the reference's slope is a division by zero it never uses
    occ[4, 5] = 1
    wp, kept = waypoints.select_ccst([(4, 0), (4, 4), (4, 9)], occ, 1.0, (0.0, 0.0), (100.0, 100.0, 0.0), (9.0, 9.0, 1.0))
    assert kept.tolist() == [[4, 0], [4, 9]]
    with pytest.raises(ValueError):
        waypoints.select_ccst([], occ, 1.0, (0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    with pytest.raises(fx.FxjpsError):
        waypoints.select_ccst([(-1, 2)], occ, 1.0, (0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))


def test_host_atan2_is_odd_in_its_first_argument():
    """The device form of the st rule (k_waypoint_st) looks atan2(dx, dy) of integer pairs up in a table the host fills for
    dx >= 0 only and negates for dx < 0: exact iff libm's atan2(-a, b) == -atan2(a, b) bit for bit.  Checked here on a dense
    range and on random large pairs (math.atan2 is the libm the reference's math.atan2 calls)."""
    import math
    import random
    import struct
    bits = lambda v: struct.pack("<d", v)
    for a in range(1, 400):
        for b in range(-400, 401):
            assert bits(math.atan2(-a, b)) == bits(-math.atan2(a, b)), (a, b)
    rnd = random.Random(5)
    for _ in range(200000):
        a, b = rnd.randint(1, 8192), rnd.randint(-8192, 8192)
        assert bits(math.atan2(-a, b)) == bits(-math.atan2(a, b)), (a, b)
    assert math.atan2(0, -3) == math.pi and math.atan2(0, 3) == 0.0 and math.atan2(0, 0) == 0.0
