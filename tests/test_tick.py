"""One planner tick end to end: grid preparation on the device (N1), the search (hot path), waypoint
selection (N2) -- against vectors produced by running the reference's own lines and the real jps1.py in sequence
(tests/golden/make_golden_tick.py)."""
import numpy as np
import pytest

from conftest import load_golden


def unpack(bits_hex, shape):
    W, H = shape
    return np.unpackbits(np.frombuffer(bytes.fromhex(bits_hex), dtype=np.uint8))[:W * H].reshape(W, H)


@pytest.mark.gpu
def test_tick_matches_reference_sequence():
    import fuxi_planner_amd as fx
    from fuxi_planner_amd import waypoints
    recs = load_golden("tick.json")
    assert len(recs) >= 100 and {r["variant"] for r in recs} == {0, 1}
    planned = inter = held = 0
    with fx.Planner([0]) as p:
        for r in recs:
            raw = unpack(r["raw_bits"], r["raw_shape"])
            s, g, d, shape, end_occu = p.prepare_grid(raw, r["start"], r["goal"], r["ifa"], r["variant"])  # N1
            assert list(s) == r["map_start"] and list(g) == r["map_goal"] and list(shape) == r["grid_shape"]
            assert end_occu == r["end_occu"]
            origin = fx.Planner.shifted_origin(r["origin"], d, r["reso"])
            path = p.plan(s, g, 2)                                                                        # hot path
            if r["path"] is None:
                assert path == []
                continue
            assert [list(c) for c in path] == r["path"]
            planned += 1
            if r["variant"] == 0:                                                                          # N2
                wp, goal_out, _ = waypoints.select_st(path, s, r["reso"], origin, r["pos"], r["goal3"], end_occu, r["prev_wp"])
            else:
                wp, _, goal_out = waypoints.select_ccst(path, p.get_grid(), r["reso"], origin, r["pos"], r["goal3"], end_occu,
                                                        return_goal=True)
            assert goal_out.tolist() == r["goal_out"]
            assert wp.tolist() == r["wp"], (r["variant"], wp, r["wp"])
            inter += r["wp"][:2] != r["goal3"][:2]
            held += end_occu
    assert planned > 50 and inter > 20 and held > 15  # ticks with a relocated goal (end_occu = 1) included


def test_tick_glue_on_host():
    """The same sequence without a device: prepared grid from the numpy restatement, path from the vectors -- checks
    the conventions between the stages (shifted start, shifted origin, the +1 of the path cells)."""
    import fuxi_planner_amd as fx
    from fuxi_planner_amd import waypoints
    from oracle import gridprep
    n = 0
    for r in load_golden("tick.json"):
        if r["path"] is None:
            continue
        raw = unpack(r["raw_bits"], r["raw_shape"])
        grid, s, g, d, eo = gridprep.prepare_full(raw, r["start"], r["goal"], r["ifa"], r["variant"])
        assert list(s) == r["map_start"] and list(g) == r["map_goal"] and eo == r["end_occu"]
        origin = fx.Planner.shifted_origin(r["origin"], d, r["reso"])
        if r["variant"] == 0:
            wp, goal_out, _ = waypoints.select_st(r["path"], s, r["reso"], origin, r["pos"], r["goal3"], eo, r["prev_wp"])
        else:
            wp, _, goal_out = waypoints.select_ccst(r["path"], grid, r["reso"], origin, r["pos"], r["goal3"], eo, return_goal=True)
        assert goal_out.tolist() == r["goal_out"]
        assert wp.tolist() == r["wp"]
        n += 1
    assert n > 50
