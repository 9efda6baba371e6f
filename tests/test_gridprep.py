"""Grid preparation: global_planner_st.py:230-275 / global_planner_ccst.py:415-464 (through end_occu).
Golden vectors come from executing those reference lines (tests/golden/make_golden_gridprep.py)."""
import numpy as np
import pytest

from conftest import load_golden


def unpack(bits_hex, shape):
    W, H = shape
    return np.unpackbits(np.frombuffer(bytes.fromhex(bits_hex), dtype=np.uint8))[:W * H].reshape(W, H)


def test_oracle_restatement_matches_reference_lines():
    from oracle import gridprep
    recs = load_golden("gridprep.json")
    assert len(recs) >= 200 and {r["variant"] for r in recs} == {0, 1}
    moved = 0
    eo = [0, 0]
    for r in recs:
        raw = unpack(r["raw_bits"], r["raw_shape"])
        g, s, go, d, e = gridprep.prepare_full(raw, r["start"], r["goal"], r["ifa"], r["variant"])
        assert list(g.shape) == r["grid_shape"] and np.array_equal(g, unpack(r["grid_bits"], r["grid_shape"]))
        assert list(s) == r["start_out"] and list(go) == r["goal_out"] and list(d) == r["map_d"] and e == r["end_occu"]
        sh = 1 if r["variant"] == 0 else 0
        moved += list(go) != [r["goal"][0] + d[0] - sh, r["goal"][1] + d[1] - sh]
        eo[r["variant"]] += e
    assert moved > 20  # the goal-on-obstacle relocation is exercised
    assert eo[0] > 10 and eo[1] > 10  # ... and so is end_occu = 1, for both nodes


@pytest.mark.gpu
def test_device_grid_preparation_matches_goldens():
    import fuxi_planner_amd as fx
    with fx.Planner([0]) as p:
        for r in load_golden("gridprep.json"):
            raw = unpack(r["raw_bits"], r["raw_shape"])
            s, g, d, shape, eo = p.prepare_grid(raw, r["start"], r["goal"], r["ifa"], r["variant"])
            assert list(shape) == r["grid_shape"] and list(s) == r["start_out"] and list(g) == r["goal_out"] and list(d) == r["map_d"]
            assert eo == r["end_occu"]
            assert np.array_equal(p.get_grid(), unpack(r["grid_bits"], r["grid_shape"]))


@pytest.mark.gpu
def test_prepare_then_plan_equals_host_prepared_grid(oracle):
    """The node's tick: prepare on the device, then plan on the resident grid -- same path as planning on the
    grid the host-side restatement prepares."""
    from oracle import gridprep
    import fuxi_planner_amd as fx
    rng = np.random.default_rng(9)
    with fx.Planner([0]) as p:
        for it in range(30):
            W0, H0 = int(rng.integers(20, 160)), int(rng.integers(20, 160))
            raw = (rng.random((W0, H0)) < 0.08).astype(np.uint8)
            start = (int(rng.integers(-3, W0)), int(rng.integers(-3, H0)))
            goal = (int(rng.integers(0, W0 + 3)), int(rng.integers(0, H0 + 3)))
            variant, ifa = it & 1, 1 + (it & 1)
            try:
                eg, es, ego, ed, eeo = gridprep.prepare_full(raw, start, goal, ifa, variant)
            except (IndexError, ValueError):
                continue
            s, g, d, shape, eo = p.prepare_grid(raw, start, goal, ifa, variant)
            assert (s, g, d, eo) == (es, ego, ed, eeo) and shape == eg.shape
            if not (0 <= s[0] < shape[0] and 0 <= s[1] < shape[1]):
                continue
            off, cells, cost, st = p.plan_batch([s], [g], 2)
            path, ocost, _ = oracle.plan(eg, s, g, 2, literal=False)
            if path == 0:
                assert st[0] == 0
            else:
                assert [tuple(map(int, c)) for c in cells] == path and cost[0] == ocost


def test_occupancy_message_goldens_are_self_consistent():
    """map_callback (global_planner_st.py:16-20): data[y*w + x] lands at map[x][y]; 100 -> 1, -1 -> 0, the rest kept."""
    for r in load_golden("occupancy_msg.json"):
        w, h = r["width"], r["height"]
        d = np.array(r["data"], dtype=np.int64).reshape(h, w).T
        m = np.array(r["map"], dtype=np.int64).reshape(w, h)
        exp = d.copy()
        exp[d == 100] = 1
        exp[d == -1] = 0
        assert np.array_equal(m, exp)


@pytest.mark.gpu
def test_prepare_from_occupancy_message():
    """The fused adapter equals map_callback followed by the grid preparation."""
    from oracle import gridprep
    import fuxi_planner_amd as fx
    with fx.Planner([0]) as p:
        for k, r in enumerate(load_golden("occupancy_msg.json")):
            w, h = r["width"], r["height"]
            m = np.array(r["map"], dtype=np.int64).reshape(w, h)  # what map_callback stores
            start, goal, ifa, variant = (0, 0), (w - 1, h - 1), 1 + (k & 1), k & 1
            try:
                eg, es, ego, ed, eeo = gridprep.prepare_full(m, start, goal, ifa, variant)
            except (IndexError, ValueError):
                continue
            s, g, d, shape, eo = p.prepare_occupancy_msg(np.array(r["data"], dtype=np.int8), w, h, start, goal, ifa, variant)
            assert (s, g, d, eo) == (es, ego, ed, eeo) and shape == eg.shape
            assert np.array_equal(p.get_grid(), eg)
