"""GPU suite (-m gpu): waypoint selection over whole batches -- the device kernel behind
fxjps_waypoint_ccst_batch and the threaded fxjps_waypoint_st_batch -- against the vectors produced by executing the
reference's own lines (tests/golden/waypoints.json), against the one-path host functions those vectors pin, and against
the numpy restatement oracle/waypoints.py on the paths of BASELINE config 2.  Everything bit for bit."""
import numpy as np
import pytest

from test_waypoints import cases, grid_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planner():
    import fuxi_planner_amd as fx
    p = fx.Planner([0])
    yield p
    p.close()


def test_batch_forms_on_the_reference_vectors(planner):
    from fuxi_planner_amd import waypoints
    n1 = n0 = 0
    for rec in cases():
        path = np.array(rec["path"], dtype=np.int32)
        off = np.array([0, len(path)], dtype=np.int64)
        exp = rec["out"]
        if rec["variant"] == 1:
            planner.set_grid_occ(grid_of(rec))
            wp, goal, nk, kept = waypoints.select_ccst_batch(planner, 1, rec["reso"], rec["origin"], rec["pos"], rec["goal"], [rec["end_occu"]],
                                                             paths=(off, path), return_kept=True)
            assert kept[:nk[0]].tolist() == exp["kept"], (n1, kept[:nk[0]].tolist(), exp["kept"])
            assert wp[0].tolist() == exp["wp"] and goal[0].tolist() == exp["goal_out"], n1
            n1 += 1
        else:
            prev = rec["prev_wp"]
            pw = None if prev is None else [prev + [0.0] * (3 - len(prev))]
            wp, dim, goal, ang = waypoints.select_st_batch(planner, 1, rec["map_start"], rec["reso"], rec["origin"], rec["pos"], rec["goal"],
                                                           [rec["end_occu"]], pw, None if prev is None else [len(prev)], paths=(off, path))
            assert wp[0, :dim[0]].tolist() == exp["wp"] and goal[0].tolist() == exp["goal_out"] and ang[0] == exp["ang_wp"], n0
            n0 += 1
    assert n1 == 150 and n0 == 150


def test_config2_paths_in_one_call(planner):
    """The 10 000 paths of BASELINE config 2, still resident on the device after plan_batch: one kernel launch prunes
    them all against the resident grid.  Every path against the one-path host function (pinned by the reference vectors),
    600 of them against the numpy restatement; the same through caller-supplied CSR paths; queries without a path get
    the goal; the st rule for the whole batch on host threads."""
    from fuxi_planner_amd import synth, waypoints
    from oracle import waypoints as ow
    occ = synth.synth_grid(1024, 1024, 1, 0.20)
    nq = 10000
    s, g = synth.synth_queries(occ, 1, nq)
    planner.set_grid_occ(occ)
    off, cells, cost, st = planner.plan_batch(s, g, 2, 1024)
    rng = np.random.default_rng(12)
    reso, origin = 0.2, np.array([-3.25, 7.5])
    pos = np.c_[(s[:, 0] + 1) * reso + origin[0] + rng.normal(0, 0.4, nq), s[:, 1] * reso + origin[1] + rng.normal(0, 0.4, nq), rng.choice([0.0, 0.5, 1.2], nq)]
    pos[::7] = np.c_[rng.uniform(-10, 200, len(pos[::7])), rng.uniform(-10, 200, len(pos[::7])), rng.uniform(0, 2, len(pos[::7]))]
    goal = np.c_[(g[:, 0] + 1) * reso + origin[0], g[:, 1] * reso + origin[1], np.full(nq, 1.5)]
    eo = (rng.random(nq) < 0.1).astype(np.int32)
    wp, gout, nk = waypoints.select_ccst_batch(planner, nq, reso, origin, pos, goal, eo)            # the resident paths
    wp2, gout2, nk2, kept = waypoints.select_ccst_batch(planner, nq, reso, origin, pos, goal, eo, paths=(off, cells), return_kept=True)
    assert wp.tobytes() == wp2.tobytes() and gout.tobytes() == gout2.tobytes() and np.array_equal(nk, nk2)
    nopath = np.flatnonzero(st <= 0)
    assert len(nopath) > 10 and np.array_equal(wp[nopath], goal[nopath]) and (nk[nopath] == 0).all()
    pruned = 0
    for q in np.flatnonzero(st > 0):
        p = cells[off[q]:off[q + 1]]
        w1, k1, g1 = waypoints.select_ccst(p, occ, reso, origin, pos[q], goal[q], int(eo[q]), return_goal=True)
        assert np.array_equal(kept[off[q]:off[q] + nk[q]], k1), q
        assert wp[q].tobytes() == w1.tobytes() and gout[q].tobytes() == g1.tobytes(), q
        pruned += len(k1) < len(p)
    assert pruned > 9000
    mapu = occ.astype(np.float64)
    for q in np.flatnonzero(st > 0)[::16][:600]:
        w1, k1, g1 = ow.select_ccst(cells[off[q]:off[q + 1]], mapu, reso, origin, pos[q], goal[q], int(eo[q]))
        assert np.array_equal(kept[off[q]:off[q] + nk[q]], k1) and wp[q].tobytes() == w1.tobytes() and gout[q].tobytes() == g1.tobytes(), q
    # the st rule, whole batch: resident paths == supplied paths == the one-path function
    ms = s + 1 + rng.integers(-1, 2, (nq, 2))
    a = waypoints.select_st_batch(planner, nq, ms, reso, origin, pos, goal, eo)
    b = waypoints.select_st_batch(planner, nq, ms, reso, origin, pos, goal, eo, paths=(off, cells), nthreads=3)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    for q in range(0, nq, 5):
        if st[q] <= 0:
            assert a[1][q] == 3 and np.array_equal(a[0][q], goal[q])
            continue
        w1, g1, ang1 = waypoints.select_st(cells[off[q]:off[q + 1]], ms[q], reso, origin, pos[q], goal[q], int(eo[q]))
        assert a[0][q, :a[1][q]].tobytes() == w1.tobytes() and a[2][q].tobytes() == g1.tobytes() and a[3][q] == ang1, q


def test_waypoints_follow_a_streaming_frame_and_shards(planner):
    """The batch form reads the paths where the last planning call left them: behind fxjps_replan_frame too, and shard by
    shard on a handle with several contexts."""
    import fuxi_planner_amd as fx
    from fuxi_planner_amd import synth, waypoints
    occ = synth.synth_grid(300, 260, 5, 0.2)
    s, g = synth.synth_queries(occ, 5, 901)
    pos = np.c_[s + 0.5, np.zeros(len(s))]
    goal = np.c_[g + 0.0, np.ones(len(s))]
    planner.set_grid_occ(occ)
    ref_res = planner.plan_batch(s, g, 2, 512)
    ref = waypoints.select_ccst_batch(planner, len(s), 1.0, (0.0, 0.0), pos, goal)
    with fx.Planner([0, 0, 0]) as p3:
        p3.set_grid_occ(occ)
        p3.set_queries(s, g, 2, 512)
        res = p3.replan_frame()
        assert all(np.array_equal(x, y) for x, y in zip(res, ref_res))
        got = waypoints.select_ccst_batch(p3, len(s), 1.0, (0.0, 0.0), pos, goal)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(got, ref))
        with pytest.raises(fx.FxjpsError):
            waypoints.select_ccst_batch(p3, len(s) - 1, 1.0, (0.0, 0.0), pos[1:], goal[1:])  # not the last batch's size


def test_batch_edges(planner):
    """Empty batch, paths of one, two and three points, every flag set, a path of several hundred collinear points (the
    kernel's 64-point windows): the batch kernel equals the one-path function."""
    from fuxi_planner_amd import waypoints
    occ = np.zeros((400, 40), dtype=np.uint8)
    occ[200, 5:35] = 1
    planner.set_grid_occ(occ)
    wp, g, nk = waypoints.select_ccst_batch(planner, 0, 1.0, (0.0, 0.0), np.zeros((0, 3)), np.zeros((0, 3)), paths=(np.zeros(1, np.int64), np.zeros((0, 2), np.int32)))
    assert wp.shape == (0, 3) and nk.shape == (0,)
    paths = [[(3, 3)], [(3, 3), (9, 9)], [(3, 3), (9, 9), (9, 20)], [(x, 2) for x in range(0, 399)], [(x, 2) for x in range(0, 190)] + [(190, 3), (199, 12), (199, 36), (201, 38), (230, 38)],
             [(0, 0), (1, 1)], []]
    off = np.zeros(len(paths) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(p) for p in paths])
    cells = np.array([c for p in paths for c in p], dtype=np.int32).reshape(-1, 2)
    n = len(paths)
    rng = np.random.default_rng(1)
    pos = np.c_[rng.uniform(0, 10, n), rng.uniform(0, 10, n), rng.uniform(0, 2, n)]
    goal = np.c_[rng.uniform(0, 400, n), rng.uniform(0, 40, n), np.ones(n)]
    for eo in (np.zeros(n, np.int32), np.ones(n, np.int32)):
        wp, gout, nk, kept = waypoints.select_ccst_batch(planner, n, 0.5, (1.0, -2.0), pos, goal, eo, paths=(off, cells), return_kept=True)
        for q, p in enumerate(paths):
            if not p:
                assert nk[q] == 0 and np.array_equal(wp[q], goal[q])
                continue
            w1, k1, g1 = waypoints.select_ccst(p, occ, 0.5, (1.0, -2.0), pos[q], goal[q], int(eo[q]), return_goal=True)
            assert np.array_equal(kept[off[q]:off[q] + nk[q]], k1) and wp[q].tobytes() == w1.tobytes() and gout[q].tobytes() == g1.tobytes(), q
