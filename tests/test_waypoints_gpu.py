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


def test_st_rule_on_the_device_equals_the_host_form(planner, monkeypatch):
    """fxjps_waypoint_st_batch runs on the device (round 6: one wavefront per path, the angles out of a table the host's own
    atan2 filled).  Against the threaded host form (FXJPS_WAYPOINT_ST_HOST=1) on every path of a planned batch with last
    tick's waypoints of 0 / 2 / 3 components; against the one-path function on long paths built so that the loop ends at
    the 1st ... 300th point (the kernel takes 64 points per step and carries the angle over); with a map_start so far off
    the grid that the table would not fit (the call then takes the host form by itself); on a handle with three contexts."""
    import fuxi_planner_amd as fx
    from fuxi_planner_amd import synth, waypoints
    occ = synth.synth_grid(640, 512, 3, 0.2)
    nq = 6000
    s, g = synth.synth_queries(occ, 3, nq)
    planner.set_grid_occ(occ)
    off, cells, cost, st = planner.plan_batch(s, g, 2, 1024)
    rng = np.random.default_rng(3)
    reso, origin = 0.25, np.array([1.5, -40.0])
    pos = np.c_[(s[:, 0] + 1) * reso + origin[0] + rng.normal(0, 0.6, nq), (s[:, 1] + 1) * reso + origin[1] + rng.normal(0, 0.6, nq), rng.choice([0.0, 1.0], nq)]
    goal = np.c_[(g[:, 0] + 1) * reso + origin[0], (g[:, 1] + 1) * reso + origin[1], np.full(nq, 1.5)]
    eo = (rng.random(nq) < 0.1).astype(np.int32)
    ms = s + 1 + rng.integers(-3, 4, (nq, 2))
    prev = rng.uniform(-5, 200, (nq, 3))
    pdim = rng.choice([0, 2, 3], nq).astype(np.int32)
    for tre in ((2.0, np.pi / 4), (0.5, 0.05)):
        dev = waypoints.select_st_batch(planner, nq, ms, reso, origin, pos, goal, eo, prev, pdim, dis_wp_tre=tre[0], ang_wp_tre=tre[1])
        dev2 = waypoints.select_st_batch(planner, nq, ms, reso, origin, pos, goal, eo, prev, pdim, paths=(off, cells), dis_wp_tre=tre[0], ang_wp_tre=tre[1])
        monkeypatch.setenv("FXJPS_WAYPOINT_ST_HOST", "1")
        host = waypoints.select_st_batch(planner, nq, ms, reso, origin, pos, goal, eo, prev, pdim, dis_wp_tre=tre[0], ang_wp_tre=tre[1])
        monkeypatch.delenv("FXJPS_WAYPOINT_ST_HOST")
        for x, y, z in zip(dev, dev2, host):
            assert x.tobytes() == z.tobytes() and y.tobytes() == z.tobytes()
        assert len(set(dev[1].tolist())) == 2 and (dev[3] > 0).sum() > nq // 2  # (both kinds of waypoint, angles that are not trivial)
    # long paths: the angle to the goal's bearing grows point by point and turns back at point `turn`
    paths, mss = [], []
    for n, turn in ((300, 299), (300, 64), (300, 65), (300, 63), (300, 128), (300, 129), (129, 128), (65, 64), (64, 63), (2, 1), (1, 0), (200, 1), (300, 250)):
        R = 2500.0
        phi = np.linspace(0.0, 1.1, n)
        pts = np.stack([np.rint(R * np.sin(phi)), np.rint(R * np.cos(phi))], 1).astype(np.int32) + 7
        if 0 < turn < n - 1:  # (an angle no larger than the one in front of it)
            pts[turn] = pts[turn - 1] + np.array([0, 3] if turn > 1 else [0, 5], np.int32)
        pts[-1] = (7, 7 + 2400)  # the goal straight ahead
        paths.append(pts)
        mss.append((8, 8))
    off2 = np.zeros(len(paths) + 1, dtype=np.int64)
    off2[1:] = np.cumsum([len(p) for p in paths])
    cells2 = np.concatenate(paths).astype(np.int32)
    n2 = len(paths)
    pos2 = np.c_[rng.uniform(0, 10, n2), rng.uniform(0, 10, n2), np.zeros(n2)]
    goal2 = np.c_[rng.uniform(0, 400, n2), rng.uniform(0, 400, n2), np.ones(n2)]
    got = waypoints.select_st_batch(planner, n2, np.array(mss), 0.1, (0.0, 0.0), pos2, goal2, paths=(off2, cells2), dis_wp_tre=0.0, ang_wp_tre=0.0)
    kinds = set()
    for q, p in enumerate(paths):
        w1, g1, a1 = waypoints.select_st(p, mss[q], 0.1, (0.0, 0.0), pos2[q], goal2[q], 0, dis_wp_tre=0.0, ang_wp_tre=0.0)
        assert got[0][q, :got[1][q]].tobytes() == w1.tobytes() and got[2][q].tobytes() == g1.tobytes() and got[3][q] == a1, (q, got[3][q], a1)
        kinds.add(int(got[1][q]))
    assert kinds == {2, 3}
    # a map_start far off the grid: the table of angles would not fit, the host form answers
    far = np.tile(np.array([[3000000, -2000000]], np.int32), (n2, 1))
    got = waypoints.select_st_batch(planner, n2, far, 0.1, (0.0, 0.0), pos2, goal2, paths=(off2, cells2))
    for q, p in enumerate(paths):
        w1, g1, a1 = waypoints.select_st(p, far[q], 0.1, (0.0, 0.0), pos2[q], goal2[q], 0)
        assert got[0][q, :got[1][q]].tobytes() == w1.tobytes() and got[3][q] == a1, q
    # several contexts: each shard's paths stay where they were planned
    with fx.Planner([0, 0, 0]) as p3:
        p3.set_grid_occ(occ)
        res = p3.plan_batch(s[:901], g[:901], 2, 1024)
        a = waypoints.select_st_batch(p3, 901, ms[:901], reso, origin, pos[:901], goal[:901], eo[:901], prev[:901], pdim[:901])
        b = waypoints.select_st_batch(planner, 901, ms[:901], reso, origin, pos[:901], goal[:901], eo[:901], prev[:901], pdim[:901],
                                      paths=(res[0], res[1]))
        assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
