"""GPU suite (-m gpu): the HIP planner, called through the C ABI, against the golden vectors
captured from the real jps1.py and against the CPU oracle on the same seeded inputs.
Integer/index outputs (jump-point cells, lengths) and the float64 path cost must be bit-exact."""
import contextlib
import io
import math

import numpy as np
import pytest

from conftest import grid_from_bits, load_golden, pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planner():
    import fuxi_planner_amd as fx
    p = fx.Planner([0])  # raises loudly if libfxjps.so or the GPU is missing
    yield p
    p.close()


def gpu_vs_oracle(planner, oracle, occ, s, g, h, max_len=None):
    off, cells, cost, st = planner.plan_batch(s, g, h, max_len)
    ml = max(int(st.max()) if len(st) else 1, 1) + 8
    oc, ol, ocost, _ = oracle.plan_batch(occ, s, g, h, literal=False, max_len=ml if max_len is None else max_len, nthreads=8)
    assert np.array_equal(st, ol)
    assert cost.tobytes() == ocost.tobytes()
    for q in range(len(st)):
        n = max(int(ol[q]), 0)
        assert np.array_equal(cells[off[q]:off[q + 1]], oc[q, :n]), "query %d" % q
    return off, cells, cost, st


def check_rec(planner, rec):
    off, cells, cost, st = planner.plan_batch([rec["start"]], [rec["goal"]], rec["hchoice"])
    if rec["path"] is None:
        assert st[0] == 0 and off[1] == 0
    else:
        assert [tuple(map(int, c)) for c in cells] == pairs(rec["path"])
        if rec["start"] != rec["goal"]:
            assert float(cost[0]).hex() == rec["cost_hex"]


# ------------------------------------------------------------------ arithmetic
def test_device_sqrt_is_correctly_rounded(planner):
    """Every argument the planner can pass to sqrt on a grid up to 4096 a side: n <= 2*4095^2."""
    hi = 2 * 4095 * 4095 + 1
    step = 1 << 23
    for n0 in range(0, hi, step):
        n1 = min(hi, n0 + step)
        got = planner.selftest_sqrt(n0, n1)
        assert np.array_equal(got, np.sqrt(np.arange(n0, n1, dtype=np.float64)))
    assert planner.selftest_sqrt(2, 3)[0] == math.sqrt(2)


def test_wavefront_minimum(planner):
    """The DPP reduction behind every open-list pop equals the shuffle form and a host minimum."""
    assert planner.selftest_wavemin(4096, 1) == 0
    assert planner.selftest_wavemin(1000, 99) == 0


def _openlist_script(planner, keys_f, keys_x, step_pops, step_off, expect_fail=False, **cfg):
    """Run the script on the device, replay it on a binary heap: every pop is the minimum, every entry comes out once."""
    import heapq
    kf = np.ascontiguousarray(keys_f, dtype=np.float64).view(np.uint64)
    kx = np.ascontiguousarray(keys_x, dtype=np.uint32)
    of, ox, os_, ok, info = planner.selftest_openlist(kf, kx, step_pops, step_off, **cfg)
    heap, pos, used = [], 0, np.zeros(len(kf), dtype=bool)

    def pop_check():
        f, x = heapq.heappop(heap)
        assert (int(of[pos]), int(ox[pos])) == (f, x), (pos, hex(int(of[pos])), hex(f), hex(int(ox[pos])), hex(x))
        sl = int(os_[pos])
        assert int(kf[sl]) == f and int(kx[sl]) == x and not used[sl], (pos, sl)  # the payload follows its key, once
        used[sl] = True

    for i in range(len(step_pops)):
        if info["fail"] and pos >= len(of):  # (the script stopped here: a far-tier region was full)
            break
        assert int(ok[i]) <= int(step_pops[i]) and (int(ok[i]) >= 1 or not heap or int(step_pops[i]) == 0), (i, int(ok[i]))
        for _ in range(int(ok[i])):
            pop_check()
            pos += 1
        for j in range(int(step_off[i]), int(step_off[i + 1])):
            heapq.heappush(heap, (int(kf[j]), int(kx[j])))
    if expect_fail:
        assert info["fail"] in (1, 2), info
        return info
    assert info["fail"] == 0, (info, len(heap), pos, len(kf))
    while pos < len(of):
        pop_check()
        pos += 1
    assert not heap and used.all() and len(of) == len(kf)
    return info


def _script(rng, n, key_fn, max_pop=16, max_push=24, bursts=True, grow=150):
    """Phases: `grow` steps that push more than they pop (the list reaches a few thousand entries), then steps that
    mostly pop until about as much has been asked for as was pushed (a step pops at most what the register tier holds,
    so the list never quite empties before the end)."""
    pops, off, kf, kx = [], [0], [], []
    t = 0
    while len(kf) < n:
        grown = 0
        for _ in range(grow):
            m = int(rng.integers(0, max_push + 1))
            if bursts and rng.random() < 0.02:
                m = 64
            m = min(m, n - len(kf))
            f, x = key_fn(t, m)
            kf.extend(f)
            kx.extend(x)
            off.append(len(kf))
            k = int(rng.integers(0, max_pop + 1))
            pops.append(k)
            grown += m - k
            t += 1
        while grown > 0 and len(kf) < n:
            m = min(int(rng.integers(0, 4)), n - len(kf))
            f, x = key_fn(t, m)
            kf.extend(f)
            kx.extend(x)
            off.append(len(kf))
            pops.append(64)
            grown -= 24  # (what such a step really pops, give or take)
            t += 1
    return np.array(kf, np.float64), np.array(kx, np.uint32), np.array(pops, np.uint32), np.array(off, np.uint32)


def test_open_list_alone(planner):
    """The three-tier open list of the search kernel, driven directly (fxjps_selftest_openlist): A*-like key streams,
    integer keys with hundreds of ties per level, adversarial orders, far tiers of a few hundred entries -- and what no
    map produces: thousands of entries with one and the same full key, which only the direct pop from the global tier
    can hand out (`slow_pops` asserted), with the lazily deleted near band in front of it."""
    rng = np.random.default_rng(31)

    def xy(m, lim=8192):
        return ((rng.integers(0, lim, m) << 17) | (rng.integers(0, lim, m) << 4) | rng.integers(0, 11, m)).astype(np.uint32)

    # 1. an A* frontier: keys a little above a slowly rising floor, more pushes than pops, then the drain
    def astar(t, m):
        return 100.0 + 0.01 * t + rng.random(m) * rng.choice([0.5, 5.0, 50.0]), xy(m)
    for cfg in ({"far_cap": 131072}, {"far_cap": 65536, "near_max": 48}, {"far_cap": 65536, "near_max": 8, "delta0": 0.05}):
        info = _openlist_script(planner, *_script(rng, 60000, astar), **cfg)
        assert info["far_refills"] > 50, info
    # 2. octile-like integer keys: few levels, hundreds of entries each (refills slice one level on (x, y))
    def octile(t, m):
        return (1000 + 10 * (t // 200) + 4 * rng.integers(0, 6, m)).astype(np.float64), xy(m)
    for cfg in ({"delta0": 20.0, "far_cap": 131072}, {"delta0": 20.0, "far_cap": 65536, "near_max": 32}):
        _openlist_script(planner, *_script(rng, 50000, octile), **cfg)
    # 3. no order at all, and keys far below everything popped so far
    def wild(t, m):
        return np.abs(rng.normal(0.0, 1.0, m)) * 10.0 ** rng.integers(-3, 6), xy(m)
    _openlist_script(planner, *_script(rng, 40000, wild), far_cap=131072, near_max=64)
    # 4. one and the same full key, thousands of times, among others: the tiers cannot split them
    def flood(t, m):
        f = 50.0 + rng.random(m)
        x = xy(m)
        same = rng.random(m) < 0.7
        f[same] = 50.5
        x[same] = (77 << 17) | (99 << 4) | 3
        return f, x
    info = _openlist_script(planner, *_script(rng, 20000, flood, max_pop=6), far_cap=65536, near_max=128)
    assert info["slow_pops"] > 1000, info
    info = _openlist_script(planner, *_script(rng, 6000, lambda t, m: (np.full(m, 7.0), np.full(m, 5 << 4, np.uint32)), max_pop=3), far_cap=32768)
    assert info["slow_pops"] > 1000, info
    # 5. the far band as a ring of f bands (large grids): set-up, whole bands into M, oversized bands, the last region
    #    dealt out again; cells below 64 x 64 (the ring reads cell infos from the map)
    def wide(t, m):
        return 10.0 + 0.05 * t + rng.random(m) * rng.choice([1.0, 20.0, 300.0]), xy(m, 64)
    for cfg in ({"far_cap": 262144, "near_max": 128}, {"far_cap": 131072, "near_max": 32, "delta0": 0.5}):
        info = _openlist_script(planner, *_script(rng, 80000, wide, grow=400), banded=True, **cfg)
        assert info["far_refills"] > 100, info
    # ... and regions that overflow: the failure is reported (the search re-runs such a query on the large pool),
    # what was popped until then is right
    _openlist_script(planner, *_script(rng, 60000, wide, max_pop=4, grow=2000), banded=True, far_cap=4096, near_max=16, expect_fail=True)
    _openlist_script(planner, *_script(rng, 30000, astar, max_pop=2, grow=2000), far_cap=256, near_max=16, expect_fail=True)


# ------------------------------------------------------------------ golden vectors from jps1.py
def test_known_answers_through_the_jps1_shim(planner):
    """The drop-in module: same return tuple, same printed cost, `path1[0] is 0` for no path."""
    from fuxi_planner_amd import jps1
    for rec in load_golden("known_answers.json"):
        grid = np.array(rec["grid"], dtype=np.float64).reshape(rec["shape"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            r = jps1.method(grid, tuple(rec["start"]), tuple(rec["goal"]), rec["hchoice"])
        assert isinstance(r, tuple) and len(r) == 2 and isinstance(r[1], float)
        if rec["path"] is None:
            assert r[0] is 0  # noqa: F632  (the callers' own idiom, global_planner_st.py:287)
            assert buf.getvalue() == ""
        else:
            assert r[0] == pairs(rec["path"]) and all(type(v) is int for c in r[0] for v in c)
            assert buf.getvalue().strip() == rec["printed"]
            assert (np.array(r[0]) + np.array([1, 1])).shape == (len(r[0]), 2)  # global_planner_st.py:292


def test_shim_error_behaviour(planner):
    from fuxi_planner_amd import jps1
    with pytest.raises(IndexError):
        jps1.method(np.zeros((5, 5)), (5, 5), (1, 1), 2)
    with pytest.raises(TypeError):
        jps1.method(np.zeros((5, 5)), (0, 0), (1, 1), 3)
    r = jps1.method(np.zeros((5, 5)), (np.int64(0), np.int64(0)), (np.int64(9), np.int64(9)), 2)
    assert r[0] is 0  # noqa: F632


def test_shim_on_a_map_that_changes_between_ticks(oracle):
    """The node calls jps1.method with a freshly built matrix every tick (global_planner_st.py:246-285).  The shim uploads it
    only when it differs from the resident grid: ticks with the same map, a changed cell, the map of another size, the
    first map again, a grid that something else made resident in between -- every answer is the oracle's on THAT tick's map."""
    from fuxi_planner_amd import jps1, default_planner
    rng = np.random.default_rng(21)
    a = (rng.random((147, 112)) < 0.25).astype(np.float64)
    b = a.copy()
    b[rng.integers(0, 147, 30), rng.integers(0, 112, 30)] = 1
    c = (rng.random((60, 90)) < 0.2).astype(np.float64)
    ticks = [a, a.copy(), np.where(a == 1, 1.0, 100.0), b, b, c, a, a]
    for t, m in enumerate(ticks):
        occ = (m == 1).astype(np.uint8)
        free = np.argwhere(occ == 0)
        k = rng.integers(0, len(free), 2)
        s, g = tuple(int(v) for v in free[k[0]]), tuple(int(v) for v in free[k[1]])
        if t == 7:  # (somebody else used the process-wide planner in between)
            default_planner().set_grid_occ(np.zeros((9, 9), np.uint8))
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            r = jps1.method(m, s, g, 2)
        want, cost, _ = oracle.plan(occ, s, g, 2, literal=False)
        assert (r[0] if r[0] else 0) == want, (t, s, g)
        if want:
            assert float(buf.getvalue()) == cost or s == g


def test_random_small_goldens(planner):
    for rec in load_golden("random_small.json"):
        planner.set_grid_occ(grid_from_bits(rec["grid_bits"], rec["shape"]))
        check_rec(planner, rec)


def test_reference_maps_goldens(planner, map_grids):
    cur = None
    for rec in load_golden("maps_png.json"):
        key = (rec["map"], "canvas" in rec)
        if key != cur:
            bits = np.unpackbits(map_grids[rec["map"]])
            if "canvas" in rec:
                occ = np.zeros((256, 256), dtype=np.uint8)
                occ[:147, :112] = bits[:147 * 112].reshape(147, 112)
            else:
                W, H = rec["shape"]
                occ = bits[:W * H].reshape(W, H)
            planner.set_grid_occ(occ)
            cur = key
        check_rec(planner, rec)


def test_synth1024_goldens(planner):
    from fuxi_planner_amd import synth
    g = load_golden("synth1024.json")
    occ = synth.synth_grid(g["W"], g["H"], g["grid_seed"], g["p"])
    planner.set_grid_occ(occ)
    recs = g["queries"]
    for h in (1, 2):
        sel = [r for r in recs if r["hchoice"] == h]
        s = np.array([r["start"] for r in sel], dtype=np.int32)
        t = np.array([r["goal"] for r in sel], dtype=np.int32)
        off, cells, cost, st = planner.plan_batch(s, t, h, 1024)
        for q, r in enumerate(sel):
            if r["path"] is None:
                assert st[q] == 0
            else:
                assert cells[off[q]:off[q + 1]].reshape(-1).tolist() == r["path"]
                assert float(cost[q]).hex() == r["cost_hex"]


# ------------------------------------------------------------------ seeded differential tests
@pytest.mark.parametrize("seed", range(6))
def test_random_grids_vs_oracle(planner, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for it in range(8):
        W, H = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        occ = (rng.random((W, H)) < float(rng.choice([0, .05, .2, .35, .5, .7]))).astype(np.uint8)
        n = 150
        s = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        g = np.stack([rng.integers(-1, W + 1, n), rng.integers(-1, H + 1, n)], 1).astype(np.int32)
        planner.set_grid_occ(occ)
        gpu_vs_oracle(planner, oracle, occ, s, g, 1 + (it + seed) % 2)


def test_open_and_sparse_maps_vs_oracle(planner, oracle):
    """Long rays (multi-word scans) and wide-open areas (many equal-f ties)."""
    rng = np.random.default_rng(77)
    for W, H, dens, h in [(300, 200, 0.0, 2), (200, 300, 0.0, 1), (700, 900, 0.01, 2), (512, 512, 0.05, 1), (1, 500, 0.1, 2),
                          (500, 1, 0.1, 2), (2, 2, 0.0, 2)]:
        occ = (rng.random((W, H)) < dens).astype(np.uint8)
        n = 300
        s = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        g = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int32)
        planner.set_grid_occ(occ)
        gpu_vs_oracle(planner, oracle, occ, s, g, h)


def test_structured_maps_vs_oracle(planner, oracle):
    """Rooms / corridors / mazes: long detours, dead ends and unreachable pockets."""
    W = H = 193
    occ = np.zeros((W, H), dtype=np.uint8)
    occ[::16, :] = 1
    occ[:, ::16] = 1
    rng = np.random.default_rng(3)
    for k in range(1, 12):  # doors
        for j in range(12):
            occ[16 * k, 16 * j + int(rng.integers(1, 16))] = 0
            occ[16 * j + int(rng.integers(1, 16)), 16 * k] = 0
    occ[100:110, 100:110] = 1
    occ[103:106, 103:106] = 0  # sealed pocket
    free = np.argwhere(occ == 0)
    n = 400
    s = free[rng.integers(0, len(free), n)].astype(np.int32)
    g = free[rng.integers(0, len(free), n)].astype(np.int32)
    g[:20] = [104, 104]  # into the sealed pocket: exhaustive (0, t) in the reference
    s[20:30] = [104, 105]  # out of it
    planner.set_grid_occ(occ)
    off, cells, cost, st = gpu_vs_oracle(planner, oracle, occ, s, g, 2)
    assert (st[:30] == 0).all() and (st > 0).any()


# ------------------------------------------------------------------ edge cases of the reference
def test_edge_cases(planner, oracle):
    from fuxi_planner_amd import _lib
    occ = np.zeros((6, 7), dtype=np.uint8)
    occ[2, 3] = 1
    planner.set_grid_occ(occ)
    off, cells, cost, st = planner.plan_batch(np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32))  # empty batch
    assert len(st) == 0 and off.tolist() == [0] and cells.shape == (0, 2)
    s = np.array([[0, 0], [6, 0], [-1, 3], [1, 1], [2, 3], [0, 0], [0, 0], [2, 3]], dtype=np.int32)
    g = np.array([[5, 6], [1, 1], [1, 1], [1, 1], [5, 6], [2, 3], [6, 7], [2, 3]], dtype=np.int32)
    off, cells, cost, st = planner.plan_batch(s, g, 2)
    assert st[1] == _lib.Q_BAD_START and st[2] == _lib.Q_BAD_START  # IndexError / wrap-around in the reference
    assert st[3] == 1 and cells[off[3]].tolist() == [1, 1] and cost[3] == 0.0  # start == goal
    assert st[5] == 0 and st[6] == 0  # goal occupied / out of bounds: (0, t)
    assert st[7] == 1  # start == goal on an obstacle still returns [start]
    keep = [0, 3, 4, 5, 6, 7]
    _, ol, ocost, _ = oracle.plan_batch(occ, s[keep], g[keep], 2, max_len=64)
    assert st[keep].tolist() == ol.tolist() and cost[keep].tobytes() == ocost.tobytes()
    # a path longer than the caller's slot is reported, not truncated
    off, cells, cost, st = planner.plan_batch(s[:1], g[:1], 2, max_path_len=1)
    assert st[0] == _lib.Q_PATH_TOO_LONG and off[1] == 0 and cost[0] == ocost[0]
    with pytest.raises(TypeError):
        planner.plan_batch(s, g, 0)


def test_a_path_longer_than_the_default_slot(planner, oracle):
    """64 x 64 cells, a serpentine of corridors with a notch above every other cell: 671 jump points, where the default slot
    of a path holds max(256, 4 x 64).  plan_batch with the default slot grows it and plans again; the one-query call
    (Planner.plan / jps1.method: their own arrays, the cells back with the call) hands over to it; an explicit slot that is
    too small is reported, never truncated."""
    from fuxi_planner_amd import _lib, jps1
    W = H = 64
    occ = np.ones((W, H), np.uint8)
    rows = list(range(1, H - 2, 3))
    for i, y in enumerate(rows):
        occ[1:W - 1, y] = 0
        occ[2:W - 2:2, y + 1] = 0  # the notches: a forced neighbour at every other cell of the corridor
        if i + 1 < len(rows):
            occ[W - 2 if i % 2 == 0 else 1, y:rows[i + 1] + 1] = 0
    s, g = (1, rows[0]), ((W - 2) if (len(rows) - 1) % 2 == 0 else 1, rows[-1])
    want, cost, _ = oracle.plan(occ, s, g, 2, literal=False)
    assert len(want) > max(256, 4 * W)
    planner.set_grid_occ(occ)
    assert planner.default_max_path_len() == 256
    off, cells, c, st = planner.plan_batch([s], [g], 2)
    assert st[0] == len(want) and [tuple(v) for v in cells.tolist()] == want and c[0] == cost
    assert planner.plan(s, g, 2) == want and planner.last_cost == cost
    st1, c1, cells1 = planner.plan_one(s, g, 2)
    assert st1 == len(want) and c1 == cost and [tuple(v) for v in np.asarray(cells1).tolist()] == want
    off, cells, c, st = planner.plan_batch([s], [g], 2, max_path_len=300)
    assert st[0] == _lib.Q_PATH_TOO_LONG and off[1] == 0 and c[0] == cost
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        r = jps1.method(occ.astype(np.float64), s, g, 2)
    assert r[0] == want and float(buf.getvalue()) == cost


def test_dense_and_csr_entry_points_agree(planner):
    import ctypes as C
    from fuxi_planner_amd import _lib, synth
    occ = synth.synth_grid(128, 96, 4, 0.22)
    planner.set_grid_occ(occ)
    s, g = synth.synth_queries(occ, 4, 333)
    off, cells, cost, st = planner.plan_batch(s, g, 2, 256)
    dense = np.full((333, 256, 2), -7, dtype=np.int32)
    ln = np.zeros(333, dtype=np.int32)
    co = np.zeros(333, dtype=np.float64)
    secs = C.c_double()
    rc = planner._L.fxjps_plan_batch(planner._h, _lib.ptr(s, C.c_int32), _lib.ptr(g, C.c_int32), 333, 2, 256,
                                     _lib.ptr(dense, C.c_int32), _lib.ptr(ln, C.c_int32), _lib.ptr(co, C.c_double), C.byref(secs))
    assert rc == 0 and np.array_equal(ln, st) and co.tobytes() == cost.tobytes() and secs.value > 0
    for q in range(333):
        n = max(int(st[q]), 0)
        assert np.array_equal(dense[q, :n], cells[off[q]:off[q + 1]]) and (dense[q, n:] == -7).all()


def test_update_cells_equals_fresh_upload(planner, oracle):
    from fuxi_planner_amd import synth
    occ = synth.synth_grid(200, 160, 8, 0.20)
    planner.set_grid_occ(occ)
    rng = np.random.default_rng(5)
    idx = rng.choice(200 * 160, 3000, replace=False)
    xy = np.stack([idx // 160, idx % 160], 1).astype(np.int32)
    val = (1 - occ[xy[:, 0], xy[:, 1]]).astype(np.uint8)
    planner.update_cells(xy, val)
    occ2 = occ.copy()
    occ2[xy[:, 0], xy[:, 1]] = val
    s, g = synth.synth_queries(occ2, 9, 400)
    gpu_vs_oracle(planner, oracle, occ2, s, g, 2)


def test_deferred_cell_updates(planner, oracle):
    """fxjps_update_cells_deferred: three updates in a row without rebuilding the maps (a cell may change in each of
    them: the calls are applied in order), then a plan -- which rebuilds them once -- equals a fresh upload; so does a
    streaming frame and a rebuilding update behind deferred ones."""
    from fuxi_planner_amd import synth
    occ = synth.synth_grid(200, 160, 8, 0.20)
    rng = np.random.default_rng(6)
    s, g = synth.synth_queries(occ, 9, 300)

    def some_updates(grid, k):
        out = []
        for _ in range(k):
            idx = rng.choice(200 * 160, 2500, replace=False)
            xy = np.stack([idx // 160, idx % 160], 1).astype(np.int32)
            val = rng.integers(0, 2, len(xy)).astype(np.uint8)
            grid[xy[:, 0], xy[:, 1]] = val
            out.append((xy, val))
        return out

    planner.set_grid_occ(occ)
    cur = occ.copy()
    for xy, val in some_updates(cur, 3):
        planner.update_cells(xy, val, rebuild=False)
    gpu_vs_oracle(planner, oracle, cur, s, g, 2)
    assert np.array_equal(planner.get_grid(), cur)
    ups = some_updates(cur, 3)  # deferred, deferred, then a rebuilding one
    planner.update_cells(*ups[0], rebuild=False)
    planner.update_cells(*ups[1], rebuild=False)
    planner.update_cells(*ups[2])
    gpu_vs_oracle(planner, oracle, cur, s, g, 1)
    ups = some_updates(cur, 3)  # deferred ones in front of a streaming frame
    planner.set_queries(s, g, 2, 512)
    planner.update_cells(*ups[0], rebuild=False)
    planner.update_cells(*ups[1], rebuild=False)
    off, cells, cost, st = planner.replan_frame(*ups[2])
    oc, ol, ocost, _ = oracle.plan_batch(cur, s, g, 2, literal=False, max_len=512)
    assert np.array_equal(st, ol) and cost.tobytes() == ocost.tobytes()
    for q in range(len(s)):
        assert np.array_equal(cells[off[q]:off[q + 1]], oc[q, :max(int(ol[q]), 0)])

    # readers of the grid are ordered behind queued updates (the stream is non-blocking: a null-stream copy is not),
    # and the update arrays are the library's the moment the call returns: temporaries, overwritten at once here
    big = synth.synth_grid(1024, 1024, 3, 0.20)
    planner.set_grid_occ(big)
    for rep in range(4):
        idx = rng.choice(1024 * 1024, 300000, replace=False)
        xy = np.stack([idx // 1024, idx % 1024], 1).astype(np.int32)
        val = rng.integers(0, 2, len(xy)).astype(np.uint8)
        big[xy[:, 0], xy[:, 1]] = val
        planner.update_cells(xy, val, rebuild=False)
        xy[:] = 0
        val[:] = 1 - big[0, 0]
        del xy, val
        assert np.array_equal(planner.get_grid(), big), rep
    nb = planner.debug_nbmask()  # (rebuilds the maps the deferred updates left stale)
    pad = np.ones((1026, 1026), dtype=np.uint8)
    pad[1:-1, 1:-1] = big
    assert np.array_equal(nb[1:-1, 1:-1] & 1, pad[0:-2, 0:-2])  # bit 0 = the (-1, -1) neighbour


def test_neighbour_mask_map(planner):
    """Derived map K2: bit k of nb8[x+1][y+1] is the occupancy of the k-th neighbour (border = occupied)."""
    rng = np.random.default_rng(2)
    occ = (rng.random((37, 70)) < 0.3).astype(np.uint8)
    planner.set_grid_occ(occ)
    nb = planner.debug_nbmask()
    pad = np.ones((39 + 2, 72 + 2), dtype=np.uint8)
    pad[2:-2, 2:-2] = occ
    exp = np.zeros((39, 72), dtype=np.uint8)
    k = 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dx or dy:
                exp |= pad[1 + dx:40 + dx, 1 + dy:73 + dy] << k
                k += 1
    assert np.array_equal(nb, exp)


# ------------------------------------------------------------------ BASELINE configurations at full size
def path_invariants(occ, s, g, off, cells, cost, st, hchoice=2):
    """Size-independent properties: waypoints are collinear along one of 8 directions over free cells,
    start/goal are the end points, and the float64 cost is the in-order sum of the segment lengths."""
    for q in np.flatnonzero(st > 0):
        p = cells[off[q]:off[q + 1]].astype(np.int64)
        assert tuple(p[0]) == tuple(s[q]) and tuple(p[-1]) == tuple(g[q])
        d = np.diff(p, axis=0)
        ad = np.abs(d)
        assert ((ad[:, 0] == ad[:, 1]) | (ad[:, 0] == 0) | (ad[:, 1] == 0)).all() and (ad.sum(1) > 0).all()
        acc = 0.0
        for dx, dy in d:
            acc = acc + (math.sqrt(dx * dx + dy * dy) if hchoice == 2 else float(max(abs(dx), abs(dy)) * (14 if dx and dy else 10)))
        assert acc == cost[q]
        for (x, y), (dx, dy) in zip(p[:-1], d):  # every intermediate cell is free
            n = max(abs(dx), abs(dy))
            xs = x + np.sign(dx) * np.arange(1, n + 1)
            ys = y + np.sign(dy) * np.arange(1, n + 1)
            assert (occ[xs, ys] == 0).all()


def test_config2_full_batch(planner, oracle):
    """BASELINE config 2: 1024x1024, 20 % obstacles, 10 000 queries -- every path against the oracle."""
    from fuxi_planner_amd import synth
    occ = synth.synth_grid(1024, 1024, 1, 0.20)
    planner.set_grid_occ(occ)
    s, g = synth.synth_queries(occ, 1, 10000)
    off, cells, cost, st = gpu_vs_oracle(planner, oracle, occ, s, g, 2, 1024)
    assert (st >= 0).all() and (st > 0).sum() > 9900
    sel = np.arange(0, 10000, 25)
    o2 = np.zeros(len(sel) + 1, dtype=np.int64)
    o2[1:] = np.cumsum(np.maximum(st[sel], 0))
    c2 = np.concatenate([cells[off[q]:off[q + 1]] for q in sel])
    path_invariants(occ, s[sel], g[sel], o2, c2, cost[sel], st[sel])
    # determinism: a second run returns byte-identical buffers
    off2, cells2, cost2, st2 = planner.plan_batch(s, g, 2, 1024)
    assert np.array_equal(off, off2) and np.array_equal(cells, cells2) and cost.tobytes() == cost2.tobytes()


def test_config3_sample(planner, oracle):
    """BASELINE config 3 grid (4096x4096): a sample the oracle finishes in seconds, plus invariants."""
    from fuxi_planner_amd import synth
    occ = synth.synth_grid(4096, 4096, 2, 0.20)
    planner.set_grid_occ(occ)
    s, g = synth.synth_queries(occ, 2, 96)
    off, cells, cost, st = gpu_vs_oracle(planner, oracle, occ, s, g, 2, 4096)
    path_invariants(occ, s, g, off, cells, cost, st)
