"""GPU suite (-m gpu): cell updates rebuild only what the changed cells can reach -- the neighbour bytes and
scan words of their box, the cell infos of the rows and columns through it -- and unite small updates into the component
labels instead of relabelling the map.  After every update the derived device maps are compared byte for byte with those
of a fresh upload of the same grid on a second handle; the component forest must keep together whatever the fresh labels
keep together (it may be coarser where an update split a component: those queries are searched, not answered at once);
plans on the updated handle equal the oracle's."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def roots(par):
    r = par.astype(np.int64).ravel().copy()
    idx = np.flatnonzero(r >= 0)
    while True:
        nxt = r[r[idx]]
        if np.array_equal(nxt, r[idx]):
            return r
        r[idx] = nxt


def check_against_fresh(p, q, cur, tag):
    q.set_grid_occ(cur)
    a, b = p.debug_maps(), q.debug_maps()
    for k in ("nb8", "bm", "ci", "dbm", "jd"):
        assert np.array_equal(a[k], b[k]), (tag, k, int((a[k] != b[k]).sum()))
    free = np.flatnonzero(cur.ravel() == 0)
    ra, rb = roots(a["comp"])[free], roots(b["comp"])[free]
    assert (ra >= 0).all() and (rb >= 0).all(), tag
    pairs = np.unique(np.stack([rb, ra]), axis=1)
    assert pairs.shape[1] == len(np.unique(rb)), (tag, "a fresh component is split over several roots")
    return len(np.unique(ra)), len(np.unique(rb))


@pytest.mark.parametrize("jd_mode", ["walk", "walk_max_3", "walk_max_3_no_list", "stream"])
def test_partial_rebuild_equals_fresh_upload(oracle, jd_mode, monkeypatch):
    """jd_mode: how the jump distances follow an update -- from the changed cells backwards (k_jd_walk; on the open maps
    below walks outgrow their bound and a wavefront of k_jd_finish carries each of them on), the same with walks of at
    most 3 steps (that hand-over on every map, nearly every walk), the same without room for a single handed-on walk (the
    records are streamed after all), every record read (round 4's k_update_jd)."""
    import fuxi_planner_amd as fx
    from fuxi_planner_amd import synth
    from test_gpu_parity import gpu_vs_oracle
    if jd_mode.startswith("walk_max_3"):
        monkeypatch.setenv("FXJPS_JD_WALK_MAX", "3")
        monkeypatch.setenv("FXJPS_JD_STREAM_DIV", "1")  # (walk however many cells changed: by default many of them means streaming up front)
        if jd_mode.endswith("no_list"):
            monkeypatch.setenv("FXJPS_JD_OVF_CAP", "0")
    elif jd_mode == "stream":
        monkeypatch.setenv("FXJPS_JD_WALK", "0")
    rng = np.random.default_rng(77)
    shapes = ((1024, 1024, 0.20), (700, 333, 0.3), (130, 2100, 0.15), (65, 64, 0.4), (600, 500, 0.0), (333, 800, 0.01), (512, 512, 0.04))
    with fx.Planner([0]) as p, fx.Planner([0]) as q:
        for (W, H, dens) in (shapes if jd_mode == "walk" else shapes[1:6]):
            # (the two largest maps: four rounds of the nine kinds instead of ten -- each step there reads 2 x 25 MB of maps back)
            n_steps = 36 if W * H >= 512 * 512 else 90
            cur = (rng.random((W, H)) < dens).astype(np.uint8)
            p.set_grid_occ(cur)
            s, g = synth.synth_queries(cur, 3, 300)
            coarse = 0
            for step in range(n_steps):
                kind = step % 9
                rebuild = True
                if kind in (0, 1):      # a sensor window, all its cells sent, fresh values
                    w = int(rng.integers(1, min(64, W, H) + 1))
                    x0, y0 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - w + 1))
                    xs, ys = np.meshgrid(np.arange(x0, x0 + w), np.arange(y0, y0 + w), indexing="ij")
                    xy = np.stack([xs.ravel(), ys.ravel()], 1)
                    val = (rng.random(len(xy)) < dens).astype(np.uint8)
                elif kind == 2:         # corners and borders
                    xy = np.array([[0, 0], [W - 1, H - 1], [0, H - 1], [W - 1, 0], [W // 2, 0], [0, H // 2]])
                    val = rng.integers(0, 2, len(xy)).astype(np.uint8)
                elif kind == 3:         # a wall across the map, then (next time) gone again: components split and merge
                    x = int(rng.integers(1, W - 1))
                    xy = np.stack([np.full(H, x), np.arange(H)], 1)
                    val = np.full(H, (step // 9) % 2 == 0, dtype=np.uint8)
                elif kind == 4:         # nothing changes
                    xy = np.stack([rng.integers(0, W, 40), rng.integers(0, H, 40)], 1)
                    val = cur[xy[:, 0], xy[:, 1]].copy()
                elif kind == 5:         # two deferred updates, the next one rebuilds
                    xy = np.unique(np.stack([rng.integers(0, W, 30), rng.integers(0, H, 30)], 1), axis=0)
                    val = rng.integers(0, 2, len(xy)).astype(np.uint8)
                    rebuild = False
                elif kind == 6:
                    xy = np.unique(np.stack([rng.integers(0, W, 30), rng.integers(0, H, 30)], 1), axis=0)
                    val = rng.integers(0, 2, len(xy)).astype(np.uint8)
                    rebuild = False
                elif kind == 7:         # a single cell
                    xy = np.array([[int(rng.integers(0, W)), int(rng.integers(0, H))]])
                    val = 1 - cur[xy[:, 0], xy[:, 1]]
                else:                   # large: more cells than the labels take incrementally (full relabelling)
                    k = min(W * H // 3, 20000)
                    idx = rng.choice(W * H, k, replace=False)
                    xy = np.stack([idx // H, idx % H], 1)
                    val = rng.integers(0, 2, k).astype(np.uint8)
                cur[xy[:, 0], xy[:, 1]] = val
                p.update_cells(xy.astype(np.int32), val.astype(np.uint8), rebuild=rebuild)
                if rebuild:
                    na, nb = check_against_fresh(p, q, cur, (W, H, step, kind))
                    coarse += na < nb
            gpu_vs_oracle(p, oracle, cur, s, g, 2)  # and the searches on the updated handle are the oracle's
            gpu_vs_oracle(p, oracle, cur, s, g, 1)
            print("partial rebuilds on %dx%d: %d of %d states with coarser labels than a fresh relabelling" % (W, H, coarse, n_steps * 7 // 9))
        # 70 small updates in a row: the 65th asks for the full relabelling
        cur = (rng.random((512, 512)) < 0.35).astype(np.uint8)
        p.set_grid_occ(cur)
        for step in range(70):
            xy = np.array([[int(rng.integers(0, 512)), int(rng.integers(0, 512))]])
            val = 1 - cur[xy[:, 0], xy[:, 1]]
            cur[xy[:, 0], xy[:, 1]] = val
            p.update_cells(xy.astype(np.int32), val.astype(np.uint8))
        na, nb = check_against_fresh(p, q, cur, "70 small")
        assert na <= nb


def test_fused_build_equals_separate_kernels(monkeypatch):
    """A small grid is built in four launches (k_build_1 .. 3 + k_derive_jd: kernels that need nothing of each other share
    a launch, the neighbour byte of the column words is computed instead of read); FXJPS_FUSED_BUILD=0 runs the eight separate
    kernels.  Neighbour bytes, straight and diagonal scan words, cell infos, jump distances byte for byte, and the same
    components, on the reference's 35 maps, the 256 x 256 canvas of config 1 and random grids up to the 2^18 cells the merged
    form takes -- built twice in a row each."""
    import json
    import fuxi_planner_amd as fx
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "maps_png.npz"))
    with open(os.path.join(root, "tests", "golden", "maps_png.json")) as f:
        recs = json.load(f)
    grids = []
    for nm in z.files:
        W, H = [r for r in recs if r["map"] == nm and "canvas" not in r][0]["shape"]
        grids.append(np.unpackbits(z[nm])[:W * H].reshape(W, H).astype(np.uint8))
    canvas = np.zeros((256, 256), np.uint8)
    canvas[:147, :112] = grids[[i for i, nm in enumerate(z.files) if nm == "-16.20-11.40_out.png"][0]]
    grids.append(canvas)
    rng = np.random.default_rng(9)
    for W, H, dens in ((1, 1, 0.0), (3, 70, 0.3), (65, 64, 0.4), (512, 512, 0.2), (400, 655, 0.05), (2100, 124, 0.35), (300, 300, 0.0), (64, 4096, 1.0)):
        grids.append((rng.random((W, H)) < dens).astype(np.uint8))
    monkeypatch.setenv("FXJPS_SETGRID_WAIT", "1")
    with fx.Planner([0]) as p, fx.Planner([0]) as q:
        for g in grids:
            for rep in range(2):
                monkeypatch.delenv("FXJPS_FUSED_BUILD", raising=False)
                p.set_grid_occ(g)
                monkeypatch.setenv("FXJPS_FUSED_BUILD", "0")
                q.set_grid_occ(g)
                a, b = p.debug_maps(), q.debug_maps()
                for k in ("nb8", "bm", "ci", "dbm", "jd"):
                    assert np.array_equal(a[k], b[k]), (g.shape, k, int((a[k] != b[k]).sum()))
                free = np.flatnonzero(g.ravel() == 0)
                ra, rb = roots(a["comp"])[free], roots(b["comp"])[free]
                assert (ra >= 0).all() and (rb >= 0).all()
                pairs = np.unique(np.stack([rb, ra]), axis=1)  # the same partition of the free cells
                assert pairs.shape[1] == len(np.unique(rb)) == len(np.unique(ra)), g.shape
                assert ((roots(a["comp"]) < 0) == (g.ravel() != 0)).all()


def test_window_update_on_a_large_map_is_cheap():
    """A 64 x 64 window re-observed on 4096 x 4096: the rebuild touches 66 lines, the rows / columns through them and the
    jump distances whose rays reach a changed cell (found from the changed cells backwards), not 16 M cells, their labels
    and their 268 MB of records (round 2: about 2 ms; round 4, every record read: 0.27 ms; round 5: 0.09 ms)."""
    import fuxi_planner_amd as fx
    from fuxi_planner_amd import synth
    occ = synth.synth_grid(4096, 4096, 2, 0.20)
    rng = np.random.default_rng(5)
    with fx.Planner([0]) as p:
        p.set_grid_occ(occ)
        ts = []
        for rep in range(40):
            x0, y0 = int(rng.integers(0, 4096 - 64)), int(rng.integers(0, 4096 - 64))
            xs, ys = np.meshgrid(np.arange(x0, x0 + 64), np.arange(y0, y0 + 64), indexing="ij")
            xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
            val = (rng.random(len(xy)) < 0.2).astype(np.uint8)
            t = time.perf_counter()
            p.update_cells(xy, val)
            ts.append(time.perf_counter() - t)
        ts = np.array(ts[8:]) * 1e3
        print("64x64 window update at 4096^2: median %.3f ms, max %.3f ms (host call, blocking)" % (np.median(ts), ts.max()))
        # A wall-clock bound inside the parity run: past it the test WARNS (a shared or down-clocked box is not a parity
        # failure and must not hide the tests behind it under `pytest -x`); it fails only at three times the bound.
        bound = 0.6 if os.environ.get("FXJPS_JD_WALK") == "0" else 0.16
        if np.median(ts) >= bound:
            import warnings
            warnings.warn("64 x 64 window update at 4096^2: median %.3f ms, expected below %.2f ms" % (np.median(ts), bound))
        assert np.median(ts) < 3 * bound


def test_odd_update_lists_and_several_contexts(oracle):
    """Cells outside the grid in the list (ignored), an empty list, a list that repeats the resident values, a deferred
    update followed by a new grid of another size (the leftover box must not reach into the new maps) -- on a handle with
    three contexts on the device (each applies the list and rebuilds its own maps): plans equal the oracle's and those
    of a one-context handle, the maps of the first context equal a fresh upload."""
    import fuxi_planner_amd as fx
    from fuxi_planner_amd import synth
    from test_gpu_parity import gpu_vs_oracle
    rng = np.random.default_rng(9)
    with fx.Planner([0, 0, 0]) as p3, fx.Planner([0]) as q:
        cur = synth.synth_grid(300, 200, 8, 0.25)
        s, g = synth.synth_queries(cur, 8, 400)
        p3.set_grid_occ(cur)
        xy = np.array([[-1, 5], [300, 5], [5, -1], [5, 200], [10, 10], [299, 199], [0, 0]], dtype=np.int32)
        val = np.array([1, 1, 1, 1, 1, 1, 1], dtype=np.uint8)
        ok = (xy[:, 0] >= 0) & (xy[:, 0] < 300) & (xy[:, 1] >= 0) & (xy[:, 1] < 200)
        cur[xy[ok, 0], xy[ok, 1]] = val[ok]
        p3.update_cells(xy, val)
        check_against_fresh(p3, q, cur, "outside cells")
        p3.update_cells(np.zeros((0, 2), np.int32), np.zeros(0, np.uint8))
        p3.update_cells(np.array([[-5, -5]], np.int32), np.array([1], np.uint8))       # nothing on the grid at all
        p3.update_cells(xy[ok], cur[xy[ok, 0], xy[ok, 1]])                               # the resident values again
        check_against_fresh(p3, q, cur, "no-op lists")
        gpu_vs_oracle(p3, oracle, cur, s, g, 2)
        # cells named several times with different values: the last entry of the list decides (`grid[xs, ys] = vals`)
        for rep in range(5):
            cells = np.stack([rng.integers(0, 300, 60), rng.integers(0, 200, 60)], 1)
            xy = cells[rng.integers(0, 60, 4000)].astype(np.int32)
            val = rng.integers(0, 2, 4000).astype(np.uint8)
            cur[xy[:, 0], xy[:, 1]] = val
            p3.update_cells(xy, val, rebuild=rep % 2 == 0)
        p3.update_cells(np.zeros((0, 2), np.int32), np.zeros(0, np.uint8))
        assert np.array_equal(p3.get_grid(), cur)
        check_against_fresh(p3, q, cur, "lists with duplicates")
        # a deferred update, then a new (smaller) grid: the old box is forgotten
        p3.update_cells(np.array([[290, 190]], np.int32), np.array([1], np.uint8), rebuild=False)
        cur = synth.synth_grid(90, 120, 3, 0.2)
        s, g = synth.synth_queries(cur, 3, 300)
        p3.set_grid_occ(cur)
        check_against_fresh(p3, q, cur, "new grid behind a deferred update")
        for step in range(6):
            k = int(rng.integers(1, 200))
            idx = rng.choice(90 * 120, k, replace=False)
            xy = np.stack([idx // 120, idx % 120], 1).astype(np.int32)
            val = rng.integers(0, 2, k).astype(np.uint8)
            cur[xy[:, 0], xy[:, 1]] = val
            p3.update_cells(xy, val, rebuild=step % 2 == 0)
        gpu_vs_oracle(p3, oracle, cur, s, g, 2)
        check_against_fresh(p3, q, cur, "after mixed updates")
