"""Replan throttle against the reference's own condition (tests/golden/make_golden_replan.py)."""
import json
import os

import numpy as np
import pytest

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


def test_frame_pipeline_hands_every_handle_every_update_in_order(monkeypatch):
    """FramePipeline (host logic, no GPU): with schedule "turn" frame f goes to handle f % K, with "free" to an idle one;
    either way the handle first applies the updates of the frames the other handles planned, oldest first -- so that the
    grid a frame is planned on is the grid after ITS update."""
    import threading
    import fuxi_planner_amd.planner as planner_mod
    from fuxi_planner_amd.replan import FramePipeline

    class FakePlanner(object):
        made = []

        def __init__(self, devices):
            self.grid = None
            self.calls = []
            self.closed = False
            self.lock = threading.Lock()
            FakePlanner.made.append(self)

        def set_memory_share(self, k):
            self.share = k

        def set_grid_occ(self, occ):
            self.grid = occ.copy()

        def set_queries(self, starts, goals, hchoice, max_path_len):
            self.nq = len(starts)

        def update_cells(self, xy, val, rebuild=True):
            assert not rebuild, "the frame's own call rebuilds the maps"
            assert self.lock.acquire(blocking=False), "a handle is used by one thread at a time"
            self.grid[xy[:, 0], xy[:, 1]] = val
            self.calls.append("u")
            self.lock.release()

        def replan_frame(self, xy, val):
            assert self.lock.acquire(blocking=False), "a handle is used by one thread at a time"
            self.grid[xy[:, 0], xy[:, 1]] = val
            self.calls.append("r")
            out = self.grid.copy()
            self.lock.release()
            return out

        def close(self):
            self.closed = True

    monkeypatch.setattr(planner_mod, "Planner", FakePlanner)
    rng = np.random.default_rng(3)
    occ = (rng.random((12, 9)) < 0.3).astype(np.uint8)
    frames, want, g = [], [], occ.copy()
    for fr in range(17):
        n = int(rng.integers(0, 20))
        xy = np.stack([rng.integers(0, 12, n), rng.integers(0, 9, n)], 1).astype(np.int32)
        xy = np.unique(xy, axis=0).astype(np.int32).reshape(-1, 2)  # ((one update per cell within a frame: not required any more))
        val = rng.integers(0, 2, len(xy)).astype(np.uint8)
        g[xy[:, 0], xy[:, 1]] = val
        frames.append((xy, val))
        want.append(g.copy())
    for k in (1, 3, 5):  # whichever handle is idle takes the frame: every frame still sees the grid after its own update
        FakePlanner.made = []
        with FramePipeline(0, k, occ, np.zeros((4, 2), np.int32), np.ones((4, 2), np.int32)) as pipe:
            assert pipe.schedule == "free"
            futs = [pipe.submit(xy, val) for xy, val in frames]
            got = [f.result() for f in futs]
        assert len(FakePlanner.made) == k and all(p.closed for p in FakePlanner.made)
        for fr in range(len(frames)):
            assert np.array_equal(got[fr], want[fr]), (k, fr)
        assert sum(p.calls.count("r") for p in FakePlanner.made) == len(frames)
    for k in (1, 3, 5):
        FakePlanner.made = []
        with FramePipeline(0, k, occ, np.zeros((4, 2), np.int32), np.ones((4, 2), np.int32), schedule="turn") as pipe:
            futs = [pipe.submit(xy, val) for xy, val in frames]
            got = [f.result() for f in futs]
        assert len(FakePlanner.made) == k and all(p.closed for p in FakePlanner.made)
        for fr in range(len(frames)):
            assert np.array_equal(got[fr], want[fr]), (k, fr)
        # handle j planned frames j, j + k, ...: before each of them the k - 1 updates of the others (j for the first)
        for j, p in enumerate(FakePlanner.made):
            assert "".join(p.calls).startswith("u" * j + "r")
            assert p.calls.count("r") == len(range(j, len(frames), k))
            assert p.share == k

    # a handle that fails stops the pipeline: its frame's future carries the exception, the submit whose turn is that
    # handle raises it BEFORE anything of the new frame is queued, later submits are refused, close() still closes
    class Boom(RuntimeError):
        pass

    FakePlanner.made = []
    pipe = FramePipeline(0, 2, occ, np.zeros((4, 2), np.int32), np.ones((4, 2), np.int32), schedule="turn")
    orig = FakePlanner.made[1].replan_frame
    FakePlanner.made[1].replan_frame = lambda xy, val: (_ for _ in ()).throw(Boom("device lost"))
    f0 = pipe.submit(*frames[0])
    f1 = pipe.submit(*frames[1])
    f2 = pipe.submit(*frames[2])
    assert np.array_equal(f0.result(), want[0]) and np.array_equal(f2.result(), want[2])
    with pytest.raises(Boom):
        f1.result()
    n_before = pipe._n
    with pytest.raises(Boom):
        pipe.submit(*frames[3])          # handle 1's turn
    assert pipe._n == n_before and all(len(b) == 0 or b is pipe._backlog[1] or len(b) == 1 for b in pipe._backlog)
    with pytest.raises(RuntimeError):
        pipe.submit(*frames[3])
    FakePlanner.made[1].replan_frame = orig
    pipe.close()
    assert all(p.closed for p in FakePlanner.made)


def test_batch_pipeline_hands_the_batches_out_in_turn(monkeypatch):
    """BatchPipeline (host logic, no GPU): batch b goes to handle b % K, every handle has the grid, a handle is used by
    one thread at a time, every future carries its own batch's answer."""
    import threading
    import fuxi_planner_amd.planner as planner_mod
    from fuxi_planner_amd.replan import BatchPipeline

    class FakePlanner(object):
        made = []

        def __init__(self, devices):
            self.lock = threading.Lock()
            self.batches = []
            self.closed = False
            FakePlanner.made.append(self)

        def set_memory_share(self, k):
            self.share = k

        def set_grid_occ(self, occ):
            self.grid = occ.copy()

        def plan_batch(self, starts, goals, hchoice=2, max_path_len=None):
            assert self.lock.acquire(blocking=False), "a handle is used by one thread at a time"
            self.batches.append(int(starts[0, 0]))
            out = (int(starts[0, 0]), int(self.grid.sum()), hchoice, max_path_len)
            self.lock.release()
            return out

        def close(self):
            self.closed = True

    monkeypatch.setattr(planner_mod, "Planner", FakePlanner)
    occ = (np.random.default_rng(4).random((10, 7)) < 0.3).astype(np.uint8)
    for k in (1, 2, 4):
        FakePlanner.made = []
        with BatchPipeline(0, k, occ, schedule="turn") as pipe:
            futs = [pipe.submit(np.full((3, 2), b, np.int32), np.zeros((3, 2), np.int32), 1 + b % 2, 64) for b in range(11)]
            got = [f.result() for f in futs]
        assert got == [(b, int(occ.sum()), 1 + b % 2, 64) for b in range(11)]
        assert len(FakePlanner.made) == k and all(p.closed and p.share == k for p in FakePlanner.made)
        for j, p in enumerate(FakePlanner.made):
            assert p.batches == list(range(j, 11, k))
        FakePlanner.made = []
        with BatchPipeline(0, k, occ) as pipe:  # (the default: whichever handle is idle)
            assert pipe.schedule == "free"
            futs = [pipe.submit(np.full((3, 2), b, np.int32), np.zeros((3, 2), np.int32), 1 + b % 2, 64) for b in range(11)]
            got = [f.result() for f in futs]
        assert got == [(b, int(occ.sum()), 1 + b % 2, 64) for b in range(11)]
        assert sorted(b for p in FakePlanner.made for b in p.batches) == list(range(11))
