"""Replan throttle of the ccst node: scripts/global_planner_ccst.py:311-312, 476-480.

The node searches again only when the vehicle has moved more than 0.3 m since the last search, or more than 0.3 s
have passed, or no search has been recorded yet -- which the reference encodes as `last_jps_pos[0] == 0`, so a
vehicle whose last search happened at x == 0.0 exactly always searches again (kept).

    throttle = ReplanThrottle()
    ...
    if throttle.due((px, py, pz)):
        path1 = jps1.method(mapu, tuple(map_start), tuple(map_goal), 2)
        throttle.mark(planner.parse_local_position(planner.pos))      # the position is read again after the search
"""
import os
import sys
import time
import warnings
from concurrent.futures import FIRST_COMPLETED, ThreadPoolExecutor, wait

import numpy as np

HW_QUEUES = 16  # hardware queues the pipelines below want the HIP runtime to hand out (its default is 4)


def configure_hw_queues(n=HW_QUEUES):
    """Frames / batches in flight need hardware queues: the HIP runtime maps every stream onto one of GPU_MAX_HW_QUEUES
    (default 4) queues, and the persistent search kernels of two handles whose streams share a queue run one after the
    other.  The variable is read ONCE, when the runtime initialises in this process -- so it has to be in the environment
    before the first GPU call (of this package or of anything else, torch included).  Call this first thing in a process
    that will build a `FramePipeline` / `BatchPipeline`; the pipelines call it themselves and warn when it is too late.
    -> the number of queues the runtime will use (or is assumed to use), None when that cannot be arranged any more."""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        try:
            return int(cur)
        except ValueError:
            return None
    from . import _lib
    torch = sys.modules.get("torch")
    hip_up = _lib._lib is not None or (torch is not None and getattr(getattr(torch, "cuda", None), "is_initialized", lambda: False)())
    if hip_up:
        return None
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return int(n)


def _want_queues(k, what):
    got = configure_hw_queues()
    if got is None:
        warnings.warn("%s with %d planner handles: the HIP runtime of this process was initialised without GPU_MAX_HW_QUEUES "
                      "(4 hardware queues): handles that share a queue plan one after the other.  Call "
                      "fuxi_planner_amd.replan.configure_hw_queues() before the first GPU call, or set GPU_MAX_HW_QUEUES=%d in "
                      "the environment." % (what, k, HW_QUEUES), RuntimeWarning, stacklevel=3)
    elif got < k:
        warnings.warn("%s with %d planner handles on %d hardware queues (GPU_MAX_HW_QUEUES): handles that share a queue plan "
                      "one after the other" % (what, k, got), RuntimeWarning, stacklevel=3)


class ReplanThrottle(object):
    def __init__(self, min_move=0.3, min_interval=0.3, clock=time.time):
        self.min_move = min_move
        self.min_interval = min_interval
        self.clock = clock
        self.last_pos = np.array([0, 0, 0])  # ccst:311
        self.last_time = 0                   # ccst:312

    def due(self, pos):
        """ccst:476."""
        p = np.asarray(pos, dtype=np.float64)
        return bool(self.last_pos[0] == 0 or np.linalg.norm(self.last_pos - p) > self.min_move or
                    self.clock() - self.last_time > self.min_interval)

    def mark(self, pos):
        """ccst:479-480."""
        self.last_pos = np.asarray(pos, dtype=np.float64).copy()
        self.last_time = self.clock()


class FramePipeline(object):
    """Streaming replan with several frames in flight (BASELINE config 5: a 60 Hz *rate*).

    One frame -- cell updates, map rebuild, search of the persistent queries -- takes as long as its longest query:
    73 ms for 1 000 queries at 1024^2, during most of which the chip is nearly idle.  The frames do not depend on each
    other's results, only on the grid, so K planner handles on one device (own grid copy, own scratch, own stream) share
    the frames: the handle that takes frame f applies the updates of the frames since its last one, in order, and plans
    frame f while the frames before it are still being searched.  Every frame's result is what fxjps_replan_frame returns
    for that frame (same library calls, same kernels); the latency of a frame stays what it was, the rate is K times
    higher until the chip is full.

        pipe = FramePipeline(0, 6, occ, starts, goals)
        futures = [pipe.submit(xy, val) for xy, val in frames]      # returns at once while a handle is free
        for f in futures: offsets, cells, cost, status = f.result()
    """

    def __init__(self, device, k, occ, starts, goals, hchoice=2, max_path_len=None, schedule="free"):
        """schedule: "free" (default) -- a frame goes to whichever handle is idle (the one that has waited longest), submit
        waits only while none is: a frame that takes long holds up its own handle, not the frames whose turn would have
        come behind it (config 5, twelve handles: 155.8 -> 159 frames/s, p99 104 - 109 -> 98 ms); "turn" -- frame f goes to
        handle f % K and submit waits for that handle (round 4's).  The results are the same either way."""
        from .planner import Planner
        self.k = int(k)
        if schedule not in ("turn", "free"):
            raise ValueError("schedule must be 'turn' or 'free'")
        self.schedule = schedule
        _want_queues(self.k, "FramePipeline")
        self.planners = [Planner([device]) for _ in range(self.k)]
        for p in self.planners:
            p.set_memory_share(self.k)  # K handles on one device: each sizes its scratch for a K-th of it
            p.set_grid_occ(occ)
            p.set_queries(starts, goals, hchoice, max_path_len)
        self._backlog = [[] for _ in range(self.k)]  # updates a handle has not applied yet, oldest first
        self._busy = [None] * self.k
        self._pool = ThreadPoolExecutor(max_workers=self.k)
        self._n = 0
        self._broken = None  # the exception that stopped the pipeline: a handle that failed has lost updates

    def _run(self, j, updates):
        p = self.planners[j]
        for xy, val in updates[:-1]:  # the frames the other handles planned: the grid follows them (the maps are
            p.update_cells(xy, val, rebuild=False)  # rebuilt once, by the frame's own call)
        return p.replan_frame(*updates[-1])

    def submit(self, xy, val):
        """Queue one frame.  Blocks only while no handle is idle ("turn": while the handle whose turn it is still plans)."""
        if self._broken is not None:
            raise RuntimeError("FramePipeline stopped by an earlier failure: %r" % (self._broken,))
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        val = np.ascontiguousarray(val, dtype=np.uint8).reshape(-1)
        if self.schedule == "free":
            idle = [i for i in range(self.k) if self._busy[i] is None or self._busy[i].done()]
            if not idle:
                wait([f for f in self._busy if f is not None], return_when=FIRST_COMPLETED)
                idle = [i for i in range(self.k) if self._busy[i] is None or self._busy[i].done()]
            # the one that has waited longest (the longest list of updates to catch up on): no handle falls behind for good
            j = max(idle, key=lambda i: (len(self._backlog[i]), -i))
        else:
            j = self._n % self.k
        # the handle's previous frame first: if it failed, the handle has popped updates it never applied and plans on
        # a stale grid from here on -- the pipeline stops instead (nothing of the new frame has been queued yet)
        if self._busy[j] is not None:
            try:
                self._busy[j].result()
            except BaseException as e:
                self._broken = e
                raise
        for b in self._backlog:
            b.append((xy, val))
        self._n += 1
        updates, self._backlog[j] = self._backlog[j], []
        self._busy[j] = self._pool.submit(self._run, j, updates)
        return self._busy[j]

    def close(self):
        try:
            for f in self._busy:
                if f is not None:
                    try:
                        f.result()
                    except BaseException as e:  # (the caller saw it through the frame's own future)
                        self._broken = self._broken or e
        finally:
            self._pool.shutdown(wait=True)
            for p in self.planners:
                p.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BatchPipeline(object):
    """Independent query batches on one resident grid with several batches in flight.

    A batch of config-2 size ends with the latency of its slowest queries: for the last third of
    its 68 ms the chip is nearly idle.  Batches do not depend on each other, so K planner handles on one device (own
    grid copy, own scratch -- a K-th of the device's each --, own streams) take them in turn; every result is what
    `Planner.plan_batch` returns for that batch (same library call, same kernels).  A batch's latency grows (it shares
    the chip), the rate is what the chip's vector units allow.

        pipe = BatchPipeline(0, 3, occ)
        futures = [pipe.submit(starts, goals) for starts, goals in batches]   # blocks only while its handle is busy
        for f in futures: offsets, cells, cost, status = f.result()
    """

    def __init__(self, device, k, occ, schedule="free"):
        """schedule: as FramePipeline's ("turn": batch b to handle b % K; "free": to an idle one)."""
        from .planner import Planner
        self.k = int(k)
        if schedule not in ("turn", "free"):
            raise ValueError("schedule must be 'turn' or 'free'")
        self.schedule = schedule
        _want_queues(self.k, "BatchPipeline")
        self.planners = [Planner([device]) for _ in range(self.k)]
        for p in self.planners:
            p.set_memory_share(self.k)
            p.set_grid_occ(occ)
        self._busy = [None] * self.k
        self._pool = ThreadPoolExecutor(max_workers=self.k)
        self._n = 0

    def submit(self, starts, goals, hchoice=2, max_path_len=None):
        """Queue one batch.  Blocks only while the handle whose turn it is still plans its previous batch (whose failure,
        if any, is the caller's through that batch's own future)."""
        if self.schedule == "free":
            idle = [i for i in range(self.k) if self._busy[i] is None or self._busy[i].done()]
            if not idle:
                wait([f for f in self._busy if f is not None], return_when=FIRST_COMPLETED)
                idle = [i for i in range(self.k) if self._busy[i] is None or self._busy[i].done()]
            j = idle[0]
        else:
            j = self._n % self.k
        if self._busy[j] is not None:
            try:
                self._busy[j].result()
            except BaseException:
                pass
        self._n += 1
        self._busy[j] = self._pool.submit(self.planners[j].plan_batch, starts, goals, hchoice, max_path_len)
        return self._busy[j]

    def close(self):
        try:
            for f in self._busy:
                if f is not None:
                    try:
                        f.result()
                    except BaseException:
                        pass
        finally:
            self._pool.shutdown(wait=True)
            for p in self.planners:
                p.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
