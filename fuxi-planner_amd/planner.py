"""Host-side planner object: owns one libfxjps handle (one or more MI355X) and
mirrors the reference's call surface for the grid search.

Reference interface being replaced: jps1.method(matrix, start, goal, hchoice)
(scripts/jps1.py:183-230), called from scripts/global_planner_st.py:285 and
scripts/global_planner_ccst.py:477.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FxjpsError


def as_occ(matrix):
    """The reference treats a cell as an obstacle iff it compares equal to 1
    (jps1.py:20-29); 100, 0.5, -1 ... are free."""
    return np.ascontiguousarray(np.asarray(matrix) == 1).view(np.uint8)  # (bool -> uint8 is a view: no second pass over the cells)


class Planner(object):
    """A resident occupancy grid plus batched (start, goal) planning on the GPU."""

    def __init__(self, devices=None):
        L = _lib.load()
        n = L.fxjps_device_count()
        if n <= 0:
            raise FxjpsError(_lib.E_NODEV, "no HIP device visible: fuxi-planner_amd has no CPU fallback")
        if devices is None:
            devices = [0]
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = L.fxjps_create(_lib.BACKEND_HIP, ids, len(devices), C.byref(h))
        if rc != 0:
            raise FxjpsError(rc, (L.fxjps_last_error(None) or b"").decode())
        self._L = L
        self._h = h
        self.devices = list(devices)
        self.shape = None

    @classmethod
    def for_rank(cls, device, rank, world, unique_id=None):
        """One process per GPU without torch: this process is `rank` of `world` and plans on `device`; `unique_id` is the
        128-byte RCCL id rank 0 got from `Planner.rank_unique_id()` (None for world == 1).  Collective: every rank calls it."""
        L = _lib.load()
        if L.fxjps_device_count() <= 0:
            raise FxjpsError(_lib.E_NODEV, "no HIP device visible: fuxi-planner_amd has no CPU fallback")
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        rc = L.fxjps_create_rank(int(device), int(rank), int(world), buf, C.byref(h))
        if rc != 0:
            raise FxjpsError(rc, (L.fxjps_last_error(None) or b"").decode())
        self = cls.__new__(cls)
        self._L = L
        self._h = h
        self.devices = [int(device)]
        self.shape = None
        self.rank, self.world = int(rank), int(world)
        return self

    @staticmethod
    def rank_unique_id():
        """The id of a new RCCL communicator (ncclGetUniqueId): rank 0 makes it and hands it to the other ranks."""
        L = _lib.load()
        buf = C.create_string_buffer(128)
        rc = L.fxjps_rank_unique_id(buf)
        if rc != 0:
            raise FxjpsError(rc, (L.fxjps_last_error(None) or b"").decode())
        return buf.raw

    @staticmethod
    def rank_preflight(device):
        """What can keep THIS process out of the ranks' collectives, checked without one: the device exists and takes an
        allocation, librccl loads and has the entry points used.  -> None, or the reason as text."""
        try:
            L = _lib.load()
            rc = L.fxjps_rank_preflight(int(device))
        except (OSError, FxjpsError) as e:
            return str(e)
        return None if rc == 0 else "fxjps error %d: %s" % (rc, (L.fxjps_last_error(None) or b"").decode())

    def reserve_grid(self, W, H):
        """Allocate the device buffers of a W x H grid (what the next set_grid* call would allocate), nothing else."""
        self._chk(self._L.fxjps_reserve_grid(self._h, int(W), int(H)))
        self._resident = None

    def set_grid_rank(self, occ, W, H):
        """Collective over the ranks of `for_rank`: rank 0 passes the uint8 [W][H] occupancy, the others None; ONE
        ncclBroadcast of the W*H bytes inside the library, then every rank builds its maps."""
        if occ is not None:
            occ = np.ascontiguousarray(occ, dtype=np.uint8)
            if occ.shape != (W, H):
                raise ValueError("grid shape %r is not (%d, %d)" % (occ.shape, W, H))
        self._chk(self._L.fxjps_set_grid_rank(self._h, _lib.ptr(occ, C.c_uint8) if occ is not None else None, int(W), int(H)))
        self.shape = (int(W), int(H))

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.fxjps_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise FxjpsError(rc, (self._L.fxjps_last_error(self._h) or b"").decode())

    # -- grid
    @property
    def shape(self):
        """(W, H) of the resident grid, None before the first one."""
        return self._shape

    @shape.setter
    def shape(self, v):  # (every call that replaces the resident grid sets it: what set_grid remembers is void then)
        self._shape = v
        self._resident = None

    def set_grid(self, matrix):
        """Upload `matrix` (any 2-D array, matrix[x][y], obstacle iff == 1).

        The node hands jps1.method its map anew every tick (global_planner_st.py:246-262 builds `mapu` from scratch each
        time) whether or not a map message arrived in between: a matrix equal to the grid that is resident is not uploaded
        again -- the comparison costs microseconds, the upload and the map build ~ 120 us at the node's map size."""
        occ = as_occ(matrix)
        r = self._resident
        if r is not None and r.shape == occ.shape and np.array_equal(r, occ):
            return
        self.set_grid_occ(occ)
        self._resident = occ  # (as_occ made a new array: nobody else holds it)

    def set_grid_occ(self, occ):
        """Upload a uint8 [W][H] occupancy array (non-zero = obstacle)."""
        occ = np.ascontiguousarray(occ, dtype=np.uint8)
        if occ.ndim != 2:
            raise ValueError("grid must be 2-D")
        W, H = occ.shape
        self._resident = None  # (a failing upload leaves the handle without a grid, or with half of this one: nothing is resident)
        self._chk(self._L.fxjps_set_grid(self._h, _lib.ptr(occ, C.c_uint8), W, H))
        self.shape = (W, H)

    def set_grid_device(self, dev_ptr, W, H):
        """Adopt a uint8 [W][H] grid that already lives in device memory
        (e.g. the output of an RCCL broadcast done by the host framework)."""
        self._chk(self._L.fxjps_set_grid_device(self._h, C.c_void_p(int(dev_ptr)), int(W), int(H)))
        self.shape = (int(W), int(H))

    def prepare_grid(self, raw, start, goal, ifa, variant="st"):
        """The callers' grid preparation on the device (global_planner_st.py:230-272 / global_planner_ccst.py:415-458):
        pad `raw` (> 0 = occupied) so that start and goal fit, dilate by `ifa`, keep the result resident.
        -> (start', goal', map_d, (W, H), end_occu) with start'/goal' in the prepared grid (goal moved off obstacles)
        and the reference's end_occu flag (global_planner_st.py:268-275 / global_planner_ccst.py:461-464)."""
        raw = np.ascontiguousarray(np.asarray(raw) > 0, dtype=np.uint8)
        if raw.ndim != 2:
            raise ValueError("grid must be 2-D")
        v = {"st": 0, "ccst": 1}[variant] if isinstance(variant, str) else int(variant)
        s = (C.c_int32 * 2)(int(start[0]), int(start[1]))
        g = (C.c_int32 * 2)(int(goal[0]), int(goal[1]))
        W, H, eo = C.c_int32(), C.c_int32(), C.c_int32()
        md = (C.c_int32 * 2)()
        self._chk(self._L.fxjps_prepare_grid(self._h, _lib.ptr(raw, C.c_uint8), raw.shape[0], raw.shape[1], int(ifa), v,
                                             s, g, C.byref(W), C.byref(H), md, C.byref(eo)))
        self.shape = (W.value, H.value)
        return (s[0], s[1]), (g[0], g[1]), (md[0], md[1]), self.shape, eo.value

    @staticmethod
    def shifted_origin(map_o, map_d, map_reso):
        """The map origin after the preparation (global_planner_st.py:235-236 / global_planner_ccst.py:420-421): the
        padding moves it by -map_d cells.  Same expression, same rounding, as the reference's."""
        return list(-np.asarray(map_d) * map_reso + np.asarray(map_o, dtype=np.float64))

    def prepare_occupancy_msg(self, data, width, height, start, goal, ifa, variant="st"):
        """prepare_grid straight from a nav_msgs/OccupancyGrid (`data` = msg.data, int8, row-major [y][x]):
        map_callback (global_planner_st.py:15-20) is fused into the device kernel."""
        data = np.ascontiguousarray(data, dtype=np.int8).reshape(-1)
        if data.size != width * height:
            raise ValueError("data has %d cells, expected %d" % (data.size, width * height))
        v = {"st": 0, "ccst": 1}[variant] if isinstance(variant, str) else int(variant)
        s = (C.c_int32 * 2)(int(start[0]), int(start[1]))
        g = (C.c_int32 * 2)(int(goal[0]), int(goal[1]))
        W, H, eo = C.c_int32(), C.c_int32(), C.c_int32()
        md = (C.c_int32 * 2)()
        self._chk(self._L.fxjps_prepare_occupancy_msg(self._h, _lib.ptr(data, C.c_int8), int(width), int(height), int(ifa), v,
                                                      s, g, C.byref(W), C.byref(H), md, C.byref(eo)))
        self.shape = (W.value, H.value)
        return (s[0], s[1]), (g[0], g[1]), (md[0], md[1]), self.shape, eo.value

    def get_grid(self, context=0):
        """The resident uint8 [W][H] occupancy grid (e.g. the prepared map the node publishes); of a multi-device handle:
        the bytes context `context` holds (equal on every device after the broadcast)."""
        W, H = C.c_int32(), C.c_int32()
        self._chk(self._L.fxjps_get_grid_context(self._h, int(context), None, C.byref(W), C.byref(H)))
        out = np.empty((W.value, H.value), dtype=np.uint8)
        self._chk(self._L.fxjps_get_grid_context(self._h, int(context), _lib.ptr(out, C.c_uint8), None, None))
        return out

    # -- wire / on-disk adapters
    def publish_map(self):
        """What publish_map (global_planner_st.py:102-115) sends: -> (data int8[W*H] row-major [y][x] with 100 = occupied,
        width, height) of the resident grid."""
        w, h = C.c_int32(), C.c_int32()
        self._chk(self._L.fxjps_publish_map(self._h, None, C.byref(w), C.byref(h)))
        data = np.empty(w.value * h.value, dtype=np.int8)
        self._chk(self._L.fxjps_publish_map(self._h, _lib.ptr(data, C.c_int8), None, None))
        return data, w.value, h.value

    def set_grid_image(self, gray):
        """Adopt a decoded 8-bit grey image (rows x cols) as the resident grid with the prior-map convention of
        global_planner_st.py:176-182: > 200 free, else occupied, grid = img[::-1].T."""
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        if gray.ndim != 2:
            raise ValueError("image must be 2-D (convert('L'))")
        self._chk(self._L.fxjps_set_grid_image(self._h, _lib.ptr(gray, C.c_uint8), gray.shape[0], gray.shape[1]))
        self.shape = (gray.shape[1], gray.shape[0])

    def snapshot_image(self, channels=1):
        """The snapshot convention of global_planner_st.py:365-374: uint8 [H][W] (or [H][W][3]) image, 255 = free."""
        r, c = C.c_int32(), C.c_int32()
        self._chk(self._L.fxjps_snapshot_image(self._h, None, int(channels), C.byref(r), C.byref(c)))
        out = np.empty((r.value, c.value) if channels == 1 else (r.value, c.value, channels), dtype=np.uint8)
        self._chk(self._L.fxjps_snapshot_image(self._h, _lib.ptr(out, C.c_uint8), int(channels), None, None))
        return out

    def update_cells(self, xy, val, rebuild=True):
        """Set cells of the resident grid.  rebuild=False: the derived maps are rebuilt by the next planning call (or
        the next update with rebuild=True) instead of now -- for several updates in a row."""
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        val = np.ascontiguousarray(val, dtype=np.uint8).reshape(-1)
        if len(val) != len(xy):
            raise ValueError("xy and val lengths differ")
        self._resident = None
        fn = self._L.fxjps_update_cells if rebuild else self._L.fxjps_update_cells_deferred
        self._chk(fn(self._h, _lib.ptr(xy, C.c_int32), _lib.ptr(val, C.c_uint8), len(val)))

    # -- planning
    def default_max_path_len(self):
        W, H = self.shape
        return int(min(W * H + 1, max(256, 4 * max(W, H))))

    def plan_batch(self, starts, goals, hchoice=2, max_path_len=None):
        """-> (offsets int64[n+1], cells int32[total,2], cost float64[n], status int32[n]).

        status[q] > 0: number of jump points of query q (cells[offsets[q]:offsets[q+1]]),
        0: no path, < 0: a per-query error code (_lib.Q_*)."""
        if self.shape is None:
            raise FxjpsError(_lib.E_NOGRID, "plan_batch before set_grid")
        starts = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, 2)
        goals = np.ascontiguousarray(goals, dtype=np.int32).reshape(-1, 2)
        if len(starts) != len(goals):
            raise ValueError("starts and goals lengths differ")
        if hchoice not in (1, 2):
            # heuristic() returns None and jps1.py:188/227 then fails on None + float
            raise TypeError("unsupported operand type(s) for +: 'float' and 'NoneType' (hchoice must be 1 or 2)")
        n = len(starts)
        auto = max_path_len is None
        mpl = self.default_max_path_len() if auto else int(max_path_len)
        limit = self.shape[0] * self.shape[1] + 1
        while True:
            offsets = np.zeros(n + 1, dtype=np.int64)
            status = np.zeros(n, dtype=np.int32)
            cost = np.zeros(n, dtype=np.float64)
            secs = C.c_double(0.0)
            # sizing call (cells stay in the handle), then one copy into an exactly sized array
            self._chk(self._L.fxjps_plan_batch_csr(self._h, _lib.ptr(starts, C.c_int32), _lib.ptr(goals, C.c_int32), n,
                                                   int(hchoice), mpl, _lib.ptr(offsets, C.c_int64), None, 0,
                                                   _lib.ptr(status, C.c_int32), _lib.ptr(cost, C.c_double),
                                                   C.byref(secs)))
            cells = np.empty((int(offsets[n]), 2), dtype=np.int32)
            if offsets[n] > 0:
                self._chk(self._L.fxjps_last_cells(self._h, _lib.ptr(cells, C.c_int32), int(offsets[n])))
            if auto and mpl < limit and (status == _lib.Q_PATH_TOO_LONG).any():
                mpl = min(limit, mpl * 8)  # rare: a path with more jump points than the default slot
                continue
            break
        self.last_seconds = secs.value
        return offsets, cells, cost, status

    # -- streaming replan (persistent goals, one call per frame)
    def set_queries(self, starts, goals, hchoice=2, max_path_len=None):
        """Store the persistent (start, goal) set that replan_frame plans every frame."""
        if self.shape is None:
            raise FxjpsError(_lib.E_NOGRID, "set_queries before set_grid")
        starts = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, 2)
        goals = np.ascontiguousarray(goals, dtype=np.int32).reshape(-1, 2)
        if len(starts) != len(goals):
            raise ValueError("starts and goals lengths differ")
        if hchoice not in (1, 2):
            raise TypeError("unsupported operand type(s) for +: 'float' and 'NoneType' (hchoice must be 1 or 2)")
        mpl = self.default_max_path_len() if max_path_len is None else int(max_path_len)
        self._chk(self._L.fxjps_set_queries(self._h, _lib.ptr(starts, C.c_int32), _lib.ptr(goals, C.c_int32), len(starts),
                                            int(hchoice), mpl))
        self._nq = len(starts)

    def replan_frame(self, xy=None, val=None):
        """One frame: apply the cell updates (xy int32[n, 2], val uint8[n]; may be empty), rebuild the maps, plan the
        stored queries.  -> (offsets, cells, cost, status) as plan_batch."""
        if xy is None:
            xy, val = np.zeros((0, 2), np.int32), np.zeros(0, np.uint8)
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        val = np.ascontiguousarray(val, dtype=np.uint8).reshape(-1)
        if len(val) != len(xy):
            raise ValueError("xy and val lengths differ")
        n = self._nq
        self._resident = None
        offsets = np.zeros(n + 1, dtype=np.int64)
        status = np.zeros(n, dtype=np.int32)
        cost = np.zeros(n, dtype=np.float64)
        secs = C.c_double(0.0)
        self._chk(self._L.fxjps_replan_frame(self._h, _lib.ptr(xy, C.c_int32), _lib.ptr(val, C.c_uint8), len(val),
                                             _lib.ptr(offsets, C.c_int64), None, 0, _lib.ptr(status, C.c_int32),
                                             _lib.ptr(cost, C.c_double), C.byref(secs)))
        cells = np.empty((int(offsets[n]), 2), dtype=np.int32)
        if offsets[n] > 0:
            self._chk(self._L.fxjps_last_cells(self._h, _lib.ptr(cells, C.c_int32), int(offsets[n])))
        self.last_seconds = secs.value
        return offsets, cells, cost, status

    def plan_one(self, start, goal, hchoice=2):
        """One query -- the node's call, once per tick -- without the per-call allocations of plan_batch: the arrays of a
        one-query call are kept, the cells come back with the call itself (no sizing call).  -> (status, cost, cells
        int32[n, 2] view valid until the next call)."""
        if self.shape is None:
            raise FxjpsError(_lib.E_NOGRID, "plan before set_grid")
        if hchoice not in (1, 2):
            raise TypeError("unsupported operand type(s) for +: 'float' and 'NoneType' (hchoice must be 1 or 2)")
        mpl = self.default_max_path_len()
        o = getattr(self, "_one", None)
        if o is None or o["mpl"] != mpl:
            o = {"mpl": mpl, "s": np.zeros((1, 2), np.int32), "g": np.zeros((1, 2), np.int32), "off": np.zeros(2, np.int64),
                 "st": np.zeros(1, np.int32), "cost": np.zeros(1, np.float64), "cells": np.zeros((mpl, 2), np.int32), "secs": C.c_double(0.0)}
            o["args"] = (_lib.ptr(o["s"], C.c_int32), _lib.ptr(o["g"], C.c_int32), _lib.ptr(o["off"], C.c_int64), _lib.ptr(o["cells"], C.c_int32),
                         _lib.ptr(o["st"], C.c_int32), _lib.ptr(o["cost"], C.c_double), C.byref(o["secs"]))
            self._one = o
        o["s"][0, 0], o["s"][0, 1] = start[0], start[1]
        o["g"][0, 0], o["g"][0, 1] = goal[0], goal[1]
        a = o["args"]
        self._chk(self._L.fxjps_plan_batch_csr(self._h, a[0], a[1], 1, int(hchoice), mpl, a[2], a[3], mpl, a[4], a[5], a[6]))
        st = int(o["st"][0])
        if st == _lib.Q_PATH_TOO_LONG and mpl < self.shape[0] * self.shape[1] + 1:  # (rare: the general path grows the slot)
            offsets, cells, cost, status = self.plan_batch([start], [goal], hchoice)
            return int(status[0]), float(cost[0]), cells[offsets[0]:offsets[1]]
        self.last_seconds = o["secs"].value
        return st, float(o["cost"][0]), o["cells"][:max(st, 0)]

    def plan(self, start, goal, hchoice=2):
        """plan(start, goal) -> waypoint list [(x, y), ...] (jump points, start and
        goal inclusive); [] when there is no path."""
        st, cost, cells = self.plan_one(start, goal, hchoice)
        self.last_cost = cost
        if st == _lib.Q_BAD_START:
            raise IndexError("start %r is outside the %dx%d grid" % (tuple(start), self.shape[0], self.shape[1]))
        if st < 0:
            raise FxjpsError(st, "query failed")
        return [(int(x), int(y)) for x, y in cells]

    def timing(self):
        t = _lib.Timing()
        self._chk(self._L.fxjps_last_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.Timing._fields_}

    def timing_per_context(self):
        """Per context of the handle (one per entry of `devices`): device, queries of its shard, search-kernel ms,
        resident wavefronts of the last batch."""
        out = []
        for r in range(len(self.devices)):
            dev, nq, ms, wv = C.c_int32(), C.c_int64(), C.c_double(), C.c_int64()
            self._chk(self._L.fxjps_last_timing_device(self._h, r, C.byref(dev), C.byref(nq), C.byref(ms), C.byref(wv)))
            out.append({"device": dev.value, "queries": nq.value, "kernel_ms": ms.value, "waves": wv.value})
        return out

    def comm_info(self):
        """-> {"contexts", "devices", "rccl_ranks"}: rccl_ranks is ncclCommCount of the handle's communicator (0 while
        no collective has run: one device, or contexts sharing a device)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.fxjps_comm_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"contexts": a.value, "devices": b.value, "rccl_ranks": c.value}

    def set_memory_share(self, handles_per_device):
        """This handle is one of `handles_per_device` on its device(s): scratch pools and resident wavefronts are sized
        for that share (FramePipeline sets it)."""
        self._chk(self._L.fxjps_set_memory_share(self._h, int(handles_per_device)))

    # -- test hooks
    def selftest_sqrt(self, n0, n1):
        out = np.empty(n1 - n0, dtype=np.float64)
        self._chk(self._L.fxjps_selftest_sqrt(self._h, n0, n1, _lib.ptr(out, C.c_double)))
        return out

    def selftest_wavemin(self, rounds=4096, seed=1):
        bad = C.c_int64(-1)
        self._chk(self._L.fxjps_selftest_wavemin(self._h, rounds, seed, C.byref(bad)))
        return bad.value

    def selftest_openlist(self, keys_f, keys_x, step_pops, step_off, banded=False, far_cap=8192, near_max=512, delta0=2.0):
        """Run a push / pop script through the open list of the search kernel (fxjps_selftest_openlist).
        -> (popped f bits uint64[n], popped x uint32[n], pushed-entry index uint32[n], pops per step uint32[nsteps],
        {fail, far_refills, slow_pops})"""
        kf = np.ascontiguousarray(keys_f, dtype=np.uint64)
        kx = np.ascontiguousarray(keys_x, dtype=np.uint32)
        sp = np.ascontiguousarray(step_pops, dtype=np.uint32)
        so = np.ascontiguousarray(step_off, dtype=np.uint32)
        n, ns = len(kf), len(sp)
        of, ox, os_ = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        ok, info = np.zeros(max(ns, 1), np.uint32), np.zeros(8, np.uint32)
        self._chk(self._L.fxjps_selftest_openlist(self._h, int(bool(banded)), int(far_cap), int(near_max), float(delta0),
                                                  _lib.ptr(kf, C.c_uint64), _lib.ptr(kx, C.c_uint32), n, _lib.ptr(sp, C.c_uint32),
                                                  _lib.ptr(so, C.c_uint32), ns, _lib.ptr(of, C.c_uint64), _lib.ptr(ox, C.c_uint32),
                                                  _lib.ptr(os_, C.c_uint32), _lib.ptr(ok, C.c_uint32), _lib.ptr(info, C.c_uint32)))
        t = int(info[0])
        return of[:t], ox[:t], os_[:t], ok[:ns], {"fail": int(info[1]), "far_refills": int(info[2]), "slow_pops": int(info[3]), "held": [int(v) for v in info[4:8]]}

    def debug_maps(self):
        """The derived device maps (fxjps_debug_read_maps): {"bm": uint64[4, LINES, WORDS, 2], "ci": uint16[W+2, H+2],
        "comp": int32[W, H] (union-find parent links), "nb8": uint8[W+2, H+2], "dbm": uint64[4, W+H+3, WORDS, 2] (the
        diagonal scan words), "jd": uint16[W+2, H+2, 8] (the jump distances)}."""
        W, H = self.shape
        PW, PH = W + 2, H + 2
        NS = (PH + 63) & ~63
        LINES = max(PW, PH)
        WORDS = (LINES + 63) // 64
        out = {}
        for which, name, dt, shape in ((0, "bm", np.uint64, (4, LINES, WORDS, 2)), (1, "ci", np.uint16, (PW, NS)), (2, "comp", np.int32, (W, H)),
                                       (3, "nb8", np.uint8, (PW, NS)), (4, "dbm", np.uint64, (4, PW + PH - 1, WORDS, 2)), (5, "jd", np.uint16, (PW, NS, 8))):
            a = np.zeros(shape, dtype=dt)
            nb = C.c_int64(0)
            self._chk(self._L.fxjps_debug_read_maps(self._h, which, a.ctypes.data_as(C.c_void_p), a.nbytes, C.byref(nb)))
            assert nb.value == a.nbytes, (name, nb.value, a.nbytes)
            out[name] = a[:, :PH] if name in ("ci", "nb8", "jd") else a
        # (lines the kernels neither write nor read -- the +-x scans have a line per padded y, the +-y scans per padded
        # x, the array has max(PW, PH) of each -- hold whatever the allocation held)
        out["bm"][0:2, PH:] = 0
        out["bm"][2:4, PW:] = 0
        return out

    def debug_nbmask(self):
        W, H = self.shape
        buf = np.empty((W + 2, H + 2), dtype=np.uint8)
        self._chk(self._L.fxjps_debug_read_nbmask(self._h, _lib.ptr(buf, C.c_uint8)))
        return buf


_default = None


def default_planner():
    """Process-wide planner on device 0 (what the jps1 shim uses)."""
    global _default
    if _default is None:
        _default = Planner([0])
    return _default


def plan(grid, start, goal, hchoice=2):
    """north_star call surface: plan(grid, start, goal) -> waypoint list."""
    p = default_planner()
    p.set_grid(grid)
    return p.plan(start, goal, hchoice)


def plan_batch(grid, starts, goals, hchoice=2, max_path_len=None):
    p = default_planner()
    p.set_grid(grid)
    return p.plan_batch(starts, goals, hchoice, max_path_len)
