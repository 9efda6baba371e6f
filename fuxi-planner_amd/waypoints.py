"""Waypoint selection after a plan: the step the reference's nodes run on the path that
jps1.method returned -- scripts/global_planner_st.py:292-327 and scripts/global_planner_ccst.py:487-526 (with
map_line_col, :258-283).  Thin wrappers over the C ABI (fxjps_waypoint_st / fxjps_waypoint_ccst); the functions are
host code and need no device.

    path1 = jps1.method(mapu, tuple(map_start), tuple(map_goal), 2)
    if path1[0] is not 0:
        wp, global_goal, ang_wp = waypoints.select_st(path1[0], map_start, map_reso, map_o, (px, py, pz), global_goal,
                                                      end_occu, prev_wp=wp)                          # st
        wp, kept, global_goal = waypoints.select_ccst(path1[0], mapu, map_reso, map_o, (px, py, pz), global_goal,
                                                      end_occu, return_goal=True)                  # ccst
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .planner import as_occ


_HOST_LIB = None  # test hook: a library with the host-only entry points (e.g. the sanitizer build); None = libfxjps.so


def _L():
    return _HOST_LIB if _HOST_LIB is not None else _lib.load()


def _cells(path):
    c = np.ascontiguousarray(np.asarray(path, dtype=np.int32).reshape(-1, 2))
    if c.shape[0] < 1:
        raise ValueError("empty path")
    return c


def _vec(v, n):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel())
    if a.shape[0] != n:
        raise ValueError("expected %d components" % n)
    return a


def select_st(path, map_start, map_reso, map_o, pos, global_goal, end_occu=0, prev_wp=None, dis_wp_tre=2.0,
              ang_wp_tre=math.pi / 4):
    """global_planner_st.py:292-327.  -> (wp ndarray of 2 or 3 components, global_goal ndarray[3], ang_wp)."""
    L = _L()
    c = _cells(path)
    ms = np.ascontiguousarray(np.asarray(map_start, dtype=np.int32).ravel())
    o, p, g = _vec(map_o, 2), _vec(pos, 3), _vec(global_goal, 3)
    pw = None if prev_wp is None else np.ascontiguousarray(np.asarray(prev_wp, dtype=np.float64).ravel())
    wp = np.zeros(3)
    gout = np.zeros(3)
    dim = C.c_int32(0)
    ang = C.c_double(0.0)
    rc = L.fxjps_waypoint_st(_lib.ptr(c, C.c_int32), c.shape[0], _lib.ptr(ms, C.c_int32), float(map_reso), _lib.ptr(o, C.c_double),
                             _lib.ptr(p, C.c_double), _lib.ptr(g, C.c_double), int(end_occu), float(dis_wp_tre), float(ang_wp_tre),
                             None if pw is None else _lib.ptr(pw, C.c_double), 0 if pw is None else int(pw.shape[0]),
                             _lib.ptr(wp, C.c_double), C.byref(dim), _lib.ptr(gout, C.c_double), C.byref(ang))
    if rc != 0:
        raise _lib.FxjpsError(rc, "fxjps_waypoint_st: bad argument")
    return wp[:dim.value].copy(), gout, ang.value


def select_ccst(path, mapu, map_reso, map_o, pos, global_goal, end_occu=0, return_goal=False):
    """global_planner_ccst.py:487-544.  -> (wp ndarray[3], kept cells int32[m, 2]) [+ global_goal ndarray[3] after the block
    when return_goal]."""
    L = _L()
    c = _cells(path)
    occ = as_occ(mapu)
    o, p, g = _vec(map_o, 2), _vec(pos, 3), _vec(global_goal, 3)
    wp = np.zeros(3)
    gout = np.zeros(3)
    kept = np.zeros((c.shape[0], 2), dtype=np.int32)
    nk = C.c_int32(0)
    rc = L.fxjps_waypoint_ccst(_lib.ptr(c, C.c_int32), c.shape[0], _lib.ptr(occ, C.c_uint8), occ.shape[0], occ.shape[1], float(map_reso),
                               _lib.ptr(o, C.c_double), _lib.ptr(p, C.c_double), _lib.ptr(g, C.c_double), int(end_occu),
                               _lib.ptr(wp, C.c_double), _lib.ptr(gout, C.c_double), _lib.ptr(kept, C.c_int32), C.byref(nk))
    if rc != 0:
        raise _lib.FxjpsError(rc, "fxjps_waypoint_ccst: bad argument")
    if return_goal:
        return wp, kept[:nk.value].copy(), gout
    return wp, kept[:nk.value].copy()


def _batch_vec(v, n):
    a = np.asarray(v, dtype=np.float64)
    if a.ndim == 1:
        a = np.broadcast_to(a, (n, 3))
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != (n, 3):
        raise ValueError("expected %d x 3 values" % n)
    return a


def _batch_paths(planner, paths, n_hint):
    """paths: None (the planner's most recent batch, resident on the device) or (offsets, cells) as plan_batch returns."""
    if paths is None:
        return n_hint, None, None
    off = np.ascontiguousarray(paths[0], dtype=np.int64)
    cells = np.ascontiguousarray(np.asarray(paths[1], dtype=np.int32).reshape(-1, 2))
    return len(off) - 1, off, cells


def select_ccst_batch(planner, nq, map_reso, map_o, pos, global_goal, end_occu=None, paths=None, return_kept=False):
    """global_planner_ccst.py:487-544 for every path of a batch, on the device, against the planner's resident grid
    (fxjps_waypoint_ccst_batch).  paths=None: the paths of the planner's last plan_batch / replan_frame (`nq` of them).
    pos / global_goal: one (x, y, z) for all, or nq x 3.  -> (wp float64[nq, 3], global_goal float64[nq, 3], n_kept int32[nq])
    [+ kept cells int32[total, 2], aligned with the offsets of the paths, when return_kept]."""
    nq, off, cells = _batch_paths(planner, paths, int(nq))
    o = _vec(map_o, 2)
    p, g = _batch_vec(pos, nq), _batch_vec(global_goal, nq)
    eo = None if end_occu is None else np.ascontiguousarray(np.broadcast_to(np.asarray(end_occu, dtype=np.int32), (nq,)))
    wp, gout, nk = np.zeros((nq, 3)), np.zeros((nq, 3)), np.zeros(nq, dtype=np.int32)
    kept = None
    cap = 0
    if return_kept:
        if off is None:
            raise ValueError("return_kept needs the paths (their offsets place the kept cells)")
        cap = int(off[-1])
        kept = np.zeros((max(cap, 1), 2), dtype=np.int32)
    planner._chk(planner._L.fxjps_waypoint_ccst_batch(
        planner._h, nq, None if off is None else _lib.ptr(off, C.c_int64), None if cells is None else _lib.ptr(cells, C.c_int32),
        float(map_reso), _lib.ptr(o, C.c_double), _lib.ptr(p, C.c_double), _lib.ptr(g, C.c_double),
        None if eo is None else _lib.ptr(eo, C.c_int32), _lib.ptr(wp, C.c_double), _lib.ptr(gout, C.c_double), _lib.ptr(nk, C.c_int32),
        None if kept is None else _lib.ptr(kept, C.c_int32), cap))
    if return_kept:
        return wp, gout, nk, kept[:cap]
    return wp, gout, nk


def select_st_batch(planner, nq, map_start, map_reso, map_o, pos, global_goal, end_occu=None, prev_wp=None, prev_dim=None, paths=None,
                    dis_wp_tre=2.0, ang_wp_tre=math.pi / 4, nthreads=0):
    """global_planner_st.py:292-327 for every path of a batch, one wavefront per path on the device (fxjps_waypoint_st_batch;
    the angles come out of a table of the host's own atan2, filled by `nthreads` host threads once per range of map_start).
    -> (wp float64[nq, 3], dim int32[nq] (2 or 3 valid components), global_goal float64[nq, 3], ang_wp float64[nq])"""
    nq, off, cells = _batch_paths(planner, paths, int(nq))
    o = _vec(map_o, 2)
    p, g = _batch_vec(pos, nq), _batch_vec(global_goal, nq)
    ms = np.ascontiguousarray(np.broadcast_to(np.asarray(map_start, dtype=np.int32), (nq, 2)))
    eo = None if end_occu is None else np.ascontiguousarray(np.broadcast_to(np.asarray(end_occu, dtype=np.int32), (nq,)))
    pw = pd = None
    if prev_wp is not None:
        pw = np.ascontiguousarray(np.asarray(prev_wp, dtype=np.float64).reshape(nq, 3))
        pd = np.ascontiguousarray(np.asarray(prev_dim, dtype=np.int32).reshape(nq))
    wp, gout, dim, ang = np.zeros((nq, 3)), np.zeros((nq, 3)), np.zeros(nq, dtype=np.int32), np.zeros(nq)
    planner._chk(planner._L.fxjps_waypoint_st_batch(
        planner._h, nq, None if off is None else _lib.ptr(off, C.c_int64), None if cells is None else _lib.ptr(cells, C.c_int32),
        _lib.ptr(ms, C.c_int32), float(map_reso), _lib.ptr(o, C.c_double), _lib.ptr(p, C.c_double), _lib.ptr(g, C.c_double),
        None if eo is None else _lib.ptr(eo, C.c_int32), float(dis_wp_tre), float(ang_wp_tre),
        None if pw is None else _lib.ptr(pw, C.c_double), None if pd is None else _lib.ptr(pd, C.c_int32),
        _lib.ptr(wp, C.c_double), _lib.ptr(dim, C.c_int32), _lib.ptr(gout, C.c_double), _lib.ptr(ang, C.c_double), int(nthreads)))
    return wp, dim, gout, ang
