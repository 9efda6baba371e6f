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
