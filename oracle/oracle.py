"""ctypes binding of oracle/liboracle_jps.so -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

The library is a plain-C restatement of the reference's scripts/jps1.py:1-246
(see oracle/jps_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the shipped planner never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_jps.so")


class Stats(C.Structure):
    _fields_ = [("cells", C.c_int64), ("pushes", C.c_int64), ("pops", C.c_int64),
                ("open_peak", C.c_int64), ("jumps", C.c_int64)]


STATS_DTYPE = np.dtype([("cells", "<i8"), ("pushes", "<i8"), ("pops", "<i8"),
                        ("open_peak", "<i8"), ("jumps", "<i8")])


def build(force=False):
    """Compile the checker with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "jps_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle_jps.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        p_u8 = C.POINTER(C.c_uint8)
        p_i32 = C.POINTER(C.c_int32)
        p_f64 = C.POINTER(C.c_double)
        L.fxo_version.restype = C.c_int
        L.fxo_plan.restype = C.c_int
        L.fxo_plan.argtypes = [p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, p_i32, C.c_int32, p_f64,
                               C.POINTER(Stats)]
        L.fxo_plan_batch.restype = C.c_int
        L.fxo_plan_batch.argtypes = [p_u8, C.c_int32, C.c_int32, p_i32, p_i32, C.c_int64,
                                     C.c_int32, C.c_int32, C.c_int32, p_i32, p_i32, p_f64,
                                     C.c_void_p, C.c_int32]
        L.fxo_splitmix64.restype = C.c_uint64
        L.fxo_splitmix64.argtypes = [C.c_uint64]
        L.fxo_synth_grid.restype = None
        L.fxo_synth_grid.argtypes = [p_u8, C.c_int32, C.c_int32, C.c_uint64, C.c_double]
        L.fxo_synth_queries.restype = None
        L.fxo_synth_queries.argtypes = [p_u8, C.c_int32, C.c_int32, C.c_uint64, C.c_int64,
                                        C.c_int64, p_i32, p_i32]
        _lib = L
    return _lib


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def as_occ(matrix):
    """Obstacle iff the element compares equal to 1 (jps1.py:20-29)."""
    return np.ascontiguousarray(np.asarray(matrix) == 1, dtype=np.uint8)


def plan(matrix, start, goal, hchoice=2, literal=True, max_len=None):
    """-> (path list[(x, y)] | 0, cost float, stats dict)."""
    occ = as_occ(matrix)
    W, H = occ.shape
    if max_len is None:
        max_len = W * H + 1 if W * H < 1 << 16 else 1 << 16
    out = np.zeros((max_len, 2), dtype=np.int32)
    cost = C.c_double(0.0)
    st = Stats()
    n = lib().fxo_plan(_ptr(occ, C.c_uint8), W, H, int(start[0]), int(start[1]), int(goal[0]),
                       int(goal[1]), int(hchoice), 1 if literal else 0, _ptr(out, C.c_int32),
                       max_len, C.byref(cost), C.byref(st))
    stats = {k: getattr(st, k) for k, _ in Stats._fields_}
    if n < 0:
        raise ValueError("oracle error %d" % n)
    if n == 0:
        return 0, 0.0, stats
    return [(int(x), int(y)) for x, y in out[:n]], cost.value, stats


def plan_batch(matrix, starts, goals, hchoice=2, literal=False, max_len=1024, nthreads=1,
               want_stats=False):
    """-> (cells int32[nq,max_len,2], length int32[nq], cost f64[nq], stats | None)."""
    occ = as_occ(matrix)
    W, H = occ.shape
    starts = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, 2)
    goals = np.ascontiguousarray(goals, dtype=np.int32).reshape(-1, 2)
    nq = starts.shape[0]
    cells = np.zeros((nq, max_len, 2), dtype=np.int32)
    length = np.zeros(nq, dtype=np.int32)
    cost = np.zeros(nq, dtype=np.float64)
    stats = np.zeros(nq, dtype=STATS_DTYPE) if want_stats else None
    lib().fxo_plan_batch(_ptr(occ, C.c_uint8), W, H, _ptr(starts, C.c_int32),
                         _ptr(goals, C.c_int32), nq, int(hchoice), 1 if literal else 0,
                         int(max_len), _ptr(cells, C.c_int32), _ptr(length, C.c_int32),
                         _ptr(cost, C.c_double),
                         stats.ctypes.data_as(C.c_void_p) if want_stats else None, int(nthreads))
    return cells, length, cost, stats


def synth_grid(W, H, seed, p=0.20):
    occ = np.zeros((W, H), dtype=np.uint8)
    lib().fxo_synth_grid(_ptr(occ, C.c_uint8), W, H, int(seed), float(p))
    return occ


def synth_queries(occ, qseed, n, first=0):
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    W, H = occ.shape
    s = np.zeros((n, 2), dtype=np.int32)
    g = np.zeros((n, 2), dtype=np.int32)
    lib().fxo_synth_queries(_ptr(occ, C.c_uint8), W, H, int(qseed), int(first), int(n),
                            _ptr(s, C.c_int32), _ptr(g, C.c_int32))
    return s, g
