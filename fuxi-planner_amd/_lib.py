"""ctypes binding of libfxjps.so (include/fxjps.h).  No CPU fallback: if the HIP
library cannot be loaded, or no MI355X is visible, every entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FXJPS_LIB: bring-up override (e.g. the traced build); the package default is the in-tree library
LIB_PATH = os.environ.get("FXJPS_LIB") or os.path.join(_HERE, "libfxjps.so")

OK = 0
E_ARG, E_NODEV, E_HIP, E_NOGRID, E_NOMEM, E_COMM = -1, -2, -3, -4, -5, -6
Q_NOPATH, Q_PATH_TOO_LONG, Q_BAD_START, Q_CAPACITY = 0, -1, -2, -3
BACKEND_HIP = 1

# every symbol include/fxjps.h declares (tests check the .so exports all of them)
VERSION = 600  # FXJPS_VERSION of include/fxjps.h
SYMBOLS = ("fxjps_version", "fxjps_timing_size", "fxjps_last_timing_sized", "fxjps_rank_preflight", "fxjps_reserve_grid",
           "fxjps_device_count", "fxjps_create", "fxjps_rank_unique_id", "fxjps_create_rank", "fxjps_set_grid_rank", "fxjps_destroy", "fxjps_last_error",
           "fxjps_set_grid", "fxjps_set_grid_device", "fxjps_prepare_grid", "fxjps_prepare_occupancy_msg", "fxjps_get_grid", "fxjps_get_grid_context", "fxjps_publish_map", "fxjps_set_grid_image", "fxjps_snapshot_image", "fxjps_update_cells", "fxjps_update_cells_deferred", "fxjps_set_queries", "fxjps_replan_frame", "fxjps_plan_batch",
           "fxjps_plan_batch_csr", "fxjps_last_cells", "fxjps_last_timing", "fxjps_last_timing_device", "fxjps_comm_info", "fxjps_set_memory_share", "fxjps_selftest_sqrt", "fxjps_selftest_wavemin", "fxjps_selftest_openlist", "fxjps_debug_read_nbmask", "fxjps_debug_read_maps", "fxjps_debug_counters", "fxjps_debug_qstat",
           "fxjps_waypoint_st", "fxjps_waypoint_ccst", "fxjps_waypoint_ccst_batch", "fxjps_waypoint_st_batch")


class Timing(C.Structure):
    _fields_ = [("search_kernel_ms", C.c_double), ("total_ms", C.c_double), ("search_launches", C.c_int64),
                ("retried", C.c_int64), ("pops", C.c_int64), ("pushes", C.c_int64), ("far_refills", C.c_int64),
                ("slow_pops", C.c_int64), ("table_wipes", C.c_int64), ("reused", C.c_int64), ("table_direct", C.c_int64),
                ("waves", C.c_int64), ("waves_short", C.c_int64), ("head_launch_ms", C.c_double), ("batch_launch_ms", C.c_double),
                ("solo_timeouts", C.c_int64)]


class FxjpsError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "fxjps error %d: %s" % (code, msg))
        self.code = code


_lib = None


def bind_host(L):
    """Prototypes of the host-only entry points (csrc/fxjps_waypoints.cpp); also used on the sanitizer build of that
    translation unit by the CPU test-suite."""
    p_i32 = C.POINTER(C.c_int32)
    p_u8 = C.POINTER(C.c_uint8)
    p_f64 = C.POINTER(C.c_double)
    L.fxjps_waypoint_st.restype = C.c_int
    L.fxjps_waypoint_st.argtypes = [p_i32, C.c_int32, p_i32, C.c_double, p_f64, p_f64, p_f64, C.c_int32, C.c_double, C.c_double,
                                    p_f64, C.c_int32, p_f64, p_i32, p_f64, p_f64]
    L.fxjps_waypoint_ccst.restype = C.c_int
    L.fxjps_waypoint_ccst.argtypes = [p_i32, C.c_int32, p_u8, C.c_int32, C.c_int32, C.c_double, p_f64, p_f64, p_f64, C.c_int32, p_f64, p_f64,
                                      p_i32, p_i32]
    return L


def load():
    """Load libfxjps.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FxjpsError(E_NODEV, "%s is missing: build it with `make -C %s` (hipcc, gfx950); "
                         "there is no CPU fallback" % (LIB_PATH, _HERE))
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    p_i32 = C.POINTER(C.c_int32)
    p_i64 = C.POINTER(C.c_int64)
    p_u8 = C.POINTER(C.c_uint8)
    p_f64 = C.POINTER(C.c_double)
    L.fxjps_version.restype = C.c_int
    L.fxjps_timing_size.restype = C.c_int
    # (neither call touches a device)
    if L.fxjps_version() != VERSION or L.fxjps_timing_size() != C.sizeof(Timing):
        raise FxjpsError(E_ARG, "%s is version %d with a %d-byte timing record; this binding is for version %d / %d bytes: rebuild it"
                         % (LIB_PATH, L.fxjps_version(), L.fxjps_timing_size(), VERSION, C.sizeof(Timing)))
    L.fxjps_last_timing_sized.restype = C.c_int
    L.fxjps_last_timing_sized.argtypes = [vp, vp, C.c_int64]
    L.fxjps_rank_preflight.restype = C.c_int
    L.fxjps_rank_preflight.argtypes = [C.c_int]
    L.fxjps_reserve_grid.restype = C.c_int
    L.fxjps_reserve_grid.argtypes = [vp, C.c_int32, C.c_int32]
    L.fxjps_device_count.restype = C.c_int
    L.fxjps_create.restype = C.c_int
    L.fxjps_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.fxjps_destroy.restype = None
    L.fxjps_destroy.argtypes = [vp]
    L.fxjps_last_error.restype = C.c_char_p
    L.fxjps_last_error.argtypes = [vp]
    L.fxjps_set_grid.restype = C.c_int
    L.fxjps_set_grid.argtypes = [vp, p_u8, C.c_int32, C.c_int32]
    L.fxjps_set_grid_device.restype = C.c_int
    L.fxjps_set_grid_device.argtypes = [vp, vp, C.c_int32, C.c_int32]
    L.fxjps_prepare_grid.restype = C.c_int
    L.fxjps_prepare_grid.argtypes = [vp, p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32, p_i32, p_i32, p_i32, p_i32, p_i32, p_i32]
    L.fxjps_prepare_occupancy_msg.restype = C.c_int
    L.fxjps_prepare_occupancy_msg.argtypes = [vp, C.POINTER(C.c_int8), C.c_int32, C.c_int32, C.c_int32, C.c_int32, p_i32, p_i32,
                                              p_i32, p_i32, p_i32, p_i32]
    L.fxjps_publish_map.restype = C.c_int
    L.fxjps_publish_map.argtypes = [vp, C.POINTER(C.c_int8), p_i32, p_i32]
    L.fxjps_set_grid_image.restype = C.c_int
    L.fxjps_set_grid_image.argtypes = [vp, p_u8, C.c_int32, C.c_int32]
    L.fxjps_snapshot_image.restype = C.c_int
    L.fxjps_snapshot_image.argtypes = [vp, p_u8, C.c_int32, p_i32, p_i32]
    L.fxjps_get_grid.restype = C.c_int
    L.fxjps_get_grid.argtypes = [vp, p_u8, p_i32, p_i32]
    L.fxjps_get_grid_context.restype = C.c_int
    L.fxjps_get_grid_context.argtypes = [vp, C.c_int32, p_u8, p_i32, p_i32]
    L.fxjps_update_cells.restype = C.c_int
    L.fxjps_update_cells.argtypes = [vp, p_i32, p_u8, C.c_int64]
    L.fxjps_update_cells_deferred.restype = C.c_int
    L.fxjps_update_cells_deferred.argtypes = [vp, p_i32, p_u8, C.c_int64]
    L.fxjps_set_queries.restype = C.c_int
    L.fxjps_set_queries.argtypes = [vp, p_i32, p_i32, C.c_int64, C.c_int32, C.c_int32]
    L.fxjps_replan_frame.restype = C.c_int
    L.fxjps_replan_frame.argtypes = [vp, p_i32, p_u8, C.c_int64, p_i64, p_i32, C.c_int64, p_i32, p_f64, p_f64]
    L.fxjps_plan_batch.restype = C.c_int
    L.fxjps_plan_batch.argtypes = [vp, p_i32, p_i32, C.c_int64, C.c_int32, C.c_int32, p_i32, p_i32, p_f64, p_f64]
    L.fxjps_plan_batch_csr.restype = C.c_int
    L.fxjps_plan_batch_csr.argtypes = [vp, p_i32, p_i32, C.c_int64, C.c_int32, C.c_int32, p_i64, p_i32,
                                       C.c_int64, p_i32, p_f64, p_f64]
    L.fxjps_last_cells.restype = C.c_int
    L.fxjps_last_cells.argtypes = [vp, p_i32, C.c_int64]
    L.fxjps_last_timing.restype = C.c_int
    L.fxjps_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.fxjps_last_timing_device.restype = C.c_int
    L.fxjps_last_timing_device.argtypes = [vp, C.c_int32, p_i32, p_i64, p_f64, p_i64]
    L.fxjps_rank_unique_id.restype = C.c_int
    L.fxjps_rank_unique_id.argtypes = [vp]
    L.fxjps_create_rank.restype = C.c_int
    L.fxjps_create_rank.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.fxjps_set_grid_rank.restype = C.c_int
    L.fxjps_set_grid_rank.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
    L.fxjps_comm_info.restype = C.c_int
    L.fxjps_comm_info.argtypes = [vp, p_i32, p_i32, p_i32]
    L.fxjps_set_memory_share.restype = C.c_int
    L.fxjps_set_memory_share.argtypes = [vp, C.c_int32]
    L.fxjps_selftest_sqrt.restype = C.c_int
    L.fxjps_selftest_sqrt.argtypes = [vp, C.c_uint32, C.c_uint32, p_f64]
    L.fxjps_selftest_wavemin.restype = C.c_int
    L.fxjps_selftest_wavemin.argtypes = [vp, C.c_int32, C.c_uint64, p_i64]
    L.fxjps_debug_read_maps.restype = C.c_int
    L.fxjps_debug_read_maps.argtypes = [vp, C.c_int32, vp, C.c_int64, p_i64]
    p_u32, p_u64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.fxjps_selftest_openlist.restype = C.c_int
    L.fxjps_selftest_openlist.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, p_u64, p_u32, C.c_int64, p_u32, p_u32, C.c_int32,
                                          p_u64, p_u32, p_u32, p_u32, p_u32]
    L.fxjps_debug_read_nbmask.restype = C.c_int
    L.fxjps_debug_read_nbmask.argtypes = [vp, p_u8]
    L.fxjps_waypoint_ccst_batch.restype = C.c_int
    L.fxjps_waypoint_ccst_batch.argtypes = [vp, C.c_int64, p_i64, p_i32, C.c_double, p_f64, p_f64, p_f64, p_i32, p_f64, p_f64, p_i32, p_i32,
                                            C.c_int64]
    L.fxjps_waypoint_st_batch.restype = C.c_int
    L.fxjps_waypoint_st_batch.argtypes = [vp, C.c_int64, p_i64, p_i32, p_i32, C.c_double, p_f64, p_f64, p_f64, p_i32, C.c_double, C.c_double,
                                          p_f64, p_i32, p_f64, p_i32, p_f64, p_f64, C.c_int32]
    bind_host(L)
    _lib = L
    return L


def ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))
