import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
os.environ["FXJPS_LIB"] = os.path.join(os.getcwd(), "fuxi-planner_amd", "libfxjps_hwid.so")
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
p = fx.Planner([0]); L = _lib.load()
L.fxjps_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
occ = synth.synth_grid(1024, 1024, 1, 0.2); p.set_grid_occ(occ)
s, g = synth.synth_queries(occ, 1, 10000)
for rep in range(3):
    p.plan_batch(s[9206:9207], g[9206:9207], 2, 1024)
    c = (C.c_uint64 * 64)(); L.fxjps_debug_counters(p._h, c)
    tm = p.timing()
    print("kernel %.2f ms, pops %d; takes %d, waited %.0f cycles each, late entries %.1f each, block %.1f entries each; refills by the searching wavefront itself %d" % (
        tm["search_kernel_ms"], tm["pops"], c[43], c[42] / max(c[43], 1), c[45] / max(c[43], 1), c[46] / max(c[43], 1), c[44]))
    for w in (0, 1):
        v = c[40 + w]; print("wave", w, "hw_id %08x" % v, "wave_id", v & 15, "simd", (v >> 4) & 3, "cu", (v >> 8) & 15, "sh", (v >> 12) & 1, "se", (v >> 13) & 7)
