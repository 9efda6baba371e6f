#!/bin/bash
# Cycles per phase of the search loop from one-interval builds (-DFXJPS_PHASE_S / _E: two clock reads per iteration).
#   tools/phase_times.sh build "9:8 8:0 0:5 ..."      (here: cross-compiles fuxi-planner_amd/libfxjps_ph_<s>_<e>.so, 4 at a time)
#   tools/phase_times.sh run "9:8 8:0 ..." [query id]  (on the GPU box)
MODE=$1; IV=${2:-"9:8 8:0 0:5 5:1 1:7 7:2 7:12 12:13 13:9"}; Q=${3:-9206}
cd "$(dirname "$0")/.."
if [ "$MODE" = build ]; then
  n=0
  for iv in $IV; do s=${iv%%:*}; e=${iv#*:}
    ( cd fuxi-planner_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -shared -fvisibility=hidden $FX_PHASE_FLAGS -DFXJPS_PHASE_S=$s -DFXJPS_PHASE_E=$e -o libfxjps_ph_${s}_${e}.so csrc/fxjps_waypoints.o csrc/fxjps.hip -Wl,--version-script=csrc/exports.map -ldl 2>&1 | grep -E "error" ) &
    n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
  done; wait; ls fuxi-planner_amd/libfxjps_ph_*.so
else
  for iv in $IV; do s=${iv%%:*}; e=${iv#*:}
    FXJPS_COOP=0 FXJPS_LIB=$PWD/fuxi-planner_amd/libfxjps_ph_${s}_${e}.so python3 - $s $e $Q <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import fuxi_planner_amd as fx
from fuxi_planner_amd import synth, _lib
s_, e_, q = sys.argv[1], sys.argv[2], int(sys.argv[3])
p = fx.Planner([0]); L = _lib.load()
L.fxjps_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
occ = synth.synth_grid(1024, 1024, 1, 0.2); p.set_grid_occ(occ)
s, g = synth.synth_queries(occ, 1, 10000)
for rep in range(2):
    p.plan_batch(s[q:q + 1], g[q:q + 1], 2, 1024)
tm = p.timing(); c = (C.c_uint64 * 64)(); L.fxjps_debug_counters(p._h, c)
print("interval %2s -> %2s: %8.1f cycles per pop (%5.1f %% of the kernel's %.2f ms at 2.4 GHz), kernel %.2f ms" % (
    s_, e_, c[8] / tm["pops"], 100.0 * c[8] / (tm["search_kernel_ms"] * 2.4e6), tm["search_kernel_ms"], tm["search_kernel_ms"]))
PY
  done
fi
