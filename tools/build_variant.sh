#!/bin/bash
# tools/build_variant.sh <name> [-Dmacro ...]   ->  fuxi-planner_amd/libfxjps_<name>.so (cross-compiles here, gfx950)
# The A/B builds of a round: one library per candidate, compared on the GPU box with tools/gpu_round6.sh abl / abone / abpmc.
cd "$(dirname "$0")/../fuxi-planner_amd" || exit 9
n=$1; shift
[ -f csrc/fxjps_waypoints.o ] || make -s csrc/fxjps_waypoints.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function "$@" \
  -shared -o libfxjps_$n.so csrc/fxjps_waypoints.o csrc/fxjps.hip -Wl,--version-script=csrc/exports.map -ldl 2>&1 | grep -E "error|warning: v" ; ls -la libfxjps_$n.so | awk '{print $5, $9}'
