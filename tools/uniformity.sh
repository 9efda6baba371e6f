#!/bin/bash
# Counts the branches of fx::k_search<2> that LLVM's uniformity analysis treats as divergent (each one costs
# exec-mask bookkeeping and keeps loop-carried state in VGPRs).  Usage: tools/uniformity.sh [kernels.inc] [extra flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
INC=${1:-$ROOT/fuxi-planner_amd/csrc/fxjps_kernels.hip.inc}
shift || true
W=${TMPDIR:-/tmp}/fxuni.$$
mkdir -p $W
sed "s#fxjps_kernels.hip.inc#$INC#; s#\"../../include/fxjps.h\"#\"$ROOT/include/fxjps.h\"#" $ROOT/fuxi-planner_amd/csrc/fxjps.hip > $W/fx.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -S -emit-llvm --cuda-device-only -o $W/fx.ll $W/fx.hip
/opt/rocm/lib/llvm/bin/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes='print<uniformity>' -disable-output $W/fx.ll 2> $W/uni.txt
awk '/UniformityInfo for function .*k_searchILi2ELb0ELb1E/{f=1;next} /UniformityInfo for function/{f=0} f' $W/uni.txt > $W/u2.txt
echo "k_search<2, false, true>: divergent br $(grep -c 'DIVERGENT:.* br i1' $W/u2.txt), uniform br $(grep ' br i1' $W/u2.txt | grep -vc DIVERGENT), divergent phi $(grep 'DIVERGENT' $W/u2.txt | grep -c ' phi ')"
echo "dump: $W/u2.txt"
