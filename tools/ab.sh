#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh "<lib1> <lib2> ..." "<workloads>" [steps]
# Prints plans/s and ms per step of bench.py for each (library, workload); alternates the libraries per workload.
LIBS=$1; WLS=${2:-"c2 c4shard"}; ST=${3:-4}
mkdir -p gpurun_out/ab
for w in $WLS; do for l in $LIBS; do
  FXJPS_LIB=$PWD/fuxi-planner_amd/$l timeout -k 10 600 python bench.py --workload $w --steps $ST --warmup 1 --no-cpu-baseline > gpurun_out/ab/${w}_$l.json 2> gpurun_out/ab/${w}_$l.err
  if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump" gpurun_out/ab/${w}_$l.err 2>/dev/null; then echo "GPU FAULT with $l on $w: stopping"; exit 3; fi
  echo "$l $w $(python3 -c "import json; d=json.load(open('gpurun_out/ab/${w}_$l.json')); print(round(d['value']), round(d['ms_per_step'],1))" 2>&1 | tail -1)"
done; done
