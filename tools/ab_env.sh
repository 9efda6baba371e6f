#!/bin/bash
# A/B of run-time knobs (environment variables read by libfxjps.so) on the GPU box:
#   tools/ab_env.sh "<name>:<VAR=val,VAR=val> ..." "<workloads>" [steps]       e.g. "base: lazy64:FXJPS_LAZY_ADD=64"
# Prints plans/s, ms per step and kernel ms of bench.py for each (setting, workload).
SETS=$1; WLS=${2:-"c2 c4shard"}; ST=${3:-4}
mkdir -p gpurun_out/abenv
for w in $WLS; do for s in $SETS; do
  n=${s%%:*}; e=${s#*:}; e=${e//,/ }
  env $e timeout -k 10 600 python bench.py --workload $w --steps $ST --warmup 1 --no-cpu-baseline --no-also > gpurun_out/abenv/${w}_$n.json 2> gpurun_out/abenv/${w}_$n.err
  echo "$n $w $(python3 -c "import json; d=json.loads(open('gpurun_out/abenv/${w}_$n.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1))" 2>&1 | tail -1)"
done; done
