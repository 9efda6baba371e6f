#!/bin/bash
# A/B of library builds on lone queries: tools/ab_one.sh "<lib1> <lib2> ..." [query ids] [workload]
LIBS=$1; IDS=${2:-9206,606}; W=${3:-c2}
for l in $LIBS; do
  echo "== $l"; FXJPS_LIB=$PWD/fuxi-planner_amd/$l timeout -k 10 300 python tools/one_query.py $W $IDS 2 > /tmp/ab_one.$$ 2>&1; cut -c1-100 /tmp/ab_one.$$
  if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump" /tmp/ab_one.$$; then echo "GPU FAULT with $l: stopping"; exit 3; fi
done
