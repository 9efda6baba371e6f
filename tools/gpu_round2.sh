#!/bin/bash
# One GPU-box session of round 2 (run through gpurun): the GPU suite, literal operation counts of every bench workload
# (CPU work, but the box has 256 threads), the FETCH_SIZE / WRITE_SIZE calibration, baseline bench lines.
# Usage: tools/gpu_round2.sh <tag> [steps...]   steps: tests algo calib bench pmc
TAG=${1:-r2}; shift
STEPS=${@:-tests algo calib bench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in $STEPS; do
case $s in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log;;
algo)
  timeout 1500 python tools/algo_bytes.py > $OUT/algo_bytes.log 2>&1; cp fuxi-planner_amd/workloads.json $OUT/workloads.json; tail -8 $OUT/algo_bytes.log | cut -c1-300;;
calib)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/calib_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -- ./tools/calib_scatter > $OUT/calib_$c.log 2>&1
  done
  python3 tools/calib_report.py $OUT > $OUT/calib_report.txt 2>&1; cat $OUT/calib_report.txt;;
bench)
  for w in c2 c2h1 c4shard c5 c5low c3; do
    timeout 900 python bench.py --workload $w --steps 5 --warmup 2 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"; cut -c1-420 $OUT/bench_$w.json
  done;;
pmc)
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    rm -rf $OUT/pmc_$n
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$n.log 2>&1
  done
  rm -rf $OUT/stats
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
  python3 tools/pmc_report.py $OUT > $OUT/pmc_report.txt 2>&1; cat $OUT/pmc_report.txt;;
profiles)
  # one kernel-stats run + instruction / HBM counter passes per bench workload, on the build that is in the tree
  for w in ${FX_PROFILE_WORKLOADS:-c2 c2h1 c4shard c5 c5local c3}; do
    P=$OUT/prof_$w; rm -rf $P; mkdir -p $P
    ST=3; [ $w = c3 ] && ST=1
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --workload $w --steps $ST --warmup 1 --no-cpu-baseline > $P/stats.log 2>&1
    for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      n=$(echo $c | cut -d' ' -f1)
      timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$n -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $P/pmc_$n.log 2>&1
    done
    timeout 900 python bench.py --workload $w --steps 5 --warmup 2 > $P/bench.json 2> $P/bench.err
    python3 tools/profile_summary.py $P $w > $P/summary.json 2> $P/summary.err; cut -c1-400 $P/summary.json
  done;;
esac
done
