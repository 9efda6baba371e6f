#!/bin/bash
# One GPU-box session of round 3 (run through gpurun).  Usage: tools/gpu_round3.sh <tag> [steps...]
#   tests      the -m gpu suite
#   bench      the driver's default line (c2 + config.also), the config-4 anchor (1 M queries, one GPU), config 5 with
#              frames in flight over its 600 frames
#   lines      one bench line per workload
#   algo5      literal operation counts of the 616 config-5 frames (CPU work on the box's 256 threads)
#   profiles   per workload: rocprofv3 kernel stats + counter passes + bench line (tools/collect_profiles.py reads it)
#   busy       VALUBusy / SALUBusy passes of c2 and c4shard
TAG=${1:-r3}; shift
STEPS=${@:-tests bench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# a GPU fault ends the session: nothing else is started on a device that has just faulted
chk() { if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump" "$@" 2>/dev/null; then echo "GPU FAULT reported in $*: stopping"; exit 3; fi; }
for s in $STEPS; do
case $s in
tests)
  timeout 2400 python -m pytest tests -m gpu -q -s -x --durations=15 > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
  chk $OUT/pytest.log; if [ $rc -ne 0 ]; then exit 4; fi;;
quick)
  timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 -k "${FX_TESTS:-open_list or map_updates or partial or window or deferred or streaming or config5 or update_cells}" > $OUT/pytest_quick.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest_quick.log; tail -12 $OUT/pytest_quick.log
  chk $OUT/pytest_quick.log; if [ $rc -ne 0 ]; then exit 4; fi;;
bench)
  timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"; cut -c1-3000 $OUT/bench_default.json
  timeout 600 python bench.py --workload c4 --inlib --gpus 1 --steps 2 --warmup 1 > $OUT/bench_c4_n1.json 2> $OUT/bench_c4_n1.err; echo "c4 rc=$?"; cut -c1-1200 $OUT/bench_c4_n1.json
  timeout 900 python bench.py --workload c5pipe --no-cpu-baseline > $OUT/bench_c5pipe.json 2> $OUT/bench_c5pipe.err; echo "c5pipe rc=$?"; cut -c1-2000 $OUT/bench_c5pipe.json;;
lines)
  for w in ${FX_WORKLOADS:-c2 c2h1 c4shard c5 c5local c3}; do
    timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-also > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"; cut -c1-420 $OUT/bench_$w.json
  done;;
algo5)
  timeout 1500 python tools/algo_bytes.py c5 c5local4k > $OUT/algo_bytes.log 2>&1; cp fuxi-planner_amd/workloads.json $OUT/workloads.json; tail -3 $OUT/algo_bytes.log | cut -c1-300;;
profiles)
  for w in ${FX_PROFILE_WORKLOADS:-c2 c2h1 c4shard c5 c5local c3}; do
    P=$OUT/prof_$w; rm -rf $P; mkdir -p $P
    ST=3; [ $w = c3 ] && ST=1
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --workload $w --steps $ST --warmup 1 --no-cpu-baseline --no-also > $P/stats.log 2>&1
    for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      n=$(echo $c | cut -d' ' -f1)
      timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$n -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $P/pmc_$n.log 2>&1
    done
    timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-also > $P/bench.json 2> $P/bench.err
    python3 tools/profile_summary.py $P $w > $P/summary.json 2> $P/summary.err; cut -c1-400 $P/summary.json
  done;;
multi)
  # the two N > 1 code paths of bench.py on a one-GPU box: every rank / context on device 0 (everything but the collective)
  FXJPS_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_inlib2.json 2> $OUT/bench_inlib2.err; echo "inlib x2 rc=$?"; cut -c1-1500 $OUT/bench_inlib2.json
  chk $OUT/bench_inlib2.err
  FXJPS_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_torchrun2.json 2> $OUT/bench_torchrun2.err; echo "torchrun x2 (gloo, one device) rc=$?"; grep "^{" $OUT/bench_torchrun2.json | cut -c1-1500
  chk $OUT/bench_torchrun2.err;;
busy)
  for w in c2 c4shard; do
    P=$OUT/busy_$w; rm -rf $P
    timeout 600 rocprofv3 --kernel-trace --pmc VALUBusy SALUBusy --output-format csv -d $P -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $OUT/busy_$w.log 2>&1
    python3 tools/busy_report.py $P $w | tee $OUT/busy_$w.txt
  done;;
esac
done
