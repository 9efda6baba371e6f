#!/bin/bash
# One GPU-box session of round 4 (run through gpurun).  Usage: tools/gpu_round4.sh <tag> [steps...]
#   tests      the -m gpu suite                       quick     a subset by -k (FX_TESTS)
#   bench      the driver's default line (c2 + config.also: c4shard, c3, c5pipe over its 600 frames, c1)
#   lines      one bench line per workload (FX_WORKLOADS)
#   one        the slowest queries of config 2 alone on the chip + single-call latencies
#   profiles   per workload: rocprofv3 kernel stats + counter passes + bench line (tools/collect_profiles.py reads it)
#   busy       VALUBusy / SALUBusy passes of c2 and c4shard
#   multi      both `bench.py --gpus 2` code paths with every rank / context on device 0
#   stress     tools/gpu_stress.py + tools/gpu_stress_updates.py (FX_STRESS_S seconds each)
# Every step checks its logs for a GPU fault before the next program is started: nothing runs on a device that has just faulted.
: ${GRAFT_REPO_ROOT:?must run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-r4}; shift
STEPS=${@:-tests bench}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 9
mkdir -p $OUT
chk() { if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump\|Aborted (core dumped)" "$@" 2>/dev/null; then echo "GPU FAULT reported in $*: stopping"; exit 3; fi; }
for s in $STEPS; do
case $s in
tests)
  timeout 2700 python -m pytest tests -m gpu -q -s -x --durations=15 > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
  chk $OUT/pytest.log; if [ $rc -ne 0 ]; then exit 4; fi;;
quick)
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 -k "${FX_TESTS:-parity or map_updates}" > $OUT/pytest_quick.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest_quick.log; tail -12 $OUT/pytest_quick.log
  chk $OUT/pytest_quick.log; if [ $rc -ne 0 ]; then exit 4; fi;;
bench)
  timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"; cut -c1-4000 $OUT/bench_default.json
  chk $OUT/bench_default.err;;
lines)
  for w in ${FX_WORKLOADS:-c2 c2h1 c4shard c5 c5local c3}; do
    timeout 900 python bench.py --workload $w --steps ${FX_STEPS:-5} --warmup 2 --no-also --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"; cut -c1-420 $OUT/bench_$w.json
    chk $OUT/bench_$w.err
  done;;
one)
  timeout 600 python tools/one_query.py c2 ${FX_QIDS:-9206,606,5866,1020} 2 > $OUT/one.txt 2>&1; cat $OUT/one.txt; chk $OUT/one.txt
  timeout 600 python tools/latency.py > $OUT/latency.txt 2>&1; cat $OUT/latency.txt; chk $OUT/latency.txt;;
profiles)
  for w in ${FX_PROFILE_WORKLOADS:-c2 c2h1 c4shard c5 c5local c3}; do
    P=$OUT/prof_$w; rm -rf $P; mkdir -p $P
    ST=3; [ $w = c3 ] && ST=1
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --workload $w --steps $ST --warmup 1 --no-cpu-baseline --no-also > $P/stats.log 2>&1
    chk $P/stats.log
    for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      n=$(echo $c | cut -d' ' -f1)
      timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$n -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $P/pmc_$n.log 2>&1
      chk $P/pmc_$n.log
    done
    timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-also > $P/bench.json 2> $P/bench.err
    chk $P/bench.err
    python3 tools/profile_summary.py $P $w > $P/summary.json 2> $P/summary.err; cut -c1-400 $P/summary.json
  done;;
multi)
  FXJPS_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_inlib2.json 2> $OUT/bench_inlib2.err; echo "inlib x2 rc=$?"; cut -c1-1500 $OUT/bench_inlib2.json
  chk $OUT/bench_inlib2.err
  FXJPS_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_torchrun2.json 2> $OUT/bench_torchrun2.err; echo "torchrun x2 (one device) rc=$?"; grep "^{" $OUT/bench_torchrun2.json | cut -c1-1500
  chk $OUT/bench_torchrun2.err;;
busy)
  for w in c2 c4shard; do
    P=$OUT/busy_$w; rm -rf $P
    timeout 600 rocprofv3 --kernel-trace --pmc VALUBusy SALUBusy --output-format csv -d $P -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $OUT/busy_$w.log 2>&1
    chk $OUT/busy_$w.log
    python3 tools/busy_report.py $P $w | tee $OUT/busy_$w.txt
  done;;
stress)
  timeout 900 python tools/gpu_stress.py ${FX_STRESS_S:-240} ${FX_SEED:-4} > $OUT/stress.txt 2>&1; tail -4 $OUT/stress.txt; chk $OUT/stress.txt
  FXJPS_DIRECT=0 timeout 900 python tools/gpu_stress.py $(( ${FX_STRESS_S:-240} * 2 / 3 )) $(( ${FX_SEED:-4} + 1 )) > $OUT/stress_hashed.txt 2>&1; tail -2 $OUT/stress_hashed.txt; chk $OUT/stress_hashed.txt
  FXJPS_COOP=1 timeout 900 python tools/gpu_stress.py $(( ${FX_STRESS_S:-240} / 2 )) $(( ${FX_SEED:-4} + 2 )) > $OUT/stress_coop.txt 2>&1; tail -2 $OUT/stress_coop.txt; chk $OUT/stress_coop.txt
  timeout 900 python tools/gpu_stress_updates.py ${FX_STRESS_S:-240} ${FX_SEED:-4} > $OUT/stress_updates.txt 2>&1; tail -4 $OUT/stress_updates.txt; chk $OUT/stress_updates.txt;;
esac
done
