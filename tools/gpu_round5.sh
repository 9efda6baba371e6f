#!/bin/bash
# One GPU-box session of round 5 (run through gpurun).  Usage: tools/gpu_round5.sh <tag> [steps...]
#   every step of tools/gpu_round4.sh (tests quick bench lines one profiles busy multi stress), plus:
#   cumask     tools/cumask_probe: where the blocks of CU-masked launches land (XCD / CU)
#   headxcd    A/B of FXJPS_HEAD_XCC x FXJPS_SOLO on config 2 (FX_HEAD="<0|1>:<solo> ..."; needs `make -C fuxi-planner_amd libfxjps_xcc.so`)
#   queues     c5pipe under rocprofv3 --kernel-trace for "<queues>:<K>[:one]" settings (FX_QK), tools/queue_map.py on each
#   window     the 64 x 64 window update at 4096^2, walk / stream (tests/test_map_updates_gpu.py prints it), its kernels
#   proxy2q    instruction counts and commits per iteration of batches of 16 and of 8 nodes (two queries per wavefront, by proxy)
#   clock      tools/clock_probe.py: cycles and wall time of the longest queries, alone and inside their batch
# Every step checks its logs for a GPU fault before the next program is started.
: ${GRAFT_REPO_ROOT:?must run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-r5}; shift
STEPS=${@:-tests bench}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 9
mkdir -p $OUT
chk() { if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump\|Aborted (core dumped)" "$@" 2>/dev/null; then echo "GPU FAULT reported in $*: stopping"; exit 3; fi; }
for s in $STEPS; do
case $s in
cumask)
  timeout -k 10 120 tools/cumask_probe > $OUT/cumask.txt 2>&1; echo "cumask rc=$?"; cat $OUT/cumask.txt; chk $OUT/cumask.txt;;
headxcd)
  for hs in ${FX_HEAD:-0:16 1:16 1:24 1:32 0:32}; do
    x=${hs%%:*}; n=${hs##*:}
    FXJPS_LIB=$PWD/fuxi-planner_amd/libfxjps_xcc.so FXJPS_HEAD_XCC=$x FXJPS_SOLO=$n timeout -k 10 300 python bench.py --workload ${FX_HEAD_WL:-c2} --steps ${FX_STEPS:-8} --warmup 3 --no-also --no-cpu-baseline > $OUT/head_${x}_$n.json 2> $OUT/head_${x}_$n.err; rc=$?
    chk $OUT/head_${x}_$n.err
    python3 - $OUT/head_${x}_$n.json $x $n $rc <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("head xcds %s solo %s: %.1f k plans/s, %.2f ms/step, kernel %.2f ms, launches %s" % (sys.argv[2], sys.argv[3], d["value"] / 1e3, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["launch_ms"]))
except Exception as e:
    print("head xcds %s solo %s: rc %s, no line (%r)" % (sys.argv[2], sys.argv[3], sys.argv[4], e))
PY
  done;;
queues)
  for qk in ${FX_QK:-16:12 16:16}; do
    IFS=: read q k one <<< "$qk"
    P=$OUT/queues_${q}_${k}${one:+_one}; rm -rf $P
    GPU_MAX_HW_QUEUES=$q FXJPS_ONE_STREAM=${one:+1} timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $P -- python3 bench.py --workload c5pipe --steps ${FX_QSTEPS:-240} --warmup 24 --frames-in-flight $k --no-also --no-cpu-baseline > $P.json 2> $P.err
    chk $P.err
    python3 - $P.json $q $k "$one" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); c = d["config"]
    print("queues", sys.argv[2], "K", sys.argv[3], sys.argv[4], ":", round(c["frames_per_s"], 1), "frames/s (under the tracer)", {a: round(b, 1) for a, b in c["submit_to_paths_latency_ms"].items()})
except Exception as e:
    print("no line", repr(e))
PY
    python3 tools/queue_map.py $P > $P.map.txt 2>&1; cat $P.map.txt
    rm -rf $P  # (the traces are large; the map is what is kept)
  done;;
window)
  timeout -k 10 600 python -m pytest tests/test_map_updates_gpu.py -m gpu -q -s -x -k "window" > $OUT/window.log 2>&1; grep "window update\|passed\|failed" $OUT/window.log
  FXJPS_JD_WALK=0 timeout -k 10 600 python -m pytest tests/test_map_updates_gpu.py -m gpu -q -s -x -k "window" > $OUT/window_stream.log 2>&1; grep "window update\|passed\|failed" $OUT/window_stream.log
  chk $OUT/window.log $OUT/window_stream.log
  rm -rf $OUT/window_prof; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/window_prof -- python3 -m pytest tests/test_map_updates_gpu.py -m gpu -q -x -k "window" > $OUT/window_prof.log 2>&1
  chk $OUT/window_prof.log
  python3 - $OUT/window_prof <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("kernels of the window updates (rocprofv3 --stats): name, calls, average us")
    for r in rows[:14]:
        print("  %-60s %6s %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/window_prof;;
proxy2q)
  # Two queries per wavefront, by proxy: what a batch of 8 nodes commits and what its iteration costs,
  # beside the batch of 16 -- on a full chip (40 000 queries of config 4's stream: no head launch).  The variant libraries
  # are built with -DFXJPS_KN=8 (tools/README.md).
  tools/gpu_pmc.sh kn16 ${FX_PROXY_NQ:-40000} > $OUT/pmc_kn16.txt 2>&1; cat $OUT/pmc_kn16.txt; chk $OUT/pmc_kn16.txt gpurun_out/pmc_kn16.log
  FXJPS_LIB=$PWD/fuxi-planner_amd/libfxjps_kn8.so FX_PROF_LIB=$PWD/fuxi-planner_amd/libfxjps_kn8_prof.so tools/gpu_pmc.sh kn8 ${FX_PROXY_NQ:-40000} > $OUT/pmc_kn8.txt 2>&1; cat $OUT/pmc_kn8.txt; chk $OUT/pmc_kn8.txt gpurun_out/pmc_kn8.log
  for l in "" kn8; do
    FXJPS_LIB=${l:+$PWD/fuxi-planner_amd/libfxjps_$l.so} timeout -k 10 300 python bench.py --workload c4shard --steps 3 --warmup 1 --no-also --no-cpu-baseline 2> $OUT/c4shard_${l:-kn16}.err | cut -c1-200
    chk $OUT/c4shard_${l:-kn16}.err
  done;;
clock)
  timeout -k 10 300 python tools/clock_probe.py c2 6 > $OUT/clock.txt 2>&1; cat $OUT/clock.txt; chk $OUT/clock.txt;;
*)
  tools/gpu_round4.sh $TAG $s || exit $?;;
esac
done
