#!/bin/bash
# One GPU-box session of round 6 (run through gpurun).  Usage: tools/gpu_round6.sh <tag> [steps...]
#   every step of tools/gpu_round5.sh / gpu_round4.sh, plus:
#   isupply    instruction supply of the search kernel: I-cache requests / hits / misses, instruction fetches, cycles a wavefront
#              waits for an instruction, on c4shard-like (40 000 queries), c3-like (8 000 queries at 4096^2) and query 9206 alone
#   events     event counters of the diagnostic build on a full chip (trip counts of the data-dependent loops per iteration)
#   abl        A/B of library builds: FX_LIBS="libfxjps.so libfxjps_x.so ..." on FX_AB_WL (default "c4shard c2"), alternating,
#              FX_AB_REP repetitions (default 2); prints plans/s, ms per step, kernel ms
#   abone      the same on lone queries (FX_QIDS, default 9206,606)
#   abpmc      instruction counters per pop and per iteration of each library of FX_LIBS (40 000 queries, full chip)
#   smoke      FX_LIBS: 2 000 config-2 queries of each library against the oracle (cells, lengths, cost bytes)
#   multi6     the N = 2 rehearsals on one device: in-library and ranks, default line and streaming frames; self-verification
# Every step checks its logs for a GPU fault before the next program is started.
: ${GRAFT_REPO_ROOT:?must run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-r6}; shift
STEPS=${@:-tests bench}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 9
mkdir -p $OUT
chk() { if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU coredump\|Aborted (core dumped)" "$@" 2>/dev/null; then echo "GPU FAULT reported in $*: stopping"; exit 3; fi; }
pmc_pass() {  # name, counters, env..., -- program
  local n=$1 c=$2; shift 2
  local P=$OUT/pmc_$n; rm -rf $P
  env "$@" timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P -- python3 tools/gpu_prof.py --noprof ${FX_NQ:-40000} > $P.log 2>&1
  chk $P.log
  python3 tools/pmc_sum.py $P "$n" $P.log | tee -a $OUT/isupply.txt
  rm -rf $P
}
for s in $STEPS; do
case $s in
isupply)
  rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_WAIT_ANY\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INSTS_[A-Z_]*\|SQ_WAVES\b" | sort -u | tr '\n' ' ' > $OUT/counters_available.txt; cat $OUT/counters_available.txt; echo
  : > $OUT/isupply.txt
  for cfg in "c4like:FX_W=1024:40000" "c3like:FX_W=4096:8000" "q9206:FX_QIDS=9206:10000"; do
    IFS=: read name envv nq <<< "$cfg"
    for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
      g1=$(echo $grp | cut -d' ' -f1)
      FX_NQ=$nq pmc_pass ${name}_$g1 "$grp" $envv
    done
  done;;
events)
  timeout -k 10 400 python3 tools/gpu_prof.py ${FX_NQ:-40000} > $OUT/events_c4like.txt 2>&1; chk $OUT/events_c4like.txt; cat $OUT/events_c4like.txt
  FX_QIDS=9206 timeout -k 10 300 python3 tools/gpu_prof.py 10000 > $OUT/events_q9206.txt 2>&1; chk $OUT/events_q9206.txt; cat $OUT/events_q9206.txt;;
abl)
  for rep in $(seq 1 ${FX_AB_REP:-2}); do for w in ${FX_AB_WL:-c4shard c2}; do for l in ${FX_LIBS:-libfxjps.so}; do
    F=$OUT/ab_${w}_${l%.so}_$rep
    FXJPS_LIB=$PWD/fuxi-planner_amd/$l timeout -k 10 600 python bench.py --workload $w --steps ${FX_STEPS:-4} --warmup 1 --no-cpu-baseline --no-also > $F.json 2> $F.err
    chk $F.err
    echo "$l $w rep $rep: $(python3 -c "import json; d=json.loads(open('$F.json').read().strip().splitlines()[-1]); print(round(d['value']), 'plans/s', round(d['ms_per_step'],2), 'ms/step, kernel', round(d['roofline']['kernel_ms'],2), 'launches', d['roofline'].get('launch_ms'))" 2>&1 | tail -1)" | tee -a $OUT/ab.txt
  done; done; done;;
abone)
  for l in ${FX_LIBS:-libfxjps.so}; do
    echo "== $l" | tee -a $OUT/abone.txt
    FXJPS_LIB=$PWD/fuxi-planner_amd/$l timeout -k 10 300 python tools/one_query.py c2 ${FX_QIDS:-9206,606} 3 > $OUT/abone_${l%.so}.txt 2>&1; chk $OUT/abone_${l%.so}.txt
    cut -c1-110 $OUT/abone_${l%.so}.txt | tee -a $OUT/abone.txt
  done;;
abpmc)
  for l in ${FX_LIBS:-libfxjps.so}; do
    pl=${l%.so}_prof.so; [ -f fuxi-planner_amd/$pl ] || pl=none
    FXJPS_LIB=$PWD/fuxi-planner_amd/$l FX_PROF_LIB=$PWD/fuxi-planner_amd/$pl tools/gpu_pmc.sh ${l%.so} ${FX_NQ:-40000} > $OUT/abpmc_${l%.so}.txt 2>&1
    chk $OUT/abpmc_${l%.so}.txt gpurun_out/pmc_${l%.so}.log; cat $OUT/abpmc_${l%.so}.txt | tee -a $OUT/abpmc.txt
  done;;
smoke)
  for l in ${FX_LIBS:-libfxjps.so}; do
    FXJPS_LIB=$PWD/fuxi-planner_amd/$l timeout -k 10 300 python3 tools/lib_smoke.py ${FX_SMOKE_NQ:-2000} > $OUT/smoke_${l%.so}.txt 2>&1; rc=$?
    chk $OUT/smoke_${l%.so}.txt; echo "$l smoke rc=$rc: $(tail -1 $OUT/smoke_${l%.so}.txt)" | tee -a $OUT/smoke.txt
    if [ $rc -ne 0 ]; then echo "PARITY FAILURE with $l: stopping"; exit 4; fi
  done;;
multi6)
  # both `bench.py --gpus 2` code paths with every rank / context on device 0 (one-device rehearsal): the default line (c2 +
  # config 4 split two ways) and the streaming workload (the toggle list travels, the grid never does); every line must
  # carry config.verified (grid hashes equal, every shard sampled against the oracle) and cpu_baseline
  for wl in c2 c5; do
    extra=""; [ $wl = c5 ] && extra="--workload c5 --steps 3 --warmup 1"
    [ $wl = c2 ] && extra="--steps 2 --warmup 1"
    FXJPS_BENCH_ONE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 2 $extra > $OUT/multi_inlib2_$wl.json 2> $OUT/multi_inlib2_$wl.err; echo "inlib x2 $wl rc=$?"
    chk $OUT/multi_inlib2_$wl.err
    FXJPS_BENCH_ONE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 $extra > $OUT/multi_ranks2_$wl.json 2> $OUT/multi_ranks2_$wl.err; echo "ranks x2 $wl (one device) rc=$?"
    chk $OUT/multi_ranks2_$wl.err
    for f in $OUT/multi_inlib2_$wl.json $OUT/multi_ranks2_$wl.json; do
      python3 - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); c = d["config"]
    print(sys.argv[1].split("/")[-1], ": %.0f %s, verified %s, cpu_baseline %s, torch_imported %s, also.c4.verified %s, %s" % (
        d["value"], d["unit"], c.get("verified"), "yes" if "cpu_baseline" in d else "NO", c.get("torch_imported"),
        (c.get("also", {}).get("c4") or {}).get("verified"), c.get("rehearsal") or c.get("error") or ""))
except Exception as e:
    print(sys.argv[1], "no line:", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
    done
  done
  # the failure path: without the rehearsal flag an N = 2 line on a one-GPU box must FAIL (no RCCL communicator of two ranks)
  timeout -k 10 300 python bench.py --gpus 2 --steps 1 --warmup 0 --no-also --no-cpu-baseline > $OUT/multi_must_fail.json 2> $OUT/multi_must_fail.err; echo "inlib x2 without the rehearsal flag on one GPU: rc=$? (expected non-zero)";;
*)
  tools/gpu_round5.sh $TAG $s || exit $?;;
esac
done
