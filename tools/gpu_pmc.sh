#!/bin/bash
# Instruction counters of one launch of the search kernel (run on the GPU box): VALU / SALU / LDS / VMEM instructions
# per pop and per loop iteration.  Usage: [FX_QIDS=9206] tools/gpu_pmc.sh [tag] [nq]   (honours FXJPS_LIB)
# The pops come from the run's own counters; iterations from a second run on the diagnostic build when it exists
# (FX_PROF_LIB: the diagnostic build that goes with FXJPS_LIB, e.g. a -DFXJPS_KN=8 pair).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-cur}
NQ=${2:-10000}
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 tools/gpu_prof.py --noprof $NQ > $OUT.log 2>&1
BATCHES=0
if [ -f "${FX_PROF_LIB:-fuxi-planner_amd/libfxjps_prof.so}" ]; then BATCHES=$(timeout -k 10 200 python3 tools/gpu_prof.py $NQ 2>/dev/null | sed -n 's/.*batches \([0-9]*\):.*/\1/p' | head -1); fi
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$f" "$TAG" "$OUT.log" "${BATCHES:-0}" <<'PY'
import csv,sys,re
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_search' in r['Kernel_Name']]
last=max(int(r['Dispatch_Id']) for r in rows)  # last launch of the run (the timed repetition)
c={r['Counter_Name']:float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last}
pops=float(re.search(r'pops (\d+)', open(sys.argv[3]).read()).group(1))
nb=float(sys.argv[4] or 0)
r0=[r for r in rows if int(r['Dispatch_Id'])==last][0]
print("%s: vgpr %s sgpr %s pops %d | per pop: VALU %.1f SALU %.1f LDS %.2f VMEM_RD %.2f VMEM_WR %.2f | kernel %.1f ms" % (
    sys.argv[2], r0['VGPR_Count'], r0['SGPR_Count'], pops, c['SQ_INSTS_VALU']/pops, c['SQ_INSTS_SALU']/pops, c['SQ_INSTS_LDS']/pops,
    c['SQ_INSTS_VMEM_RD']/pops, c['SQ_INSTS_VMEM_WR']/pops, (int(r0['End_Timestamp'])-int(r0['Start_Timestamp']))/1e6))
if nb:
    print("   per iteration (%d): VALU %.0f SALU %.0f LDS %.1f VMEM_RD %.1f VMEM_WR %.1f, pops %.2f" % (nb, c['SQ_INSTS_VALU']/nb, c['SQ_INSTS_SALU']/nb,
          c['SQ_INSTS_LDS']/nb, c['SQ_INSTS_VMEM_RD']/nb, c['SQ_INSTS_VMEM_WR']/nb, pops/nb))
PY
