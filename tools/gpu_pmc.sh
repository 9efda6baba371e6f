#!/bin/bash
# Instruction counters of one config-2 launch (run on the GPU box): VALU / SALU / LDS / VMEM instructions per pop.
# Usage: tools/gpu_pmc.sh [tag]   (honours FXJPS_LIB)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-cur}
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 tools/gpu_prof.py --noprof 10000 > $OUT.log 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$f" "$TAG" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_search' in r['Kernel_Name']]
# last launch of the run (the timed repetition)
last=max(int(r['Dispatch_Id']) for r in rows)
c={r['Counter_Name']:float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last}
pops=332391044.0
r0=[r for r in rows if int(r['Dispatch_Id'])==last][0]
print("%s: vgpr %s sgpr %s | per pop: VALU %.1f SALU %.1f LDS %.2f VMEM_RD %.2f VMEM_WR %.2f | kernel %.1f ms" % (
    sys.argv[2], r0['VGPR_Count'], r0['SGPR_Count'], c['SQ_INSTS_VALU']/pops, c['SQ_INSTS_SALU']/pops, c['SQ_INSTS_LDS']/pops,
    c['SQ_INSTS_VMEM_RD']/pops, c['SQ_INSTS_VMEM_WR']/pops, (int(r0['End_Timestamp'])-int(r0['Start_Timestamp']))/1e6))
PY
