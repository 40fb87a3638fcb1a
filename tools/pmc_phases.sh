#!/bin/bash
# diagnostics: VALU/SALU/LDS instruction counts of IMPLSCH with phases ablated (ECWAM_HIP_DEBUG_SKIP)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mask in ${MASKS:-0 1 2 4 8 16 32 127}; do
  export ECWAM_HIP_DEBUG_SKIP=$mask
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/ph_$mask -- python3 tools/prof_implsch.py sp 32768 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/ph_$mask/*/*counter_collection.csv')[0]
agg=collections.defaultdict(float); n=0
for r in csv.DictReader(open(f)):
    if 'implsch' in r['Kernel_Name']:
        agg[r['Counter_Name']]+=float(r['Counter_Value'])
print('mask',$mask,{k:round(v/3/32768) for k,v in agg.items()})
PY
done
