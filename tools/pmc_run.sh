#!/bin/bash
# diagnostics: arbitrary PMC sets on the IMPLSCH profiling driver.  usage: PMC="A B C" [N=32768] [PREC=sp] [GEN=0|2|4] bash tools/pmc_run.sh
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out
N=${N:-32768}; PREC=${PREC:-sp}; TAG=${TAG:-x}; GEN=${GEN:-0}
rm -rf gpurun_out/pmc_$TAG
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -- python3 tools/prof_implsch.py $PREC $N $GEN > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_$TAG/*/*counter_collection.csv')[0]
agg=collections.defaultdict(float); launches=set()
for r in csv.DictReader(open(f)):
    if 'implsch' in r['Kernel_Name']:
        agg[r['Counter_Name']]+=float(r['Counter_Value'])
        if 'implsch4_pre' not in r['Kernel_Name'] and 'implsch4_fin' not in r['Kernel_Name']: launches.add(r['Dispatch_Id'])
print({k:round(v/len(launches)/$N,1) for k,v in agg.items()})   # per IMPLSCH call (k_implsch4 with its two scalar kernels) and sea point
PY
